export TMPDIR=/tmp
for lpl in 2 3 4; do
  GSPLAT_EXTRA_HIPCC_FLAGS=-DTBK_LPL=$lpl python 3dgs-avatar-release_amd/build.py --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  cd /tmp; rm -rf /tmp/st_$lpl; rocprofv3 --kernel-trace --stats -d /tmp/st_$lpl --output-format csv -- python3 /root/repo/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-upstream-leg > /dev/null 2>&1; cd /root/repo
  python -c "
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/st_$lpl/*/*kernel_stats.csv')[0])))
print('LPL=$lpl', {r['Name'][:17]: round(float(r['AverageNs'])/1e3,1) for r in rows if r['Name'].startswith(('tile_write','tile_count'))})"
done
GSPLAT_EXTRA_HIPCC_FLAGS=-DTBK_LPL=2 python 3dgs-avatar-release_amd/build.py --force > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
