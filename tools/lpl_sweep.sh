#!/bin/bash
# Sweeps build-time parameters of the tile-list writer (binning.hip) on the GPU box: rebuilds the library with
# -D flags, reads the kernel's duration from rocprofv3 --kernel-trace --stats, restores the default build.
export TMPDIR=/tmp
for flags in "" "-DTB_H=2 -DTBK_THREADS=512 -DTB_TARGET_WGS=1024" "-DTB_H=2 -DTBK_THREADS=1024 -DTB_TARGET_WGS=512" "-DTB_H=1 -DTBK_THREADS=256 -DTB_TARGET_WGS=2048" "-DTB_H=2 -DTBK_THREADS=256 -DTB_TARGET_WGS=1024" "-DTB_H=4 -DTBK_THREADS=512 -DTB_TARGET_WGS=512" "-DTBK_LPL=1"; do
  GSPLAT_EXTRA_HIPCC_FLAGS="$flags" python 3dgs-avatar-release_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  python -m pytest tests -m gpu -q -x -k "preprocess_and_binning or config5 or config2" > /tmp/t.log 2>&1 || { echo "TESTS FAILED: $flags"; tail -3 /tmp/t.log; continue; }
  cd /tmp; rm -rf /tmp/st; rocprofv3 --kernel-trace --stats -d /tmp/st --output-format csv -- python3 /root/repo/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-upstream-leg > /dev/null 2>&1; cd /root/repo
  python -c "
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/st/*/*kernel_stats.csv')[0])))
print('[$flags]', {r['Name'][:17]: round(float(r['AverageNs'])/1e3,1) for r in rows if r['Name'].startswith(('tile_write','tile_count','seg_prefix'))})"
done
GSPLAT_EXTRA_HIPCC_FLAGS="" python 3dgs-avatar-release_amd/build.py --force > /dev/null 2>&1
