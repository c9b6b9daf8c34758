#!/bin/bash
# Per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats), printed as a table and kept as
# gpurun_out/<TAG>_kernel_stats.csv.  Usage (on the GPU box): tools/kstats.sh TAG [bench.py args...]
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats --output-format csv -- python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-pmc --no-upstream-leg "$@" > $OUT/${TAG}_stats.log 2>&1 || { tail -20 $OUT/${TAG}_stats.log; exit 1; }
cd $ROOT
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_stats
python3 - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-70s %8s %10s %7s" % ("kernel", "calls", "avg_us", "pct"))
for r in rows[:28]:
    print("%-70s %8s %10.2f %6.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -c 400 $OUT/${TAG}_stats.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2
