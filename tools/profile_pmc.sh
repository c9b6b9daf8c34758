#!/bin/bash
# SQ / HBM counter passes over bench.py (run on the GPU box).  Counters are collected in passes of
# their own (no trace options next to --pmc).  Usage: tools/profile_pmc.sh TAG [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pass$i --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-upstream-leg "$@" > $OUT/pass$i.log 2>&1 || { tail -5 $OUT/pass$i.log; echo "pass $i ($set) failed"; }
done
python3 tools/pmc_summary.py $OUT/summary.json $OUT/pass*/
