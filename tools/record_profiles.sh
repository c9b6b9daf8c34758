#!/bin/bash
# Records one evidence set under gpurun_out/<TAG>_<NAME>_*: the bench line (with its own PMC passes), rocprofv3 kernel stats of
# the same command, and SQ counter passes.  Usage (on the GPU box):
#   tools/record_profiles.sh TAG [workload] [NAME] [extra bench.py args...]
# e.g. tools/record_profiles.sh r03_v1 config3
#      tools/record_profiles.sh r03_v1 config3 config3_two_call --opacity second-call --train-step
TAG=$1; WL=${2:-config3}; NAME=${3:-$WL}; shift; shift; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python bench.py --steps 200 --warmup 20 --workload $WL "$@" > $OUT/${TAG}_${NAME}_bench.json 2> $OUT/${TAG}_${NAME}_bench.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${NAME}_stats --output-format csv -- python3 $OLDPWD/bench.py --steps 30 --warmup 3 --workload $WL --no-cpu-baseline --no-pmc "$@" > /dev/null 2>&1
cd $OLDPWD
cp $(ls $OUT/${TAG}_${NAME}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_${NAME}_kernel_stats.csv
rm -rf $OUT/${TAG}_${NAME}_stats
tools/profile_pmc.sh ${TAG}_${NAME} --workload $WL "$@" > /dev/null 2>&1
cp $OUT/pmc_${TAG}_${NAME}/summary.json $OUT/${TAG}_${NAME}_sq_counters.json
rm -rf $OUT/pmc_${TAG}_${NAME}
python -c "
import json; d=json.load(open('$OUT/${TAG}_${NAME}_bench.json')); print(d['value'], d['ms_per_step'], d.get('upstream_rect',{}).get('value'), d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic')); print(d['stages_ms'])"
