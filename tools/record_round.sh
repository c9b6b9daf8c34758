#!/bin/bash
# The secondary evidence of a round, on the GPU box: bench lines of the other BASELINE shapes and of the neighbouring rows,
# the two-rank rehearsals, the graph-capture log.  Usage: tools/record_round.sh TAG   (files: gpurun_out/<TAG>_*)
TAG=$1
OUT=$PWD/gpurun_out
B="python bench.py --no-cpu-baseline --no-pmc"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("upstream_rect", {}).get("value"), d["stages_ms"], flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, flush=True)
PY
}
for wl in config2 config4 config5 avatar avatar50k; do
  steps=300; [ $wl = config2 ] && steps=500; [ $wl = config5 ] && steps=100
  $B --workload $wl --steps $steps --warmup 10 > $OUT/${TAG}_${wl}_bench.json 2> $OUT/${TAG}_${wl}_bench.err; line $OUT/${TAG}_${wl}_bench.json
done
$B --steps 200 --warmup 10 --l1 separate > $OUT/${TAG}_config3_l1_separate_bench.json 2>/dev/null; line $OUT/${TAG}_config3_l1_separate_bench.json
$B --steps 200 --warmup 10 --opacity second-call --train-step > $OUT/${TAG}_config3_two_call_bench.json 2>/dev/null; line $OUT/${TAG}_config3_two_call_bench.json
$B --steps 200 --warmup 10 --opacity fused --train-step > $OUT/${TAG}_config3_fused_opacity_bench.json 2>/dev/null; line $OUT/${TAG}_config3_fused_opacity_bench.json
$B --steps 200 --warmup 10 --loss l1+dssim --prepass --opacity second-call --train-step > $OUT/${TAG}_config3_full_step_bench.json 2>/dev/null; line $OUT/${TAG}_config3_full_step_bench.json
$B --workload avatar --steps 200 --warmup 10 --opacity second-call --train-step > $OUT/${TAG}_avatar_two_call_bench.json 2>/dev/null; line $OUT/${TAG}_avatar_two_call_bench.json
GSPLAT_BENCH_REHEARSAL=1 $B --gpus 2 --steps 50 --warmup 5 --no-upstream-leg > $OUT/${TAG}_rehearsal2_config3_weak.json 2>/dev/null; line $OUT/${TAG}_rehearsal2_config3_weak.json
GSPLAT_BENCH_REHEARSAL=1 $B --gpus 2 --workload config4 --total-frames 300 --warmup 5 --no-upstream-leg > $OUT/${TAG}_rehearsal2_config4_strong.json 2>/dev/null; line $OUT/${TAG}_rehearsal2_config4_strong.json
{ timeout -k 10 300 python tools/graph_capture_check.py config2 500; timeout -k 10 300 python tools/graph_capture_check.py avatar50k 500; } > $OUT/${TAG}_graph_capture_check.txt 2>&1
grep -E "^stage|us/frame|N=" $OUT/${TAG}_graph_capture_check.txt
