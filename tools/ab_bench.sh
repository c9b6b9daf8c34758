#!/bin/bash
# A/B of whole bench runs on the GPU box: tools/ab_bench.sh OUTDIR "label|ENV=.. ENV=..|bench args" ...  (two interleaved passes)
OUT=$1; shift
mkdir -p $OUT
for pass in 1 2; do
  for spec in "$@"; do
    label=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
    env $envs python bench.py --no-pmc --no-cpu-baseline $args > $OUT/${label}_$pass.json 2> $OUT/${label}_$pass.err
    python - "$OUT/${label}_$pass.json" "$label" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    up = d.get("upstream_rect", {}).get("value")
    print(sys.argv[2], d["value"], d["ms_per_step"], up, d["stages_ms"], flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
  done
done
