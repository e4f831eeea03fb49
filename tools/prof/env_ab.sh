#!/bin/bash
# A/B of an environment switch on one box: bash tools/prof/env_ab.sh VAR=off_value reps bench-args...
KV=$1; REPS=$2; shift 2
mkdir -p gpurun_out/envab
for rep in $(seq $REPS); do
  python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', round(d['value'],1), round(d['roofline']['kernel_ms'],4))"
  env $KV python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$KV', round(d['value'],1), round(d['roofline']['kernel_ms'],4))"
done
