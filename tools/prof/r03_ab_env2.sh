#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_ab_env2"; mkdir -p "$OUT"; rm -f $OUT/*.json
cd $R
for rep in 1 2; do
for d in 0 1; do
  SVOSLAM_RUNNER_DEFERRED=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench3drv_d${d}_$rep.json 2> $OUT/bench3drv_d${d}_$rep.err
  SVOSLAM_RUNNER_DEFERRED=$d python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench4_d${d}_$rep.json 2> $OUT/bench4_d${d}_$rep.err
done; done
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"; done
