#!/bin/bash
# same-box A/B of environment switches: AB_ENVS="name1:VAR=val,VAR2=val name2:..." ; each runs bench cfg3 (and cfg4 if AB_CFG4)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_ab_env"; mkdir -p "$OUT"; rm -f $OUT/*.json
cd $R
for rep in 1 2; do
for spec in base: $AB_ENVS; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do export "$kv"; done
    python bench.py --no-cpu-baseline --allow-missing-traffic > $OUT/bench3_${name}_$rep.json 2> $OUT/bench3_${name}_$rep.err
    [ -n "$AB_CFG4" ] && python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench4_${name}_$rep.json 2> $OUT/bench4_${name}_$rep.err )
done; done
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"; done
