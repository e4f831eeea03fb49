#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks5"; mkdir -p "$OUT"
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
for m in 2 1; do for b in 1 0; do
  SVOSLAM_MARCH_XCD=$m SVOSLAM_MARCH_BRICKS=$b python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench4_x${m}_b$b.json 2> $OUT/bench4_x${m}_b$b.err
done; done
tail -12 $OUT/pytest.log
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"; done
