#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks4"; mkdir -p "$OUT"
cd $R
for m in 0 1 2; do
  SVOSLAM_MARCH_XCD=$m python tools/prof/render_only.py 300 > $OUT/render_x$m.txt 2>&1
  SVOSLAM_MARCH_XCD=$m python bench.py --no-cpu-baseline --allow-missing-traffic > $OUT/bench_x$m.json 2> $OUT/bench_x$m.err
done
grep -H "mode 0" $OUT/render_x*.txt
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"; done
