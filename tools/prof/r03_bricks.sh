#!/bin/bash
# occupancy bricks: the suite, then same-box A/B of the march with and without them (SVOSLAM_MARCH_BRICKS=0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks"; mkdir -p "$OUT"
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest.log 2>&1
for rep in 1 2; do
  for b in 1 0; do
    SVOSLAM_MARCH_BRICKS=$b python tools/prof/render_only.py 300 > $OUT/render_b${b}_$rep.txt 2>&1
    SVOSLAM_MARCH_BRICKS=$b python bench.py --no-cpu-baseline --allow-missing-traffic > $OUT/bench_b${b}_$rep.json 2> $OUT/bench_b${b}_$rep.err
  done
done
SVOSLAM_MARCH_BRICKS=1 python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench4_b1.json 2> $OUT/bench4_b1.err
SVOSLAM_MARCH_BRICKS=0 python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench4_b0.json 2> $OUT/bench4_b0.err
tail -5 $OUT/pytest.log
grep -H "standalone" $OUT/render_*.txt
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"; done
