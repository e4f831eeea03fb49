#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_state"; mkdir -p "$OUT"
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
python bench.py --no-cpu-baseline --allow-missing-traffic --stages > $OUT/bench3_stages.json 2> $OUT/bench3_stages.err
python bench.py --no-cpu-baseline --allow-missing-traffic > $OUT/bench3.json 2> $OUT/bench3.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench3_driver.json 2> $OUT/bench3_driver.err
python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench4.json 2> $OUT/bench4.err
python tools/prof/runner_timeline.py > $OUT/timeline.txt 2>&1
grep -n "passed\|failed" $OUT/pytest.log
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']], d.get('stages'))"; done
tail -20 $OUT/timeline.txt
