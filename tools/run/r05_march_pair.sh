#!/bin/bash
# RECORD of a round-5 A/B call: the SVO_EXP_* environment knobs it sets existed only at the commit of the experiment (git log); the library now ignores them
# EXPERIMENT: the two 32 x 8 strips of a march tile half the render apart (wavefronts w and w + 4 share a SIMD)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f %s march in loop %.3f alone %.3f' % (d['value'], [round(x) for x in d['runs']], d['roofline_stages'][0]['kernel_ms'], d['stages_sequential']['march_ms']))"; }
for rep in 1 2; do
for v in 0 1; do
  echo "== rep $rep pair $v"
  SVO_EXP_PAIR=$v python tools/prof/render_only.py 300 2>&1 | grep -E "standalone" | head -1
  echo -n "cfg3 100:"; SVO_EXP_PAIR=$v python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "cfg3 20: "; SVO_EXP_PAIR=$v python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "cfg4 40: "; SVO_EXP_PAIR=$v python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
done
done
SVO_EXP_PAIR=1 timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_bricks.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
