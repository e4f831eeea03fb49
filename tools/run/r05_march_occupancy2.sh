#!/bin/bash
# RECORD of a round-5 A/B call: the SVO_EXP_* environment knobs it sets existed only at the commit of the experiment (git log); the library now ignores them
# EXPERIMENT (second pass): two workgroups per CU + costliest-first tiles against the baseline, alternated; cfg4 and the driver's 20 frames too
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f %s march in loop %.3f alone %.3f' % (d['value'], [round(x) for x in d['runs']], d['roofline_stages'][0]['kernel_ms'], d['stages_sequential']['march_ms']))"; }
for rep in 1 2; do
for cfg in "0 768" "6144 100" "4096 100" "12288 100"; do
  set -- $cfg
  echo "== rep $rep lds pad $1 B, tile order above $2 tiles"
  echo -n "cfg3 100:"; SVO_EXP_MARCH_LDS_PAD=$1 SVO_EXP_TILE_ORDER_MIN=$2 python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "cfg3 20: "; SVO_EXP_MARCH_LDS_PAD=$1 SVO_EXP_TILE_ORDER_MIN=$2 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "cfg4 40: "; SVO_EXP_MARCH_LDS_PAD=$1 SVO_EXP_TILE_ORDER_MIN=$2 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
done
done
