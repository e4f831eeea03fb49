#!/bin/bash
# RECORD of a round-5 A/B call: the SVO_EXP_* environment knobs it sets existed only at the commit of the experiment (git log); the library now ignores them
# EXPERIMENT: the brick march with fewer resident workgroups per CU (dynamic LDS padding) and its tiles costliest-first at 640x480
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "0 768" "0 100" "6144 768" "6144 100" "33000 100" "33000 768"; do
  set -- $cfg
  echo "== lds pad $1 B, tile order above $2 tiles"
  SVO_EXP_MARCH_LDS_PAD=$1 SVO_EXP_TILE_ORDER_MIN=$2 python tools/prof/render_only.py 300 2>&1 | grep -E "standalone|ms" | head -3
  SVO_EXP_MARCH_LDS_PAD=$1 SVO_EXP_TILE_ORDER_MIN=$2 python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench 100 frames: %.1f %s march in loop %.3f alone %.3f' % (d['value'], [round(x) for x in d['runs']], d['roofline_stages'][0]['kernel_ms'], d['stages_sequential']['march_ms']))"
done
