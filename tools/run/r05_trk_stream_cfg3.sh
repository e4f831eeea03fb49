#!/bin/bash
# A/B: 640x480 with the tracker's worker count capped (svoslam_config.track_workers) so that its finest level no longer fits the registers
# and the STREAMING one-launch form (168 VGPRs: other kernels' wavefronts fit beside it) runs instead of the register-resident one (213)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for v in 0 120 100 74 50; do
  SVOSLAM_CONFIG=track_workers=$v timeout 300 python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=[r for r in d['roofline_stages'] if r['stage']=='tracker'][0]
print('track_workers=$v: %.1f frames/s %s tracker %s in loop %.3f ms, alone %.3f ms' % (d['value'], [round(x) for x in d['runs']], t['kernel'][:44], t['kernel_ms'], d['stages_sequential']['tracker_ms']))"
done
done
