#!/bin/bash
# A/B at 1080p with the round's march settings: deferred commits / stream priorities (both default off above 640x480-class images)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for v in "" "runner_deferred=1" "runner_prio=1" "runner_deferred=1,runner_prio=1"; do
  SVOSLAM_CONFIG=$v timeout 300 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --lean 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v]: %.1f frames/s %s' % (d['value'], [round(x) for x in d['runs']]), ' '.join('%s %.3f' % (r['stage'], r['kernel_ms']) for r in d['roofline_stages']))"
done
done
