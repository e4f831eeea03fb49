#!/bin/bash
# RECORD of a round-5 A/B call: the svoslam_config switch it sets existed only at the commit of the experiment (git log); the library now ignores it
# streaming tracker: the last frame's vertex (1) / vertex + normal (2) recomputed from its filtered depth instead of read from the maps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
for v in 1 2; do
  SVOSLAM_CONFIG=track_recompute=$v timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_sensor.py tests/test_gpu_corrected.py tests/test_gpu_fullsize.py -x -q -k "1080 or cfg4 or tracker or track" > $O/pytest_$v.log 2>&1; echo "recompute=$v rc=$?"; tail -2 $O/pytest_$v.log
done
for v in 0 1 2 0 1 2; do
  SVOSLAM_CONFIG=track_recompute=$v timeout 300 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --lean 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=[r for r in d['roofline_stages'] if r['stage']=='tracker'][0]
print('recompute $v: %.1f frames/s %s tracker in loop %.3f ms, alone %.3f ms' % (d['value'], [round(x) for x in d['runs']], t['kernel_ms'], d['stages_sequential']['tracker_ms']))"
done
