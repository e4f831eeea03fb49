#!/bin/bash
# the sort's fused histograms (svoslam_config.sort_fuse_hist): parity tests, then A/B by the config switch on one library
O=gpurun_out/r06j; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
timeout 1200 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ss = d.get('stages_sequential', {})
print('%.1f (%.1f..%.1f) march %.4f trk %.4f | alone: sort %.3f plan %.3f commit %.3f march %.3f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms'], ss.get('fuse_sort_ms', 0), ss.get('fuse_plan_ms', 0), ss.get('fuse_commit_ms', 0), ss.get('march_ms', 0)))"; }
{
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "sort_fuse_hist=$v rep $rep  20: "; SVOSLAM_CONFIG=sort_fuse_hist=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "sort_fuse_hist=$v rep $rep 100: "; SVOSLAM_CONFIG=sort_fuse_hist=$v python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "sort_fuse_hist=$v rep $rep cfg4: "; SVOSLAM_CONFIG=sort_fuse_hist=$v python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  done
done
} 2>&1 | tee $O/ab.txt
python tools/prof/runner_timeline.py 2>&1 | grep -v amdgpu.ids | tail -4
