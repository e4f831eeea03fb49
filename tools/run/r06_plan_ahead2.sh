#!/bin/bash
# plan-ahead (SVOSLAM_PLAN_AHEAD=1) x refresh shape (2048 workgroups x 4 bricks per wavefront | 4096 x 2): does a shorter refresh pay once the
# apply -> plan -> commit cycle is gone?
O=gpurun_out/r06p; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.1f (%.1f..%.1f) march %.4f trk %.4f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms']))"; }
echo "== tests with SVOSLAM_PLAN_AHEAD=1"
SVOSLAM_PLAN_AHEAD=1 timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_reentrancy.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
{
for rep in 1 2 3; do
  for v in base r4096x2; do
    for m in 0 1; do
      case $v in base) cp /tmp/base.so $L;; *) cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L;; esac
      echo -n "$v ahead=$m rep $rep  20: "; SVOSLAM_PLAN_AHEAD=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
      echo -n "$v ahead=$m rep $rep 100: "; SVOSLAM_PLAN_AHEAD=$m python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    done
  done
done
cp /tmp/base.so $L
} 2>&1 | tee $O/plan_ahead_refresh_ab.txt
