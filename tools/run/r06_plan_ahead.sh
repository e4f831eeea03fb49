#!/bin/bash
# A/B: deferred schedule with the plans as a structure chain of PENDING links (SVOSLAM_PLAN_AHEAD = 1: on S behind the sort, 2: on C ahead
# of the commit) against the schedule where plan k+1 waits for apply k (0)
O=gpurun_out/r06p; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.1f (%.1f..%.1f) march %.4f trk %.4f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms']))"; }
timeout 600 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu -k "pending_structure or structure_chain or deferred" 2>&1 | tail -3
for m in 1 2; do
  echo "== tests with SVOSLAM_PLAN_AHEAD=$m"
  SVOSLAM_PLAN_AHEAD=$m timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_reentrancy.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
done
{
for rep in 1 2 3; do
  for m in 0 1 2; do
    echo -n "ahead=$m rep $rep  20: "; SVOSLAM_PLAN_AHEAD=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "ahead=$m rep $rep 100: "; SVOSLAM_PLAN_AHEAD=$m python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  done
done
} 2>&1 | tee $O/plan_ahead_ab.txt
