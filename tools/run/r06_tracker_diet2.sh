#!/bin/bash
# A/B: tracker rows without the zero / one terms (jonly), plus the pending matrix in scalar registers (new), against the library before (base)
O=gpurun_out/r06t; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/new.so
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d.get('stages_sequential') or {}
print('%.1f (%.1f..%.1f) march %.4f trk in loop %.4f alone %.4f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms'], s.get('tracker_ms', 0)))"; }
timeout 1700 python -m pytest tests/test_gpu_rgbd.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_corrected.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
{
for rep in 1 2 3; do
  for v in base jonly new; do
    case $v in new) cp /tmp/new.so $L;; *) cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L;; esac
    echo -n "$v rep $rep cfg3 100: "; python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs 2>/dev/null | line
    echo -n "$v rep $rep cfg4  40: "; python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs 2>/dev/null | line
  done
done
cp /tmp/new.so $L
} 2>&1 | tee $O/tracker_diet2_ab.txt
