#!/bin/bash
# A/B: the merged kernel (phase 1 = the tuned loop, phase 2 = bursts) against the separate ahead kernel (variant w6b3) and the plain march
O=gpurun_out/r06h; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.1f (%.1f..%.1f) march %.4f trk %.4f alone %.4f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms'], d.get('stages_sequential', {}).get('march_ms', 0)))"; }
SVOSLAM_CONFIG=march_ahead=90 timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_bricks.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2
SVOSLAM_CONFIG=march_ahead=0 timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_bricks.py -x -q -m gpu 2>&1 | tail -2
{
for rep in 1 2 3; do
  for v in plain sep merged merged60 merged130; do
    a=90
    case $v in plain) cp /tmp/base.so $L; a=-1;; sep) cp octree-slam_amd/_variants/libsvoslam_hip_w6b3.so $L;; merged) cp /tmp/base.so $L;; merged60) cp /tmp/base.so $L; a=60;; merged130) cp /tmp/base.so $L; a=130;; esac
    echo -n "$v rep $rep  20: "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "$v rep $rep 100: "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    [ $rep = 1 ] && { echo -n "$v rep $rep cfg4: "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line; }
  done
done
cp /tmp/base.so $L
} 2>&1 | tee $O/merged_ab.txt
for a in -1 90; do echo "== cfg2 march_ahead=$a"; SVOSLAM_CONFIG=march_ahead=$a timeout 250 python tools/prof/mesh_ray_anatomy.py cfg2 2>&1 | grep "reference" | cut -c1-80; done | tee $O/cfg2.txt
