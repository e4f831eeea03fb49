#!/bin/bash
# confirmation A/B: base vs two variants at march_ahead=90, alternating, driver's command (20 frames) and 100 frames
O=gpurun_out/r06f; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.1f (%.1f..%.1f) march %.4f trk %.4f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms']))"; }
{
for rep in 1 2 3; do
  for v in base w6b3 w6b2; do
    if [ $v = base ]; then cp /tmp/base.so $L; a=-1; else cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L; a=90; fi
    echo -n "$v rep $rep  20: "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "$v rep $rep 100: "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  done
done
cp /tmp/base.so $L
} 2>&1 | tee $O/confirm.txt
for v in base w6b3; do
  if [ $v = base ]; then cp /tmp/base.so $L; a=-1; else cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L; a=0; fi
  echo "== cfg2 anatomy $v march_ahead=$a"; SVOSLAM_CONFIG=march_ahead=$a timeout 250 python tools/prof/mesh_ray_anatomy.py cfg2 2>&1 | grep "reference" | cut -c1-80
done | tee $O/cfg2.txt
cp /tmp/base.so $L
timeout 300 python tools/prof/mesh_ray_anatomy.py cfg5 2>&1 | grep view > $O/cfg5_anat.txt; grep -o "view [0-9] ([^)]*) [a-z]*: [0-9.]* ms\|lit pixels [0-9]*" $O/cfg5_anat.txt | paste - -
timeout 600 python -m pytest tests/test_gpu_mesh.py -x -q -m gpu 2>&1 | tail -3
