#!/bin/bash
# A/B of the march in bursts (svoslam_config.march_ahead) on one box: parity tests with it on, then bench lines per setting
O=gpurun_out/r06d; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
SVOSLAM_CONFIG=march_ahead=0 timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_bricks.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests_ahead0.log 2>&1; tail -3 $O/tests_ahead0.log
SVOSLAM_CONFIG=march_ahead=40 timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_bricks.py -x -q -m gpu > $O/tests_ahead40.log 2>&1; tail -3 $O/tests_ahead40.log
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.1f frames/s (%.1f..%.1f) march %.4f ms tracker %.4f ms' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms']), d.get('stages_sequential', {}).get('march_ms'))"; }
for a in ${AHEADS:--1 0 60 120 1000000 -1 0 90}; do
  echo "== march_ahead=$a"
  echo -n "cfg3 100: "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "cfg3 20:  "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "cfg4 40:  "; SVOSLAM_CONFIG=march_ahead=$a python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
done 2>&1 | tee $O/ab.txt
for a in -1 0 100; do echo "== cfg2 anatomy march_ahead=$a"; SVOSLAM_CONFIG=march_ahead=$a timeout 250 python tools/prof/mesh_ray_anatomy.py cfg2 2>&1 | grep "reference" | cut -c1-80,330-420; done | tee $O/cfg2.txt
timeout 300 python tools/prof/mesh_ray_anatomy.py cfg5 2>&1 | grep view > $O/cfg5_anat.txt; grep -o "view [0-9] ([^)]*) [a-z]*: [0-9.]* ms\|lit pixels [0-9]*" $O/cfg5_anat.txt | paste - -
