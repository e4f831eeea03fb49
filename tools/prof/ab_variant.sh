#!/bin/bash
export SVOSLAM_BENCH_FULL_LINE=1
# Same-box A/B of a variant library against the built one (frames/s differ by +-2 % from box to box):
#   octree-slam_amd/_variants/libsvoslam_hip_spec.so = the variant (built by hand with extra -D flags, git-ignored)
# Usage (on the GPU box): bash tools/prof/ab_variant.sh [reps] [pytest-files...]
REPS=${1:-2}; shift
mkdir -p gpurun_out/ab
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
for rep in $(seq $REPS); do
  for v in base ${AB_VARIANTS:-spec}; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L; fi
    python tools/prof/render_only.py 100 > gpurun_out/ab/render_${v}_$rep.txt 2>&1
    python bench.py --no-cpu-baseline > gpurun_out/ab/bench_${v}_$rep.json 2>/dev/null
    if [ -n "$AB_20" ]; then python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/bench20_${v}_$rep.json 2>/dev/null; fi
    if [ -n "$AB_CFG4" ]; then python bench.py --workload cfg4 --steps 40 --no-cpu-baseline > gpurun_out/ab/bench4_${v}_$rep.json 2>/dev/null; fi
  done
done
if [ $# -gt 0 ]; then
  cp octree-slam_amd/_variants/libsvoslam_hip_spec.so $L
  python -m pytest "$@" -q -m gpu -x 2>&1 | grep -E "passed|failed|error" > gpurun_out/ab/pytest_spec.txt
fi
cp /tmp/base.so $L
grep -H "standalone" gpurun_out/ab/render_*.txt
for f in gpurun_out/ab/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), round(d['roofline']['kernel_ms'],4))"; done
[ -f gpurun_out/ab/pytest_spec.txt ] && cat gpurun_out/ab/pytest_spec.txt
