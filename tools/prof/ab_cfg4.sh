#!/bin/bash
# cfg4 only: several alternating runs of the built library and the variants (see ab_variant.sh)
REPS=${1:-3}; STEPS=${2:-40}
mkdir -p gpurun_out/ab4
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
for rep in $(seq $REPS); do
  for v in base ${AB_VARIANTS:-orig}; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L; fi
    python bench.py --workload cfg4 --steps $STEPS --no-cpu-baseline $AB_ARGS > gpurun_out/ab4/bench4_${v}_$rep.json 2>/dev/null
    [ -n "$AB_20" ] && python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab4/bench20_${v}_$rep.json 2>/dev/null
  done
done
cp /tmp/base.so $L
for f in gpurun_out/ab4/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), round(d['roofline']['kernel_ms'],4), d.get('stages'))"; done
