#!/bin/bash
# generic same-box A/B: base library against the named variants, alternating; VARIANTS="a b" REPS=3 CFG4=1
O=gpurun_out/${OUTDIR:-r06i}; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ss = d.get('stages_sequential', {})
print('%.1f (%.1f..%.1f) march %.4f trk %.4f | alone: sort %.3f plan %.3f commit %.3f march %.3f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms'], ss.get('fuse_sort_ms', 0), ss.get('fuse_plan_ms', 0), ss.get('fuse_commit_ms', 0), ss.get('march_ms', 0)))"; }
{
for rep in $(seq ${REPS:-3}); do
  for v in base ${VARIANTS}; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L; fi
    echo -n "$v rep $rep  20: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "$v rep $rep 100: "; python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    [ -n "$CFG4" ] && { echo -n "$v rep $rep cfg4: "; python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line; }
  done
done
cp /tmp/base.so $L
} 2>&1 | tee $O/ab.txt
if [ -n "$TESTS" ]; then for v in ${VARIANTS}; do cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L; echo "== tests with $v"; timeout 1200 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -2; done; cp /tmp/base.so $L; fi
