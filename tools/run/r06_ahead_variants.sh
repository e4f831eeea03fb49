#!/bin/bash
# same-box A/B of cone_trace variants (tools/prof/build_variant.sh <name> cone_trace <flags>) x march_ahead settings
O=gpurun_out/r06e; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.1f frames/s (%.1f..%.1f) march %.4f ms tracker %.4f ms' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms']), d.get('stages_sequential', {}).get('march_ms'))"; }
run() {
  echo -n "  cfg3 100: "; SVOSLAM_CONFIG=march_ahead=$1 python bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "  cfg3 20:  "; SVOSLAM_CONFIG=march_ahead=$1 python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  echo -n "  cfg4 40:  "; SVOSLAM_CONFIG=march_ahead=$1 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
}
{
echo "== base, march_ahead=-1"; run -1
for v in ${VARIANTS:-w6b2 w5b2 w4b3 w6b3}; do
  cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L
  for a in ${AHEADS:-0 90}; do echo "== $v, march_ahead=$a"; run $a; done
done
cp /tmp/base.so $L
echo "== base, march_ahead=-1"; run -1
} 2>&1 | tee $O/ab.txt
