#!/bin/bash
# where the brick march's steps go (diag variant), its anatomy (band timing), with and without bricks
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks2"; mkdir -p "$OUT"
cd $R
L=octree-slam_amd/libsvoslam_hip.so
if [ -n "$WITH_TESTS" ]; then ( time timeout 900 python -m pytest $WITH_TESTS -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1; fi
cp $L /tmp/base.so
cp octree-slam_amd/_variants/libsvoslam_hip_diag.so $L
python tools/prof/render_only.py 300 > $OUT/render_diag.txt 2>&1
cp /tmp/base.so $L
python tools/prof/render_only.py 300 > $OUT/render_b1.txt 2>&1
SVOSLAM_MARCH_BRICKS=1 python tools/prof/ray_anatomy.py 300 > $OUT/anatomy_b1.txt 2>&1
if [ -n "$WITH_BENCH" ]; then
  for b in 1 0; do SVOSLAM_MARCH_BRICKS=$b python bench.py --no-cpu-baseline --allow-missing-traffic > $OUT/bench_b${b}.json 2> $OUT/bench_b${b}.err; done
  for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"; done
fi
[ -f $OUT/pytest.log ] && tail -5 $OUT/pytest.log
cat $OUT/render_diag.txt $OUT/render_b1.txt; tail -34 $OUT/anatomy_b1.txt
