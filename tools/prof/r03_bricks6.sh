#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks6"; mkdir -p "$OUT"
cd $R
( time timeout 900 python -m pytest ${TESTS:-tests/test_gpu_render.py tests/test_gpu_pipeline.py} -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
cp octree-slam_amd/_variants/libsvoslam_hip_diag.so $L
python tools/prof/band_diag.py 300 > $OUT/band_diag.txt 2>&1
cp /tmp/base.so $L
python tools/prof/render_only.py 300 > $OUT/render_b1.txt 2>&1
for b in 1 ${AB0}; do
  SVOSLAM_MARCH_BRICKS=$b python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic > $OUT/bench4_b$b.json 2> $OUT/bench4_b$b.err
  SVOSLAM_MARCH_BRICKS=$b python bench.py --no-cpu-baseline --allow-missing-traffic > $OUT/bench3_b$b.json 2> $OUT/bench3_b$b.err
done
grep -n "passed\|failed" $OUT/pytest.log
cat $OUT/band_diag.txt; grep "mode 0" $OUT/render_b1.txt
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), [ (s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"; done
