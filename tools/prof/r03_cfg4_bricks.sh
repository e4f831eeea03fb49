#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_cfg4_bricks"; mkdir -p "$OUT"
cd $R
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
cp octree-slam_amd/_variants/libsvoslam_hip_diag.so $L
SVOSLAM_BRICK_MAX_DEPTH=14 DIAG_CFG4=1 python tools/prof/band_diag.py 45 2>&1 | grep -v "brick diag" > $OUT/band_diag4.txt
cp /tmp/base.so $L
for d in 14 12; do
  echo "== cold/hot march, bricks up to depth $d"; SVOSLAM_BRICK_MAX_DEPTH=$d LIKE_BENCH=1 COUNT=1 DIAG_CFG4=1 python tools/prof/cold_march.py 45 2>&1 | grep "frame\|map built"
  SVOSLAM_BRICK_MAX_DEPTH=$d python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --allow-missing-traffic 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench cfg4', round(d['value'],1), [(s['stage'], round(s['kernel_ms'],4)) for s in d['roofline_stages']])"
done > $OUT/cold.txt 2>&1
cat $OUT/band_diag4.txt $OUT/cold.txt
