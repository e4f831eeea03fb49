#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks9"; mkdir -p "$OUT"
cd $R
for b in 1 0; do
  echo "== cfg4 like bench bricks=$b"; LIKE_BENCH=1 COUNT=1 DIAG_CFG4=1 SVOSLAM_MARCH_BRICKS=$b python tools/prof/cold_march.py 45 2>&1 | grep "frame\|map built"
  echo "== cfg4 plain bricks=$b"; COUNT=1 DIAG_CFG4=1 SVOSLAM_MARCH_BRICKS=$b python tools/prof/cold_march.py 45 2>&1 | grep "frame\|map built"
done > $OUT/cold.txt 2>&1
cat $OUT/cold.txt
