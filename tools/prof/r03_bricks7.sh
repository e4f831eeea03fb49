#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks7"; mkdir -p "$OUT"
cd $R
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
cp octree-slam_amd/_variants/libsvoslam_hip_diag.so $L
DIAG_CFG4=1 python tools/prof/band_diag.py 45 > $OUT/band_diag4.txt 2>&1
cp /tmp/base.so $L
cat $OUT/band_diag4.txt
