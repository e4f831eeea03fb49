#!/bin/bash
# builds octree-slam_amd/_variants/libsvoslam_hip_diag.so (-DSVO_BRICK_DIAG: clock64 stamps + step classes in the brick march) and
# then the plain library again; run in the build container before tools/prof/profile_round.sh goes to the GPU box
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/octree-slam_amd/_variants
SVOSLAM_EXTRA_HIPCC_FLAGS="-DSVO_BRICK_DIAG" python $R/octree-slam_amd/build.py --force > /dev/null || exit 1
cp $R/octree-slam_amd/libsvoslam_hip.so $R/octree-slam_amd/_variants/libsvoslam_hip_diag.so
python $R/octree-slam_amd/build.py --force > /dev/null || exit 1
ls -la $R/octree-slam_amd/_variants/libsvoslam_hip_diag.so $R/octree-slam_amd/libsvoslam_hip.so
