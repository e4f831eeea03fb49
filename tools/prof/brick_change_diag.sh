cd $GRAFT_REPO_ROOT
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
cp octree-slam_amd/_variants/libsvoslam_hip_diag.so $L
python bench.py --no-cpu-baseline --no-stage-pass 2>&1 | grep "brick diag" | tail -4
cp /tmp/base.so $L
