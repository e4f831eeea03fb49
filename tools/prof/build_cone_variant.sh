#!/bin/bash
# builds octree-slam_amd/_variants/libsvoslam_hip_<name>.so = the built library with cone_trace.hip recompiled with extra -D flags
#   bash tools/prof/build_cone_variant.sh w6b2 -DSVO_AHEAD_WAVES=6 -DSVO_AHEAD_BURST=2
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
C=$R/octree-slam_amd/csrc; V=$R/octree-slam_amd/_variants; mkdir -p $V /tmp/variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -munsafe-fp-atomics "$@" -c $C/cone_trace.hip -o /tmp/variant_$name/cone_trace.o || exit 1
objs=$(ls $C/_obj/*.o | grep -v cone_trace.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libsvoslam_hip_$name.so $objs /tmp/variant_$name/cone_trace.o || exit 1
ls -la $V/libsvoslam_hip_$name.so
