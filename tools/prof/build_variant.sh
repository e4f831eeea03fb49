#!/bin/bash
# builds octree-slam_amd/_variants/libsvoslam_hip_<name>.so = the built library with the named translation units recompiled with extra flags
#   bash tools/prof/build_variant.sh sortprio "radix_sort svo_build" -DSVO_EXP_SORT_PRIO
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; files=$2; shift 2
C=$R/octree-slam_amd/csrc; V=$R/octree-slam_amd/_variants; T=/tmp/variant_$name; mkdir -p $V $T
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -munsafe-fp-atomics"
skip=""
for f in $files; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$f.hip -o $T/$f.o || exit 1; skip="$skip|$f.o"; done
objs=$(ls $C/_obj/*.o | grep -v -E "/(${skip#|})$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libsvoslam_hip_$name.so $objs $(for f in $files; do echo $T/$f.o; done) || exit 1
ls -la $V/libsvoslam_hip_$name.so
