// model_depth.hpp -- see model_depth.hip
#pragma once
#include "common.hpp"

namespace svoslam {
// exactly one of cam_to_world (host, 16 floats) / d_cam_to_world (device, 16 floats, read when the kernel runs) is given
int raycast_model_depth(uint16_t *d_depth, int width, int height, float fx, float fy, const float *cam_to_world,
                        const float *d_cam_to_world, const uint32_t *d_octree, const float center[3], float size,
                        unsigned long long *d_steps, hipStream_t stream);
}  // namespace svoslam
