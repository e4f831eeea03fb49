// cone_trace.hpp -- see cone_trace.hip
#pragma once
#include "common.hpp"

namespace svoslam {
int cone_trace_svo(uint8_t *d_pos, int width, int height, float fov, const float view[16], const uint32_t *d_octree,
                   const float center[3], float size, int mode, unsigned long long *d_steps, hipStream_t stream);
}  // namespace svoslam
