// cone_trace.hpp -- see cone_trace.hip
#pragma once
#include "common.hpp"

namespace svoslam {
// traces image rows [row_first, row_first + rows) of a width x height frame into d_pos (full-frame buffer)
int cone_trace_svo(uint8_t *d_pos, int width, int height, int row_first, int rows, float fov, const float view[16],
                   const uint32_t *d_octree, const float center[3], float size, int mode, unsigned long long *d_steps,
                   hipStream_t stream);
// frees the per-stream acceleration buffer(s); the caller has synchronised the stream(s)
int cone_trace_release(hipStream_t stream, bool all);
}  // namespace svoslam
