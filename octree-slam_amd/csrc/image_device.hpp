// image_device.hpp -- per-pixel device functions shared by the image kernels and the fused fusion front end
#pragma once
#include <math.h>

#include "common.hpp"

namespace svoslam {

// generateVertexMap (image_kernels.cu:24-58) for one pixel
__device__ inline void vertex_from_depth(int depth, int x, int y, int width, int height, float fx, float fy, int img_w,
                                         int img_h, float &vx, float &vy, float &vz) {
  if (depth == 0 || depth > 15000) { vx = vy = vz = INFINITY; return; }
  const float milli = 0.001f;
  vx = (float)((img_w / width) * x - img_w / 2) * (float)depth / fx * milli;
  vy = (float)(img_h / 2 - (img_h / height) * y) * (float)depth / fy * milli;
  vz = (float)depth * milli;
}

// normalize(-cross(v1, v2)), glm operation order (func_geometric.inl)
__device__ inline void normal_from_vertices(float cx, float cy, float cz, float ax, float ay, float az, float bx, float by,
                                            float bz, float &nx, float &ny, float &nz) {
  const float v1x = ax - cx, v1y = ay - cy, v1z = az - cz;
  const float v2x = bx - cx, v2y = by - cy, v2z = bz - cz;
  const float crx = -(v1y * v2z - v2y * v1z), cry = -(v1z * v2x - v2z * v1x), crz = -(v1x * v2y - v2x * v1y);
  const float inv = 1.0f / sqrtf((crx * crx + cry * cry) + crz * crz);
  nx = crx * inv; ny = cry * inv; nz = crz * inv;
}

}  // namespace svoslam
