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

}  // namespace svoslam
