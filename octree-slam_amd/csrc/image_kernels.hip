// image_kernels.hip -- RGB-D front-end image kernels on gfx950.
// Contract = src/sensor/image_kernels.cu of the reference (every function cites
// its lines).  All of them are HBM-streaming stencils/maps; the only reuse worth
// staging is the 7x7 bilateral window, which goes through LDS.
#include <math.h>

#include "image_kernels.hpp"
#include "image_device.hpp"

namespace svoslam {

// ----------------------------------------------------------------------------
// vertex / normal maps (image_kernels.cu:24-58, 104-139)
// ----------------------------------------------------------------------------

__global__ __launch_bounds__(256) void vertex_map_kernel(const uint16_t *__restrict__ depth, float *__restrict__ vmap,
                                                         int width, int height, float fx, float fy, int img_w, int img_h,
                                                         int first_idx, int end_idx) {
  const int idx = first_idx + blockIdx.x * 256 + threadIdx.x;
  if (idx >= end_idx) return;
  float vx, vy, vz;
  vertex_from_depth(depth[idx], idx % width, idx / width, width, height, fx, fy, img_w, img_h, vx, vy, vz);
  vmap[3 * (size_t)idx] = vx; vmap[3 * (size_t)idx + 1] = vy; vmap[3 * (size_t)idx + 2] = vz;
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

__global__ __launch_bounds__(256) void normal_map_kernel(const float *__restrict__ vmap, float *__restrict__ nmap, int width,
                                                         int height) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= width * height) return;
  const int x = idx % width, y = idx / width;
  float nx, ny, nz;
  if (x == width - 1 || y == height - 1) {
    nx = ny = nz = INFINITY;
  } else {
    const float *c = vmap + 3 * (size_t)idx, *a = c + 3, *b = c + 3 * (size_t)width;
    normal_from_vertices(c[0], c[1], c[2], a[0], a[1], a[2], b[0], b[1], b[2], nx, ny, nz);
  }
  nmap[3 * (size_t)idx] = nx; nmap[3 * (size_t)idx + 1] = ny; nmap[3 * (size_t)idx + 2] = nz;
}

// Fused: one pass over the depth image writes both maps (3 depth reads per pixel,
// L1/L2-served, instead of re-reading the 12-byte vertices from HBM).
__global__ __launch_bounds__(256) void vertex_normal_kernel(const uint16_t *__restrict__ depth, float *__restrict__ vmap,
                                                            float *__restrict__ nmap, int width, int height, float fx,
                                                            float fy, int img_w, int img_h) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= width * height) return;
  const int x = idx % width, y = idx / width;
  float cx, cy, cz;
  vertex_from_depth(depth[idx], x, y, width, height, fx, fy, img_w, img_h, cx, cy, cz);
  vmap[3 * (size_t)idx] = cx; vmap[3 * (size_t)idx + 1] = cy; vmap[3 * (size_t)idx + 2] = cz;
  float nx, ny, nz;
  if (x == width - 1 || y == height - 1) {
    nx = ny = nz = INFINITY;
  } else {
    float ax, ay, az, bx, by, bz;
    vertex_from_depth(depth[idx + 1], x + 1, y, width, height, fx, fy, img_w, img_h, ax, ay, az);
    vertex_from_depth(depth[idx + width], x, y + 1, width, height, fx, fy, img_w, img_h, bx, by, bz);
    normal_from_vertices(cx, cy, cz, ax, ay, az, bx, by, bz, nx, ny, nz);
  }
  nmap[3 * (size_t)idx] = nx; nmap[3 * (size_t)idx + 1] = ny; nmap[3 * (size_t)idx + 2] = nz;
}

// ----------------------------------------------------------------------------
// bilateral filter (image_kernels.cu:142-186)
// ----------------------------------------------------------------------------
// exp() of the range/space weight.  The reference calls the __expf fast-math
// intrinsic (:170, ~2 ulp, NVIDIA-specific bits); this fixed Cody-Waite +
// degree-6 polynomial in explicit fmaf (<= 1 ulp) is evaluated identically by
// the CPU oracle, so the filtered depth agrees bit for bit.  Sub-normal results
// flush to zero as the .ftz intrinsic does.
__device__ inline float det_expf(float x) {
  if (!(x >= -87.0f)) return 0.0f;
  if (x > 88.0f) return INFINITY;
  const float kf = rintf(x * 1.44269504088896341f);
  float r = fmaf(kf, -0.693359375f, x);
  r = fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500E-4f;
  p = fmaf(p, r, 1.3981999507E-3f);
  p = fmaf(p, r, 8.3334519073E-3f);
  p = fmaf(p, r, 4.1665795894E-2f);
  p = fmaf(p, r, 1.6666665459E-1f);
  p = fmaf(p, r, 5.0000001201E-1f);
  const float e = fmaf(p, r * r, r) + 1.0f;
  return ldexpf(e, (int)kf);
}

constexpr int kBilTileW = 64, kBilTileH = 4, kBilR = 3;
constexpr int kBilLdsW = kBilTileW + 2 * kBilR, kBilLdsH = kBilTileH + 2 * kBilR;

// 64x4 output pixels per workgroup (one image row segment per wavefront: 128-byte
// coalesced stores), 70x10 halo tile staged in LDS.
__global__ __launch_bounds__(256) void bilateral_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int width,
                                                        int height, float sig_spat, float sig_dep) {
  __shared__ uint16_t tile[kBilLdsH][kBilLdsW + 2];
  const int x0 = blockIdx.x * kBilTileW - kBilR, y0 = blockIdx.y * kBilTileH - kBilR;
  for (int i = threadIdx.x; i < kBilLdsW * kBilLdsH; i += 256) {
    const int ly = i / kBilLdsW, lx = i % kBilLdsW;
    const int gx = x0 + lx, gy = y0 + ly;
    tile[ly][lx] = (gx >= 0 && gx < width && gy >= 0 && gy < height) ? in[(size_t)gy * width + gx] : (uint16_t)0;
  }
  __syncthreads();
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int x = blockIdx.x * kBilTileW + lx, y = blockIdx.y * kBilTileH + ly;
  if (x >= width || y >= height) return;
  const int value = tile[ly + kBilR][lx + kBilR];
  // Q12: window [x-3, min(x+4, W-1)) x [y-3, min(y+4, H-1)) -- last column/row excluded
  int tx = x + 4; if (tx > width - 1) tx = width - 1;
  int ty = y + 4; if (ty > height - 1) ty = height - 1;
  int cx0 = x - 3; if (cx0 < 0) cx0 = 0;
  int cy0 = y - 3; if (cy0 < 0) cy0 = 0;
  float sum1 = 0, sum2 = 0;
  for (int cy = cy0; cy < ty; ++cy)
    for (int cx = cx0; cx < tx; ++cx) {
      const int depth = tile[cy - y0][cx - x0];
      const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
      // the reference's product is a 32-bit `int` one (:168, mul.lo.s32): a difference beyond 46340 wraps it, negative for
      // 65535 next to an ordinary depth.  Written as an unsigned product converted back, so that the wrap is defined
      // behaviour here (the compiler had turned the signed form into an unsigned convert: no wrap, 60 % of a test image off)
      const unsigned diff = (unsigned)(value - depth);
      const float color2 = (float)(int)(diff * diff);
      const float weight = det_expf(-(space2 * sig_spat + color2 * sig_dep));
      sum1 = fmaf((float)depth, weight, sum1);
      sum2 += weight;
    }
  const float q = sum1 / sum2;
  const int r = (q != q) ? 0 : (int)rintf(q);  // __float2int_rn
  out[(size_t)y * width + x] = (uint16_t)r;
}

// ----------------------------------------------------------------------------
// pyramids (image_kernels.cu:236-326); width/height below are OUTPUT dims
// ----------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void subsample_depth_kernel(const T *__restrict__ in, T *__restrict__ out, int width,
                                                              int height, float sigma_depth) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= width * height) return;
  const int x = idx % width, y = idx / width;
  const int D = 5;
  const float center = (float)in[4 * (size_t)y * width + 2 * x];
  int tx = 2 * x - D / 2 + D; if (tx > 2 * width - 1) tx = 2 * width - 1;
  int ty = 2 * y - D / 2 + D; if (ty > 2 * height - 1) ty = 2 * height - 1;
  int cx0 = 2 * x - D / 2; if (cx0 < 0) cx0 = 0;
  int cy0 = 2 * y - D / 2; if (cy0 < 0) cy0 = 0;
  float sum = 0, count = 0;
  for (int cy = cy0; cy < ty; ++cy)
    for (int cx = cx0; cx < tx; ++cx) {
      const float val = (float)in[2 * (size_t)cy * width + cx];
      if (fabsf(val - center) < sigma_depth) { sum += val; ++count; }
    }
  out[idx] = (T)((count == 0) ? 0 : sum / count);
}

template <class T, int C>
__global__ __launch_bounds__(256) void subsample_kernel(const T *__restrict__ in, T *__restrict__ out, int width, int height) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= width * height) return;
  const int x = idx % width, y = idx / width;
#pragma unroll
  for (int c = 0; c < C; c++) out[(size_t)idx * C + c] = in[(4 * (size_t)y * width + 2 * x) * C + c];
}

// colorToIntensity :188-203 (Q13: the green weight multiplies .b)
__global__ __launch_bounds__(256) void color_to_intensity_kernel(const uint8_t *__restrict__ rgb, float *__restrict__ out,
                                                                 int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float r = (float)rgb[3 * (size_t)i] / 255.0f, b = (float)rgb[3 * (size_t)i + 2] / 255.0f;
  out[i] = (r * 0.299f + b * 0.587f) + b * 0.114f;
}

// gradient / difference: declared by the reference (image_kernels.h:45-49), never defined; this build's own
// specification (oracle: ora_gradient / ora_difference): Sobel 3x3 / 8 on interior pixels, (0,0) on the border
__global__ __launch_bounds__(256) void gradient_kernel(const float *__restrict__ in, float2 *__restrict__ grad, int w, int h) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  const int x = i % w, y = i / w;
  float2 g = make_float2(0.0f, 0.0f);
  if (x > 0 && y > 0 && x < w - 1 && y < h - 1) {
    const float *r0 = in + (size_t)(y - 1) * w + x, *r1 = in + (size_t)y * w + x, *r2 = in + (size_t)(y + 1) * w + x;
    const float gx = ((r0[1] - r0[-1]) + 2.0f * (r1[1] - r1[-1])) + (r2[1] - r2[-1]);
    const float gy = ((r2[-1] - r0[-1]) + 2.0f * (r2[0] - r0[0])) + (r2[1] - r0[1]);
    g = make_float2(gx * 0.125f, gy * 0.125f);
  }
  grad[i] = g;
}

__global__ __launch_bounds__(256) void difference_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                         float *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}

// ----------------------------------------------------------------------------
// rigid transforms (image_kernels.cu:206-234) ; Q19: INF * 0 -> NaN
// ----------------------------------------------------------------------------
template <bool FROM_DEVICE>
__global__ __launch_bounds__(256) void transform_kernel(float *__restrict__ v, mat4 hm, const float *__restrict__ dm, int n,
                                                        float w) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *m = FROM_DEVICE ? dm : hm.m;
  float ox, oy, oz;
  mat4_mul_point(m, v[3 * (size_t)i], v[3 * (size_t)i + 1], v[3 * (size_t)i + 2], w, ox, oy, oz);
  v[3 * (size_t)i] = ox; v[3 * (size_t)i + 1] = oy; v[3 * (size_t)i + 2] = oz;
}

// ----------------------------------------------------------------------------
// bounding box (image_kernels.cu:60-102): min/max over the points whose x and z
// are finite (Q1).  min/max are exact, so any reduction tree gives the same bits.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_partial_kernel(const float *__restrict__ pts, int n, float *__restrict__ partial) {
  __shared__ float sm[4][7];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float cnt = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    if (!finitef_(x) || !finitef_(z)) continue;
    lo[0] = fminf(x, lo[0]); lo[1] = fminf(y, lo[1]); lo[2] = fminf(z, lo[2]);
    hi[0] = fmaxf(x, hi[0]); hi[1] = fmaxf(y, hi[1]); hi[2] = fmaxf(z, hi[2]);
    cnt = 1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      lo[k] = fminf(lo[k], __shfl_down(lo[k], o));
      hi[k] = fmaxf(hi[k], __shfl_down(hi[k], o));
    }
    cnt = fmaxf(cnt, __shfl_down(cnt, o));
  }
  const unsigned wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 3; k++) { sm[wave][k] = lo[k]; sm[wave][3 + k] = hi[k]; }
    sm[wave][6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      for (int k = 0; k < 3; k++) { sm[0][k] = fminf(sm[0][k], sm[w][k]); sm[0][3 + k] = fmaxf(sm[0][3 + k], sm[w][3 + k]); }
      sm[0][6] = fmaxf(sm[0][6], sm[w][6]);
    }
    for (int k = 0; k < 7; k++) partial[blockIdx.x * 7 + k] = sm[0][k];
  }
}

// one wavefront: lane l folds rows l, l+64, ... then a shuffle tree (min/max are exact: any order)
__global__ __launch_bounds__(64) void bbox_final_kernel(const float *__restrict__ partial, int blocks, float *__restrict__ out) {
  float r[7] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, 0};
  for (int b = threadIdx.x; b < blocks; b += 64) {
#pragma unroll
    for (int k = 0; k < 3; k++) { r[k] = fminf(r[k], partial[b * 7 + k]); r[3 + k] = fmaxf(r[3 + k], partial[b * 7 + 3 + k]); }
    r[6] = fmaxf(r[6], partial[b * 7 + 6]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) { r[k] = fminf(r[k], __shfl_down(r[k], o)); r[3 + k] = fmaxf(r[3 + k], __shfl_down(r[3 + k], o)); }
    r[6] = fmaxf(r[6], __shfl_down(r[6], o));
  }
  if (threadIdx.x == 0)
    for (int k = 0; k < 7; k++) out[k] = r[k];
}

// ----------------------------------------------------------------------------
// host wrappers
// ----------------------------------------------------------------------------
#define CHECK_DIMS(w, h) if ((w) <= 0 || (h) <= 0) return SVOSLAM_ERR_INVALID_ARG

int generate_vertex_map(const uint16_t *d_depth, float *d_vertex, int w, int h, float fx, float fy, int img_w, int img_h, hipStream_t s) {
  CHECK_DIMS(w, h);
  if (!d_depth || !d_vertex) return SVOSLAM_ERR_INVALID_ARG;
  vertex_map_kernel<<<cdiv((long long)w * h, 256), 256, 0, s>>>(d_depth, d_vertex, w, h, fx, fy, img_w, img_h, 0, w * h);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int generate_vertex_map_rows(const uint16_t *d_depth, float *d_vertex, int w, int h, int first_row, int rows, float fx,
                             float fy, int img_w, int img_h, hipStream_t s) {
  CHECK_DIMS(w, h);
  if (!d_depth || !d_vertex || first_row < 0 || rows < 0 || first_row + rows > h) return SVOSLAM_ERR_INVALID_ARG;
  if (rows == 0) return SVOSLAM_OK;
  vertex_map_kernel<<<cdiv((long long)w * rows, 256), 256, 0, s>>>(d_depth, d_vertex, w, h, fx, fy, img_w, img_h, first_row * w,
                                                                  (first_row + rows) * w);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int generate_normal_map(const float *d_vertex, float *d_normal, int w, int h, hipStream_t s) {
  CHECK_DIMS(w, h);
  if (!d_vertex || !d_normal) return SVOSLAM_ERR_INVALID_ARG;
  normal_map_kernel<<<cdiv((long long)w * h, 256), 256, 0, s>>>(d_vertex, d_normal, w, h);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int generate_vertex_normal_maps(const uint16_t *d_depth, float *d_vertex, float *d_normal, int w, int h, float fx, float fy,
                                int img_w, int img_h, hipStream_t s) {
  CHECK_DIMS(w, h);
  vertex_normal_kernel<<<cdiv((long long)w * h, 256), 256, 0, s>>>(d_depth, d_vertex, d_normal, w, h, fx, fy, img_w, img_h);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int bilateral_filter(const uint16_t *d_in, uint16_t *d_out, int w, int h, hipStream_t s) {
  CHECK_DIMS(w, h);
  if (!d_in || !d_out) return SVOSLAM_ERR_INVALID_ARG;
  const float spatial = 0.5f / (4.5f * 4.5f);            // BILATERAL_SIGMA_SPATIAL, :20,182
  const float depth = (float)(0.5 / (40.0f * 40.0f));   // BILATERAL_SIGMA_DEPTH,   :19,183
  dim3 grid(cdiv(w, kBilTileW), cdiv(h, kBilTileH));
  bilateral_kernel<<<grid, 256, 0, s>>>(d_in, d_out, w, h, spatial, depth);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int subsample_depth_u16_to(const uint16_t *d_in, uint16_t *d_out, int w, int h, hipStream_t s) {
  CHECK_DIMS(w / 2, h / 2);
  subsample_depth_kernel<uint16_t><<<cdiv((long long)(w / 2) * (h / 2), 256), 256, 0, s>>>(d_in, d_out, w / 2, h / 2, 40.0f * 3.0f);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int subsample_depth_u16(uint16_t *d_data, uint16_t *d_tmp, int w, int h, hipStream_t s) {
  if (!d_data || !d_tmp) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(subsample_depth_u16_to(d_data, d_tmp, w, h, s));
  SVO_HIP(hipMemcpyAsync(d_data, d_tmp, (size_t)(w / 2) * (h / 2) * 2, hipMemcpyDeviceToDevice, s));
  return SVOSLAM_OK;
}

int subsample_depth_f32(float *d_data, float *d_tmp, int w, int h, hipStream_t s) {
  if (!d_data || !d_tmp) return SVOSLAM_ERR_INVALID_ARG;
  CHECK_DIMS(w / 2, h / 2);
  subsample_depth_kernel<float><<<cdiv((long long)(w / 2) * (h / 2), 256), 256, 0, s>>>(d_data, d_tmp, w / 2, h / 2, 40.0f * 3.0f);
  SVO_LAUNCH_CHECK();
  SVO_HIP(hipMemcpyAsync(d_data, d_tmp, (size_t)(w / 2) * (h / 2) * 4, hipMemcpyDeviceToDevice, s));
  return SVOSLAM_OK;
}

int subsample_f32(float *d_data, float *d_tmp, int w, int h, hipStream_t s) {
  if (!d_data || !d_tmp) return SVOSLAM_ERR_INVALID_ARG;
  CHECK_DIMS(w / 2, h / 2);
  subsample_kernel<float, 1><<<cdiv((long long)(w / 2) * (h / 2), 256), 256, 0, s>>>(d_data, d_tmp, w / 2, h / 2);
  SVO_LAUNCH_CHECK();
  SVO_HIP(hipMemcpyAsync(d_data, d_tmp, (size_t)(w / 2) * (h / 2) * 4, hipMemcpyDeviceToDevice, s));
  return SVOSLAM_OK;
}

int subsample_rgb8(uint8_t *d_data, uint8_t *d_tmp, int w, int h, hipStream_t s) {
  if (!d_data || !d_tmp) return SVOSLAM_ERR_INVALID_ARG;
  CHECK_DIMS(w / 2, h / 2);
  subsample_kernel<uint8_t, 3><<<cdiv((long long)(w / 2) * (h / 2), 256), 256, 0, s>>>(d_data, d_tmp, w / 2, h / 2);
  SVO_LAUNCH_CHECK();
  SVO_HIP(hipMemcpyAsync(d_data, d_tmp, (size_t)(w / 2) * (h / 2) * 3, hipMemcpyDeviceToDevice, s));
  return SVOSLAM_OK;
}

int color_to_intensity(const uint8_t *d_rgb, float *d_out, int n, hipStream_t s) {
  if (!d_rgb || !d_out || n < 0) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  color_to_intensity_kernel<<<cdiv(n, 256), 256, 0, s>>>(d_rgb, d_out, n);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int gradient(const float *d_in, float *d_grad2, int w, int h, hipStream_t s) {
  if (!d_in || !d_grad2) return SVOSLAM_ERR_INVALID_ARG;
  CHECK_DIMS(w, h);
  gradient_kernel<<<cdiv((long long)w * h, 256), 256, 0, s>>>(d_in, reinterpret_cast<float2 *>(d_grad2), w, h);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int difference(const float *d_in1, const float *d_in2, float *d_out, int n, hipStream_t s) {
  if (!d_in1 || !d_in2 || !d_out || n < 0) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  difference_kernel<<<cdiv(n, 256), 256, 0, s>>>(d_in1, d_in2, d_out, n);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

static int transform_host(float *d_v, const float trans[16], int n, float w, hipStream_t s) {
  if (!d_v || !trans || n < 0) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  mat4 m;
  for (int i = 0; i < 16; i++) m.m[i] = trans[i];
  transform_kernel<false><<<cdiv(n, 256), 256, 0, s>>>(d_v, m, nullptr, n, w);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}
int transform_vertex_map(float *d_v, const float trans[16], int n, hipStream_t s) { return transform_host(d_v, trans, n, 1.0f, s); }
int transform_normal_map(float *d_v, const float trans[16], int n, hipStream_t s) { return transform_host(d_v, trans, n, 0.0f, s); }

int transform_vertex_map_dmat(float *d_v, const float *d_trans, int n, hipStream_t s) {
  if (!d_v || !d_trans || n < 0) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  mat4 unused = {};
  transform_kernel<true><<<cdiv(n, 256), 256, 0, s>>>(d_v, unused, d_trans, n, 1.0f);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int point_cloud_bbox_device(svoslam::DeviceBuffer &scratch, const float *d_points, int n, float *d_out7, hipStream_t s) {
  if (!d_points || !d_out7 || n <= 0) return SVOSLAM_ERR_INVALID_ARG;
  int blocks = (int)cdiv(n, 256);
  if (blocks > 1024) blocks = 1024;
  SVO_TRY(scratch.reserve((size_t)(1024 + 1) * 7 * 4));
  float *partial = scratch.as<float>() + 7;
  bbox_partial_kernel<<<blocks, 256, 0, s>>>(d_points, n, partial);
  bbox_final_kernel<<<1, 64, 0, s>>>(partial, blocks, d_out7);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int point_cloud_bbox(svoslam::DeviceBuffer &scratch, const float *d_points, int n, float h_bbox0[3], float h_bbox1[3], hipStream_t s) {
  if (!d_points || !h_bbox0 || !h_bbox1 || n < 0) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  SVO_TRY(scratch.reserve((size_t)(1024 + 1) * 7 * 4));
  float *result = scratch.as<float>();
  SVO_TRY(point_cloud_bbox_device(scratch, d_points, n, result, s));
  float r[7];
  SVO_HIP(hipMemcpyAsync(r, result, sizeof(r), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  if (r[6] == 0.0f) return SVOSLAM_OK;  // no valid point: bbox unchanged
  // min_vec3 / max_vec3 (:60-94): a zero vector on the left means "unset"
  const bool unset0 = h_bbox0[0] == 0.0f && h_bbox0[1] == 0.0f && h_bbox0[2] == 0.0f;
  const bool unset1 = h_bbox1[0] == 0.0f && h_bbox1[1] == 0.0f && h_bbox1[2] == 0.0f;
  for (int k = 0; k < 3; k++) {
    h_bbox0[k] = unset0 ? r[k] : fminf(r[k], h_bbox0[k]);
    h_bbox1[k] = unset1 ? r[3 + k] : fmaxf(r[3 + k], h_bbox1[k]);
  }
  return SVOSLAM_OK;
}

}  // namespace svoslam
