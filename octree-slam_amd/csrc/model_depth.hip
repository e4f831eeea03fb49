// model_depth.hip -- the map ray-cast into a DEPTH image in the sensor's pixel grid: the model a frame-to-model ICP tracks
// against (second half).  OWN SPECIFICATION: the reference has no such function -- rgbd_camera.cpp:185 only
// leaves the TODO "ICP should not swap, as last_frame should be updated by a different function".  The specification is
// stated with oracle/svoslam_oracle.c ora_raycast_model_depth, which this kernel follows operation by operation:
//   pixel (x, y) looks along d = ((x - w/2) / fx, (h/2 - y) / fy, 1) of the sensor frame (the direction generateVertexMap
//   gives the pixel, image_kernels.cu:24-58), carried into the map by cam_to_world (operator*(mat4, vec4), as main.cpp:40
//   applies it to the vertex map); the ray is marched as coneTrace marches (cone_tracing_kernels.cu:53-146: START_DIST,
//   LOD = ceil(log2(size / (ray length x pixel scale))), pixel scale 1 / fy, descent to the first childless node or the
//   LOD, step = size / 2^level, MAX_RANGE) and stops at the first sample whose node carries A >= 254 -- what retires a ray
//   there; rint(1000 x ray length / |d|) is the pixel (uint16 millimetres along the optical axis), 0 = nothing met.
// One ray per lane, the descent from the root with the reference's own centre arithmetic (no level grid, no bricks): 0.19 ms
// for a 640x480 image of the saturated first frame of config 3 (5.5 M steps).  Resuming a step's descent at the level-8 node of
// the previous step (faces kept as lo < t <= hi per axis; exact, bit-equal) was built and measured: 0.21 ms -- the upper levels
// are L1 / L2 hits whose loads overlap, the twelve extra registers and six compares cost more.  The mode is opt-in.
#include "model_depth.hpp"

namespace svoslam {

namespace {

constexpr float kModelMaxRange = 10.0f;    // cone_tracing_kernels.cu:24
constexpr float kModelStartDist = 0.002f;  // :27
constexpr int kModelMaxSteps = 1 << 20;    // guard only
constexpr uint32_t kFlagBit = 0x40000000u, kLinkMask = 0x3FFFFFFFu;

// ceil(log2(q)) from the bits of the rounded quotient (the oracle's ceil_log2_pos)
__device__ inline int ceil_log2_bits(float q) {
  const uint32_t u = f2bits(q);
  if ((int32_t)u <= 0) return 0;
  const int ex = (int)(u >> 23);
  const uint32_t man = u & 0x7FFFFFu;
  if (ex == 255) return 128;
  if (ex == 0) { const int hb = 31 - __clz((int)man); return (hb - 149) + ((man & (man - 1)) != 0); }
  return (ex - 127) + (man != 0);
}

struct ModelParams {
  float m[16];       // cam_to_world when d_m == nullptr
  float cx, cy, cz, size, fx, fy;
  int w, h;
};

__global__ __launch_bounds__(256) void model_depth_kernel(uint16_t *__restrict__ depth_out, const uint32_t *__restrict__ octree,
                                                          const float *__restrict__ d_m, ModelParams P,
                                                          unsigned long long *__restrict__ d_steps) {
  // 16 x 16 pixel tiles: neighbouring rays walk neighbouring nodes
  const int tiles_x = (P.w + 15) >> 4;
  const int tx = (int)(blockIdx.x % (unsigned)tiles_x), ty = (int)(blockIdx.x / (unsigned)tiles_x);
  const int px = tx * 16 + (int)(threadIdx.x & 15u), py = ty * 16 + (int)(threadIdx.x >> 4);
  unsigned long long steps = 0ull;
  if (px < P.w && py < P.h) {
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = d_m ? d_m[i] : P.m[i];
    float ox, oy, oz, qx, qy, qz;
    mat4_mul_point(m, 0.0f, 0.0f, 0.0f, 1.0f, ox, oy, oz);
    const float dcx = (float)(px - P.w / 2) / P.fx, dcy = (float)(P.h / 2 - py) / P.fy;
    mat4_mul_point(m, dcx, dcy, 1.0f, 1.0f, qx, qy, qz);
    const float dx = qx - ox, dy = qy - oy, dz = qz - oz;
    const float sqr = dot3(dx, dy, dz, dx, dy, dz);
    const float len_d = sqrtf(sqr);
    uint16_t out = 0;
    if (finitef_(len_d) && len_d > 0.0f) {
      const float inv = 1.0f / len_d;  // normalize(): v * (1 / sqrt(dot(v, v)))
      float rx = kModelStartDist * (dx * inv), ry = kModelStartDist * (dy * inv), rz = kModelStartDist * (dz * inv);
      const float pix_scale = 1.0f / P.fy;
      for (int step = 0; step < kModelMaxSteps; step++) {
        steps++;
        const float tgx = ox + rx, tgy = oy + ry, tgz = oz + rz;
        const float ray_len = sqrtf(dot3(rx, ry, rz, rx, ry, rz));
        const float pix_size = ray_len * pix_scale;
        int depth = ceil_log2_bits(P.size / pix_size);
        uint32_t node = 0u, child = 0u;
        float ts = P.size, cx = P.cx, cy = P.cy, cz = P.cz;
        for (int i = 0; i < depth; i++) {
          const bool x = tgx > cx, y = tgy > cy, z = tgz > cz;
          node = child + ((x ? 1u : 0u) + (y ? 2u : 0u) + (z ? 4u : 0u));
          const uint32_t w0 = octree[2 * (size_t)node];
          if (!(w0 & kFlagBit)) { depth = i + 1; break; }
          child = w0 & kLinkMask;
          ts = ts / 2.0f;
          cx += x ? ts : -ts;  // temp_size * (+-1): exact
          cy += y ? ts : -ts;
          cz += z ? ts : -ts;
        }
        const uint32_t val = octree[2 * (size_t)node + 1];
        if ((val >> 24) >= 254u) {
          const float mm = (ray_len / len_d) * 1000.0f;
          if (mm < 65535.0f) out = (uint16_t)rintf(mm);
          break;
        }
        const float new_dist = P.size / ldexpf(1.0f, depth);
        const float s = (ray_len + new_dist) / ray_len;
        rx *= s; ry *= s; rz *= s;
        if (sqrtf(dot3(rx, ry, rz, rx, ry, rz)) > kModelMaxRange) break;
      }
    }
    depth_out[(size_t)py * P.w + px] = out;
  }
  if (d_steps) {  // one atomic per wavefront
    for (int off = 32; off > 0; off >>= 1) steps += __shfl_down(steps, off);
    if ((threadIdx.x & 63u) == 0u && steps) atomicAdd(d_steps, steps);
  }
}

}  // namespace

int raycast_model_depth(uint16_t *d_depth, int width, int height, float fx, float fy, const float *cam_to_world,
                        const float *d_cam_to_world, const uint32_t *d_octree, const float center[3], float size,
                        unsigned long long *d_steps, hipStream_t stream) {
  if (!d_depth || !d_octree || !center || width <= 0 || height <= 0) return SVOSLAM_ERR_INVALID_ARG;
  if ((cam_to_world == nullptr) == (d_cam_to_world == nullptr)) return SVOSLAM_ERR_INVALID_ARG;  // exactly one of the two
  if (!(fx > 0.0f) || !(fy > 0.0f) || !(size > 0.0f)) return SVOSLAM_ERR_INVALID_ARG;
  ModelParams P;
  for (int i = 0; i < 16; i++) P.m[i] = cam_to_world ? cam_to_world[i] : 0.0f;
  P.cx = center[0]; P.cy = center[1]; P.cz = center[2]; P.size = size; P.fx = fx; P.fy = fy; P.w = width; P.h = height;
  const unsigned tiles = (unsigned)(((width + 15) >> 4) * ((height + 15) >> 4));
  model_depth_kernel<<<tiles, 256, 0, stream>>>(d_depth, d_octree, d_cam_to_world, P, d_steps);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

}  // namespace svoslam
