// cone_trace.hip -- per-pixel octree ray march / voxel cone trace on gfx950.
//
// Contract = rendering::coneTraceSVO (src/rendering/cone_tracing_kernels.cu:
// 24-198): same rays, same LOD rule, same step rule, same byte arithmetic, into
// an offscreen uchar4 buffer in ray-index order.
//
// Organisation: the reference launches ONE kernel + ONE thrust::remove_if + a
// D2H count per march step and round-trips every ray's state through HBM
// (rays[], ind[], pos[]).  Here one launch marches every ray to retirement with
// its state in registers; a wavefront is an 8x8 pixel tile so the 64 rays walk
// neighbouring nodes (one L1/L2 line serves most of the wave at the top levels).
// Traffic left: (levels descended + 1) node reads per step + 4 B per pixel.
#include "cone_trace.hpp"

namespace svoslam {

__device__ constexpr float kMaxRange = 10.0f;   // cone_tracing_kernels.cu:24
__device__ constexpr float kStartDist = 0.002f; // :27
constexpr int kMaxSteps = 1 << 20;              // guard only; the reference loops until retirement
constexpr int kPathCache = 18;                  // levels kept in the per-ray path cache (depth 16 + Q4 level + 1)

// ceil(log(q)/log(2)) of :69 from the binary32 exponent (exact; the reference's
// float log() can be one level off within an ulp of a power of two)
__device__ inline int ceil_log2_pos(float q) {
  const uint32_t u = f2bits(q);
  if ((int32_t)u <= 0) return 0;
  const int ex = (int)(u >> 23);
  const uint32_t man = u & 0x7FFFFFu;
  if (ex == 255) return 128;
  if (ex == 0) {
    const int hb = 31 - __clz((int)man);
    return (hb - 149) + ((man & (man - 1)) != 0);
  }
  return (ex - 127) + (man != 0);
}

// float -> uint8_t of :110-112,133-135: cvt.rzi.u32.f32 (negative / NaN -> 0,
// saturating), then the low byte
__device__ inline uint32_t f2u8(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFu;
  return (uint32_t)f & 0xFFu;
}

__device__ inline float length3(float x, float y, float z) { return sqrtf(dot3(x, y, z, x, y, z)); }

struct TraceParams {
  float origin[3], x_dir[3], y_dir[3];
  float center[3];
  float size, pix_scale;
  int width, height, mode;
  int row_first, row_end;  // rows [row_first, row_end) are traced (row band of a multi-GPU tile split)
};

__global__ __launch_bounds__(256) void cone_trace_kernel(uchar4 *__restrict__ pos, const uint32_t *__restrict__ octree,
                                                         TraceParams P, unsigned long long *__restrict__ counters) {
  // 16x16 pixel workgroup, one 8x8 tile per wavefront
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const int px = blockIdx.x * 16 + (int)(wave & 1u) * 8 + (int)(lane & 7u);
  const int py = P.row_first + blockIdx.y * 16 + (int)(wave >> 1) * 8 + (int)(lane >> 3);
  unsigned long long my_steps = 0, my_levels = 0;
  if (px < P.width && py < P.row_end) {
    const int idx = py * P.width + px;
    // createRays :29-51 (hard-coded Kinect focal lengths; fov is unused there)
    const float res_x = (float)P.width, res_y = (float)P.height;
    const float magx = ((float)px - res_x / 2.0f) / 532.57f;
    const float magy = ((float)py - res_y / 2.0f) / 531.54f;
    // cross(x_dir, -y_dir)
    const float nyx = -P.y_dir[0], nyy = -P.y_dir[1], nyz = -P.y_dir[2];
    const float fx = P.x_dir[1] * nyz - nyy * P.x_dir[2];
    const float fy = P.x_dir[2] * nyx - nyz * P.x_dir[0];
    const float fz = P.x_dir[0] * nyy - nyx * P.x_dir[1];
    const float dx = ((magx * P.x_dir[0]) + (magy * P.y_dir[0])) + fx;
    const float dy = ((magx * P.x_dir[1]) + (magy * P.y_dir[1])) + fy;
    const float dz = ((magx * P.x_dir[2]) + (magy * P.y_dir[2])) + fz;
    const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
    float rx = kStartDist * (dx * inv), ry = kStartDist * (dy * inv), rz = kStartDist * (dz * inv);
    uint32_t vx = 0, vy = 0, vz = 0, vw = 0;  // local uchar4 pixel
    uint32_t out = 0;
    // Path cache: octant, word0 and word1 of the node visited at each level of the previous
    // descent.  The pool is read-only while rendering, so when the new sample takes the same
    // octant at level i (same comparisons against the same centres, recomputed in registers) the
    // node is the same and its words are reused without a load; the first differing level
    // invalidates the rest.  Consecutive samples of a ray share most of their ancestors, which
    // removes most of the dependent loads of the walk; the comparisons and centre updates are
    // executed exactly as in the reference, so the visited nodes are identical.
    uint32_t c_w0[kPathCache], c_w1[kPathCache];
    unsigned long long c_oct = 0;  // 3 bits per level
    int c_len = 0;
    const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
    for (int step = 0; step < kMaxSteps; step++) {
      my_steps++;
      // Q9: the reference re-reads pos[index], which stays 0 until retirement
      if (P.mode == SVOSLAM_RENDER_REFERENCE) vx = vy = vz = vw = 0;
      const float tx = P.origin[0] + rx, ty = P.origin[1] + ry, tz = P.origin[2] + rz;
      const float ray_len = length3(rx, ry, rz);
      const float pix_size = ray_len * P.pix_scale;
      int depth = ceil_log2_pos(P.size / pix_size);
      uint32_t node_idx = 0, child_idx = 0;
      uint32_t oct_val = 0;
      bool have_val = false;
      float temp_size = P.size, cx = P.center[0], cy = P.center[1], cz = P.center[2];
      bool stopped = false;
#pragma unroll
      for (int i = 0; i < kPathCache; i++) {
        if (!stopped && i < depth) {
          const bool x = tx > cx, y = ty > cy, z = tz > cz;
          const uint32_t oct = (uint32_t)(x + 2 * y + 4 * z);
          node_idx = child_idx + oct;
          my_levels++;
          uint32_t w0, w1;
          if (i < c_len && ((uint32_t)(c_oct >> (3 * i)) & 7u) == oct) {
            w0 = c_w0[i]; w1 = c_w1[i];
          } else {
            const uint2 nd = nodes[node_idx];
            w0 = nd.x; w1 = nd.y;
            c_w0[i] = w0; c_w1[i] = w1;
            c_oct = (c_oct & ~(7ull << (3 * i))) | ((unsigned long long)oct << (3 * i));
            c_len = i + 1;
          }
          oct_val = w1; have_val = true;
          if (!(w0 & kFlag)) { depth = i + 1; stopped = true; }
          else {
            child_idx = w0 & kMask;
            temp_size /= 2.0f;
            cx += temp_size * (x ? 1 : -1);
            cy += temp_size * (y ? 1 : -1);
            cz += temp_size * (z ? 1 : -1);
          }
        }
      }
      if (!stopped) {
        // deeper than the cache (cannot happen for pools of depth <= 16 + the Q4 level): plain walk
        for (int i = kPathCache; i < depth; i++) {
          const bool x = tx > cx, y = ty > cy, z = tz > cz;
          node_idx = child_idx + (uint32_t)(x + 2 * y + 4 * z);
          my_levels++;
          const uint2 nd = nodes[node_idx];
          oct_val = nd.y; have_val = true;
          if (!(nd.x & kFlag)) { depth = i + 1; break; }
          child_idx = nd.x & kMask;
          temp_size /= 2.0f;
          cx += temp_size * (x ? 1 : -1);
          cy += temp_size * (y ? 1 : -1);
          cz += temp_size * (z ? 1 : -1);
        }
      }
      if (!have_val) oct_val = octree[1];  // depth <= 0: the reference reads node 0 (:107 with node_idx = 0)
      // :108 max(0, unsigned) is the (int, unsigned) overload: no clamp, alpha = A - 127 signed
      const int alpha = (int)((oct_val >> 24) - 127u);
      const float af = (float)alpha / 127.0f;
      vx = (vx + f2u8(af * (float)(oct_val & 0xFF))) & 0xFFu;
      vy = (vy + f2u8(af * (float)((oct_val >> 8) & 0xFF))) & 0xFFu;
      vz = (vz + f2u8(af * (float)((oct_val >> 16) & 0xFF))) & 0xFFu;
      bool retired = false;
      if ((int)vw + alpha < 127) {
        vw = (uint32_t)((int)vw + alpha) & 0xFFu;
      } else {
        vw = 255u;
        retired = true;
      }
      if (!retired) {
        const float new_dist = P.size / ldexpf(1.0f, depth);  // pow(2.0f, depth) :126
        const float s = (ray_len + new_dist) / ray_len;
        rx *= s; ry *= s; rz *= s;
        if (length3(rx, ry, rz) > kMaxRange) {
          const float sc = 127.0f / (float)vw;
          vx = f2u8((float)vx * sc);
          vy = f2u8((float)vy * sc);
          vz = f2u8((float)vz * sc);
          vw = 255u;
          retired = true;
        }
      }
      if (retired) { out = vx | (vy << 8) | (vz << 16) | (vw << 24); break; }
    }
    uchar4 o;
    o.x = (unsigned char)(out & 0xFF); o.y = (unsigned char)((out >> 8) & 0xFF);
    o.z = (unsigned char)((out >> 16) & 0xFF); o.w = (unsigned char)(out >> 24);
    pos[idx] = o;
  }
  if (counters) {
    // wave-level sums, one atomic pair per wavefront
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      my_steps += __shfl_down(my_steps, o);
      my_levels += __shfl_down(my_levels, o);
    }
    if (lane == 0) {
      atomicAdd(&counters[0], my_steps);
      atomicAdd(&counters[1], my_levels);
    }
  }
}

// ---- host side: glm::inverse(view) products of :161-167, pix_scale of :171 ----
static void mat4_inverse_host(const float *m, float *out);  // below

int cone_trace_svo(uint8_t *d_pos, int width, int height, int row_first, int rows, float fov, const float view[16],
                   const uint32_t *d_octree, const float center[3], float size, int mode, unsigned long long *d_steps,
                   hipStream_t stream) {
  if (!d_pos || !view || !d_octree || !center || width <= 0 || height <= 0) return SVOSLAM_ERR_INVALID_ARG;
  if (row_first < 0 || rows < 0 || row_first + rows > height) return SVOSLAM_ERR_INVALID_ARG;
  if (rows == 0) return SVOSLAM_OK;
  if (mode != SVOSLAM_RENDER_REFERENCE && mode != SVOSLAM_RENDER_CARRY) return SVOSLAM_ERR_INVALID_ARG;
  float inv[16];
  mat4_inverse_host(view, inv);
  TraceParams P;
  mat4_mul_point(inv, 0.0f, 0.0f, 0.0f, 1.0f, P.origin[0], P.origin[1], P.origin[2]);
  mat4_mul_point(inv, -1.0f, 0.0f, 0.0f, 0.0f, P.x_dir[0], P.x_dir[1], P.x_dir[2]);
  mat4_mul_point(inv, 0.0f, -1.0f, 0.0f, 0.0f, P.y_dir[0], P.y_dir[1], P.y_dir[2]);
  for (int k = 0; k < 3; k++) P.center[k] = center[k];
  P.size = size;
  P.pix_scale = tanf(fov * 3.14159f / 180.0f) / (float)height;
  P.width = width; P.height = height; P.mode = mode;
  P.row_first = row_first; P.row_end = row_first + rows;
  dim3 grid(cdiv(width, 16), cdiv(rows, 16));
  cone_trace_kernel<<<grid, 256, 0, stream>>>(reinterpret_cast<uchar4 *>(d_pos), d_octree, P, d_steps);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

// glm compute_inverse<tmat4x4> (external/include/glm/detail/type_mat4x4.inl:477-534),
// cofactor expansion with glm's operation order
static void mat4_inverse_host(const float *m, float *out) {
#define E(c, r) m[4 * (c) + (r)]
  const float c00 = E(2,2) * E(3,3) - E(3,2) * E(2,3), c02 = E(1,2) * E(3,3) - E(3,2) * E(1,3), c03 = E(1,2) * E(2,3) - E(2,2) * E(1,3);
  const float c04 = E(2,1) * E(3,3) - E(3,1) * E(2,3), c06 = E(1,1) * E(3,3) - E(3,1) * E(1,3), c07 = E(1,1) * E(2,3) - E(2,1) * E(1,3);
  const float c08 = E(2,1) * E(3,2) - E(3,1) * E(2,2), c10 = E(1,1) * E(3,2) - E(3,1) * E(1,2), c11 = E(1,1) * E(2,2) - E(2,1) * E(1,2);
  const float c12 = E(2,0) * E(3,3) - E(3,0) * E(2,3), c14 = E(1,0) * E(3,3) - E(3,0) * E(1,3), c15 = E(1,0) * E(2,3) - E(2,0) * E(1,3);
  const float c16 = E(2,0) * E(3,2) - E(3,0) * E(2,2), c18 = E(1,0) * E(3,2) - E(3,0) * E(1,2), c19 = E(1,0) * E(2,2) - E(2,0) * E(1,2);
  const float c20 = E(2,0) * E(3,1) - E(3,0) * E(2,1), c22 = E(1,0) * E(3,1) - E(3,0) * E(1,1), c23 = E(1,0) * E(2,1) - E(2,0) * E(1,1);
  const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11};
  const float F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
  const float V0[4] = {E(1,0), E(0,0), E(0,0), E(0,0)}, V1[4] = {E(1,1), E(0,1), E(0,1), E(0,1)};
  const float V2[4] = {E(1,2), E(0,2), E(0,2), E(0,2)}, V3[4] = {E(1,3), E(0,3), E(0,3), E(0,3)};
  const float SA[4] = {+1, -1, +1, -1}, SB[4] = {-1, +1, -1, +1};
  float inv[16];
  for (int i = 0; i < 4; i++) {
    inv[0 + i] = ((V1[i] * F0[i] - V2[i] * F1[i]) + V3[i] * F2[i]) * SA[i];
    inv[4 + i] = ((V0[i] * F0[i] - V2[i] * F3[i]) + V3[i] * F4[i]) * SB[i];
    inv[8 + i] = ((V0[i] * F1[i] - V1[i] * F3[i]) + V3[i] * F5[i]) * SA[i];
    inv[12 + i] = ((V0[i] * F2[i] - V1[i] * F4[i]) + V2[i] * F5[i]) * SB[i];
  }
  const float d0 = E(0,0) * inv[0], d1 = E(0,1) * inv[4], d2 = E(0,2) * inv[8], d3 = E(0,3) * inv[12];
  const float ood = 1.0f / ((d0 + d1) + (d2 + d3));
  for (int i = 0; i < 16; i++) out[i] = inv[i] * ood;
#undef E
}

}  // namespace svoslam
