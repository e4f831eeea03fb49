// cone_trace.hip -- per-pixel octree ray march / voxel cone trace on gfx950.
//
// Contract = rendering::coneTraceSVO (src/rendering/cone_tracing_kernels.cu:
// 24-198): same rays, same LOD rule, same step rule, same byte arithmetic, into
// an offscreen uchar4 buffer in ray-index order.
//
// Organisation: the reference launches ONE kernel + ONE thrust::remove_if + a
// D2H count per march step and round-trips every ray's state through HBM
// (rays[], ind[], pos[]).  Here one launch marches every ray to retirement with
// its state in registers; a wavefront is an 8x8 pixel tile.
//
// What bounds this kernel (measured, scratch/ray_steps.py): its run time equals the
// critical path of its SLOWEST ray -- time tracks max(steps per ray), not the ray
// count (half the rows: same time) -- i.e. steps x (dependent loads + dependent ALU)
// per step.  Hence (1) the first kTop levels of every walk come from a dense "top
// tree" whose addresses need no loads (one latency instead of kTop), (2) the
// per-step arithmetic is reduced to one division and one square root without
// changing a single result bit (see step_lod / the notes in the loop).
#include "cone_trace.hpp"
#include "workspace.hpp"

namespace svoslam {

__device__ constexpr float kMaxRange = 10.0f;   // cone_tracing_kernels.cu:24
__device__ constexpr float kStartDist = 0.002f; // :27
constexpr int kMaxSteps = 1 << 20;              // guard only; the reference loops until retirement

// ceil(log(size/pix)/log(2)) of :69 WITHOUT the division.  For positive normal binary32 a, b with
// a = ma*2^ea, b = mb*2^eb (ma, mb in [1,2)): a/b = (ma/mb)*2^(ea-eb).  If ma == mb the quotient
// is the power of two 2^(ea-eb); if ma > mb it lies strictly inside (2^(ea-eb), 2^(ea-eb+1)) and,
// since ma/mb >= 1 + 2^-24 * (1+eps), it never rounds down to the power of two; if ma < mb it lies
// in (2^(ea-eb-1), 2^(ea-eb)] after rounding.  ceil(log2(fl(a/b))) is therefore
// (ea - eb) + (ma > mb), identical to evaluating ceil_log2 on the rounded quotient.
// (The reference's float log() may be one level off within an ulp of a power of two; the oracle
// uses the exact value, as here.)
__device__ inline int step_lod(float size, float pix_size) {
  const uint32_t ua = f2bits(size), ub = f2bits(pix_size);
  const int ea = (int)((ua >> 23) & 0xFF), eb = (int)((ub >> 23) & 0xFF);
  if (ea == 0 || eb == 0 || ea == 255 || eb == 255 || (int32_t)ua < 0 || (int32_t)ub < 0 || ea - eb > 120 || eb - ea > 120) {
    // zero / subnormal / inf / nan / negative operands: evaluate the quotient itself (never on real data)
    const float q = size / pix_size;
    const uint32_t u = f2bits(q);
    if ((int32_t)u <= 0) return 0;
    const int ex = (int)(u >> 23);
    const uint32_t man = u & 0x7FFFFFu;
    if (ex == 255) return 128;
    if (ex == 0) { const int hb = 31 - __clz((int)man); return (hb - 149) + ((man & (man - 1)) != 0); }
    return (ex - 127) + (man != 0);
  }
  return (ea - eb) + ((ua & 0x7FFFFFu) > (ub & 0x7FFFFFu) ? 1 : 0);
}

// float -> uint8_t of :110-112,133-135: cvt.rzi.u32.f32 (negative / NaN -> 0, saturating), then the
// low byte.  v_cvt_u32_f32 has exactly these semantics, so no range branches are needed.
__device__ inline uint32_t f2u8(float f) { return __float2uint_rz(f) & 0xFFu; }

__device__ inline float length3(float x, float y, float z) { return sqrtf(dot3(x, y, z, x, y, z)); }

// ---- top tree --------------------------------------------------------------
// Levels 1..kTop of the pool mirrored as a dense, heap-ordered array: entry
// (d, path) = both words of the node reached by the octant path `path` of length d
// (zeros if that node does not exist).  The octant bits of a sample depend only on
// float comparisons against centres recomputed in registers, so the addresses of
// its first kTop nodes are known WITHOUT loading anything: the kTop loads are
// issued together (one memory latency) instead of kTop dependent round trips.
// Rebuilt from the pool at the start of every render (the pool is const during it).
constexpr int kTop = 6;
__host__ __device__ constexpr int top_offset(int d) { return d <= 1 ? 0 : 8 * ((1 << (3 * (d - 1))) - 1) / 7; }  // entries before level d
constexpr int kTopEntries = top_offset(kTop + 1);  // 8 + 64 + ... + 8^kTop = 299592

__global__ __launch_bounds__(256) void build_top_tree_kernel(const uint32_t *__restrict__ octree, uint2 *__restrict__ top,
                                                            float *__restrict__ alpha_lut) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  // (float)alpha / 127.0f of :110-112 for alpha = A - 127 in [-127, 128]: 256 IEEE quotients, computed once
  if (e < 256) alpha_lut[e] = (float)(e - 127) / 127.0f;
  if (e >= kTopEntries) return;
  int d = 1;
  while (d < kTop && e >= top_offset(d + 1)) d++;
  const uint32_t path = (uint32_t)(e - top_offset(d));
  const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
  uint32_t base = 0;
  uint2 nd = make_uint2(0u, 0u);
  for (int l = 1; l <= d; l++) {
    nd = nodes[base + ((path >> (3 * (d - l))) & 7u)];
    if (l < d) {
      if (!(nd.x & kFlag)) { nd = make_uint2(0u, 0u); break; }  // path does not exist below an unsplit node
      base = nd.x & kMask;
    }
  }
  top[e] = nd;
}

struct TraceParams {
  float origin[3], x_dir[3], y_dir[3];
  float center[3];
  float size, pix_scale;
  int width, height, mode;
  int row_first, row_end;  // rows [row_first, row_end) are traced (row band of a multi-GPU tile split)
};

// CARRY = false: SVOSLAM_RENDER_REFERENCE.  The reference re-reads pos[index] every step and that
// pixel stays 0 until the ray retires (Q9), so a sample's colour matters only on the step that
// retires the ray: the march needs alpha alone and the colour is formed once, after the loop.
// CARRY = true: the local pixel is carried across steps.
template <bool CARRY>
__global__ __launch_bounds__(256) void cone_trace_kernel(uchar4 *__restrict__ pos, const uint32_t *__restrict__ octree,
                                                         const uint2 *__restrict__ top, const float *__restrict__ alpha_lut_g,
                                                         TraceParams P, unsigned long long *__restrict__ counters) {
  __shared__ float alpha_lut[256];
  alpha_lut[threadIdx.x] = alpha_lut_g[threadIdx.x];
  __syncthreads();
  // 16x16 pixel workgroup, one 8x8 tile per wavefront
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const int px = blockIdx.x * 16 + (int)(wave & 1u) * 8 + (int)(lane & 7u);
  const int py = P.row_first + blockIdx.y * 16 + (int)(wave >> 1) * 8 + (int)(lane >> 3);
  uint32_t my_steps = 0, my_levels = 0;
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
    uint32_t vx = 0, vy = 0, vz = 0, vw = 0;  // local uchar4 pixel (CARRY only)
    const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
    // glm::length(ray) is evaluated at :67 and again at :131 on the advanced ray; the second value is
    // the first value of the next step (same operands), so it is computed once and carried.
    float ray_len = length3(rx, ry, rz);
    uint32_t oct_val = 0;
    int alpha = 0;
    bool range_exit = false;
    for (int step = 0; step < kMaxSteps; step++) {
      my_steps++;
      const float tx = P.origin[0] + rx, ty = P.origin[1] + ry, tz = P.origin[2] + rz;
      const float pix_size = ray_len * P.pix_scale;
      int depth = step_lod(P.size, pix_size);
      float temp_size = P.size, cx = P.center[0], cy = P.center[1], cz = P.center[2];
      // levels 1..kTop: octant bits by comparisons only (no loads), then kTop independent loads from the
      // top tree, all issued unconditionally (levels beyond the LOD depth read valid, unused entries)
      uint2 e[kTop];
      {
        uint32_t path = 0;
#pragma unroll
        for (int i = 0; i < kTop; i++) {
          const bool x = tx > cx, y = ty > cy, z = tz > cz;
          path = path * 8u + (uint32_t)(x + 2 * y + 4 * z);
          e[i] = top[top_offset(i + 1) + path];
          temp_size *= 0.5f;  // "/= 2.0f" (:97): exact either way
          cx += temp_size * (x ? 1 : -1);
          cy += temp_size * (y ? 1 : -1);
          cz += temp_size * (z ? 1 : -1);
        }
      }
      // first level (in walk order) whose node has no children, among the nb = min(depth, kTop) levels
      // the reference would visit; branch-free select chain
      const int nb = depth < kTop ? depth : kTop;
      int stop = kTop;
#pragma unroll
      for (int i = kTop - 1; i >= 0; i--) stop = (!(e[i].x & kFlag)) ? i : stop;
      const bool stopped = stop < nb;
      const int last = stopped ? stop : nb - 1;  // index of the last visited level (-1 if nb <= 0)
      my_levels += (uint32_t)(last + 1 > 0 ? last + 1 : 0);
      uint32_t w1 = 0;
#pragma unroll
      for (int i = 0; i < kTop; i++) w1 = (i == last) ? e[i].y : w1;
      bool have_val = last >= 0;
      if (stopped) depth = stop + 1;
      // deeper levels: the dependent walk of the reference, continued from level kTop; word0 and
      // word1 of a node are fetched together so the colour needs no further dependent load
      if (!stopped && depth > kTop) {
        uint32_t child_idx = e[kTop - 1].x & kMask;
        for (int i = kTop; i < depth; i++) {
          const bool x = tx > cx, y = ty > cy, z = tz > cz;
          my_levels++;
          const uint2 nd = nodes[child_idx + (uint32_t)(x + 2 * y + 4 * z)];
          w1 = nd.y;
          if (!(nd.x & kFlag)) { depth = i + 1; break; }
          child_idx = nd.x & kMask;
          temp_size *= 0.5f;
          cx += temp_size * (x ? 1 : -1);
          cy += temp_size * (y ? 1 : -1);
          cz += temp_size * (z ? 1 : -1);
        }
      }
      if (!have_val) w1 = octree[1];  // depth <= 0: the reference reads node 0 (:107 with node_idx = 0)
      oct_val = w1;
      // :108 max(0, unsigned) is the (int, unsigned) overload: no clamp, alpha = A - 127 signed
      alpha = (int)((oct_val >> 24) - 127u);
      bool retired;
      if (CARRY) {
        const float af = alpha_lut[alpha + 127];  // (float)alpha / 127.0f
        vx = (vx + f2u8(af * (float)(oct_val & 0xFF))) & 0xFFu;
        vy = (vy + f2u8(af * (float)((oct_val >> 8) & 0xFF))) & 0xFFu;
        vz = (vz + f2u8(af * (float)((oct_val >> 16) & 0xFF))) & 0xFFu;
        retired = !((int)vw + alpha < 127);
        vw = retired ? 255u : ((uint32_t)((int)vw + alpha) & 0xFFu);
      } else {
        retired = !(alpha < 127);  // value.w is 0 at every step (Q9)
      }
      if (retired) break;
      // oct_size / pow(2.0f, depth) (:126): division by a power of two == exact scaling
      const float new_dist = (depth >= -100 && depth <= 100) ? ldexpf(P.size, -depth) : P.size / ldexpf(1.0f, depth);
      const float s = (ray_len + new_dist) / ray_len;
      rx *= s; ry *= s; rz *= s;
      ray_len = length3(rx, ry, rz);
      if (ray_len > kMaxRange) { range_exit = true; break; }
    }
    if (!CARRY) {  // the pixel of the retiring step, formed from an all-zero pos[index]
      const float af = alpha_lut[alpha + 127];
      vx = f2u8(af * (float)(oct_val & 0xFF));
      vy = f2u8(af * (float)((oct_val >> 8) & 0xFF));
      vz = f2u8(af * (float)((oct_val >> 16) & 0xFF));
      vw = (uint32_t)alpha & 0xFFu;
    }
    uint32_t out = 0;
    if (range_exit) {  // :131-138 : scale up the colour
      const float sc = 127.0f / (float)vw;
      vx = f2u8((float)vx * sc);
      vy = f2u8((float)vy * sc);
      vz = f2u8((float)vz * sc);
      out = vx | (vy << 8) | (vz << 16) | (255u << 24);
    } else {
      out = vx | (vy << 8) | (vz << 16) | (255u << 24);
    }
    if (P.mode & 0x100) out = my_steps;  // diagnostic: per-ray step count instead of the colour
    uchar4 o;
    o.x = (unsigned char)(out & 0xFF); o.y = (unsigned char)((out >> 8) & 0xFF);
    o.z = (unsigned char)((out >> 16) & 0xFF); o.w = (unsigned char)(out >> 24);
    pos[idx] = o;
  }
  if (counters) {
    // wave-level sums, one atomic pair per wavefront
    unsigned long long s64 = my_steps, l64 = my_levels;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s64 += __shfl_down(s64, o);
      l64 += __shfl_down(l64, o);
    }
    if (lane == 0) {
      atomicAdd(&counters[0], s64);
      atomicAdd(&counters[1], l64);
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
  if ((mode & 0xFF) != SVOSLAM_RENDER_REFERENCE && (mode & 0xFF) != SVOSLAM_RENDER_CARRY) return SVOSLAM_ERR_INVALID_ARG;
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
  static DeviceBuffer top_tree;  // 2.4 MB, library-owned (calls from several host threads must be serialised)
  SVO_TRY(top_tree.reserve((size_t)kTopEntries * sizeof(uint2) + 256 * sizeof(float)));
  float *alpha_lut = reinterpret_cast<float *>(top_tree.as<uint2>() + kTopEntries);
  build_top_tree_kernel<<<cdiv(kTopEntries, 256), 256, 0, stream>>>(d_octree, top_tree.as<uint2>(), alpha_lut);
  if ((mode & 0xFF) == SVOSLAM_RENDER_CARRY)
    cone_trace_kernel<true><<<grid, 256, 0, stream>>>(reinterpret_cast<uchar4 *>(d_pos), d_octree, top_tree.as<uint2>(), alpha_lut, P, d_steps);
  else
    cone_trace_kernel<false><<<grid, 256, 0, stream>>>(reinterpret_cast<uchar4 *>(d_pos), d_octree, top_tree.as<uint2>(), alpha_lut, P, d_steps);
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
