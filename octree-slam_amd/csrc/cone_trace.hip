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
// What bounds this kernel (measured, scratch/ray_steps.py + ISA): instruction issue while
// the chip is full and, in the tail, the critical path of the SLOWEST ray (steps x
// dependent latency per step); HBM traffic is ~4 % of the algorithmic bytes (the walk
// hits in L2).  So each step is made short, without changing a result bit:
//  (1) the octant decisions of ALL levels of a sample come from one lookup per axis into
//      a sorted table of the reference's own split planes (see "split-plane table") instead
//      of a 3-op-per-axis-per-level dependent chain;
//  (2) levels 1..7 of the walk are ONE load from a dense 128^3 grid that stores, per cell,
//      where the reference's walk stops and the colour word it ends on ("level grid");
//  (3) the per-step arithmetic is one division and one square root (see step_lod / loop notes).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "config.hpp"
#include "cone_trace.hpp"
#include "pool_grid.hpp"
#include "stage_timing.hpp"
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

// The compiler's correctly rounded a / b and sqrt(x) without their range handling (v_div_scale x 2, v_div_fixup: 3 of 11
// instructions; the 2^32 pre-scaling of tiny arguments and the zero / infinity class test of sqrt: 7 of 16) -- the same
// v_rcp / v_sqrt seeds and the same FMA corrections, hence the same bits whenever that handling is the identity: b and the
// quotient far from the ends of the exponent range, x >= 2^-96 (+inf passes through).  Used on the march's length
// recurrence only, whose operands are ray_len in [0.002, 10] and ray_len + size / 2^depth; see march_update().
__device__ __forceinline__ float div_rn_midrange(float a, float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float e0 = fmaf(-b, r0, 1.0f);
  const float r1 = fmaf(e0, r0, r0);
  const float q0 = a * r1;
  const float e1 = fmaf(-b, q0, a);
  const float q1 = fmaf(e1, r1, q0);
  const float e2 = fmaf(-b, q1, a);
  return fmaf(e2, r1, q1);
}
// the same with r1 = fma(fma(-b, rcp(b), 1), rcp(b), rcp(b)) in hand
__device__ __forceinline__ float div_rn_midrange_r(float a, float b, float r1) {
  const float q0 = a * r1;
  const float e1 = fmaf(-b, q0, a);
  const float q1 = fmaf(e1, r1, q0);
  const float e2 = fmaf(-b, q1, a);
  return fmaf(e2, r1, q1);
}
__device__ __forceinline__ float sqrt_rn_midrange(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
  float out = rd <= 0.0f ? sd : s;
  out = ru > 0.0f ? su : out;
  return out;
}

// ---- split-plane table ----------------------------------------------------
// The reference decides the octant at every level by comparing the sample with a centre
// it builds on the way down: c_0 = centre, c_l = fl(c_{l-1} +/- size/2^l)
// (cone_tracing_kernels.cu:84-100).  Per axis this is a binary search tree of float
// thresholds: the node reached by the bit prefix b_1..b_{l-1} holds the plane tested at
// level l.  Its in-order sequence S[0..2^T-2] is sorted (rounding errors are far below the
// spacing; verified per lookup), so the T octant bits of a coordinate t are simply
//     bits(t) = #{ j : S[j] < t }                         ("t > c" goes to the high side)
// -- a position in a sorted array.  A float multiply guesses the cell g, the four
// thresholds S[g-2..g+1] around it are loaded in one 16-byte access, counted, and the
// bracket S[bits-1] < t <= S[bits] is CHECKED; a lane whose bracket is not confirmed
// (extreme centre/size ratios, NaN) falls back to the reference's own chain.  The table
// holds exactly the floats the chain produces, so the bits are the reference's bits.
// Two tables: the first 11 levels live in LDS (25 KB per workgroup, copied at kernel start) and are consulted every step;
// the levels below them continue the table by the reference's own chain (walk_deep_chain); a global table down to kTabDepth
// serves the lanes whose bracket is not confirmed.  (A 12-level LDS table, 49 KB, made a 1920x1080 / depth-14 tree march 8 %
// shorter on its own and the frame loop 4 % slower -- fewer workgroups of the other stages fit beside it: removed in round 4.)
constexpr int kTabDepth = 16;                   // == SVOSLAM_MAX_DEPTH: every level a pool of this library can have
constexpr int kTabCells = 1 << kTabDepth;
constexpr int kTabStride = kTabCells + 3;       // [-inf, -inf, S[0..2^T-2], +inf, +inf]
constexpr int kLdsDepthMax = 11;
constexpr int kLdsStrideMax = (1 << kLdsDepthMax) + 3;
__host__ __device__ constexpr int lds_cells(int d) { return 1 << d; }
__host__ __device__ constexpr int lds_stride(int d) { return (1 << d) + 3; }
constexpr int kTraceThreads = 512;              // 4 workgroups x 8 waves per CU (25 KB of LDS each)

// ---- level grid ------------------------------------------------------------
// Dense (2^G)^3 array, G = 7 (8 for renders of a megapixel and more), indexed by the first G octant bits of each axis (z, y, x).
// Entry = outcome of the reference's walk over levels 1..G on that path:
//   all G nodes have children:  x = flag | tile index of the level-G node's children, y = its colour word
//   first childless node at level st (1..G): x = st, y = that node's colour word
// so the first G dependent loads of the reference become one.  Rebuilt from the pool at the
// start of every render (the pool is const during it): 16.8 MB, ~9 us.  (G = 6: 2 MB / 5 us build but
// 9 % more trace time; G = 8: 14 % less trace time but 134 MB / 30 us of build per render.)
constexpr int kGridLevelSmall = 7;                // 128^3 cells, 16.8 MB, ~9 us to rebuild
constexpr int kGridLevelLarge = 8;                // 256^3 cells, 134 MB, ~30 us: pays off when the march itself is long (>= 1 M rays)
__host__ __device__ constexpr int grid_entries(int g) { return 1 << (3 * g); }

struct TraceParams {
  float origin[3], x_dir[3], y_dir[3];
  float center[3];
  float size, pix_scale;
  int width, height, mode;
  int row_first, row_end;  // rows [row_first, row_end) are traced (row band of a multi-GPU tile split)
  // lookup helpers (host-computed)
  float lo[3], inv_cell, inv_cell_lds;   // guess of the table cell: (t - lo) * inv_cell
  int lds_depth;                          // levels of the LDS table (11)
  int xcd_w, xcd_h;                       // tile -> XCD mapping (see cone_trace_kernel); brick march: xcd_h = tiles per row
  int pair_rows;                          // brick march: != 0 (= the number of tile rows): the two 32 x 8 strips of a tile lie half the render apart
  int lod_always;                         // brick march: pix_scale x [0.001, 11 + size] lies inside the fast LOD form's range
  uint32_t lod_first, lod_span, size_man;  // fast LOD: valid when bits(pix_size) - lod_first <= lod_span
  int size_exp;
  // Renders with more tiles than the chip holds at once (1920x1080: 4080 against 768): workgroup b takes tile tile_order[b]
  // -- the tiles of the stream's PREVIOUS render of this geometry, costliest first (tile_order_kernel) -- and adds its own
  // cost (wavefront-steps) to tile_cost for the next one.  nullptr: row-major, nothing recorded.
  const uint32_t *tile_order;
  uint32_t *tile_cost;
  int spec_from;   // cone_trace_brick_kernel<.., B > 0>: steps after this one are marched in bursts of B samples
};

// entry e of [fine table | LDS image | alpha LUT] (see "split-plane table")
__device__ inline void build_table_entry(int e, float *__restrict__ table, float *__restrict__ alpha_lut, const TraceParams &P) {
  if (e < 3 * (kTabStride + kLdsStrideMax)) {
    // fine table (kTabDepth levels) followed by the LDS image (P.lds_depth levels)
    const bool fine = e < 3 * kTabStride;
    const int T = fine ? kTabDepth : P.lds_depth, stride = fine ? kTabStride : lds_stride(P.lds_depth);
    if (!fine && e - 3 * kTabStride >= 3 * stride) return;
    const int r = fine ? e : e - 3 * kTabStride;
    const int axis = r / stride, i = r - axis * stride;
    float v;
    if (i < 2) v = -__builtin_inff();
    else if (i >= (1 << T) + 1) v = __builtin_inff();
    else {
      // in-order index j = i - 2 of the tree node at level l with bit prefix p: j + 1 = (2p + 1) * 2^(T - l)
      const uint32_t q = (uint32_t)(i - 1);
      const int tz = __ffs((int)q) - 1;
      const int l = T - tz;
      const uint32_t p = q >> (tz + 1);  // l - 1 bits, first decision in the top bit
      float c = P.center[axis], ts = P.size;
      for (int m = 1; m < l; m++) {
        ts *= 0.5f;  // "/= 2.0f" (:97)
        c += ts * (((p >> (l - 1 - m)) & 1u) ? 1.0f : -1.0f);
      }
      v = c;
    }
    table[e] = v;
    return;
  }
  e -= 3 * (kTabStride + kLdsStrideMax);
  // (float)alpha / 127.0f of :110-112 for alpha = A - 127 in [-127, 128]: 256 IEEE quotients, computed once
  if (e < 256) alpha_lut[e] = (float)(e - 127) / 127.0f;
  else if (e < 256 + 4) alpha_lut[e] = 0.0f;   // sixteen zero bytes: the "no brick" entry of the brick march's bursts
}

template <int GRID>
__global__ __launch_bounds__(256) void build_accel_kernel(const uint32_t *__restrict__ octree, uint2 *__restrict__ grid,
                                                         float *__restrict__ table, float *__restrict__ alpha_lut,
                                                         TraceParams P) {
  int e = blockIdx.x * 256 + threadIdx.x;
  const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
  if (e < grid_entries(GRID)) {
    constexpr uint32_t kAxisMask = (1u << GRID) - 1u;
    const uint32_t xi = (uint32_t)e & kAxisMask, yi = ((uint32_t)e >> GRID) & kAxisMask, zi = (uint32_t)e >> (2 * GRID);
    uint32_t base = 0;
    uint2 out = make_uint2(0u, 0u);
    for (int l = 1; l <= GRID; l++) {
      const int sh = GRID - l;
      const uint32_t oct = ((xi >> sh) & 1u) | (((yi >> sh) & 1u) << 1) | (((zi >> sh) & 1u) << 2);
      const uint2 nd = nodes[base + oct];
      if (!(nd.x & kFlag)) { out = make_uint2((uint32_t)l, nd.y); break; }
      base = nd.x & kMask;
      out = make_uint2(kFlag | base, nd.y);
    }
    grid[e] = out;
    return;
  }
  e -= grid_entries(GRID);
  if (e < (int)pyr_entries(GRID)) {  // the pyramid of levels 1 .. GRID - 1 behind the grid (pool_grid.hpp)
    int L = 1;
    while (L < GRID - 1 && (uint32_t)e >= pyr_offset(L + 1)) L++;
    const uint32_t c = (uint32_t)e - pyr_offset(L), m = (1u << L) - 1u;
    const uint32_t xi = c & m, yi = (c >> L) & m, zi = c >> (2 * L);
    uint32_t base = 0;
    uint2 out = make_uint2(0u, 0u);
    for (int l = 1; l <= L; l++) {
      const int sh = L - l;
      const uint32_t oct = ((xi >> sh) & 1u) | (((yi >> sh) & 1u) << 1) | (((zi >> sh) & 1u) << 2);
      const uint2 nd = nodes[base + oct];
      if (!(nd.x & kFlag)) { out = make_uint2((uint32_t)l, nd.y); break; }
      base = nd.x & kMask;
      out = make_uint2(kFlag | base, nd.y);
    }
    grid[grid_entries(GRID) + e] = out;
    return;
  }
  e -= (int)pyr_entries(GRID);
  build_table_entry(e, table, alpha_lut, P);
}

// split-plane tables + alpha LUT only: for pools whose level grid is maintained incrementally (pool_grid.hpp)
__global__ __launch_bounds__(256) void build_tables_kernel(float *__restrict__ table, float *__restrict__ alpha_lut, TraceParams P) {
  build_table_entry((int)(blockIdx.x * 256 + threadIdx.x), table, alpha_lut, P);
}

struct __attribute__((packed, aligned(4))) Float4U { float a, b, c, d; };

// LDSD octant bits of one coordinate from the LDS table
template <int LDSD>
__device__ inline uint32_t axis_bits_lds(float t, float lo, float inv_cell, const float *tab, bool &ok) {
  int g = (int)((t - lo) * inv_cell);
  g = g < 0 ? 0 : (g > lds_cells(LDSD) - 1 ? lds_cells(LDSD) - 1 : g);
  const float a = tab[g], b = tab[g + 1], c = tab[g + 2], d = tab[g + 3];  // S[g-2..g+1]
  const bool c0 = a < t, c1 = b < t, c2 = c < t, c3 = d < t;
  ok = ok && c0 && !c3;
  return (uint32_t)(g - 2) + (uint32_t)c0 + (uint32_t)c1 + (uint32_t)c2 + (uint32_t)c3;
}

// kTabDepth octant bits of one coordinate (first level in the top bit); ok = bracket confirmed
__device__ inline uint32_t axis_bits(float t, float lo, float inv_cell, const float *__restrict__ tab, bool &ok) {
  int g = (int)((t - lo) * inv_cell);  // v_cvt_i32_f32: truncating, saturating, NaN -> 0
  g = g < 0 ? 0 : (g > kTabCells - 1 ? kTabCells - 1 : g);
  const Float4U th = *reinterpret_cast<const Float4U *>(reinterpret_cast<const char *>(tab) + ((uint32_t)g << 2));  // S[g-2..g+1]
  const bool c0 = th.a < t, c1 = th.b < t, c2 = th.c < t, c3 = th.d < t;
  ok = ok && c0 && !c3;
  return (uint32_t)(g - 2) + (uint32_t)c0 + (uint32_t)c1 + (uint32_t)c2 + (uint32_t)c3;
}

// the reference's chain for `levels` levels (fallback of axis_bits and levels beyond the table)
__device__ __forceinline__ uint32_t axis_bits_chain(float t, float center, float size, int levels) {
  uint32_t bits = 0;
  float c = center, ts = size;
  for (int l = 0; l < levels; l++) {
    const bool hi = t > c;
    bits = (bits << 1) | (uint32_t)hi;
    ts *= 0.5f;
    c += ts * (hi ? 1.0f : -1.0f);
  }
  return bits;
}

// walk below the LDS table's levels: octant bits from the fine (global) table, one lookup per step
__device__ __forceinline__ void walk_deep(const uint2 *__restrict__ nodes, const float *__restrict__ table, const TraceParams &P,
                                       float tx, float ty, float tz, int level, uint32_t child_idx, int &depth, uint32_t &w1) {
  bool fok = true;
  uint32_t fxb = axis_bits(tx, P.lo[0], P.inv_cell, table, fok);
  uint32_t fyb = axis_bits(ty, P.lo[1], P.inv_cell, table + kTabStride, fok);
  uint32_t fzb = axis_bits(tz, P.lo[2], P.inv_cell, table + 2 * kTabStride, fok);
  if (!fok) {
    fxb = axis_bits_chain(tx, P.center[0], P.size, kTabDepth);
    fyb = axis_bits_chain(ty, P.center[1], P.size, kTabDepth);
    fzb = axis_bits_chain(tz, P.center[2], P.size, kTabDepth);
  }
  for (int i = level; i <= depth; i++) {
    uint32_t oct;
    if (i <= kTabDepth) {
      const int sh = kTabDepth - i;
      oct = ((fxb >> sh) & 1u) | (((fyb >> sh) & 1u) << 1) | (((fzb >> sh) & 1u) << 2);
    } else {  // deeper than any pool this library builds
      oct = (axis_bits_chain(tx, P.center[0], P.size, i) & 1u) | ((axis_bits_chain(ty, P.center[1], P.size, i) & 1u) << 1) |
            ((axis_bits_chain(tz, P.center[2], P.size, i) & 1u) << 2);
    }
    const uint2 nd = nodes[child_idx + oct];
    w1 = nd.y;
    if (!(nd.x & kFlag)) { depth = i; break; }
    child_idx = nd.x & kMask;
  }
}

// walk below the LDS table's levels, octant bits continued from the table by the reference's own chain (:93-104): the
// threshold of the level-T decision is the table entry at the even rank (the centre c_(T-1) of the level T-1 node; entry
// (2p + 2) for prefix p, see build_table_entry), and every further centre is c_l = c_(l-1) +- size / 2^l -- the very
// additions the reference performs, so the bits are the reference's without the three 16-byte gathers of the fine table
template <int LDSD>
__device__ __forceinline__ void walk_deep_chain(const uint2 *__restrict__ nodes, const float *lds_tab, const TraceParams &P, float tx, float ty,
                                             float tz, uint32_t xb, uint32_t yb, uint32_t zb, uint32_t child_idx, int &depth, uint32_t &w1) {
  constexpr int kStride = lds_stride(LDSD);
  float ts = ldexpf(P.size, -LDSD);  // size / 2^T: T exact halvings
  float cx = lds_tab[(xb & ~1u) + 2u], cy = lds_tab[kStride + (yb & ~1u) + 2u], cz = lds_tab[2 * kStride + (zb & ~1u) + 2u];
  cx += ts * ((xb & 1u) ? 1.0f : -1.0f);
  cy += ts * ((yb & 1u) ? 1.0f : -1.0f);
  cz += ts * ((zb & 1u) ? 1.0f : -1.0f);
  for (int i = LDSD + 1; i <= depth; i++) {
    const bool hx = tx > cx, hy = ty > cy, hz = tz > cz;
    const uint32_t oct = (uint32_t)hx | ((uint32_t)hy << 1) | ((uint32_t)hz << 2);
    uint2 nd = nodes[child_idx + oct];
    asm volatile("" : "+v"(nd.y));
    w1 = nd.y;
    if (!(nd.x & kFlag)) { depth = i; break; }
    child_idx = nd.x & kMask;
    ts *= 0.5f;
    cx += ts * (hx ? 1.0f : -1.0f);
    cy += ts * (hy ? 1.0f : -1.0f);
    cz += ts * (hz ? 1.0f : -1.0f);
  }
}

// an LOD depth above the grid level (1 <= depth < grid level): the reference's walk from the root ends on the level-`depth` node or on
// the first childless node above it -- ONE entry of the grid's pyramid (pool_grid.hpp; until round 6: `depth` dependent loads)
template <int LDSD, int GRID>
__device__ __forceinline__ uint32_t pyramid_index(uint32_t xb, uint32_t yb, uint32_t zb, int depth) {
  const int sh = LDSD - depth;
  return (uint32_t)grid_entries(GRID) + pyr_offset_rt(depth) + (((zb >> sh) << (2 * depth)) | ((yb >> sh) << depth) | (xb >> sh));
}
template <int LDSD, int GRID>
__device__ __forceinline__ void walk_shallow(const uint2 *__restrict__ grid, uint32_t xb, uint32_t yb, uint32_t zb, int &depth, uint32_t &w1) {
  const uint2 p = grid[pyramid_index<LDSD, GRID>(xb, yb, zb, depth)];
  w1 = p.y;
  if (!(p.x & kFlag)) depth = (int)p.x;
}

// ---- step / level counters ---------------------------------------------------------------------------------------------
// Every wavefront adds its sums to one of kCountSlots slots (64 bytes apart); the LAST workgroup of the launch to get here
// (ticket in the word behind the slots) folds the slots into the caller's two counters and leaves them zero.  (One atomic pair
// per WORKGROUP on the caller's two words was 16 200 same-address atomics for a 1920x1080 render: ~0.5 ms of L2 serialisation,
// more than the march itself; a fold launch of its own behind the march cost the map stream 11 us per frame.)
constexpr int kCountSlots = 1024, kCountSlotWords = 8;
__device__ inline void count_steps(unsigned long long *__restrict__ slots, unsigned long long *__restrict__ counters, uint32_t my_steps,
                                   uint32_t my_levels, unsigned lane) {
  unsigned long long s64 = my_steps, l64 = my_levels;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s64 += __shfl_down(s64, o);
    l64 += __shfl_down(l64, o);
  }
  if (lane == 0) {
    const unsigned slot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (unsigned)(kCountSlots - 1);
    __hip_atomic_fetch_add(&slots[slot * kCountSlotWords], s64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&slots[slot * kCountSlotWords + 1], l64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // hand-off (cdna_hip_programming.md Guideline 16): the adds have been performed at the L2 they share before the ticket is taken
  __shared__ int count_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned *ticket = reinterpret_cast<unsigned *>(slots + (size_t)kCountSlots * kCountSlotWords);
  if (threadIdx.x == 0) count_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  __syncthreads();
  if (!count_last) return;
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long a = 0, b = 0;
  for (unsigned i = threadIdx.x; i < (unsigned)kCountSlots; i += blockDim.x) {  // (atomic exchanges: read at the L2, and zeroed)
    a += __hip_atomic_exchange(&slots[i * kCountSlotWords], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b += __hip_atomic_exchange(&slots[i * kCountSlotWords + 1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_down(a, o);
    b += __shfl_down(b, o);
  }
  if (lane == 0 && (a | b)) {
    atomicAdd(&counters[0], a);
    atomicAdd(&counters[1], b);
  }
}
// CARRY = false: SVOSLAM_RENDER_REFERENCE.  The reference re-reads pos[index] every step and that
// pixel stays 0 until the ray retires (Q9), so a sample's colour matters only on the step that
// retires the ray: the march needs alpha alone and the colour is formed once, after the loop.
// CARRY = true: the local pixel is carried across steps.
template <bool CARRY, int LDSD, int THREADS, int GRID>
__global__ __launch_bounds__(THREADS) void cone_trace_kernel(uchar4 *__restrict__ pos, const uint32_t *__restrict__ octree,
                                                         const uint2 *__restrict__ grid, const float *__restrict__ table,
                                                         const float *__restrict__ alpha_lut_g, TraceParams P,
                                                         unsigned long long *__restrict__ counters, unsigned long long *__restrict__ slots) {
  __shared__ float alpha_lut[256];
  constexpr int kLdsDepth = LDSD;
  constexpr int kGridLevel = GRID;
  constexpr int kLdsStride = lds_stride(LDSD);
  __shared__ float lds_tab[3 * kLdsStride];
  if (threadIdx.x < 256) alpha_lut[threadIdx.x] = alpha_lut_g[threadIdx.x];
  {
    const float *src = table + 3 * kTabStride;  // LDS image follows the fine table
    for (int i = threadIdx.x; i < 3 * kLdsStride; i += THREADS) lds_tab[i] = src[i];
  }
  __syncthreads();
  // 32 x (THREADS / 32) pixel workgroup, one 8x8 tile per wavefront
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  // Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its own L2.  When all
  // workgroups of the render are resident at once (640x480: 600 of 1024 slots) the image is cut into 4 x 2
  // regions of xcd_w x xcd_h tiles and XCD k takes region k, so the grid cells and nodes of neighbouring rays are
  // cached in ONE L2 instead of eight (kernel -6 % in the frame loop, -12 % alone).  Larger renders keep the
  // row-major order (xcd_w == 0): their workgroups are dispatched in rounds and the in-order dispatcher waits
  // for the slowest XCD, so tiles dispatched together must cost the same, i.e. be neighbours (1080p with
  // image-sized regions: kernel -9 % but frames/s -4 %; with 2x2 .. 8x8-tile sub-blocks: no change).
  int tile_x, tile_y;
  if (P.xcd_w > 0) {
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
    const int slot_y = slot / P.xcd_w;
    tile_x = (xcd & 3) * P.xcd_w + (slot - slot_y * P.xcd_w);
    tile_y = (xcd >> 2) * P.xcd_h + slot_y;
  } else {
    tile_y = (int)blockIdx.x / P.xcd_h;  // xcd_h = tiles per row here
    tile_x = (int)blockIdx.x - tile_y * P.xcd_h;
  }
  const int px = tile_x * 32 + (int)(wave & 3u) * 8 + (int)(lane & 7u);
  const int py = P.row_first + tile_y * (THREADS / 32) + (int)(wave >> 2) * 8 + (int)(lane >> 3);
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
    // root cubes of ordinary size: size / 2^depth for depth >= -60 stays below 2^81, quotients and squares of the
    // length recurrence cannot reach the exponent range's ends except by overflowing to +inf, which both forms return
    const bool midrange = P.size >= 9.5367431640625e-07f && P.size <= 1048576.0f;
    uint32_t oct_val = 0;
    int alpha = 0;
    bool range_exit = false;
    for (int step = 0; step < kMaxSteps; step++) {
      my_steps++;
      const float tx = P.origin[0] + rx, ty = P.origin[1] + ry, tz = P.origin[2] + rz;
      const float pix_size = ray_len * P.pix_scale;
      // The grid entry is requested from the GUESSED table cells before the LDS table has answered: the rank of a
      // coordinate among the split planes equals the guess unless the sample lies within rounding of a plane, so the
      // load (the first of the step's dependent round trips) overlaps the table lookups; a lane whose confirmed
      // cell differs loads again below.  Same entry, same bits (march -3 %, frames/s +2 % at cfg3).
      int gx = (int)((tx - P.lo[0]) * P.inv_cell_lds), gy = (int)((ty - P.lo[1]) * P.inv_cell_lds), gz = (int)((tz - P.lo[2]) * P.inv_cell_lds);
      gx = gx < 0 ? 0 : (gx > lds_cells(LDSD) - 1 ? lds_cells(LDSD) - 1 : gx);
      gy = gy < 0 ? 0 : (gy > lds_cells(LDSD) - 1 ? lds_cells(LDSD) - 1 : gy);
      gz = gz < 0 ? 0 : (gz > lds_cells(LDSD) - 1 ? lds_cells(LDSD) - 1 : gz);
      constexpr int kGridShift = kLdsDepth - kGridLevel;
      // LOD depth (:69); fast form of step_lod when both operands are ordinary positive floats.  Ahead of the request since round 6:
      // an LOD coarser than the grid level asks the grid's PYRAMID (pool_grid.hpp) for the cell of its own level instead.
      int depth;
      {
        const uint32_t ub = f2bits(pix_size);
        if (ub - P.lod_first <= P.lod_span) depth = (P.size_exp - (int)(ub >> 23)) + ((ub & 0x7FFFFFu) < P.size_man ? 1 : 0);
        else depth = step_lod(P.size, pix_size);
      }
      const bool coarse = depth < kGridLevel && depth >= 1;
      const uint32_t cell_s = coarse ? pyramid_index<kLdsDepth, kGridLevel>((uint32_t)gx, (uint32_t)gy, (uint32_t)gz, depth)
                                     : (((uint32_t)gz >> kGridShift) << (2 * kGridLevel)) | (((uint32_t)gy >> kGridShift) << kGridLevel) | ((uint32_t)gx >> kGridShift);
      const uint2 g_s = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(grid) + ((size_t)cell_s << 3));
      // the part of the length recurrence's division that depends on the divisor alone (div_rn_midrange: v_rcp + 2 fma)
      float inv_len = __builtin_amdgcn_rcpf(ray_len);
      inv_len = fmaf(fmaf(-ray_len, inv_len, 1.0f), inv_len, inv_len);
      asm volatile("" : "+v"(inv_len));
      // octant bits of every level, per axis
      bool ok = true;
      // the guess itself is confirmed against its two neighbours, S[g-1] < t <= S[g] (one LDS access per axis: the rank IS
      // the guess for all but ~1e-4 of the samples); only a lane whose guess is not the rank counts over four entries
      // (axis_bits_lds), and one whose bracket is still open takes the reference's chain (march -2 %)
      uint32_t xb = (uint32_t)gx, yb = (uint32_t)gy, zb = (uint32_t)gz;
      {
        float ax = lds_tab[gx + 1], bx = lds_tab[gx + 2];
        float ay = lds_tab[kLdsStride + gy + 1], by = lds_tab[kLdsStride + gy + 2];
        float az = lds_tab[2 * kLdsStride + gz + 1], bz = lds_tab[2 * kLdsStride + gz + 2];
        // all six in flight together and one test (written as a short-circuit chain the compiler reads them one after
        // the other behind branches: march +3 %)
        asm volatile("" : "+v"(ax), "+v"(bx), "+v"(ay), "+v"(by), "+v"(az), "+v"(bz));
        const bool fast = (((int)(ax < tx) & (int)!(bx < tx)) & ((int)(ay < ty) & (int)!(by < ty)) & ((int)(az < tz) & (int)!(bz < tz))) != 0;
        if (!fast) {
          xb = axis_bits_lds<LDSD>(tx, P.lo[0], P.inv_cell_lds, lds_tab, ok);
          yb = axis_bits_lds<LDSD>(ty, P.lo[1], P.inv_cell_lds, lds_tab + kLdsStride, ok);
          zb = axis_bits_lds<LDSD>(tz, P.lo[2], P.inv_cell_lds, lds_tab + 2 * kLdsStride, ok);
        }
      }
      if (!ok) {
        xb = axis_bits_chain(tx, P.center[0], P.size, kLdsDepth);
        yb = axis_bits_chain(ty, P.center[1], P.size, kLdsDepth);
        zb = axis_bits_chain(tz, P.center[2], P.size, kLdsDepth);
      }
      asm volatile("" :: "v"(xb), "v"(yb), "v"(zb));
      // ---- the walk (:76-105) ----
      // (A per-lane cache of the previous sample's path -- resume at the first level that differs -- was
      // measured: it removes most loads of a lane but not the wavefront's latency, which is set by the one
      // lane per step that crosses a coarse boundary; 0.277 ms against 0.261 ms without it.)
      uint32_t w1 = 0;
      // octants of the levels between the grid and the end of the LDS table, formed while the grid entry is on its way
      // (the walk below is then one add, one load and a flag test per level: march -4 %)
      uint32_t oct_l[kLdsDepth - kGridLevel];
#pragma unroll
      for (int q = 0; q < kLdsDepth - kGridLevel; q++) {
        const int sh = kLdsDepth - kGridLevel - 1 - q;
        oct_l[q] = ((xb >> sh) & 1u) | (((yb >> sh) & 1u) << 1) | (((zb >> sh) & 1u) << 2);
        asm volatile("" : "+v"(oct_l[q]));
      }
      if (depth >= kGridLevel) {
        const uint32_t cell = ((zb >> (kLdsDepth - kGridLevel)) << (2 * kGridLevel)) | ((yb >> (kLdsDepth - kGridLevel)) << kGridLevel) |
                              (xb >> (kLdsDepth - kGridLevel));
        uint2 g = g_s;
        if (cell != cell_s) g = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(grid) + (cell << 3));
        w1 = g.y;
        asm volatile("" :: "v"(w1));
        if (!(g.x & kFlag)) {
          depth = (int)g.x;  // stopped at the first childless node
        } else if (depth > kGridLevel) {
          // levels 7..: the dependent walk of the reference; both words of a node are fetched together
          uint32_t child_idx = g.x & kMask;
          const int lds_end = depth < kLdsDepth ? depth : kLdsDepth;
          bool stopped = false;
#pragma unroll
          for (int q = 0; q < kLdsDepth - kGridLevel; q++) {
            const int l = kGridLevel + 1 + q;
            if (!stopped && l <= lds_end) {
              uint2 nd = nodes[child_idx + oct_l[q]];
              asm volatile("" : "+v"(nd.y));  // keep the colour word in the same 8-byte load (not a second, dependent one)
              w1 = nd.y;
              if (!(nd.x & kFlag)) { depth = l; stopped = true; }
              else child_idx = nd.x & kMask;
            }
          }
          if (!stopped && depth > kLdsDepth) {  // below the LDS table's levels
            if (ok) walk_deep_chain<LDSD>(nodes, lds_tab, P, tx, ty, tz, xb, yb, zb, child_idx, depth, w1);
            else walk_deep(nodes, table, P, tx, ty, tz, kLdsDepth + 1, child_idx, depth, w1);  // bracket not confirmed (never seen): the fine table
          }
        }
      } else if (depth >= 1) {
        // LOD coarser than the grid (sample farther than ~size/(32 pix_scale)): the reference's walk from the root ends at the LOD
        // level or on a childless node above it -- the pyramid's entry of that cell, requested above from the guessed ranks
        const uint32_t cell = pyramid_index<kLdsDepth, kGridLevel>(xb, yb, zb, depth);
        uint2 g = g_s;
        if (cell != cell_s) g = grid[cell];
        w1 = g.y;
        if (!(g.x & kFlag)) depth = (int)g.x;
      } else {
        w1 = octree[1];  // depth <= 0: the reference reads node 0 (:107 with node_idx = 0)
      }
      my_levels += (uint32_t)(depth > 0 ? depth : 0);  // levels the reference visits == the depth it ends on
      oct_val = w1;
      asm volatile("" :: "v"(w1));

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
      if (midrange && depth >= -60) {  // (uniform && per-lane, no branch: both forms are a dozen instructions)
        const float s = div_rn_midrange_r(ray_len + new_dist, ray_len, inv_len);
        rx *= s; ry *= s; rz *= s;
        ray_len = sqrt_rn_midrange(dot3(rx, ry, rz, rx, ry, rz));
      } else {
        const float s = (ray_len + new_dist) / ray_len;
        rx *= s; ry *= s; rz *= s;
        ray_len = length3(rx, ry, rz);
      }
      asm volatile("" :: "v"(ray_len));
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
  if (slots) count_steps(slots, counters, my_steps, my_levels, lane);
}


// ---- the march over occupancy bricks (SVOSLAM_RENDER_REFERENCE; pool_grid.hpp "occupancy bricks") --------------------------
// the reference's walk for one sample (:76-105) through the level grid and the tree, as cone_trace_kernel performs it:
// returns the colour word of the node it ends on, `depth` = its level
template <int LDSD, int GRID>
__device__ __forceinline__ uint32_t walk_sample(const uint2 *__restrict__ nodes, const uint32_t *__restrict__ octree, const uint2 *__restrict__ grid,
                                             const float *__restrict__ table, const float *lds_tab, const TraceParams &P, float tx, float ty, float tz,
                                             uint32_t xb, uint32_t yb, uint32_t zb, bool ok, int &depth) {
  uint32_t w1 = 0;
  if (depth >= GRID) {
    const uint32_t cell = ((zb >> (LDSD - GRID)) << (2 * GRID)) | ((yb >> (LDSD - GRID)) << GRID) | (xb >> (LDSD - GRID));
    const uint2 g = grid[cell];
    w1 = g.y;
    if (!(g.x & kFlag)) {
      depth = (int)g.x;
    } else if (depth > GRID) {
      uint32_t child_idx = g.x & kMask;
      const int lds_end = depth < LDSD ? depth : LDSD;
      bool stopped = false;
      for (int l = GRID + 1; l <= lds_end; l++) {
        const int sh = LDSD - l;
        const uint32_t oct = ((xb >> sh) & 1u) | (((yb >> sh) & 1u) << 1) | (((zb >> sh) & 1u) << 2);
        const uint2 nd = nodes[child_idx + oct];
        w1 = nd.y;
        if (!(nd.x & kFlag)) { depth = l; stopped = true; break; }
        child_idx = nd.x & kMask;
      }
      if (!stopped && depth > LDSD) {
        if (ok) walk_deep_chain<LDSD>(nodes, lds_tab, P, tx, ty, tz, xb, yb, zb, child_idx, depth, w1);
        else walk_deep(nodes, table, P, tx, ty, tz, LDSD + 1, child_idx, depth, w1);
      }
    }
  } else if (depth >= 1) {
    walk_shallow<LDSD, GRID>(grid, xb, yb, zb, depth, w1);
  } else {
    w1 = octree[1];
  }
  return w1;
}

// Same rays, same samples, same pixel as cone_trace_kernel<false, ...>.  A sample requests its level-grid entry (8 bytes)
// and -- while the ray is among nodes, i.e. the previous sample's level-8 node had children -- its brick entry (2
// bytes) side by side, both from the guessed table cells.  A sample whose LOD lies in 9..12 and whose level-8 node has
// children is answered by the brick; one whose walk stops at or above level 8 (empty space: first childless node at level
// <= 8, or an LOD of exactly 8) by the grid entry; so a step is ONE round trip to memory whatever the depth of the
// tree (two on the step that enters a level-8 node with children).  What remains -- an LOD coarser than the stop level
// or deeper than a level-12 node with children, a sample whose table bracket is not confirmed -- and the LAST sample
// of every ray (whose node's colour word forms the pixel, Q9) take walk_sample().  The rare cases sit behind
// wavefront-uniform branches: the loop is bound by the instructions it issues (profiles/r03_brick_march_anatomy.txt).
// Shape S (pool_grid.hpp "Shapes"): S = 0 is the above; S = 1 (pools fused to depth 13 / 14) answers LODs 9..13 from bricks of
// level-10 nodes with level-12 cells inside the 2048^3-cell window.  The cell of a sample is then one level finer than the LDS
// table's ranks: it is GUESSED by the same float multiply at the finer pitch, its upper 11 bits are confirmed against the table
// as before and its last bit against the reference's own next centre (c_11 = table entry +- size / 2^11: the level-12 decision of
// walk_deep_chain), which the S = 0 kernel forms anyway whenever an LOD reaches 12.
// ---- the march's second phase: BURSTS of speculated samples (round 6; template parameter B of the kernel below) ----
// What binds the kernel in the tail of a render is not what it issues but the round trip of a step's two entries: a
// wavefront alone on its SIMD parks ~680 of a step's ~1100 cycles at s_waitcnt (profiles/r05_march_sq_counters_cfg3.txt), and the
// render ends when its longest ray does (394 steps at cfg3, 768 in config 2's side view).  77 % of a long ray's steps end on the
// same level as the step before (profiles/HISTORY_r01_r03.md).  So, past the first `spec_from` steps, an iteration of the loop places
// B samples at once: the current one and B - 1 further ones reached by advancing with the PREVIOUS step's level -- the reference's own
// arithmetic (:126-131) on the same operands, hence the same bits whenever that level is the one the step then finds -- and requests
// the entries of all of them back to back: B round trips overlap.  The samples are answered in order; a lane follows the chain as
// long as each step ends on the predicted level (`hit`) and otherwise advances by the level it found and starts the next iteration
// from there.  No result depends on the prediction: a hit IS the advance the unpredicted code performs; a miss discards samples
// nobody has counted.  Lanes of a wavefront fall out of step with each other (each counts its own), which costs issue slots while
// the chip is full -- hence the plain loop (phase 1: every lane of a wavefront on the same step) for the first spec_from steps.
// (Round 6, built first and dropped: one sample ahead with the prefetched entries carried across the loop's back edge.  The
// registers of in-flight loads are then free for the step's temporaries, the compiler orders every such write behind the loads it
// still counts, and merges of paths with different numbers of loads in flight end in s_waitcnt vmcnt(0): no overlap left.  Here
// every load is issued and consumed inside one iteration.)
// Measured (profiles/r06_march_ahead_ab.txt; same-box A/B of library variants, cfg3 300-frame map, three alternations each): as a kernel
// of its own with a 128-VGPR budget, bursts of 2: the march 0.317 -> 0.279 ms in the loop but the FRAME slower (2410 -> 2200 frames/s: 107
// VGPRs x 4 wavefronts leave a SIMD no room for the tracker's); at 80 VGPRs (spilling) bursts of 2 / 3 from step 90: the march 0.319 ->
// 0.291 ms, the frame +2 % over 100 frames and +3 % over the driver's 20; merged into this kernel (phase 1 = the tuned loop; 41-58 VGPRs
// spilled, in phase 2) from step 60: the march 0.313 -> 0.281 ms, frames/s 2397 -> 2474 (+3.2 %) and 2131 -> 2190 (+2.8 %); from step 90 /
// 130: +1..3 %; bursts from step 0 lose in the loop (2254) and on short renders; 1080p: 953 -> 945 (nothing).  The SQ counters say why
// it is not more (profiles/r06_march_sq_counters_cfg3.txt, the march alone): wavefront-cycles parked at s_waitcnt 140.6 M -> 127.7 M ->
// 88.8 M (none / from step 90 / from step 0) of 259-273 M, instructions issued 88.9 M -> 108.6 M -> 124.4 M quad-cycles: the bursts buy
// their overlap with the samples they discard, almost one for one (render alone 0.274 -> 0.267 -> 0.273 ms).
// Default: bursts of 3 from step 61 (svoslam_config.march_ahead = 60).
#ifndef SVO_AHEAD_WAVES
#define SVO_AHEAD_WAVES 6
#endif
struct MarchSample {
  float rx, ry, rz, len;   // the ray to this sample and its length
  float tx, ty, tz;        // the sample
  int fx, fy, fz;          // its guessed cell at the bricks' cell level
  int lod;
  uint32_t lod_ok;
  uint2 gq;                // requested: its level-grid (or pyramid) entry
  uint32_t e, have_e;      // ... and its brick entry (have_e = 0: not requested, the wavefront was not among nodes)
};

template <int THREADS, bool LOD_ALWAYS, int S, int B>  // B: samples per burst past the first P.spec_from steps (0: none); LOD_ALWAYS: every pixel size this render can form is in the fast LOD form's range (checked by the host)
__global__ __launch_bounds__(THREADS, SVO_AHEAD_WAVES) void cone_trace_brick_kernel(uchar4 *__restrict__ pos, const uint32_t *__restrict__ octree,
                                                                   const uint2 *__restrict__ grid, const uint16_t *__restrict__ bricks,
                                                                   const float *__restrict__ table, const float *__restrict__ alpha_lut_g,
                                                                   TraceParams P, unsigned long long *__restrict__ counters,
                                                                   unsigned long long *__restrict__ slots) {
  constexpr int LDSD = 11, GRID = kPoolGridLevel;
  constexpr int kLdsStride = lds_stride(LDSD);
  constexpr int kCells = lds_cells(LDSD);
  constexpr int BL = brick_bits_level(S);                               // level of the per-octant bits
  constexpr int kFine = kCells << S;                                   // cells per axis at the bricks' cell level
  constexpr uint32_t kOrg = brick_window_origin(S);                    // first cell of the window
#ifdef SVO_BRICK_DIAG
  const long long c_entry = clock64();
#endif
  __shared__ float alpha_lut[256];
  // split planes of the three axes, then what each axis' rank contributes to the DWORD index of its brick entry
  // (brick_entry_index() >> 1: the field has 2^33 entries; bit 0 of the x rank picks the half)
  __shared__ float lds_tab[3 * kLdsStride + 3 * kCells];
  uint32_t *spread = reinterpret_cast<uint32_t *>(lds_tab + 3 * kLdsStride);
  if (threadIdx.x < 256) alpha_lut[threadIdx.x] = alpha_lut_g[threadIdx.x];
  {
    const float *src = table + 3 * kTabStride;
    for (int i = threadIdx.x; i < 3 * kLdsStride; i += THREADS) lds_tab[i] = src[i];
    for (int i = threadIdx.x; i < kCells; i += THREADS) {
      const uint32_t r = (uint32_t)i;
      spread[i] = (uint32_t)(brick_entry_index(r, 0u, 0u) >> 1);
      spread[kCells + i] = (uint32_t)(brick_entry_index(0u, r, 0u) >> 1);
      spread[2 * kCells + i] = (uint32_t)(brick_entry_index(0u, 0u, r) >> 1);
    }
  }
  __syncthreads();
#ifdef SVO_BRICK_DIAG
  const long long c_tables = clock64();
#endif
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  // Tiles in row-major order: the march is bound by instruction issue, not by its loads, and the long rays of a frame come in
  // bands of rows (grazing views of the floor, a silhouette) -- with one image region per XCD (the tree march's mapping) the
  // SIMDs of the other regions idle through the tail; dealing consecutive tiles to the eight XCDs spreads it (640x480,
  // 300-frame map: 0.388 ms with regions, 0.333 with alternate rows per XCD group, 0.324 row-major).  (Round 4, measured and
  // not kept: persistent workgroups taking tiles from a queue -- the LDS tables loaded once per workgroup instead of once per
  // tile -- gave nothing at 1920x1080: 0.336 against 0.325 ms.)
  const int tile_id = P.tile_order ? (int)P.tile_order[blockIdx.x] : (int)blockIdx.x;
  const int tile_y = tile_id / P.xcd_h;  // (xcd_h = tiles per row)
  const int tile_x = tile_id - tile_y * P.xcd_h;
  const int px = tile_x * 32 + (int)(wave & 3u) * 8 + (int)(lane & 7u);
  // pair_rows != 0 (= the number of tile rows): the tile's two 32 x 8 strips lie half the render apart -- wavefronts w and w + 4 share a
  // SIMD, and the long rays of a frame come in bands of rows, so a wavefront of the expensive band is paired with one of the cheap band
  const int py = P.pair_rows ? P.row_first + (tile_y + (int)(wave >> 2) * P.pair_rows) * 8 + (int)(lane >> 3)
                         : P.row_first + tile_y * (THREADS / 32) + (int)(wave >> 2) * 8 + (int)(lane >> 3);
  uint32_t my_steps = 0, my_levels = 0;
  if (px < P.width && py < P.row_end) {
    const int idx = py * P.width + px;
    const float res_x = (float)P.width, res_y = (float)P.height;
    const float magx = ((float)px - res_x / 2.0f) / 532.57f;
    const float magy = ((float)py - res_y / 2.0f) / 531.54f;
    const float nyx = -P.y_dir[0], nyy = -P.y_dir[1], nyz = -P.y_dir[2];
    const float fx = P.x_dir[1] * nyz - nyy * P.x_dir[2];
    const float fy = P.x_dir[2] * nyx - nyz * P.x_dir[0];
    const float fz = P.x_dir[0] * nyy - nyx * P.x_dir[1];
    const float dx = ((magx * P.x_dir[0]) + (magy * P.y_dir[0])) + fx;
    const float dy = ((magx * P.x_dir[1]) + (magy * P.y_dir[1])) + fy;
    const float dz = ((magx * P.x_dir[2]) + (magy * P.y_dir[2])) + fz;
    const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
    float rx = kStartDist * (dx * inv), ry = kStartDist * (dy * inv), rz = kStartDist * (dz * inv);
    const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
    float ray_len = length3(rx, ry, rz);
    // (a root cube outside the ordinary range of sizes never gets here: the host launches cone_trace_kernel for it)
    const float ts11 = ldexpf(P.size, -LDSD);  // size / 2^11: what the level-11 decision adds to the centre (walk_deep_chain)
    const float ts12 = ldexpf(P.size, -LDSD - 1);
    const float inv_cell_fine = S == 0 ? P.inv_cell_lds : P.inv_cell_lds * 2.0f;  // (a power of two times it: the same guess one bit finer)
    // The loop carries the ray (rx, ry, rz, ray_len), the previous grid word and the counters, nothing else: which of its
    // two exits a ray took is read off afterwards, and the last sample's node is looked up again for the pixel (once per ray).
    float tx = 0.0f, ty = 0.0f, tz = 0.0f;  // the sample of the current step (the last one, after the loop)
    int lod = 0;
    uint32_t retired = 0;  // (an integer, not a bool: a loop-carried bool is a lane mask the compiler re-blends every iteration)
    uint32_t prev_gx = 0;  // the previous sample's grid word: its children flag = "this ray is among nodes"
    int dprev = 1 << 20;    // (B > 0) the previous step's level
    bool ray_done = true;   // (B > 0) false: the loop below handed the ray over to the bursts
    auto brick_entry = [&](uint32_t x, uint32_t y, uint32_t z) -> uint32_t {  // cell (x, y, z) at the bricks' cell level, through the LDS spread tables
      if (S > 0) {  // the window: cells outside it have no entry (0 = "ask the level grid")
        x -= kOrg; y -= kOrg; z -= kOrg;
        if ((x | y | z) >= kBrickWindowCells) return 0u;
      }
      const uint32_t d = spread[x] | spread[kCells + y] | spread[2 * kCells + z];
      return *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(bricks) + (((size_t)d << 2) | ((x & 1u) << 1)));
    };
    // the answers of a sample's two entries: `depth` / `retired` when one of them decides (return value); oct12 = the
    // sample's level-12 octant (read only when the brick's walk ends at level 12; 0 unless some lane's LOD reaches 12).
    // Written in integers: as booleans every condition is a 64-bit lane mask and the loop is bound by what it issues.
    auto decode = [&](uint32_t e, uint2 gq, int lod_, uint32_t oct12, int &depth, uint32_t &ret) -> bool {
      // the brick: the path stops at level st, or goes on to the bits level BL (code 4); code 0 (no brick) gives st = 8 and never
      // qualifies.  S = 0: st = 8 | code (9 / 10 / 11 / 12).  S = 1: codes 1..4 = 10..13, code 5 = the level-9 node is childless.
      // The walk ends at min(LOD, st); levels 9.. carry their own bit (4..), the bits level one per octant (8..15).  An LOD beyond
      // BL over a node with children there (bit 3) ends deeper: not the brick's to answer.
      const int st = S == 0 ? (int)((e & 7u) | 8u) : (int)((0x009DCBA8u >> ((e & 7u) << 2)) & 15u);
      const int depth_b = lod_ < st ? lod_ : st;
      uint32_t bit = (uint32_t)(depth_b - 5);                       // 4.. for levels 9..
      // levels NL .. BL - 1 are the brick's own (its node, the two levels of its cells).  S = 1: a level ABOVE the brick node (9)
      // is answered only where the path STOPS there (a childless node's word never changes); the alpha of a level-9 node with
      // children changes whenever anything below it does, and the seven sibling bricks that also carry its bit are not rebuilt
      // then -- such a sample (LOD 9 among nodes: a ray length of tens of metres at the depths this shape serves) walks the tree
      constexpr int NLv = brick_node_level(S);
      bool by_brick = (uint32_t)(depth_b - NLv) < 3u;
      if (S > 0) by_brick = by_brick || (st > GRID && st < NLv && depth_b == st);   // (st = GRID: no brick entry)
      if (lod_ >= BL) {  // (per lane; rare except at close range)
        const bool deep = depth_b == BL && (lod_ == BL || !(e & 8u));
        by_brick = by_brick || deep;
        bit = deep ? 8u + oct12 : bit;
      }
      // the grid: a first childless node at level gq.x <= 8 ends every walk whose LOD reaches it; a level-8 node with
      // children (gq.x = flag | tile >= 2^30) ends the walk of LOD 8 only
      // (round 6: a sample whose LOD is coarser than the grid level has asked the PYRAMID for the cell of its own level lv = LOD: the
      // same rule one level up -- a childless node at gq.x <= lv ends the walk, a level-lv node with children ends the walk of LOD lv)
      const uint32_t lv = lod_ < GRID ? (uint32_t)lod_ : (uint32_t)GRID;
      const uint32_t depth_g = gq.x < kFlag ? gq.x : lv;
      const uint32_t top_g = gq.x < kFlag ? 127u : lv;   // the LODs it answers: depth_g .. top_g
      const bool by_grid = (uint32_t)lod_ - depth_g <= top_g - depth_g;
      depth = by_brick ? depth_b : (int)depth_g;
      ret = by_brick ? (e >> bit) & 1u : (uint32_t)(gq.y >= 0xFE000000u);
      return by_brick || by_grid;
    };
#ifdef SVO_BRICK_DIAG
    long long c_after_loop = 0;
    uint32_t diag[4] = {0, 0, 0, 0};
    long long clk[3] = {0, 0, 0};
    const long long c_loop = clock64();
#endif
    for (;;) {
#ifdef SVO_BRICK_DIAG
      const long long c0 = clock64();
#endif
      my_steps++;
      tx = P.origin[0] + rx; ty = P.origin[1] + ry; tz = P.origin[2] + rz;
      const float pix_size = ray_len * P.pix_scale;
      // the guessed cell at the bricks' cell level (fx..); its upper 11 bits are the guessed table ranks (gx..)
      int fx_ = (int)((tx - P.lo[0]) * inv_cell_fine), fy_ = (int)((ty - P.lo[1]) * inv_cell_fine), fz_ = (int)((tz - P.lo[2]) * inv_cell_fine);
      fx_ = fx_ < 0 ? 0 : (fx_ > kFine - 1 ? kFine - 1 : fx_);
      fy_ = fy_ < 0 ? 0 : (fy_ > kFine - 1 ? kFine - 1 : fy_);
      fz_ = fz_ < 0 ? 0 : (fz_ > kFine - 1 ? kFine - 1 : fz_);
      const int gx = fx_ >> S, gy = fy_ >> S, gz = fz_ >> S;
      const uint32_t ub = f2bits(pix_size);
      lod = (P.size_exp - (int)(ub >> 23)) + ((ub & 0x7FFFFFu) < P.size_man ? 1 : 0);
      const bool lod_ok = LOD_ALWAYS || ub - P.lod_first <= P.lod_span;
      // entries requested from the GUESSED cells (see cone_trace_kernel); an LOD coarser than the grid level asks the pyramid
      const bool coarse = (uint32_t)(lod - 1) < (uint32_t)(GRID - 1);   // 1 <= lod < GRID
      uint32_t gcell = (((uint32_t)gz >> (LDSD - GRID)) << (2 * GRID)) | (((uint32_t)gy >> (LDSD - GRID)) << GRID) | ((uint32_t)gx >> (LDSD - GRID));
      if (__builtin_expect(__any(coarse), 0)) {  // (uniform: no ray of a 640x480 / 1080p SLAM frame ever gets there)
        if (coarse) gcell = pyramid_index<LDSD, GRID>((uint32_t)gx, (uint32_t)gy, (uint32_t)gz, lod);
      }
      const uint2 gq = grid[gcell];
      const bool with_brick = __any((prev_gx & kFlag) != 0u);  // (uniform: some ray of this wavefront is among nodes)
      uint32_t e = 0;
      if (with_brick) e = brick_entry((uint32_t)fx_, (uint32_t)fy_, (uint32_t)fz_);
      float inv_len = __builtin_amdgcn_rcpf(ray_len);
      inv_len = fmaf(fmaf(-ray_len, inv_len, 1.0f), inv_len, inv_len);
      // confirmation of the guessed ranks: S[g-1] < t <= S[g] on every axis
      const float ax = lds_tab[gx + 1], bx = lds_tab[gx + 2];
      const float ay = lds_tab[kLdsStride + gy + 1], by = lds_tab[kLdsStride + gy + 2];
      const float az = lds_tab[2 * kLdsStride + gz + 1], bz = lds_tab[2 * kLdsStride + gz + 2];
      bool conf = (((int)(ax < tx) & (int)!(bx < tx)) & ((int)(ay < ty) & (int)!(by < ty)) & ((int)(az < tz) & (int)!(bz < tz))) != 0;
#ifdef SVO_BRICK_DIAG
      asm volatile("" :: "v"(conf), "v"(inv_len));
      const long long c1 = clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long c2 = clock64();
#endif
      prev_gx = coarse ? 0u : gq.x;   // (a pyramid entry's children flag is not about a level-8 node: no brick can answer that LOD)
      // the step that enters a level-8 node with children: its brick entry has not been requested yet
      if (!with_brick && __any((gq.x & kFlag) != 0u && lod > GRID && lod_ok)) e = brick_entry((uint32_t)fx_, (uint32_t)fy_, (uint32_t)fz_);
      // the octant at the bits level, where some lane's walk ends there: the level-11 decision's plane is the table entry at
      // the even rank (walk_deep_chain), i.e. S[g-1] for an odd rank and S[g] for an even one -- one of the two entries
      // the confirmation has read (valid for confirmed guesses; the others are redone below).  S = 1: the level-12 bit of the
      // guessed cell is confirmed the same way (always), and the octant one level further follows the chain.
      uint32_t oct12 = 0;
      if (S > 0 || __any(lod >= BL)) {
        float cx = (gx & 1) ? ax : bx, cy = (gy & 1) ? ay : by, cz = (gz & 1) ? az : bz;
        cx += ts11 * ((gx & 1) ? 1.0f : -1.0f);
        cy += ts11 * ((gy & 1) ? 1.0f : -1.0f);
        cz += ts11 * ((gz & 1) ? 1.0f : -1.0f);
        const uint32_t hx = (uint32_t)(tx > cx), hy = (uint32_t)(ty > cy), hz = (uint32_t)(tz > cz);
        if (S == 0) {
          oct12 = hx | (hy << 1) | (hz << 2);
        } else {
          conf = conf && (((hx ^ (uint32_t)fx_) | (hy ^ (uint32_t)fy_) | (hz ^ (uint32_t)fz_)) & 1u) == 0u;
          if (__any(lod >= BL)) {
            cx += ts12 * (hx ? 1.0f : -1.0f);
            cy += ts12 * (hy ? 1.0f : -1.0f);
            cz += ts12 * (hz ? 1.0f : -1.0f);
            oct12 = (uint32_t)(tx > cx) | ((uint32_t)(ty > cy) << 1) | ((uint32_t)(tz > cz) << 2);
          }
        }
      }
      int depth;
      const bool decided = decode(e, gq, lod, oct12, depth, retired);
      float new_dist = ldexpf(P.size, -depth);  // (decided: depth in 1..12)
      bool full_form = false;
      int depth_raw = depth;   // (the level as the reference counts it, before the clip at 0: what the bursts predict with)
#ifdef SVO_BRICK_DIAG
      if (decided && conf && lod_ok) diag[0]++;
      if (with_brick) diag[2]++;
#endif
      if (__builtin_expect(__any(!(decided && conf && lod_ok)), 0)) {
        if (!(decided && conf && lod_ok)) {
          // the rare sample: an LOD outside the fast form's range, a guess that is not the rank (within rounding of a split
          // plane), or a walk neither entry decides
          if (!lod_ok) lod = step_lod(P.size, pix_size);
          bool ok = true;
          uint32_t xb = (uint32_t)gx, yb = (uint32_t)gy, zb = (uint32_t)gz;
          uint32_t e2 = e;
          uint2 g2 = gq;
          if (!conf) {
            xb = axis_bits_lds<LDSD>(tx, P.lo[0], P.inv_cell_lds, lds_tab, ok);
            yb = axis_bits_lds<LDSD>(ty, P.lo[1], P.inv_cell_lds, lds_tab + kLdsStride, ok);
            zb = axis_bits_lds<LDSD>(tz, P.lo[2], P.inv_cell_lds, lds_tab + 2 * kLdsStride, ok);
            if (!ok) {
              xb = axis_bits_chain(tx, P.center[0], P.size, LDSD);
              yb = axis_bits_chain(ty, P.center[1], P.size, LDSD);
              zb = axis_bits_chain(tz, P.center[2], P.size, LDSD);
            }
          }
          if (!conf || !lod_ok) {  // the grid (or pyramid) entry of the confirmed ranks at the true LOD
            const bool coarse2 = (uint32_t)(lod - 1) < (uint32_t)(GRID - 1);
            g2 = grid[coarse2 ? pyramid_index<LDSD, GRID>(xb, yb, zb, lod)
                              : ((zb >> (LDSD - GRID)) << (2 * GRID)) | ((yb >> (LDSD - GRID)) << GRID) | (xb >> (LDSD - GRID))];
            prev_gx = coarse2 ? 0u : g2.x;
          }
          // the reference's centres below the table (walk_deep_chain): the level-12 octant, and for S = 1 the level-13 one
          float cx = lds_tab[(xb & ~1u) + 2u], cy = lds_tab[kLdsStride + (yb & ~1u) + 2u], cz = lds_tab[2 * kLdsStride + (zb & ~1u) + 2u];
          cx += ts11 * ((xb & 1u) ? 1.0f : -1.0f);
          cy += ts11 * ((yb & 1u) ? 1.0f : -1.0f);
          cz += ts11 * ((zb & 1u) ? 1.0f : -1.0f);
          uint32_t oct12r = (uint32_t)(tx > cx) | ((uint32_t)(ty > cy) << 1) | ((uint32_t)(tz > cz) << 2);
          if (S > 0) {
            const uint32_t cxf = (xb << 1) | (oct12r & 1u), cyf = (yb << 1) | ((oct12r >> 1) & 1u), czf = (zb << 1) | (oct12r >> 2);
            if (!conf) e2 = ok ? brick_entry(cxf, cyf, czf) : 0u;
            cx += ts12 * ((oct12r & 1u) ? 1.0f : -1.0f);
            cy += ts12 * ((oct12r & 2u) ? 1.0f : -1.0f);
            cz += ts12 * ((oct12r & 4u) ? 1.0f : -1.0f);
            oct12r = (uint32_t)(tx > cx) | ((uint32_t)(ty > cy) << 1) | ((uint32_t)(tz > cz) << 2);
          } else if (!conf) {
            e2 = ok ? brick_entry(xb, yb, zb) : 0u;
          }
          const bool decided2 = ok && decode(e2, g2, lod, oct12r, depth, retired);
          if (!decided2) {
            depth = lod;
            const uint32_t w = walk_sample<LDSD, GRID>(nodes, octree, grid, table, lds_tab, P, tx, ty, tz, xb, yb, zb, ok, depth);
            retired = (w >> 24) >= 254u ? 1u : 0u;  // == !((int)(A - 127u) < 127), :108-119 with value.w == 0
#ifdef SVO_BRICK_DIAG
            diag[3]++;
#endif
          }
          new_dist = (depth >= -100 && depth <= 100) ? ldexpf(P.size, -depth) : P.size / ldexpf(1.0f, depth);
          full_form = depth < -60;
          depth_raw = depth;
          if (depth < 0) depth = 0;  // (levels the reference visits: none)
#ifdef SVO_BRICK_DIAG
          diag[1]++;
#endif
        }
      }
      my_levels += (uint32_t)depth;
#ifdef SVO_BRICK_DIAG
      clk[0] += c1 - c0; clk[1] += c2 - c1;
#endif
      // advance (:126-131); a retired ray leaves before its advance counts: its ray_len stays the last sample's
      float s = div_rn_midrange_r(ray_len + new_dist, ray_len, inv_len);
      if (__builtin_expect(__any(full_form), 0)) {
        if (full_form) s = (ray_len + new_dist) / ray_len;
      }
      const float nx = rx * s, ny = ry * s, nz = rz * s;
      float nlen = sqrt_rn_midrange(dot3(nx, ny, nz, nx, ny, nz));
      if (__builtin_expect(__any(full_form), 0)) {
        if (full_form) nlen = length3(nx, ny, nz);
      }
#ifdef SVO_BRICK_DIAG
      asm volatile("" :: "v"(nlen));
      clk[2] += clock64() - c2;
#endif
      // one exit: a retired ray, the range exit (:131), the step guard.  The advance is committed either way (the pixel is formed
      // from tx, ty, tz and the LOD of the last sample; which exit it was is read off `retired` below)
      rx = nx; ry = ny; rz = nz; ray_len = nlen;
      if (B > 0) dprev = depth_raw;
      if (retired != 0u || nlen > kMaxRange || my_steps >= (uint32_t)kMaxSteps) break;
      if (B > 0 && (int)my_steps >= P.spec_from) { ray_done = false; break; }   // the rest of this ray in bursts (below)
    }
    if (B > 0 && !ray_done) {
      // ---- phase 2: the rest of the ray in bursts of B samples (see "the same march in BURSTS" above the kernel) ----
      const char *zero_entry = reinterpret_cast<const char *>(alpha_lut_g + 256);
      // position, guessed cell and LOD of the sample at the end of q's ray
      auto place = [&](MarchSample &q) {
        q.tx = P.origin[0] + q.rx; q.ty = P.origin[1] + q.ry; q.tz = P.origin[2] + q.rz;
        int fx_ = (int)((q.tx - P.lo[0]) * inv_cell_fine), fy_ = (int)((q.ty - P.lo[1]) * inv_cell_fine), fz_ = (int)((q.tz - P.lo[2]) * inv_cell_fine);
        q.fx = fx_ < 0 ? 0 : (fx_ > kFine - 1 ? kFine - 1 : fx_);
        q.fy = fy_ < 0 ? 0 : (fy_ > kFine - 1 ? kFine - 1 : fy_);
        q.fz = fz_ < 0 ? 0 : (fz_ > kFine - 1 ? kFine - 1 : fz_);
        const uint32_t ub = f2bits(q.len * P.pix_scale);
        q.lod = (P.size_exp - (int)(ub >> 23)) + ((ub & 0x7FFFFFu) < P.size_man ? 1 : 0);
        q.lod_ok = (LOD_ALWAYS || ub - P.lod_first <= P.lod_span) ? 1u : 0u;
      };
      // its two entries, from the guessed cell.  with_brick (wavefront-uniform): some ray of the wavefront is among nodes; the brick
      // load is issued either way (from sixteen zero bytes behind the alpha table when not wanted: entry 0 = "no brick, ask the level
      // grid") and its value taken as is, so that the number of loads in flight at every later wait is known to the compiler
      auto request = [&](MarchSample &q, bool with_brick) {
        const uint32_t gx = (uint32_t)(q.fx >> S), gy = (uint32_t)(q.fy >> S), gz = (uint32_t)(q.fz >> S);
        const bool coarse = (uint32_t)(q.lod - 1) < (uint32_t)(GRID - 1);
        uint32_t gcell = ((gz >> (LDSD - GRID)) << (2 * GRID)) | ((gy >> (LDSD - GRID)) << GRID) | (gx >> (LDSD - GRID));
        if (__builtin_expect(__any(coarse), 0)) {  // (uniform: no ray of a 640x480 / 1080p SLAM frame ever gets there)
          if (coarse) gcell = pyramid_index<LDSD, GRID>(gx, gy, gz, q.lod);
        }
        q.gq = grid[gcell];
        uint32_t x = (uint32_t)q.fx, y = (uint32_t)q.fy, z = (uint32_t)q.fz;
        bool use = with_brick;
        if (S > 0) {  // the window: cells outside it have no entry
          x -= kOrg; y -= kOrg; z -= kOrg;
          const bool inwin = (x | y | z) < kBrickWindowCells;
          use = use && inwin;
          x = inwin ? x : 0u; y = inwin ? y : 0u; z = inwin ? z : 0u;
        }
        const uint32_t d = spread[x] | spread[kCells + y] | spread[2 * kCells + z];
        const char *at = use ? reinterpret_cast<const char *>(bricks) + (((size_t)d << 2) | ((x & 1u) << 1)) : zero_entry;
        q.e = *reinterpret_cast<const uint16_t *>(at);
        q.have_e = with_brick ? 1u : 0u;
      };
      // what one sample's entries say (cone_trace_brick_kernel's step between its request and its advance): the level the walk ends
      // on (as the reference counts it: before the clip at 0), `retired`, the step length, and the LOD it was evaluated with
      auto answer = [&](const MarchSample &q, float &new_dist, bool &full_form, int &lod_used) -> int {
        const int gx = q.fx >> S, gy = q.fy >> S, gz = q.fz >> S;
        const float tx = q.tx, ty = q.ty, tz = q.tz;
        const float ax = lds_tab[gx + 1], bx = lds_tab[gx + 2];
        const float ay = lds_tab[kLdsStride + gy + 1], by = lds_tab[kLdsStride + gy + 2];
        const float az = lds_tab[2 * kLdsStride + gz + 1], bz = lds_tab[2 * kLdsStride + gz + 2];
        bool conf = (((int)(ax < tx) & (int)!(bx < tx)) & ((int)(ay < ty) & (int)!(by < ty)) & ((int)(az < tz) & (int)!(bz < tz))) != 0;
        const int lod = q.lod;
        const bool lod_ok = LOD_ALWAYS || q.lod_ok != 0u;
        const bool coarse = (uint32_t)(lod - 1) < (uint32_t)(GRID - 1);
        const uint2 gq = q.gq;
        const uint32_t e = q.e;
        prev_gx = coarse ? 0u : gq.x;
        // the step that enters a level-8 node with children without a brick entry requested: a second round trip, on the rare path
        const bool need = q.have_e == 0u && (gq.x & kFlag) != 0u && lod > GRID && !coarse;
        uint32_t oct12 = 0;
        if (S > 0 || __any(lod >= BL)) {
          float cx = (gx & 1) ? ax : bx, cy = (gy & 1) ? ay : by, cz = (gz & 1) ? az : bz;
          cx += ts11 * ((gx & 1) ? 1.0f : -1.0f);
          cy += ts11 * ((gy & 1) ? 1.0f : -1.0f);
          cz += ts11 * ((gz & 1) ? 1.0f : -1.0f);
          const uint32_t hx = (uint32_t)(tx > cx), hy = (uint32_t)(ty > cy), hz = (uint32_t)(tz > cz);
          if (S == 0) {
            oct12 = hx | (hy << 1) | (hz << 2);
          } else {
            conf = conf && (((hx ^ (uint32_t)q.fx) | (hy ^ (uint32_t)q.fy) | (hz ^ (uint32_t)q.fz)) & 1u) == 0u;
            if (__any(lod >= BL)) {
              cx += ts12 * (hx ? 1.0f : -1.0f);
              cy += ts12 * (hy ? 1.0f : -1.0f);
              cz += ts12 * (hz ? 1.0f : -1.0f);
              oct12 = (uint32_t)(tx > cx) | ((uint32_t)(ty > cy) << 1) | ((uint32_t)(tz > cz) << 2);
            }
          }
        }
        int depth;
        const bool decided = decode(e, gq, lod, oct12, depth, retired) && !need;
        new_dist = ldexpf(P.size, -depth);
        full_form = false;
        lod_used = lod;
        if (__builtin_expect(__any(!(decided && conf && lod_ok)), 0)) {
          if (!(decided && conf && lod_ok)) {
            // the rare sample (cone_trace_brick_kernel's, word for word, plus the second trip above)
            int lod2 = lod;
            if (!lod_ok) lod2 = step_lod(P.size, q.len * P.pix_scale);
            lod_used = lod2;
            bool ok = true;
            uint32_t xb = (uint32_t)gx, yb = (uint32_t)gy, zb = (uint32_t)gz;
            uint32_t e2 = e;
            uint2 g2 = gq;
            if (need && conf) e2 = brick_entry((uint32_t)q.fx, (uint32_t)q.fy, (uint32_t)q.fz);
            if (!conf) {
              xb = axis_bits_lds<LDSD>(tx, P.lo[0], P.inv_cell_lds, lds_tab, ok);
              yb = axis_bits_lds<LDSD>(ty, P.lo[1], P.inv_cell_lds, lds_tab + kLdsStride, ok);
              zb = axis_bits_lds<LDSD>(tz, P.lo[2], P.inv_cell_lds, lds_tab + 2 * kLdsStride, ok);
              if (!ok) {
                xb = axis_bits_chain(tx, P.center[0], P.size, LDSD);
                yb = axis_bits_chain(ty, P.center[1], P.size, LDSD);
                zb = axis_bits_chain(tz, P.center[2], P.size, LDSD);
              }
            }
            if (!conf || !lod_ok) {
              const bool coarse2 = (uint32_t)(lod2 - 1) < (uint32_t)(GRID - 1);
              g2 = grid[coarse2 ? pyramid_index<LDSD, GRID>(xb, yb, zb, lod2)
                                : ((zb >> (LDSD - GRID)) << (2 * GRID)) | ((yb >> (LDSD - GRID)) << GRID) | (xb >> (LDSD - GRID))];
              prev_gx = coarse2 ? 0u : g2.x;
            }
            float cx = lds_tab[(xb & ~1u) + 2u], cy = lds_tab[kLdsStride + (yb & ~1u) + 2u], cz = lds_tab[2 * kLdsStride + (zb & ~1u) + 2u];
            cx += ts11 * ((xb & 1u) ? 1.0f : -1.0f);
            cy += ts11 * ((yb & 1u) ? 1.0f : -1.0f);
            cz += ts11 * ((zb & 1u) ? 1.0f : -1.0f);
            uint32_t oct12r = (uint32_t)(tx > cx) | ((uint32_t)(ty > cy) << 1) | ((uint32_t)(tz > cz) << 2);
            if (S > 0) {
              const uint32_t cxf = (xb << 1) | (oct12r & 1u), cyf = (yb << 1) | ((oct12r >> 1) & 1u), czf = (zb << 1) | (oct12r >> 2);
              if (!conf) e2 = ok ? brick_entry(cxf, cyf, czf) : 0u;
              cx += ts12 * ((oct12r & 1u) ? 1.0f : -1.0f);
              cy += ts12 * ((oct12r & 2u) ? 1.0f : -1.0f);
              cz += ts12 * ((oct12r & 4u) ? 1.0f : -1.0f);
              oct12r = (uint32_t)(tx > cx) | ((uint32_t)(ty > cy) << 1) | ((uint32_t)(tz > cz) << 2);
            } else if (!conf) {
              e2 = ok ? brick_entry(xb, yb, zb) : 0u;
            }
            const bool decided2 = ok && decode(e2, g2, lod2, oct12r, depth, retired);
            if (!decided2) {
              depth = lod2;
              const uint32_t w = walk_sample<LDSD, GRID>(nodes, octree, grid, table, lds_tab, P, tx, ty, tz, xb, yb, zb, ok, depth);
              retired = (w >> 24) >= 254u ? 1u : 0u;
            }
            new_dist = (depth >= -100 && depth <= 100) ? ldexpf(P.size, -depth) : P.size / ldexpf(1.0f, depth);
            full_form = depth < -60;
            // every load of this block has landed before it is left: one whose result a path does not consume (a short-circuited
            // decode) would stay "in flight" in the compiler's book-keeping and turn later waits into vmcnt(0)
            __builtin_amdgcn_s_waitcnt(0x0F70);
          }
        }
        return depth;
      };
      // one iteration = a burst of N samples; returns true when the ray is done (`rx.. ray_len` = the advance past its last sample)
      auto burst = [&](auto n_tag) -> bool {
        constexpr int N = decltype(n_tag)::value;
        MarchSample q[N];
        const bool with_brick = __any((prev_gx & kFlag) != 0u);
        q[0].rx = rx; q[0].ry = ry; q[0].rz = rz; q[0].len = ray_len;
        place(q[0]);
        request(q[0], with_brick);
        if (N > 1) {
          const float nd = ldexpf(P.size, -dprev);
  #pragma unroll
          for (int j = 1; j < N; j++) {
            float il = __builtin_amdgcn_rcpf(q[j - 1].len);
            il = fmaf(fmaf(-q[j - 1].len, il, 1.0f), il, il);
            const float sp = div_rn_midrange_r(q[j - 1].len + nd, q[j - 1].len, il);
            q[j].rx = q[j - 1].rx * sp; q[j].ry = q[j - 1].ry * sp; q[j].rz = q[j - 1].rz * sp;
            q[j].len = sqrt_rn_midrange(dot3(q[j].rx, q[j].ry, q[j].rz, q[j].rx, q[j].ry, q[j].rz));
            place(q[j]);
            request(q[j], with_brick);
          }
        }
        bool done = false, chain = true;
  #pragma unroll
        for (int j = 0; j < N; j++) {
          if (chain) {
            my_steps++;
            float new_dist; bool full_form; int lod_used;
            const int depth = answer(q[j], new_dist, full_form, lod_used);
            my_levels += (uint32_t)(depth > 0 ? depth : 0);
            const bool hit = j + 1 < N && depth == dprev && !full_form;
            dprev = depth;
            if (hit) {   // the next sample of the chain IS this advance
              rx = q[j + 1 < N ? j + 1 : j].rx; ry = q[j + 1 < N ? j + 1 : j].ry; rz = q[j + 1 < N ? j + 1 : j].rz; ray_len = q[j + 1 < N ? j + 1 : j].len;
            } else {     // (:126-131)
              float il = __builtin_amdgcn_rcpf(q[j].len);
              il = fmaf(fmaf(-q[j].len, il, 1.0f), il, il);
              float s = div_rn_midrange_r(q[j].len + new_dist, q[j].len, il);
              if (full_form) s = (q[j].len + new_dist) / q[j].len;
              rx = q[j].rx * s; ry = q[j].ry * s; rz = q[j].rz * s;
              ray_len = full_form ? length3(rx, ry, rz) : sqrt_rn_midrange(dot3(rx, ry, rz, rx, ry, rz));
            }
            tx = q[j].tx; ty = q[j].ty; tz = q[j].tz; lod = lod_used;
            if (retired != 0u || ray_len > kMaxRange || my_steps >= (uint32_t)kMaxSteps) { done = true; chain = false; }
            else if (!hit) chain = false;
          }
        }
        // (entries of samples nobody answered: landed before the registers are reused -- they were requested with the first ones)
        if (N > 1) __builtin_amdgcn_s_waitcnt(0x0F70);
        return done;
      };
      bool done = false;
      while (!done) done = burst(std::integral_constant<int, (B > 0 ? B : 1)>{});
    }
#ifdef SVO_BRICK_DIAG
    if (counters) {
      // cycles of the lane that stayed longest = of its wavefront: before the entries are needed / waiting for them / after
      if (P.mode & 0x200) {  // the kernel's phases instead of the step classes: LDS tables, ray set-up, loop (the epilogue: below)
        c_after_loop = clock64();
        if (lane == 0) { atomicAdd(&counters[2], (unsigned long long)(c_tables - c_entry)); atomicAdd(&counters[3], (unsigned long long)(c_loop - c_tables)); atomicAdd(&counters[4], (unsigned long long)(c_after_loop - c_loop)); }
      } else {
      atomicAdd(&counters[2], (unsigned long long)diag[0]);
      atomicAdd(&counters[3], (unsigned long long)diag[1]);
      }
      uint32_t mx = my_steps;
      for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)mx, o); mx = v > mx ? v : mx; }
      const unsigned long long top = __ballot(my_steps == mx);
      if (!(P.mode & 0x200) && (int)lane == __ffsll((long long)top) - 1) {
        atomicAdd(&counters[4], (unsigned long long)clk[0]);
        atomicAdd(&counters[5], (unsigned long long)clk[1]);
        atomicAdd(&counters[6], (unsigned long long)clk[2]);
        atomicAdd(&counters[7], (unsigned long long)mx);
      }
    }
#endif
    // the range exit (:131): a ray that did not retire on its last sample and whose advanced length passed the range
    const bool range_exit = retired == 0u && ray_len > kMaxRange;
    // the pixel of the last sample, formed from an all-zero pos[index] (Q9): its node's colour word
    uint32_t w_last;
    {
      bool ok = true;
      uint32_t xb = axis_bits_lds<LDSD>(tx, P.lo[0], P.inv_cell_lds, lds_tab, ok);
      uint32_t yb = axis_bits_lds<LDSD>(ty, P.lo[1], P.inv_cell_lds, lds_tab + kLdsStride, ok);
      uint32_t zb = axis_bits_lds<LDSD>(tz, P.lo[2], P.inv_cell_lds, lds_tab + 2 * kLdsStride, ok);
      if (!ok) {
        xb = axis_bits_chain(tx, P.center[0], P.size, LDSD);
        yb = axis_bits_chain(ty, P.center[1], P.size, LDSD);
        zb = axis_bits_chain(tz, P.center[2], P.size, LDSD);
      }
      int d2 = lod;
      w_last = walk_sample<LDSD, GRID>(nodes, octree, grid, table, lds_tab, P, tx, ty, tz, xb, yb, zb, ok, d2);
    }
    const int alpha = (int)((w_last >> 24) - 127u);
    const float af = alpha_lut[alpha + 127];
    uint32_t vx = f2u8(af * (float)(w_last & 0xFF));
    uint32_t vy = f2u8(af * (float)((w_last >> 8) & 0xFF));
    uint32_t vz = f2u8(af * (float)((w_last >> 16) & 0xFF));
    const uint32_t vw = (uint32_t)alpha & 0xFFu;
    if (range_exit) {
      const float sc = 127.0f / (float)vw;
      vx = f2u8((float)vx * sc);
      vy = f2u8((float)vy * sc);
      vz = f2u8((float)vz * sc);
    }
    uint32_t out = vx | (vy << 8) | (vz << 16) | (255u << 24);
    if (P.mode & 0x100) out = my_steps;
    uchar4 o;
    o.x = (unsigned char)(out & 0xFF); o.y = (unsigned char)((out >> 8) & 0xFF);
    o.z = (unsigned char)((out >> 16) & 0xFF); o.w = (unsigned char)(out >> 24);
    pos[idx] = o;
#ifdef SVO_BRICK_DIAG
    if (counters && (P.mode & 0x200) && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); atomicAdd(&counters[5], (unsigned long long)(clock64() - c_after_loop)); }
#endif
  }
  if (P.tile_cost) {  // what this wavefront cost: its longest ray (every lane issues until that one is done)
    uint32_t mx = my_steps;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)mx, o); mx = v > mx ? v : mx; }
    if (lane == 0) atomicAdd(&P.tile_cost[tile_id], mx);
  }
  if (slots) count_steps(slots, counters, my_steps, my_levels, lane);
}

// Tiles of the previous render, costliest first: tile_order_block (pool_grid.hpp), run by one extra workgroup of the refresh
// launch that precedes the march (no launch of its own on the map stream: as one it took 15 us + a launch boundary per frame)
// or, where no refresh is launched, by this kernel.
// Two workgroups of the brick kernel per CU, not the three its 50 KB of tables would allow: brick_march_lds_pad() bytes of dynamic LDS
// nobody reads take the third away.  With the costliest tiles first, the long rays of a frame (640x480, 300-frame map: 12 % of the
// rays, in 647 of 4800 wavefronts, the lower half of the image -- profiles/r05_ray_anatomy_cfg3_300frames.txt) start at t = 0 with
// four wavefronts per SIMD instead of six and the short tiles fill in behind them: 640x480 in the loop 2505 / 2523 -> 2547..2585
// frames/s (2358 -> 2563 on a box in a slower state), 1080p 925 / 939 -> 940..958; either change alone gives nothing or loses
// (profiles/r05_march_occupancy.txt; one workgroup per CU: 2000).
#ifndef SVO_AHEAD_BURST
#define SVO_AHEAD_BURST 3
#endif
constexpr int kAheadBurst = SVO_AHEAD_BURST;   // samples per burst of cone_trace_brick_kernel past P.spec_from
constexpr int kBrickMarchStaticLds = 1024 + 4 * (3 * lds_stride(11) + 3 * lds_cells(11));  // alpha_lut + lds_tab of cone_trace_brick_kernel
// the pad, from the device's own LDS size (6144 bytes on gfx950's 160 KB per CU; a part with another size gets the pad
// that leaves exactly two workgroups there, or none where two do not fit anyway)
static int brick_march_lds_pad() {
  static int pad = -1;
  if (pad >= 0) return pad;
  int dev = 0;
  hipDeviceProp_t prop;
  size_t lds = 160 * 1024;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.maxSharedMemoryPerMultiProcessor > 0)
    lds = (size_t)prop.maxSharedMemoryPerMultiProcessor;
  else
    (void)hipGetLastError();
  long long p = (long long)(lds / 3) + 1 - kBrickMarchStaticLds;   // three workgroups no longer fit
  p = p < 0 ? 0 : ((p + 255) / 256) * 256;
  if (p > 0 && p < 6144 && 2 * (kBrickMarchStaticLds + 6144) <= (long long)lds) p = 6144;   // (the value the round-5 measurements were taken with)
  if (2 * (kBrickMarchStaticLds + p) > (long long)lds || kBrickMarchStaticLds + p > 64 * 1024) p = 0;
  if (getenv("SVOSLAM_DEBUG_LDS")) fprintf(stderr, "svoslam: LDS per CU %zu bytes, brick march pad %lld\n", lds, p);
  pad = (int)p;
  return pad;
}
constexpr int kPairMaxTiles = 1024;
constexpr int kTileOrderMinTiles = 512;  // resident workgroups of the brick kernel (two per CU): smaller renders start every tile at once
__global__ __launch_bounds__(256) void tile_order_kernel(uint32_t *__restrict__ cost, uint32_t *__restrict__ order, int n) {
  tile_order_block(cost, order, n);
}

// tile -> XCD mapping of a render of tiles_x x tiles_y workgroup tiles; returns the number of workgroups to launch
static unsigned xcd_mapping(TraceParams &P, int tiles_x, int tiles_y) {
  constexpr int kResidentTiles = 1024;  // 256 CUs x 4 workgroups of 512 threads
  if (tiles_x * tiles_y <= kResidentTiles) {
    P.xcd_w = (int)cdiv(tiles_x, 4); P.xcd_h = (int)cdiv(tiles_y, 2);
    return 8u * (unsigned)(P.xcd_w * P.xcd_h);
  }
  P.xcd_w = 0; P.xcd_h = tiles_x;
  return (unsigned)(tiles_x * tiles_y);
}

// ---- per-stream acceleration buffers ----
struct StreamAccel {
  DeviceBuffer buf;
  unsigned long long *count_slots = nullptr;  // [kCountSlots][kCountSlotWords] + ticket, zero between renders (count_steps)
  // what the tables in `buf` were built for (they depend on the root cube only, not on the tree)
  bool tables_valid = false;
  const float *tables_at = nullptr;
  int lds_depth = 0;
  float size = 0.0f, center[3] = {0.0f, 0.0f, 0.0f};
  // tile order of large renders (TraceParams::tile_order): [cost | order] x tiles, for the geometry of the last such render
  uint32_t *tiles = nullptr;
  int tile_cap = 0, tile_count = 0, tile_w = 0, tile_rows = 0, tile_row_first = 0;
};
static std::mutex g_accel_mu;
static std::map<hipStream_t, std::unique_ptr<StreamAccel>> g_accel;

static int accel_for_stream(hipStream_t stream, StreamAccel **out) {
  std::lock_guard<std::mutex> lock(g_accel_mu);
  auto it = g_accel.find(stream);
  if (it == g_accel.end()) it = g_accel.emplace(stream, std::unique_ptr<StreamAccel>(new StreamAccel())).first;
  *out = it->second.get();
  return SVOSLAM_OK;
}

// frees the acceleration buffer of one stream (nullptr: of every stream); the caller has synchronised the stream(s)
int cone_trace_release(hipStream_t stream, bool all) {
  if (!all) pool_accel_forget_stream(stream);  // no pool's grid is ordered behind a stream that is going away
  std::lock_guard<std::mutex> lock(g_accel_mu);
  if (all) {
    for (auto &kv : g_accel) { kv.second->buf.release(); if (kv.second->count_slots) (void)hipFree(kv.second->count_slots); if (kv.second->tiles) (void)hipFree(kv.second->tiles); }
    g_accel.clear();
  } else {
    auto it = g_accel.find(stream);
    if (it != g_accel.end()) { it->second->buf.release(); if (it->second->count_slots) (void)hipFree(it->second->count_slots); if (it->second->tiles) (void)hipFree(it->second->tiles); g_accel.erase(it); }
  }
  return SVOSLAM_OK;
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
  P.tile_order = nullptr; P.tile_cost = nullptr; P.spec_from = 0; P.pair_rows = 0;
  mat4_mul_point(inv, 0.0f, 0.0f, 0.0f, 1.0f, P.origin[0], P.origin[1], P.origin[2]);
  mat4_mul_point(inv, -1.0f, 0.0f, 0.0f, 0.0f, P.x_dir[0], P.x_dir[1], P.x_dir[2]);
  mat4_mul_point(inv, 0.0f, -1.0f, 0.0f, 0.0f, P.y_dir[0], P.y_dir[1], P.y_dir[2]);
  for (int k = 0; k < 3; k++) P.center[k] = center[k];
  P.size = size;
  P.pix_scale = tanf(fov * 3.14159f / 180.0f) / (float)height;
  P.width = width; P.height = height; P.mode = mode;
  P.row_first = row_first; P.row_end = row_first + rows;
  P.lod_always = 0;
  // lookup helpers: the table-cell guess and the operand range of the fast LOD form
  for (int k = 0; k < 3; k++) P.lo[k] = center[k] - size;
  P.inv_cell = (float)kTabCells / (2.0f * size);
  P.lds_depth = kLdsDepthMax;
  P.inv_cell_lds = (float)lds_cells(P.lds_depth) / (2.0f * size);
  {
    uint32_t us;
    memcpy(&us, &size, 4);
    const int ea = (int)((us >> 23) & 0xFF);
    P.size_exp = ea;
    P.size_man = us & 0x7FFFFFu;
    if ((int32_t)us > 0 && ea >= 1 && ea <= 254) {
      const int eb_min = ea - 99 < 1 ? 1 : ea - 99, eb_max = ea + 99 > 254 ? 254 : ea + 99;
      P.lod_first = (uint32_t)eb_min << 23;
      P.lod_span = (((uint32_t)eb_max + 1u) << 23) - 1u - P.lod_first;
    } else {  // size is not an ordinary positive float: always the general form (1 - 2 wraps; never <= 0)
      P.lod_first = 0xFFFFFFFFu;
      P.lod_span = 0u;
    }
  }
  {  // ray lengths run from kStartDist to the range exit (<= 10 + one step of at most `size`): is every pixel size ordinary?
    const float lo_pix = 0.001f * P.pix_scale, hi_pix = (11.0f + size) * P.pix_scale;
    uint32_t ul, uh;
    memcpy(&ul, &lo_pix, 4); memcpy(&uh, &hi_pix, 4);
    P.lod_always = (lo_pix > 0.0f && hi_pix >= lo_pix && ul - P.lod_first <= P.lod_span && uh - P.lod_first <= P.lod_span) ? 1 : 0;
  }
  // Acceleration data of the render.
  //  * Level grid: pools the library knows (anything it allocated: svoslam_pool_init / fusion / scene) carry a level-8
  //    grid that commits keep up to date block by block (pool_grid.hpp): the render only refreshes the dirty blocks.
  //    Foreign node memory gets a grid rebuilt for this render (level 7, or 8 for a megapixel and more) in the
  //    per-stream buffer below.
  //  * Split-plane tables + alpha LUT (0.9 MB): per STREAM, library-owned, rebuilt when centre / size / LDS depth
  //    differ from what the stream's buffer holds.  Renders on one stream are ordered by the stream and share it;
  //    renders on different streams each get their own, so none rebuilds data another is marching through.
  StreamAccel *sa = nullptr;
  SVO_TRY(accel_for_stream(stream, &sa));
  DeviceBuffer &accel = sa->buf;
  const std::shared_ptr<PoolAccel> pa_hold = pool_accel_find(d_octree);  // held until the launches below are enqueued
  PoolAccel *pa = pa_hold.get();
  const bool large = pa != nullptr || (long long)width * rows >= (1ll << 20);  // e.g. 1920x1080 frames
  const int own_cells = pa ? 0 : grid_entries(large ? kGridLevelLarge : kGridLevelSmall);
  const int own_total = pa ? 0 : own_cells + (int)pyr_entries(large ? kGridLevelLarge : kGridLevelSmall);  // the grid, then its pyramid (pool_grid.hpp)
  const size_t accel_bytes = (size_t)own_total * sizeof(uint2) + (size_t)(3 * (kTabStride + kLdsStrideMax) + 256 + 4) * sizeof(float) + 64;
  const void *before = accel.ptr;
  SVO_TRY(accel.reserve(accel_bytes));
  if (accel.ptr != before) sa->tables_valid = false;
  uint2 *own_grid = accel.as<uint2>();
  float *d_table = reinterpret_cast<float *>(own_grid + own_total);
  float *alpha_lut = d_table + 3 * (kTabStride + kLdsStrideMax);
  const uint2 *d_grid = own_grid;
  const uint16_t *d_bricks = nullptr;
  int brick_shift = -1;
  const bool tables_match = sa->tables_valid && sa->tables_at == d_table && sa->lds_depth == P.lds_depth && sa->size == size &&
                            sa->center[0] == center[0] && sa->center[1] == center[1] && sa->center[2] == center[2];
  // large reference-mode renders: the tile order of the march (TraceParams::tile_order) is brought up to date by the refresh launch
  const bool carry = (mode & 0xFF) == SVOSLAM_RENDER_CARRY;
  const int n_tiles = (int)(cdiv(width, 32) * cdiv(rows, kTraceThreads / 32));
  uint32_t *tile_cost = nullptr, *tile_order = nullptr;
  bool order_done = false;
  if (pa && !carry && n_tiles > kTileOrderMinTiles) {
    const int n = n_tiles;
    if (n != sa->tile_count || width != sa->tile_w || rows != sa->tile_rows || row_first != sa->tile_row_first) {
      // another geometry: no history (all costs zero -> row-major order)
      if (n > sa->tile_cap) {
        if (sa->tiles) { SVO_HIP(hipStreamSynchronize(stream)); SVO_HIP(hipFree(sa->tiles)); sa->tiles = nullptr; }
        SVO_HIP(hipMalloc((void **)&sa->tiles, (size_t)n * 8));
        sa->tile_cap = n;
      }
      SVO_HIP(hipMemsetAsync(sa->tiles, 0, (size_t)n * 4, stream));
      sa->tile_count = n; sa->tile_w = width; sa->tile_rows = rows; sa->tile_row_first = row_first;
    }
    tile_cost = sa->tiles; tile_order = sa->tiles + n;
  }
  if (pa) {
    SVO_TRY(pool_accel_refresh(pa, d_octree, stream, &d_grid, (mode & 0xFF) == SVOSLAM_RENDER_REFERENCE, &d_bricks, &brick_shift,
                               tile_cost, tile_order, tile_cost ? n_tiles : 0, &order_done));
    if (!tables_match) build_tables_kernel<<<(int)cdiv(3 * (kTabStride + kLdsStrideMax) + 256 + 4, 256), 256, 0, stream>>>(d_table, alpha_lut, P);
  } else {
    const int build_blocks = (int)cdiv(own_total + 3 * (kTabStride + kLdsStrideMax) + 256 + 4, 256);
    if (large) build_accel_kernel<kGridLevelLarge><<<build_blocks, 256, 0, stream>>>(d_octree, own_grid, d_table, alpha_lut, P);
    else build_accel_kernel<kGridLevelSmall><<<build_blocks, 256, 0, stream>>>(d_octree, own_grid, d_table, alpha_lut, P);
  }
  sa->tables_valid = true; sa->tables_at = d_table; sa->lds_depth = P.lds_depth; sa->size = size;
  for (int k = 0; k < 3; k++) sa->center[k] = center[k];
  unsigned long long *slots = nullptr;
  if (d_steps) {
    if (!sa->count_slots) {
      SVO_HIP(hipMalloc((void **)&sa->count_slots, ((size_t)kCountSlots * kCountSlotWords + 8) * 8));  // (+ the ticket)
      SVO_HIP(hipMemsetAsync(sa->count_slots, 0, ((size_t)kCountSlots * kCountSlotWords + 8) * 8, stream));
    }
    slots = sa->count_slots;
  }
  long long stage_token = -1;
  SVO_TRY(stage_begin(kStageMarch, stream, &stage_token));
  uchar4 *out = reinterpret_cast<uchar4 *>(d_pos);
  const bool midrange_size = size >= 9.5367431640625e-07f && size <= 1048576.0f;  // (see cone_trace_kernel: the length recurrence's short forms)
  if (d_bricks && !carry && midrange_size) {  // a pool of this library in reference mode: the march over occupancy bricks
    // renders of up to kPairMaxTiles tiles: the two 32 x 8 strips of a tile lie half the render apart (see the kernel).  640x480 in
    // the loop 2366 -> 2498 frames/s (march alone 0.313 -> 0.297 ms); 1080p (4080 tiles, in rounds): alone 0.258 -> 0.238 but in
    // the loop 0.50 -> 0.58 ms and 954 -> 944 frames/s, so large renders keep their strips together (profiles/r05_march_occupancy.txt)
    P.pair_rows = n_tiles <= kPairMaxTiles ? (int)cdiv(rows, kTraceThreads / 32) : 0; P.xcd_w = 0; P.xcd_h = (int)cdiv(width, 32);
    const dim3 grid((unsigned)(cdiv(width, 32) * cdiv(rows, kTraceThreads / 32)));
    if (tile_cost) {
      if (!order_done) tile_order_kernel<<<1, 256, 0, stream>>>(tile_cost, tile_order, n_tiles);  // (the refresh was a full build)
      P.tile_cost = tile_cost; P.tile_order = tile_order;
    }
    auto launch = [&](auto kernel) {
      kernel<<<grid, kTraceThreads, (size_t)brick_march_lds_pad(), stream>>>(out, d_octree, d_grid, d_bricks, d_table, alpha_lut, P, d_steps, slots);
    };
    // svoslam_config.march_ahead: < 0 = cone_trace_brick_kernel; n >= 0 = the march one sample ahead from step n + 1 on
    const int ahead = config().march_ahead;
    P.spec_from = ahead;
    if (ahead >= 0) {
      if (brick_shift == 0) {
        if (P.lod_always) launch(cone_trace_brick_kernel<kTraceThreads, true, 0, kAheadBurst>);
        else launch(cone_trace_brick_kernel<kTraceThreads, false, 0, kAheadBurst>);
      } else {
        if (P.lod_always) launch(cone_trace_brick_kernel<kTraceThreads, true, 1, kAheadBurst>);
        else launch(cone_trace_brick_kernel<kTraceThreads, false, 1, kAheadBurst>);
      }
    } else if (brick_shift == 0) {
      if (P.lod_always) launch(cone_trace_brick_kernel<kTraceThreads, true, 0, 0>);
      else launch(cone_trace_brick_kernel<kTraceThreads, false, 0, 0>);
    } else {
      if (P.lod_always) launch(cone_trace_brick_kernel<kTraceThreads, true, 1, 0>);
      else launch(cone_trace_brick_kernel<kTraceThreads, false, 1, 0>);
    }
  } else if (large) {
    const dim3 grid(xcd_mapping(P, (int)cdiv(width, 32), (int)cdiv(rows, kTraceThreads / 32)));
    if (carry) cone_trace_kernel<true, 11, kTraceThreads, kGridLevelLarge><<<grid, kTraceThreads, 0, stream>>>(out, d_octree, d_grid, d_table, alpha_lut, P, d_steps, slots);
    else cone_trace_kernel<false, 11, kTraceThreads, kGridLevelLarge><<<grid, kTraceThreads, 0, stream>>>(out, d_octree, d_grid, d_table, alpha_lut, P, d_steps, slots);
  } else {
    const dim3 grid(xcd_mapping(P, (int)cdiv(width, 32), (int)cdiv(rows, kTraceThreads / 32)));
    if (carry) cone_trace_kernel<true, 11, kTraceThreads, kGridLevelSmall><<<grid, kTraceThreads, 0, stream>>>(out, d_octree, d_grid, d_table, alpha_lut, P, d_steps, slots);
    else cone_trace_kernel<false, 11, kTraceThreads, kGridLevelSmall><<<grid, kTraceThreads, 0, stream>>>(out, d_octree, d_grid, d_table, alpha_lut, P, d_steps, slots);
  }
  SVO_TRY(stage_end(kStageMarch, stage_token, stream));
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
