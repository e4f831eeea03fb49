// svo_build.hip -- sparse-voxel-octree fusion on gfx950.
//
// Functional contract = src/world/svo/svo.cu of the reference (node indices and
// Morton keys bit-exact): computeKeys -> split planning -> tile allocation ->
// leaf fusion -> mip-map, and the BFS extraction.  The organisation is new:
//
//   reference (svo.cu:179-237)                 here
//   ------------------------------------------ ---------------------------------
//   splitKeys + per pass {copy, remove_if,     ONE stable radix sort of the full
//   sort, unique, malloc, rightToLeftShift}    keys, then a single "plan" sweep:
//   = D sorts of n keys + 4D host syncs        every sorted unique leaf walks the
//                                              existing pool once, finds its first
//                                              unsplit ancestor (depth t) and owns
//                                              the prefixes below its common prefix
//                                              with the previous leaf.  A split
//                                              record (pass p = d - t, depth d) is
//                                              ranked inside its (p, d) bucket by a
//                                              wave64 ballot match + popcount prefix,
//                                              an LDS cross-wave offset and one row
//                                              scan: rank order == the reference's
//                                              "sorted unique codes of pass p".
//   realloc + whole-pool copy per frame        geometric capacity, one 76-byte
//   (:663-668)                                 count readback per frame
//   fillNodes race on duplicate keys (:684)    lowest point index wins (stable sort)
//   mipmapNodes: n redundant walks x D (:450)  owner lanes only, node indices saved
//                                              by the fill walk
#include <stdlib.h>
#include <string.h>

#include <cstdio>
#include <vector>

#include "radix_sort.hpp"
#include "image_device.hpp"
#include "config.hpp"
#include "pool_grid.hpp"
#include "svo_build.hpp"
#include "stage_timing.hpp"
#include "wave_rank.hpp"

namespace svoslam {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr unsigned char kNotHead = 0xFE;  // sorted element is a duplicate or an invalid key
constexpr unsigned char kNoSplit = 0xFF;  // path fully exists, nothing to split

// ----------------------------------------------------------------------------
// keys  (svo.cu:33-66, 92-106)
// ----------------------------------------------------------------------------
// idx_bits >= 0: the word of the packed sort (radix_sort.hip), key << idx_bits | point index (idx_bits = 0: the key alone)
template <int STRIDE>
__global__ __launch_bounds__(256) void compute_keys_kernel(const float *__restrict__ pts, int n, int depth, float cx,
                                                           float cy, float cz, float edge, u64 *__restrict__ keys, int idx_bits = -1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float px = pts[(size_t)i * STRIDE], py = pts[(size_t)i * STRIDE + 1], pz = pts[(size_t)i * STRIDE + 2];
  u64 morton = 1;
  // Q1 (svo.cu:38): the finite test reads x, z, z -- y is never tested
  if (finitef_(px) && finitef_(pz)) {
    for (int l = 0; l < depth; l++) {
      const bool x = px > cx, y = py > cy, z = pz > cz;
      morton = (morton << 3) + (u64)(x + 2 * y + 4 * z);
      edge /= 2.0f;
      cx += edge * (x ? 1 : -1);
      cy += edge * (y ? 1 : -1);
      cz += edge * (z ? 1 : -1);
    }
  }
  keys[i] = idx_bits > 0 ? ((morton << idx_bits) | (u64)(unsigned)i) : morton;
}

// Keys for the packed sort (radix_sort.hip): word = key << idx_bits | point index, plus the first pass's digit histogram of
// each 4096-element tile.  FROM_DEPTH: the whole front end of a frame's fusion in one launch -- generateVertexMap,
// transformVertexMap by a device-resident pose, computePointCloudBoundingBox (main.cpp:39-44) and computeKeys, every
// expression as in the stand-alone kernels (vertex_map_kernel, transform_kernel, bbox_partial_kernel, compute_keys_kernel)
// -- without a point cloud in memory: 1.2 MB of depth in, 2.4 MB of keys out instead of 3 x 3.6 MB of points.
struct FrameSource { const uint16_t *depth; const float *pose; int w, h; float fx, fy; int first = 0; };  // first: pixel offset of a row band (its n pixels follow it)

constexpr int kKeysThreads = 512, kKeysIPT = 4;  // == the packed sort's tile (radix_packed_tile(): checked by the host)
template <bool FROM_DEPTH>
__global__ __launch_bounds__(kKeysThreads) void keys_packed_kernel(const float *__restrict__ pts, FrameSource fs, int n, int depth, float cx0,
                                                          float cy0, float cz0, float edge0, int idx_bits, int bits0,
                                                          u64 *__restrict__ packed, u32 *__restrict__ tile_hist,
                                                          float *__restrict__ bbox_partial, unsigned *__restrict__ bbox_ticket,
                                                          float *__restrict__ bbox_out) {
  __shared__ u32 hist[kPackedMaxBins];
  __shared__ float sm[kKeysThreads / 64][7];
  __shared__ int is_last;
  const int bins = 1 << bits0;
  for (int d = threadIdx.x; d < bins; d += kKeysThreads) hist[d] = 0;
  __syncthreads();
  const int tile_elems = kKeysThreads * kKeysIPT;
  float m[16];
  if (FROM_DEPTH) {
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = fs.pose[k];
  }
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float cnt = 0;
  // pixel coordinates: ONE integer division per lane, then + kKeysThreads per round (a division per pixel -- ~40 instructions
  // on this ISA, next to the 14 levels' ~170 -- was a fifth of the kernel's issue time at 1920x1080)
  int gx = 0, gy = 0;
  if (FROM_DEPTH) {
    const int g0 = fs.first + blockIdx.x * tile_elems + (int)threadIdx.x;
    gy = g0 / fs.w; gx = g0 - gy * fs.w;
  }
#pragma unroll
  for (int r = 0; r < kKeysIPT; r++) {
    const int i = blockIdx.x * tile_elems + r * kKeysThreads + (int)threadIdx.x;
    if (i >= n) break;
    float px, py, pz;
    if (FROM_DEPTH) {
      float vx, vy, vz;
      const int g = fs.first + i;  // pixel of the whole image (a row band starts at fs.first)
      if (r > 0) { gx += kKeysThreads; while (gx >= fs.w) { gx -= fs.w; gy++; } }
      vertex_from_depth(fs.depth[g], gx, gy, fs.w, fs.h, fs.fx, fs.fy, fs.w, fs.h, vx, vy, vz);
      mat4_mul_point(m, vx, vy, vz, 1.0f, px, py, pz);
      if (finitef_(px) && finitef_(pz)) {  // computePointCloudBoundingBox (image_kernels.cu:60-102, Q1)
        lo[0] = fminf(px, lo[0]); lo[1] = fminf(py, lo[1]); lo[2] = fminf(pz, lo[2]);
        hi[0] = fmaxf(px, hi[0]); hi[1] = fmaxf(py, hi[1]); hi[2] = fmaxf(pz, hi[2]);
        cnt = 1;
      }
    } else {
      px = pts[3 * (size_t)i]; py = pts[3 * (size_t)i + 1]; pz = pts[3 * (size_t)i + 2];
    }
    u64 morton = 1;
    if (finitef_(px) && finitef_(pz)) {  // Q1 (svo.cu:38): x, z, z
      float cx = cx0, cy = cy0, cz = cz0, edge = edge0;
      for (int l = 0; l < depth; l++) {
        const bool x = px > cx, y = py > cy, z = pz > cz;
        morton = (morton << 3) + (u64)(x + 2 * y + 4 * z);
        edge /= 2.0f;
        cx += edge * (x ? 1 : -1);
        cy += edge * (y ? 1 : -1);
        cz += edge * (z ? 1 : -1);
      }
    }
    packed[i] = (morton << idx_bits) | (u64)(unsigned)(FROM_DEPTH ? fs.first + i : i);
    atomicAdd(&hist[(u32)morton & ((u32)bins - 1u)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < bins; d += kKeysThreads) tile_hist[(size_t)blockIdx.x * bins + d] = hist[d];
  if (!FROM_DEPTH || !bbox_out) return;
  // bounding box: workgroup partial, the last workgroup to arrive folds the partials (min / max: exact in any order).
  // Partials travel as agent-scope (sc1) stores and loads on both sides; the ticket is taken after they have completed.
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
    for (int w = 1; w < kKeysThreads / 64; w++) {
      for (int k = 0; k < 3; k++) { sm[0][k] = fminf(sm[0][k], sm[w][k]); sm[0][3 + k] = fmaxf(sm[0][3 + k], sm[w][3 + k]); }
      sm[0][6] = fmaxf(sm[0][6], sm[w][6]);
    }
    for (int k = 0; k < 7; k++)
      __hip_atomic_store(&bbox_partial[blockIdx.x * 7 + k], sm[0][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(bbox_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  // the fold by the WHOLE workgroup (round 5): one wavefront walking 1013 partials of a 1080p frame in 16 dependent rounds of
  // agent-scope loads was 30 of the kernel's 50 us
  float r[7] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, 0};
  for (int b = threadIdx.x; b < (int)gridDim.x; b += kKeysThreads) {
    float v[7];
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] = __hip_atomic_load(&bbox_partial[b * 7 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < 3; k++) { r[k] = fminf(r[k], v[k]); r[3 + k] = fmaxf(r[3 + k], v[3 + k]); }
    r[6] = fmaxf(r[6], v[6]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) { r[k] = fminf(r[k], __shfl_down(r[k], o)); r[3 + k] = fmaxf(r[3 + k], __shfl_down(r[3 + k], o)); }
    r[6] = fmaxf(r[6], __shfl_down(r[6], o));
  }
  __syncthreads();  // (sm is reused)
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 7; k++) sm[wave][k] = r[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kKeysThreads / 64; w++) {
      for (int k = 0; k < 3; k++) { sm[0][k] = fminf(sm[0][k], sm[w][k]); sm[0][3 + k] = fmaxf(sm[0][3 + k], sm[w][3 + k]); }
      sm[0][6] = fmaxf(sm[0][6], sm[w][6]);
    }
    for (int k = 0; k < 7; k++) bbox_out[k] = sm[0][k];
    __hip_atomic_store(bbox_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
}

// ----------------------------------------------------------------------------
// planning
// ----------------------------------------------------------------------------
// number of leading 3-bit levels two distinct depth-D keys share
__device__ inline int common_levels(u64 a, u64 b, int depth) {
  const u64 x = a ^ b;  // != 0, < 2^(3D)
  const int hb = 63 - __clzll((long long)x);
  return depth - 1 - hb / 3;
}

__device__ inline bool is_head(const u64 *__restrict__ skey, int j, u64 &key, int &c, int depth) {
  key = skey[j];
  const u64 prev = j > 0 ? skey[j - 1] : 1ull;
  if (key == 1ull || key == prev) return false;
  c = (prev == 1ull) ? 0 : common_levels(key, prev, depth);
  return true;
}

// splitKeys (svo.cu:108-142) for one key: first node on the path without the
// children flag.  Q3: the last level is examined only when its octant is 7.
// t in [1, D] = depth of that node, f = its index; t = kNoSplit -> f = leaf index.
// start (round 3): the leaf kernel of the commit resumes this walk instead of repeating it from the root.  It owns the
// levels below c (those shared with the previous head belong to that head), so what it needs from here is the child
// tile its first owned level lives in: `start` = base after level s = min(c, t - 1) (s = c when the whole path exists).
// s < t, so that base is one this walk has read; below the frontier the tiles are this frame's own (n0 + 8 x rank).
__device__ inline void walk_existing(const u32 *__restrict__ pool, u64 key, int depth, int c, int &t, u32 &f, u32 &start) {
  u32 base = 0, node = 0;
  t = kNoSplit;
  start = 0;
  for (int lvl = 1; lvl <= depth; lvl++) {
    const u32 oct = (u32)(key >> (3 * (depth - lvl))) & 7u;
    node = base + oct;
    if (lvl < depth || oct == 7u) {
      const u32 w0 = pool[2 * (size_t)node];
      if (!(w0 & kFlag)) { t = lvl; break; }
      base = w0 & kMask;
      if (lvl <= c) start = base;  // (the last assignment is the base after level min(c, t - 1))
    }
  }
  f = node;
}

// record depths owned by a leaf: [lo, hi] (empty if lo > hi)
__device__ inline void record_range(int t, int c, int depth, int &lo, int &hi) {
  if (t == kNoSplit) { lo = 1; hi = 0; }
  else if (t == depth) { lo = hi = depth; }  // Q4: an octant-7 leaf gains children
  else { lo = t > c + 1 ? t : c + 1; hi = depth - 1; }
}

__device__ inline u32 bucket_id(int p, int d) { return (u32)(p * 16 + (d - 1)); }

// Workgroup id -> tile of the sorted key array.  An XCD-aware order -- tile = (id % 8) * ceil(tiles / 8) + id / 8: one
// contiguous eighth of the Morton-ordered keys, i.e. a compact part of the tree and of the colour image, per XCD and L2 --
// was built and A/B-measured (round 2): kernel durations in the sequential form within 3 % (leaf kernel 34.0 / 33.9 us,
// plan_emit 10.8 / 11.0, plan_count 9.5 / 10.5), frames/s within run-to-run noise.  The plain order stays the default.
__host__ __device__ inline int xcd_grid(int tiles) { return tiles; }
__device__ inline int xcd_tile(int tiles) { (void)tiles; return (int)blockIdx.x; }

// Plan tiles are 512 sorted keys (8 wavefronts): half the [bucket][tile] counters of 256-key tiles to write,
// scan and read back.  (1024-key tiles: same frames/s within noise, and a 16-wavefront workgroup is the hardest to place
// next to the march and the tracker.)
constexpr int kPlanThreads = 512, kPlanWaves = kPlanThreads / 64;
__global__ __launch_bounds__(kPlanThreads) void plan_count_kernel(const u64 *__restrict__ skey, int n, int depth,
                                                                  const u32 *__restrict__ pool, unsigned char *__restrict__ leaf_t,
                                                                  u32 *__restrict__ leaf_f, u32 *__restrict__ leaf_start,
                                                                  u32 *__restrict__ tile_hist, int num_tiles, int *__restrict__ any_valid) {
  __shared__ u32 hist[256];
  const int tile = xcd_tile(num_tiles);
  if (tile >= num_tiles) return;
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  __syncthreads();
  const int j = tile * kPlanThreads + threadIdx.x;
  if (j < n) {
    u64 key; int c = 0;
    if (is_head(skey, j, key, c, depth)) {
      int t; u32 f, start;
      walk_existing(pool, key, depth, c, t, f, start);
      leaf_t[j] = (unsigned char)t;
      leaf_f[j] = f;
      if (leaf_start) leaf_start[j] = start;
      int lo, hi;
      record_range(t, c, depth, lo, hi);
      for (int d = lo; d <= hi; d++) atomicAdd(&hist[bucket_id(d - t, d)], 1u);
      *any_valid = 1;  // benign race: every writer stores 1
    } else {
      leaf_t[j] = kNotHead;
      // (every lane stores: whole lines leave the CU instead of the heads' scattered words, which cost a read-modify-write each --
      // the kernel's WRITE_SIZE was 2.7 x its 9 bytes per key)
      leaf_f[j] = 0u;
      if (leaf_start) leaf_start[j] = 0u;
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) tile_hist[(size_t)threadIdx.x * num_tiles + tile] = hist[threadIdx.x];
}

// Row scan of the [bucket][tile] counters (one workgroup per bucket, as row_scan_kernel) and, in the LAST workgroup to
// arrive, what plan_finish_kernel did in a launch of its own: bucket totals -> bucket bases in reference order (pass
// major, depth minor), pass ranges, any_valid (consumed and cleared for the next plan: no memset launch either).  The
// totals cross workgroups as agent-scope stores / loads on both sides, the ticket is taken after they have completed.
__global__ __launch_bounds__(256) void plan_scan_finish_kernel(u32 *__restrict__ rows, int num_tiles, u32 *__restrict__ totals,
                                                               unsigned *__restrict__ ticket, u32 *__restrict__ bucket_base,
                                                               PlanCounts *__restrict__ counts, int *__restrict__ any_valid,
                                                               int *__restrict__ d_struct, u32 *__restrict__ n0_out) {
  // d_struct (optional; svo_fuse_plan_structure): the first tile index of THIS plan's splits is taken from, and its
  // 8 x records added to, a size that follows the plans instead of the commits
  __shared__ u32 tmp[4];
  __shared__ u32 sbase[257];
  __shared__ int is_last;
  u32 *row = rows + (size_t)blockIdx.x * num_tiles;
  u32 carry = 0;
  // eight counters per lane and round (round 5: a 1920x1080 frame has 4050 tiles -- sixteen rounds of two barriers each at one
  // counter per lane); a round of zeros is not written back (most of the 256 (pass, depth) buckets are empty; its prefixes would be
  // `carry`, but plan_emit reads the prefix of (bucket, tile) only where that tile counted a record in the bucket)
  constexpr int kPer = 8;
  for (int base = 0; base < num_tiles; base += 256 * kPer) {
    const int i0 = base + (int)threadIdx.x * kPer;
    u32 v[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) { v[k] = i0 + k < num_tiles ? row[i0 + k] : 0u; sum += v[k]; }
    u32 total;
    u32 run = carry + block256_exclusive_scan(sum, tmp, total);
    if (total) {
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        if (i0 + k < num_tiles) row[i0 + k] = run;
        run += v[k];
      }
    }
    carry += total;
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&totals[blockIdx.x], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  const u32 mine = __hip_atomic_load(&totals[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32 total;
  const u32 ex = block256_exclusive_scan(mine, tmp, total);
  bucket_base[threadIdx.x] = ex;
  sbase[threadIdx.x] = ex;
  if (threadIdx.x == 0) { sbase[256] = total; bucket_base[256] = total; }
  __syncthreads();
  if (threadIdx.x <= 16) counts->pass_start[threadIdx.x] = (int32_t)sbase[threadIdx.x * 16];
  if (threadIdx.x == 17) {
    counts->pass_start[17] = (int32_t)total; counts->total_records = (int32_t)total;
    counts->any_valid = *any_valid;
    *any_valid = 0;
    if (d_struct) {
      const int n0 = *d_struct;
      *n0_out = (u32)n0;
      *d_struct = n0 + 8 * (int)total;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kPlanThreads) void plan_emit_kernel(const u64 *__restrict__ skey, int n, int depth,
                                                        const unsigned char *__restrict__ leaf_t,
                                                        const u32 *__restrict__ leaf_f, const u32 *__restrict__ bucket_base,
                                                        const u32 *__restrict__ row_prefix, int num_tiles,
                                                        u64 *__restrict__ rec_key, u32 *__restrict__ rec_front,
                                                        unsigned char *__restrict__ rec_pass, u32 *__restrict__ leaf_rec0) {
  // leaf_rec0 (optional): for the head that owns the pass-0 record of its frontier node, that record's rank (what
  // svo_fuse_split_early's commit needs to find the node's new child tile without a search)
  __shared__ u32 cnt[kPlanWaves][256];
  const int tile = xcd_tile(num_tiles);
  if (tile >= num_tiles) return;
  if (threadIdx.x < 256) {
#pragma unroll
    for (int w = 0; w < kPlanWaves; w++) cnt[w][threadIdx.x] = 0;
  }
  __syncthreads();
  const int j = tile * kPlanThreads + threadIdx.x;
  const unsigned wave = threadIdx.x >> 6;
  const unsigned long long lt = lanemask_lt();
  u64 key = 1; int c = 0, t = kNotHead, lo = 1, hi = 0; u32 f = 0;
  if (j < n) {
    t = leaf_t[j];
    if (t != kNotHead) {
      (void)is_head(skey, j, key, c, depth);
      f = leaf_f[j];
      record_range(t, c, depth, lo, hi);
    }
  }
  // phase A: per-wave record counts per (pass, depth) bucket
  for (int d = 1; d <= SVOSLAM_MAX_DEPTH; d++) {
    if (d > depth) break;
    const bool valid = d >= lo && d <= hi;
    if (!__ballot(valid)) continue;
    const u32 b = bucket_id(d - t, d) & 255u;
    const unsigned long long peers = match_digit8(valid, b);
    if (valid && (peers & lt) == 0) cnt[wave][b] = (u32)__popcll(peers);
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    u32 run = 0;
#pragma unroll
    for (int w = 0; w < kPlanWaves; w++) { const u32 v = cnt[w][threadIdx.x]; cnt[w][threadIdx.x] = run; run += v; }
  }
  __syncthreads();
  // phase B: emit records at their reference rank
  for (int d = 1; d <= SVOSLAM_MAX_DEPTH; d++) {
    if (d > depth) break;
    const bool valid = d >= lo && d <= hi;
    if (!__ballot(valid)) continue;
    const u32 b = bucket_id(d - t, d) & 255u;
    const unsigned long long peers = match_digit8(valid, b);
    if (valid) {
      // INVARIANT: row_prefix[(b, tile)] is read only where THIS tile counted a record in bucket b (plan_count ran over
      // the same keys with the same rule), so the round of 2048 tiles it lies in had a non-zero sum and plan_scan_finish wrote its
      // prefixes; entries of all-zero rounds still hold raw zero counts, not `carry`, and nobody may read them as prefixes
      const u32 pos = bucket_base[b] + row_prefix[(size_t)b * num_tiles + tile] + cnt[wave][b] +
                      (u32)__popcll(peers & lt);
      rec_key[pos] = key >> (3 * (depth - d));  // prefix key with its leading 1
      rec_front[pos] = f;
      if (rec_pass) rec_pass[pos] = (unsigned char)(d - t);
      if (leaf_rec0 && d == t) leaf_rec0[j] = pos;
    }
  }
}

// All splits of a call in ONE launch (the asynchronous path).  The reference runs one splitNodes
// launch per pass because a pass walks the pool through the tiles the previous pass created
// (svo.cu:278-289).  Here a record's tile index is its rank (num_nodes0 + 8r), known without
// touching the pool, so every word is written exactly once by one lane, in any order:
//  * a pass-0 record sets flag + tile index in its (existing) frontier node;
//  * every record initialises its own 8-child tile, and gives child c the word0 the reference's
//    NEXT pass would give it: flag + tile index of the record for key*8+c if that prefix is being
//    split too (found by key in bucket (pass+1, depth+1), which is sorted), else 0.
// Same pool contents as the pass loop, no pass ordering, no host-side counts.
__global__ __launch_bounds__(256) void split_all_kernel(const u64 *__restrict__ rec_key, const u32 *__restrict__ rec_front,
                                                        const unsigned char *__restrict__ rec_pass,
                                                        const u32 *__restrict__ bucket_base, const PlanCounts *__restrict__ counts,
                                                        u32 *__restrict__ pool, const int *__restrict__ d_size, int depth,
                                                        u32 *__restrict__ grid_dirty, u32 *__restrict__ n0_saved, int structure) {
  SVO_HIGH_PRIO();  // also beside a march (deferred commits: apply -> plan -> commit -> apply is the cycle that can bind the frame; cfg3, 100 frames, 2209-2499 -> 2373-2511)
  const u32 total = (u32)counts->total_records;
  // structure != 0 (svo_fuse_plan_structure): the first tile index comes from the plan (*n0_saved, set by
  // plan_scan_finish_kernel from the structure-side size) and the links ARE written -- the next plan reads them
  const u32 n0 = structure ? *n0_saved : (u32)*d_size;
  // n0_saved != nullptr: deferred commit -- the links of the pass-0 records (the only words of this kernel a concurrent
  // ray march could see) are left to commit_apply_kernel, which needs the first tile index
  if (n0_saved && !structure && blockIdx.x == 0 && threadIdx.x == 0) *n0_saved = n0;
  for (u32 r = blockIdx.x * 256u + threadIdx.x; r < total; r += gridDim.x * 256u) {
    const u64 key = rec_key[r];
    const int pass = rec_pass[r];
    const int d = (63 - __clzll((long long)key)) / 3;
    // level grid of the ray march (pool_grid.hpp): a split above the block level re-labels the whole cube of its node
    if (grid_dirty && d < kPoolGridBlockLevel) pool_grid_mark(grid_dirty, key, d);
    const u32 child = n0 + 8u * r;
    if (pass == 0 && (!n0_saved || structure)) pool[2 * (size_t)rec_front[r]] = kFlag + (child & kMask);
    u32 w0[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (d + 1 <= depth - 1) {  // children at depth d+1 can only be records while d+1 < D
      const u32 b = bucket_id(pass + 1, d + 1);
      u32 lo = bucket_base[b];
      const u32 end = bucket_base[b + 1];
      u32 hi = end;
      const u64 first = key << 3;
      while (lo < hi) {  // lower bound of key*8 in the sorted bucket
        const u32 mid = (lo + hi) >> 1;
        if (rec_key[mid] < first) lo = mid + 1; else hi = mid;
      }
      for (u32 q = lo; q < end && q < lo + 8u; q++) {
        const u64 k = rec_key[q];
        if ((k >> 3) != key) break;
        w0[(u32)(k & 7ull)] = kFlag + ((n0 + 8u * q) & kMask);
      }
    }
    uint4 *tile = reinterpret_cast<uint4 *>(pool + 2 * (size_t)child);  // 64-byte aligned child tile
    const u32 a = 127u << 24;
    tile[0] = make_uint4(w0[0], a, w0[1], a);
    tile[1] = make_uint4(w0[2], a, w0[3], a);
    tile[2] = make_uint4(w0[4], a, w0[5], a);
    tile[3] = make_uint4(w0[6], a, w0[7], a);
  }
}

// size bookkeeping of the asynchronous path: *d_size += 8 * records (the pool's device-side size)
__global__ void pool_size_update_kernel(int *__restrict__ d_size, const PlanCounts *__restrict__ counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *d_size += 8 * counts->total_records;
}

// splitNodes (svo.cu:239-276) for the records of one pass: record r of the
// whole call gets tile num_nodes0 + 8r, exactly the reference's
// num_nodes + 8*index with index = rank in the pass's sorted unique list.
__global__ __launch_bounds__(256) void split_pass_kernel(const u64 *__restrict__ rec_key, const u32 *__restrict__ rec_front,
                                                         int begin, int end, int pass, u32 *__restrict__ pool,
                                                         u32 num_nodes0) {
  const int r = begin + blockIdx.x * 256 + threadIdx.x;
  if (r >= end) return;
  const u64 key = rec_key[r];
  u32 node = rec_front[r];
  for (int s = pass - 1; s >= 0; s--) node = (pool[2 * (size_t)node] & kMask) + ((u32)(key >> (3 * s)) & 7u);
  const u32 child = num_nodes0 + 8u * (u32)r;
  pool[2 * (size_t)node] = kFlag + (child & kMask);
  uint4 *tile = reinterpret_cast<uint4 *>(pool + 2 * (size_t)child);  // 64-byte aligned child tile
  const uint4 init = make_uint4(0u, 127u << 24, 0u, 127u << 24);
  tile[0] = init; tile[1] = init; tile[2] = init; tile[3] = init;
}

// ----------------------------------------------------------------------------
// leaf fusion (svo.cu:291-382) + path capture for the mip-map
// ----------------------------------------------------------------------------
__device__ inline u32 blend_color256(u32 cur, unsigned char r, unsigned char g, unsigned char b) {
  const int a = (int)(cur >> 24);
  const float f1 = (1 - ((float)a / 256.0f)), f2 = (float)a / 256.0f;
  // new*f1 + cur*f2 is exact in binary32 (<= 16-bit numerators over 256)
  const u32 nr = (u32)(unsigned char)((float)r * f1 + (float)(cur & 0xFF) * f2);
  const u32 ng = (u32)(unsigned char)((float)g * f1 + (float)((cur >> 8) & 0xFF) * f2);
  const u32 nb = (u32)(unsigned char)((float)b * f1 + (float)((cur >> 16) & 0xFF) * f2);
  const int na = a + 2 < 255 ? a + 2 : 255;
  return nr + (ng << 8) + (nb << 16) + ((u32)na << 24);
}

__device__ inline u32 blend_vec4(u32 cur, float r, float g, float b) {
  float nr = r * 256.0f, ng = g * 256.0f, nb = b * 256.0f;
  const int a = (int)(cur >> 24);
  const float f1 = 1 - ((float)a / 256.0f), f2 = (float)a / 256.0f;
  nr = nr * f1 + (float)(cur & 0xFF) * f2;
  ng = ng * f1 + (float)((cur >> 8) & 0xFF) * f2;
  nb = nb * f1 + (float)((cur >> 16) & 0xFF) * f2;
  const int na = a + 2 < 255 ? a + 2 : 255;
  // Q21: 256 carries into the next channel through the integer adds
  return (u32)((int)nr) + ((u32)((int)ng) << 8) + ((u32)((int)nb) << 16) + ((u32)na << 24);
}

template <bool VEC4>
__global__ __launch_bounds__(256) void fill_kernel(const u64 *__restrict__ skey, const u32 *__restrict__ sidx, int n,
                                                   int depth, const unsigned char *__restrict__ leaf_t,
                                                   const void *__restrict__ colors, int color_by_position,
                                                   u32 *__restrict__ pool, u32 *__restrict__ path_nodes,
                                                   unsigned char *__restrict__ leaf_c) {
  // leaf_c (round 5): the head's common-prefix length for the mip passes (0xFF: not a head), so that each of their D - 1
  // launches reads one byte per key instead of two 8-byte keys (config 5: 318 M keys x 15 levels)
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  if (leaf_t[j] == kNotHead) { leaf_c[j] = 0xFF; return; }
  u64 key; int c = 0;
  (void)is_head(skey, j, key, c, depth);
  leaf_c[j] = (unsigned char)c;
  u32 base = 0, node = 0;
  for (int lvl = 1; lvl <= depth; lvl++) {
    node = base + ((u32)(key >> (3 * (depth - lvl))) & 7u);
    if (lvl < depth) {
      if (lvl > c) path_nodes[(size_t)(lvl - 1) * n + j] = node;  // this lane owns prefix(lvl)
      base = pool[2 * (size_t)node] & kMask;
    }
  }
  // duplicates: the head of a run of equal keys is the lowest point index (stable sort)
  const size_t ci = color_by_position ? (size_t)j : (size_t)sidx[j];
  const u32 cur = pool[2 * (size_t)node + 1];
  u32 out;
  if (VEC4) {
    const float *v = reinterpret_cast<const float *>(colors) + 4 * ci;
    out = blend_vec4(cur, v[0], v[1], v[2]);
  } else {
    const unsigned char *v = reinterpret_cast<const unsigned char *>(colors) + 3 * ci;
    out = blend_color256(cur, v[0], v[1], v[2]);
  }
  pool[2 * (size_t)node + 1] = out;
}

// averageChildren (svo.cu:384-441).  Q5: all 8 children always count.
__device__ inline u32 average_tile(const u32 *__restrict__ pool, u32 child_base) {
  const uint4 *tile = reinterpret_cast<const uint4 *>(pool + 2 * (size_t)child_base);
  u32 r = 0, g = 0, b = 0, a = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint4 v = tile[q];
    const u32 w1a = v.y, w1b = v.w;
    r += (w1a & 0xFF) + (w1b & 0xFF);
    g += ((w1a >> 8) & 0xFF) + ((w1b >> 8) & 0xFF);
    b += ((w1a >> 16) & 0xFF) + ((w1b >> 16) & 0xFF);
    const u32 aa = w1a >> 24, ab = w1b >> 24;
    a = a > aa ? a : aa;
    a = a > ab ? a : ab;
  }
  // float sums / 8.0f of the reference are exact: integer floor division
  return (r >> 3) + ((g >> 3) << 8) + ((b >> 3) << 16) + (a << 24);
}

// the same with agent-scope (sc1) loads: children that another lane of this launch has just stored write-through
// (mip_straddle2_kernel, tier 1) must not come from a stale line of this CU's L1
__device__ inline u32 average_tile_agent(u32 *__restrict__ pool, u32 child_base) {
  unsigned long long *tile = reinterpret_cast<unsigned long long *>(pool + 2 * (size_t)child_base);
  unsigned long long v[8];
#pragma unroll
  for (int q = 0; q < 8; q++) v[q] = __hip_atomic_load(tile + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32 r = 0, g = 0, b = 0, a = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const u32 w1 = (u32)(v[q] >> 32);
    r += w1 & 0xFF; g += (w1 >> 8) & 0xFF; b += (w1 >> 16) & 0xFF;
    const u32 aa = w1 >> 24;
    a = a > aa ? a : aa;
  }
  return (r >> 3) + ((g >> 3) << 8) + ((b >> 3) << 16) + (a << 24);
}

// the same over the words a deferred commit sees: a child written by THIS commit has its word in the shadow array
// (entry = epoch << 32 | word), every other child keeps the pool's word
__device__ inline u32 average_tile_deferred(const u32 *__restrict__ pool, const unsigned long long *__restrict__ shadow, u32 epoch,
                                            u32 child_base) {
  const uint4 *tile = reinterpret_cast<const uint4 *>(pool + 2 * (size_t)child_base);
  const uint4 *sh = reinterpret_cast<const uint4 *>(shadow + child_base);  // 8 entries of 8 bytes: {word, epoch} pairs
  u32 r = 0, g = 0, b = 0, a = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint4 v = tile[q];
    const uint4 t = sh[q];
    const u32 w1a = t.y == epoch ? t.x : v.y, w1b = t.w == epoch ? t.z : v.w;
    r += (w1a & 0xFF) + (w1b & 0xFF);
    g += ((w1a >> 8) & 0xFF) + ((w1b >> 8) & 0xFF);
    b += ((w1a >> 16) & 0xFF) + ((w1b >> 16) & 0xFF);
    const u32 aa = w1a >> 24, ab = w1b >> 24;
    a = a > aa ? a : aa;
    a = a > ab ? a : ab;
  }
  return (r >> 3) + ((g >> 3) << 8) + ((b >> 3) << 16) + (a << 24);
}
__device__ inline void shadow_store(unsigned long long *__restrict__ shadow, u32 epoch, u32 node, u32 word) {
  shadow[node] = ((unsigned long long)epoch << 32) | word;
}

__global__ __launch_bounds__(256) void mip_level_kernel(int n, int d, const unsigned char *__restrict__ leaf_c,
                                                        const u32 *__restrict__ path_nodes, u32 *__restrict__ pool) {
  // four keys per lane (one 4-byte load of their prefix bytes): most lanes own nothing at a given level, and a launch over 318 M
  // single bytes ran at a quarter of the memory rate
  const long long j0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (j0 >= n) return;
  u32 c4 = 0xFFFFFFFFu;
  if (j0 + 3 < n) c4 = *reinterpret_cast<const u32 *>(leaf_c + j0);
  else for (int k = 0; k < 4 && j0 + k < n; k++) c4 = (c4 & ~(0xFFu << (8 * k))) | ((u32)leaf_c[j0 + k] << (8 * k));
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if ((int)((c4 >> (8 * k)) & 0xFFu) >= d) continue;  // not a head (0xFF), or an earlier leaf owns this prefix
    const u32 node = path_nodes[(size_t)(d - 1) * n + (size_t)(j0 + k)];
    pool[2 * (size_t)node + 1] = average_tile(pool, pool[2 * (size_t)node] & kMask);
  }
}

// final mip pass (Q6): the mean of root children 0..7 lands in node 0's word1.
// One thread = "every thread reads the pre-launch pool" made deterministic.
__global__ void mip_root_kernel(u32 *__restrict__ pool, const PlanCounts *__restrict__ counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && counts->any_valid) pool[1] = average_tile(pool, 0);
}

// ---- leaf blend + mip levels of the asynchronous commit in TWO launches ----------------------------
// The reference (and the blocking path above) runs one launch per mip level because a level reads what
// the level below wrote.  In the sorted key array the leaves under a node are a contiguous run that
// starts at the node's owner lane, so a node whose run ends inside its owner's (1024-lane) workgroup has
// ALL its touched descendants in that workgroup: the workgroup can finish it level by level behind
// __syncthreads() (same-CU visibility).  Only a node whose run crosses the end of its owner's
// workgroup -- at most one per workgroup and level -- is deferred to a second, single-workgroup launch
// that handles those "straddlers" deepest level first.  Same values as the level-by-level passes.
constexpr u32 kNoStraddler = 0xFFFFFFFFu;

constexpr int kFillThreads = 512;  // leaves per workgroup.  Larger: fewer straddlers for the single-workgroup second launch;
// smaller: more workgroups resident next to the tracker's (which pin 150 CUs).  Measured at cfg3, fill + straddle us:
// 1024 -> 54 + 20, 512 -> 45 + 23, 256 -> 37 + 32; 2418 / 2481 / 2477 frames/s.
template <int MAXD>  // levels a lane keeps in registers: 12 for pools of depth <= 12 (57 VGPRs: four workgroups per CU), 16 otherwise (65: three)
__global__ __launch_bounds__(kFillThreads) void fill_mip_local_kernel(const u64 *__restrict__ skey, const u32 *__restrict__ sidx, int n,
                                                             int depth, const unsigned char *__restrict__ leaf_t,
                                                             const unsigned char *__restrict__ colors, u32 *__restrict__ pool,
                                                             u32 *__restrict__ strad, int num_tiles, u32 *__restrict__ grid_dirty,
                                                             unsigned long long *__restrict__ shadow, u32 epoch,
                                                             u32 *__restrict__ apply_nodes, const u64 *__restrict__ rec_key,
                                                             const u32 *__restrict__ bucket_base, const u32 *__restrict__ n0_saved,
                                                             const u32 *__restrict__ leaf_rec0, const u32 *__restrict__ leaf_start,
                                                             int *__restrict__ strad_bc, int brick_shift, const int *__restrict__ n_live) {
  // n_live (optional; key-range sharded commit): only the first *n_live of the n sorted elements are keys, the rest is padding (key 1):
  // the workgroups past them leave their list entries empty and go, and nobody searches the padding for the next head
  if (n_live) { const int nl = *n_live; n = nl < n ? nl : n; }
  const bool early_links = leaf_rec0 != nullptr;
  // shadow != nullptr: deferred commit.  Every colour word goes to shadow[node] instead of the pool, children are read
  // through average_tile_deferred, and apply_nodes lists the nodes written, per workgroup (apply_append; the counts behind the lists),
  // (level `depth` = the leaf; kNoStraddler = none) for commit_apply_kernel.  The link from the key's frontier node
  // (the first node on its path without children, depth leaf_t[j]) to its new child tile is not in the pool yet either:
  // the tile is n0 + 8 x (rank of the pass-0 record of that prefix), found by key in the record bucket (0, leaf_t[j]).
  SVO_HIGH_PRIO();
  __shared__ int last_owner[SVOSLAM_MAX_DEPTH + 1];  // per level: last lane of this workgroup owning a node there
  __shared__ int next_pos, next_c;                   // first head lane after this workgroup and its common-prefix length
  __shared__ int min_c;                              // smallest common-prefix length of a head of this tile (99: no head)
  __shared__ u32 apply_cnt;                          // deferred commit: nodes this workgroup has listed for the apply so far
  const int tid = (int)threadIdx.x;
  const int bid = xcd_tile(num_tiles);  // this workgroup's tile of the sorted keys
  if (bid >= num_tiles) return;
  int search_tiles = num_tiles;  // where the search for the next head (below) ends
  if (n_live) {
    const int live_tiles = (n + kFillThreads - 1) / kFillThreads;
    if (bid >= live_tiles) {  // padding only
      if (tid >= 1 && tid < depth) {
        strad[2 * ((size_t)tid * num_tiles + bid)] = kNoStraddler;
        strad[2 * ((size_t)tid * num_tiles + bid) + 1] = 0u;
      }
      if (shadow && tid == 0) apply_nodes[(size_t)num_tiles * kFillThreads * (size_t)depth + bid] = 0u;
      return;
    }
    search_tiles = live_tiles;
  }
  if (tid == 0) apply_cnt = 0;  // (first used behind the set-up's barriers)
  // the workgroup's part of the apply list: `depth` entries per lane at most, written densely from its start (apply_append)
  u32 *apply_mine = apply_nodes ? apply_nodes + (size_t)bid * kFillThreads * (size_t)depth : nullptr;
  auto apply_append = [&](bool mine, u32 node) {  // all lanes of the wavefront call it; one LDS atomic per wavefront
    const unsigned long long m = __ballot(mine);
    if (!m) return;
    u32 base = 0;
    if ((tid & 63) == 0) base = atomicAdd(&apply_cnt, (u32)__popcll(m));
    base = (u32)__shfl((int)base, 0);
    if (mine) apply_mine[base + (u32)__popcll(m & lanemask_lt())] = node;
  };
  const int j = bid * kFillThreads + tid;
  // Everything the setup needs from memory is requested at once (one round trip instead of four in sequence): this
  // lane's key pair and point index, and the key pair of the lane at the same place in the next workgroup, one of
  // which is the first head after this workgroup (normally; the loop below covers a workgroup without any head).
  const int jn = j + kFillThreads;
  const unsigned char lt = j < n ? leaf_t[j] : kNotHead;
  const unsigned char ltn = jn < n ? leaf_t[jn] : kNotHead;
  const u64 key_j = j < n ? skey[j] : 1ull, key_p = (j > 0 && j < n) ? skey[j - 1] : 1ull;
  const u64 key_n = jn < n ? skey[jn] : 1ull, key_np = jn < n ? skey[jn - 1] : 1ull;
  const u32 point = j < n ? sidx[j] : 0u;
  const bool head = lt != kNotHead;
  // where the plan's walk of this key's path left off (walk_existing): the child tile of its node at level
  // min(c, leaf_t - 1) -- the walk below starts there instead of at the root (requested with the other setup loads)
  const u32 start_base = (leaf_start && head) ? leaf_start[j] : 0u;
  // early links: rank of the pass-0 record of this key's frontier node, known to the head that owns the record (the first
  // under that node); requested with the other setup loads, handed to the heads that follow below
  u32 rec0 = 0;
  if (early_links && head && lt != kNoSplit) {
    const int cc = (key_p == 1ull) ? 0 : common_levels(key_j, key_p, depth);
    if (cc < (int)lt) rec0 = leaf_rec0[j];
  }
  u64 key = 1; int c = 0;
  if (head) { key = key_j; c = (key_p == 1ull) ? 0 : common_levels(key_j, key_p, depth); }  // is_head() on the values in hand
  // level grid of the ray march (pool_grid.hpp): everything this key changes lies below its level-5 prefix; the first
  // head of a run of keys sharing that prefix marks the block
  if (grid_dirty && head && c < kPoolGridBlockLevel) pool_grid_mark(grid_dirty, key, depth);
  // occupancy bricks: likewise everything below the brick node's level (9 + brick_shift) lies under the key's prefix of that
  // level; the first head of a run lists it (pool_grid.hpp: test-and-set here, the workgroup's ring slots behind the two
  // barriers below, the store behind the walk).  Childless siblings that a split creates beside the key's path are not listed
  // as bricks: a key whose frontier lies above the brick node (new tiles at or above its level) puts its brick into the sibling
  // ring at the end of this kernel, and the refresh writes the siblings' (uniform) lines (pool_grid.hip, brick_siblings).
  __shared__ u32 brick_cnt, brick_base;
  const bool bricks_on = grid_dirty != nullptr && brick_shift >= 0 && depth >= brick_node_level(brick_shift);
  u32 brick_entry = 0, brick_off = 0;
  const bool brick_mine = bricks_on && brick_mark_test(grid_dirty, head && c < brick_node_level(brick_shift), key, depth, brick_shift, brick_entry);
  if (tid <= SVOSLAM_MAX_DEPTH) {
    last_owner[tid] = -1;
    if (tid >= 1 && tid < depth) {  // "no straddler" unless a lane says otherwise below
      strad[2 * ((size_t)tid * num_tiles + bid)] = kNoStraddler;
      strad[2 * ((size_t)tid * num_tiles + bid) + 1] = 0u;
    }
  }
  if (tid == 0) { next_pos = 0x7FFFFFFF; next_c = -1; min_c = 99; brick_cnt = 0u; }  // no later head: every run ends with the array
  __syncthreads();
  if (bricks_on) {  // (straight behind the barrier: every lane of the wavefront takes part in the ballot)
    const unsigned long long bm = __ballot(brick_mine);
    if (bm) {
      const int leader = __ffsll((long long)bm) - 1;
      u32 woff = 0;
      if ((tid & 63) == leader) woff = atomicAdd(&brick_cnt, (u32)__popcll(bm));
      brick_off = (u32)__shfl((int)woff, leader) + (u32)__popcll(bm & ((1ull << (tid & 63)) - 1ull));
    }
  }
  if (head) {
    for (int d = c + 1; d < depth; d++) atomicMax(&last_owner[d], j);
    if (strad_bc) atomicMin(&min_c, c);
  }
  if (ltn != kNotHead) atomicMin(&next_pos, jn);
  __syncthreads();
  if (bricks_on && tid == 0 && brick_cnt) brick_base = brick_ring_reserve(grid_dirty, brick_cnt);
  if (next_pos == 0x7FFFFFFF) {  // no head in the next workgroup (all duplicates / invalid points): look further
    for (int nb = bid + 2; nb < search_tiles; nb++) {
      const int jj = nb * kFillThreads + tid;
      if (jj < n && leaf_t[jj] != kNotHead) atomicMin(&next_pos, jj);
      __syncthreads();
      const bool found = next_pos != 0x7FFFFFFF;
      __syncthreads();  // every lane has read next_pos before anyone updates it again
      if (found) break;
    }
    if (tid == 0 && next_pos != 0x7FFFFFFF) {
      u64 k2; int c2 = 0;
      (void)is_head(skey, next_pos, k2, c2, depth);
      next_c = c2;
    }
  } else if (jn == next_pos) {
    next_c = (key_np == 1ull) ? 0 : common_levels(key_n, key_np, depth);
  }
  // boundary record for the straddler pass: {how many levels the first head after this tile shares with its predecessor
  // (-1: none), the smallest common-prefix length of a head IN this tile (99: none)}.  A level-d node that owns leaves at
  // the end of tile t continues into later tiles iff bc[t].x >= d, and covers the whole of a later tile b iff no level-d
  // node starts there: bc[b].y >= d.
  if (strad_bc) {
    __syncthreads();
    if (tid == 0) { strad_bc[2 * bid] = next_c; strad_bc[2 * bid + 1] = min_c; }
  }
  // early_links != 0: the child tiles of this commit were initialised ahead of it (svo_fuse_split_early: split_all_kernel
  // without its links, while the previous frame was still being ray-marched); the links of the pass-0 records -- the only
  // words of the split a ray march can see -- are written HERE, by the first head under each frontier node, and the walk
  // below takes the frontier's child tile from the record's rank like a deferred commit does.
  int frontier = 0;       // level whose link is not in the pool yet (0: none)
  int link_level = 0;     // early_links: level of the frontier node whose link this lane writes (0: none)
  u32 frontier_child = 0;
  // The heads under one frontier node are consecutive heads that all stop at it (the node has no children, so every key
  // through it has the same leaf_t), and the first of them owns its record: a head's record is that of the NEAREST owner
  // at or before it.  Within a wavefront by ballot + shuffle, across wavefronts through LDS; a run that began in an
  // earlier workgroup falls back to the search by key.
  bool have_rec0 = false;
  if (early_links) {
    __shared__ u32 wave_rec0[kFillThreads / 64];
    __shared__ int wave_has[kFillThreads / 64];
    const bool towner = head && lt != kNoSplit && c < (int)lt;
    const unsigned lane = (unsigned)tid & 63u, wv = (unsigned)tid >> 6;
    const unsigned long long om = __ballot(towner);
    const unsigned long long upto = om & ((lane == 63u) ? ~0ull : ((2ull << lane) - 1ull));
    const int src = upto ? 63 - __clzll((long long)upto) : -1;
    const u32 got = (u32)__shfl((int)rec0, src >= 0 ? src : 0);
    const int last = om ? 63 - __clzll((long long)om) : 0;
    const u32 lastv = (u32)__shfl((int)rec0, last);
    if (lane == 0) { wave_has[wv] = om != 0ull; wave_rec0[wv] = lastv; }
    __syncthreads();
    if (src >= 0) { rec0 = got; have_rec0 = true; }
    else {
      for (int w = (int)wv - 1; w >= 0; w--)
        if (wave_has[w]) { rec0 = wave_rec0[w]; have_rec0 = true; break; }
    }
  }
  if ((shadow || early_links) && head && lt != kNoSplit && ((int)lt < depth || early_links)) {
    const int ft = (int)lt;
    const u64 prefix = key >> (3 * (depth - ft));
    u32 lo = rec0;
    if (!have_rec0) {
      const u32 b = bucket_id(0, ft);
      u32 hi = bucket_base[b + 1];
      lo = bucket_base[b];
      while (lo < hi) {  // the record exists: every head's path is split down to depth - 1 (and an octant-7 leaf gains children, Q4)
        const u32 mid = (lo + hi) >> 1;
        if (rec_key[mid] < prefix) lo = mid + 1; else hi = mid;
      }
    }
    frontier_child = *n0_saved + 8u * lo;
    if (ft < depth) frontier = ft;
    if (early_links && c < ft) {  // this lane is the first head under the frontier node: it owns the pass-0 record
      link_level = ft;            // (the node's index comes out of the walk below)
      // level grid of the ray march: a split above the block level re-labels the whole cube of its node (as split_all_kernel marks it)
      if (grid_dirty && ft < kPoolGridBlockLevel) pool_grid_mark(grid_dirty, prefix, ft);
    }
  }
  // walk to the leaf (fillNodes, svo.cu:291-382), remembering the owned nodes and their child tiles
  u32 node_at[MAXD], child_at[MAXD];
#pragma unroll
  for (int l = 0; l < MAXD; l++) { node_at[l] = 0; child_at[l] = 0; }
  if (head) {
    // duplicates: the head of a run of equal keys is the lowest point index (stable sort)
    const unsigned char *v = colors + 3 * (size_t)point;
    const unsigned char cr = v[0], cg = v[1], cb = v[2];  // in flight during the walk
    // levels 1 .. skip belong to an earlier head (c) or were walked by the plan (up to the frontier): resume below them
    // (links that existed when the plan ran never change; replicas of one map have the same indices)
    int skip = 0;
    u32 base = 0, node = 0;
    if (leaf_start) {
      const int tt = lt == kNoSplit ? depth : (int)lt - 1;
      skip = c < tt ? c : tt;
      base = start_base;
    }
#pragma unroll
    for (int lvl = 1; lvl <= MAXD; lvl++) {
      if (lvl <= depth && lvl > skip) {
        node = base + ((u32)(key >> (3 * (depth - lvl))) & 7u);
        if (lvl == link_level) pool[2 * (size_t)node] = kFlag + (frontier_child & kMask);
        if (lvl < depth) {
          base = pool[2 * (size_t)node] & kMask;
          if (lvl == frontier) base = frontier_child & kMask;
          node_at[lvl] = node;
          child_at[lvl] = base;
        }
      }
    }
    const u32 word = blend_color256(pool[2 * (size_t)node + 1], cr, cg, cb);
    if (shadow) shadow_store(shadow, epoch, node, word);
    else pool[2 * (size_t)node + 1] = word;
    node_at[0] = node;  // (slot 0 is free: the root is not a lane's node)
  }
  if (shadow) apply_append(head, node_at[0]);
  __syncthreads();
  if (brick_mine) brick_ring_store(grid_dirty, brick_base + brick_off, brick_entry);
#pragma unroll
  for (int d = MAXD - 1; d >= 1; d--) {
    if (d < depth) {
      u32 wrote = kNoStraddler;
      if (head && c < d) {  // this lane owns its level-d prefix
        if (j != last_owner[d] || next_c < d) {
          if (shadow) {
            shadow_store(shadow, epoch, node_at[d], average_tile_deferred(pool, shadow, epoch, child_at[d]));
            wrote = node_at[d];
          } else {
            pool[2 * (size_t)node_at[d] + 1] = average_tile(pool, child_at[d]);
          }
        } else {  // the run continues in a later workgroup
          strad[2 * ((size_t)d * num_tiles + bid)] = node_at[d];
          strad[2 * ((size_t)d * num_tiles + bid) + 1] = child_at[d];
        }
      }
      if (shadow) apply_append(wrote != kNoStraddler, wrote);
      __syncthreads();
    }
  }
  // the sibling ring (see above).  At the END of the kernel: a divergent returning atomic between the barriers of the set-up left
  // the wavefront's lanes apart at the ballots that follow there (test_async_fusion_long_runs_of_duplicates_and_invalid_points)
  if (brick_mine && lt != kNoSplit && (int)lt < brick_node_level(brick_shift)) brick_sibling_list(grid_dirty, brick_entry);
  if (shadow && tid == 0) apply_nodes[(size_t)num_tiles * kFillThreads * (size_t)depth + bid] = apply_cnt;  // (behind the last level's barrier)
}

// straddling nodes deepest level first, then the root quirk (Q6) and the device-side size
constexpr int kStradThreads = 1024;
__global__ __launch_bounds__(kStradThreads) void mip_straddle_kernel(u32 *__restrict__ pool, const u32 *__restrict__ strad, int num_tiles,
                                                            int depth, const PlanCounts *__restrict__ counts,
                                                            int *__restrict__ d_size, u32 *__restrict__ grid_dirty,
                                                            int32_t *__restrict__ h_sizes, int *__restrict__ d_slot,
                                                            unsigned long long *__restrict__ shadow, u32 epoch, int keep_size,
                                                            const int *__restrict__ n_live) {
  SVO_HIGH_PRIO();
  // n_live (key-range sharded commit): the tiles past the rank's slice hold no straddler (fill_mip_local_kernel): not read
  const int list_stride = num_tiles;
  if (n_live) { const int live = (*n_live + kFillThreads - 1) / kFillThreads; num_tiles = live < num_tiles ? live : num_tiles; }
  // shadow != nullptr: deferred commit (see fill_mip_local_kernel); the list entries double as the apply list
  // keep_size != 0 (key-range sharded commit): the pool's size is set by keyrange_finish_kernel, from every rank's record count
  auto average = [&](u32 child_base) { return shadow ? average_tile_deferred(pool, shadow, epoch, child_base) : average_tile(pool, child_base); };
  auto store = [&](u32 node, u32 word) {
    if (shadow) shadow_store(shadow, epoch, node, word);
    else pool[2 * (size_t)node + 1] = word;
  };
  // a thread's list entries do not depend on the levels below, so those of the next level are fetched while
  // this level's tiles are averaged (one dependent load per level instead of two); kSlots entries per thread
  // in registers cover 8192 workgroups (2 M points: 1920x1080), longer lists fall back to the plain loop
  constexpr int kSlots = 8;
  const uint2 *list = reinterpret_cast<const uint2 *>(strad);
  const bool fits = num_tiles <= kSlots * kStradThreads;
  uint2 cur[kSlots], nxt[kSlots];
  auto fetch = [&](int d, uint2 *e) {
#pragma unroll
    for (int q = 0; q < kSlots; q++) {
      const int t = (int)threadIdx.x + kStradThreads * q;
      e[q] = (d >= 1 && t < num_tiles) ? list[(size_t)d * list_stride + t] : make_uint2(kNoStraddler, 0u);
    }
  };
  if (fits) fetch(depth - 1, cur);
  for (int d = depth - 1; d >= 1; d--) {
    if (fits) {
      fetch(d - 1, nxt);
#pragma unroll
      for (int q = 0; q < kSlots; q++)
        if (cur[q].x != kNoStraddler) store(cur[q].x, average(cur[q].y));
#pragma unroll
      for (int q = 0; q < kSlots; q++) cur[q] = nxt[q];
    } else {
      for (int t = (int)threadIdx.x; t < num_tiles; t += kStradThreads) {
        const uint2 e = list[(size_t)d * list_stride + t];
        if (e.x != kNoStraddler) store(e.x, average(e.y));
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (counts->any_valid) store(0u, average(0u));
    if (!keep_size) {
      const int size_now = *d_size + 8 * counts->total_records;
      *d_size = size_now;
      if (h_sizes) {  // the host learns the size from pinned memory behind the commit's event (PoolTracker)
        const int sl = *d_slot;
        h_sizes[sl] = size_now;
        *d_slot = (sl + 1) % 8;
      }
    }
  }
  // last kernel of the commit: every mark of this commit is in the bitmap (kernel boundaries); list the marked blocks
  // for the next render's refresh of the level grid (pool_grid.hpp)
  if (grid_dirty) pool_grid_compact(grid_dirty, kStradThreads);
}

// ---- straddlers in two tiers (round 3; direct commits) -----------------------------------------------------------
// mip_straddle_kernel above is ONE workgroup walking [level][tile] lists level by level: 23 us at 640x480 (600 tiles,
// 11 levels) and 127 us at 1920x1080 (4050 tiles, 13 levels) -- every level is a dependent round trip (children tiles
// -> average -> store -> barrier) over up to `tiles` entries.  Most straddlers are LOCAL: a deep node's run crosses one
// or two tile boundaries.  Tier 1: one workgroup per group of kStradGroup consecutive tiles finishes, level by level,
// every straddler whose run ends inside its group (<= kStradGroup entries per level: one lane each); a run that leaves
// the group -- at most ONE per level and group, since it covers every later tile of the group -- becomes a "super
// straddler".  Tier 2: the last group to arrive (agent-scope hand-off, cdna_hip_programming.md Guideline 16: tier-1 words are
// stored write-through, every wave drains, one lane takes the ticket; the last arriver acquires) finishes the super
// straddlers, <= groups per level, then does what the single workgroup did at its end (root quirk Q6, size, dirty
// list of the level grid).  Same values: a node is averaged after all of its touched children, whichever tier owns them.
constexpr int kStradGroup = 16;
constexpr int kStrad2Threads = 256;
__global__ __launch_bounds__(kStrad2Threads) void mip_straddle2_kernel(u32 *__restrict__ pool, const u32 *__restrict__ strad,
                                                              const int *__restrict__ strad_bc, u32 *__restrict__ sstrad,
                                                              unsigned *__restrict__ ticket, int num_tiles, int depth,
                                                              const PlanCounts *__restrict__ counts, int *__restrict__ d_size,
                                                              u32 *__restrict__ grid_dirty, int32_t *__restrict__ h_sizes,
                                                              int *__restrict__ d_slot) {
  SVO_HIGH_PRIO();
  __shared__ int bc[kStradGroup], mc[kStradGroup];
  __shared__ int is_last;
  const int tid = (int)threadIdx.x, g = (int)blockIdx.x, groups = (int)gridDim.x;
  const int t0 = g * kStradGroup, t1 = (t0 + kStradGroup < num_tiles) ? t0 + kStradGroup : num_tiles;
  const uint2 *list = reinterpret_cast<const uint2 *>(strad);
  uint2 *super = reinterpret_cast<uint2 *>(sstrad);  // [level][group]
  if (tid < kStradGroup) {
    bc[tid] = (t0 + tid < t1) ? strad_bc[2 * (t0 + tid)] : -1;
    mc[tid] = (t0 + tid < t1) ? strad_bc[2 * (t0 + tid) + 1] : 99;
  }
  __syncthreads();
  // ---- tier 1 (first wavefront; lane i = tile t0 + i)
  const int t = t0 + tid;
  for (int d = depth - 1; d >= 1; d--) {
    if (tid < kStradGroup) {
      uint2 sup = make_uint2(kNoStraddler, 0u);
      if (t < t1) {
        const uint2 e = list[(size_t)d * num_tiles + t];
        if (e.x != kNoStraddler) {
          // the run leaves tile t (that made it a straddler: bc[t] >= d).  It ends inside a later tile b as soon as
          // another level-d node starts there (mc[b] < d); otherwise it covers tile b to its end and goes on iff bc[b] >= d
          bool inside = false;
          for (int b = tid + 1; b < t1 - t0; b++)
            if (mc[b] < d || bc[b] < d) { inside = true; break; }
          if (inside) __hip_atomic_store(&pool[2 * (size_t)e.x + 1], average_tile_agent(pool, e.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else sup = e;
        }
      }
      // at most one lane holds a super straddler of this level: fold it to lane 0
      const unsigned long long m = __ballot(sup.x != kNoStraddler);
      if (m) {
        const int src = __ffsll((long long)m) - 1;
        sup.x = (u32)__shfl((int)sup.x, src); sup.y = (u32)__shfl((int)sup.y, src);
      }
      if (tid == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&super[(size_t)d * groups + g]),
                           ((unsigned long long)sup.y << 32) | sup.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the level's stores have left this CU before the next level reads them back
  }
  // ---- hand-off: tier-1 stores are write-through (sc1) and drained above by every wave; one lane takes the ticket
  if (tid == 0) {
    const unsigned k = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = k == (unsigned)groups - 1u;
  }
  __syncthreads();
  if (!is_last) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's stale lines once; plain loads below
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
  __syncthreads();
  // ---- tier 2: super straddlers deepest level first (a run that leaves its group may end anywhere: no locality left)
  for (int d = depth - 1; d >= 1; d--) {
    for (int q = tid; q < groups; q += kStrad2Threads) {
      const unsigned long long e = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&super[(size_t)d * groups + q]),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u32 node = (u32)e, child = (u32)(e >> 32);
      if (node != kNoStraddler) pool[2 * (size_t)node + 1] = average_tile(pool, child);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (tid == 0) {
    if (counts->any_valid) pool[1] = average_tile(pool, 0u);  // Q6
    const int size_now = *d_size + 8 * counts->total_records;
    *d_size = size_now;
    if (h_sizes) {  // the host learns the size from pinned memory behind the commit's event (PoolTracker)
      const int sl = *d_slot;
      h_sizes[sl] = size_now;
      *d_slot = (sl + 1) % 8;
    }
  }
  if (grid_dirty) pool_grid_compact(grid_dirty, kStrad2Threads);
}

// Second half of a deferred commit: everything the commit computed while the previous frame was being ray-marched
// becomes visible -- the colour words from the shadow array (apply_nodes: the leaf kernel's per-workgroup lists of the nodes
// it wrote -- a dense [level][key] table of mostly empty slots until round 4: 116 MB written and read per 1080p frame --; the
// straddler list; the root), and the links of the pass-0 split records to their (already initialised) child tiles.
// Every word is written once; ~0.6 M scattered 4-byte stores at 640x480.
__global__ __launch_bounds__(256) void commit_apply_kernel(u32 *__restrict__ pool, const unsigned long long *__restrict__ shadow,
                                                           const u32 *__restrict__ apply_nodes, int fill_tiles, int list_cap,
                                                           const u32 *__restrict__ strad, int strad_first, int strad_end,
                                                           const u32 *__restrict__ rec_front, const unsigned char *__restrict__ rec_pass,
                                                           const PlanCounts *__restrict__ counts, const u32 *__restrict__ n0_saved) {
  SVO_HIGH_PRIO();
  // the leaf kernel's lists: one per fill tile, `count` entries from its start (counts behind the lists)
  const u32 *list_counts = apply_nodes + (size_t)fill_tiles * list_cap;
  for (int t = (int)blockIdx.x; t < fill_tiles; t += (int)gridDim.x) {
    const u32 cnt = list_counts[t];
    const u32 *list = apply_nodes + (size_t)t * list_cap;
    for (u32 i = threadIdx.x; i < cnt; i += 256u) {
      const u32 node = list[i];
      pool[2 * (size_t)node + 1] = (u32)shadow[node];
    }
  }
  const long long stride = (long long)gridDim.x * 256, t0 = (long long)blockIdx.x * 256 + threadIdx.x;
  for (long long i = strad_first + t0; i < strad_end; i += stride) {
    const u32 node = strad[2 * i];
    if (node != kNoStraddler) pool[2 * (size_t)node + 1] = (u32)shadow[node];
  }
  const long long total = counts->total_records;
  const u32 n0 = *n0_saved;
  for (long long r = t0; r < total; r += stride)
    if (rec_pass[r] == 0) pool[2 * (size_t)rec_front[r]] = kFlag + ((n0 + 8u * (u32)r) & kMask);
  if (t0 == 0 && counts->any_valid) pool[1] = (u32)shadow[0];
}

// ----------------------------------------------------------------------------
// host driver
// ----------------------------------------------------------------------------
// ---- non-blocking size tracking of the asynchronous fusion ----------------------------------------
// Every commit copies the new size (4 bytes) to a pinned host slot behind an event.  The next plan polls
// the events: each completed one makes pool->size current up to that commit and releases its worst-case
// reservation, so the host learns the true size a frame or two late WITHOUT ever waiting for the device.
struct PoolTracker {
  static constexpr int kSlots = 8;  // == the modulus in mip_straddle_kernel
  int32_t *h_size = nullptr;  // pinned, device-visible [kSlots]: the commit's last kernel stores the new size itself
  int *d_slot = nullptr;      // device: slot the next commit writes (advances with `next` below, once per commit)
  int *d_struct = nullptr;    // device: the pool's size as the STRUCTURE chain sees it (svo_fuse_plan_structure; set from d_size by pool_structure_begin)
  hipEvent_t ev[kSlots];
  struct InFlight { int slot; int64_t bound; };
  InFlight q[kSlots];  // oldest first
  int count = 0, next = 0;
  int planned_ahead = 0;  // svo_fuse_plan_structure calls whose commit has not been enqueued yet (their reservations must survive pool_sync)
};

static PoolTracker *tracker_of(svoslam_pool *pool) { return reinterpret_cast<PoolTracker *>(pool->tracker); }

static int tracker_create(svoslam_pool *pool) {
  if (pool->tracker) return SVOSLAM_OK;
  PoolTracker *t = new PoolTracker();
  for (int i = 0; i < PoolTracker::kSlots; i++) t->ev[i] = nullptr;
  bool ok = hipHostMalloc((void **)&t->h_size, PoolTracker::kSlots * 4, hipHostMallocDefault) == hipSuccess;
  for (int i = 0; ok && i < PoolTracker::kSlots; i++) ok = hipEventCreateWithFlags(&t->ev[i], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMalloc((void **)&t->d_slot, 4) == hipSuccess && memset_sync(t->d_slot, 0, 4) == hipSuccess;
  ok = ok && hipMalloc((void **)&t->d_struct, 4) == hipSuccess && memset_sync(t->d_struct, 0, 4) == hipSuccess;
  pool->tracker = t;
  if (!ok) {  // whatever was created goes away again (events and pinned memory leaked on these paths)
    (void)hipGetLastError();
    pool_tracker_destroy(pool);
    return SVOSLAM_ERR_HIP;
  }
  return SVOSLAM_OK;
}

void pool_tracker_destroy(svoslam_pool *pool) {
  PoolTracker *t = pool ? tracker_of(pool) : nullptr;
  if (!t) return;
  for (int i = 0; i < PoolTracker::kSlots; i++) if (t->ev[i]) (void)hipEventDestroy(t->ev[i]);
  if (t->h_size) (void)hipHostFree(t->h_size);
  if (t->d_slot) (void)hipFree(t->d_slot);
  if (t->d_struct) (void)hipFree(t->d_struct);
  delete t;
  pool->tracker = nullptr;
}

// retire the completed readbacks (wait_all: block until every one has completed)
static int tracker_poll(svoslam_pool *pool, bool wait_all) {
  PoolTracker *t = tracker_of(pool);
  if (!t) return SVOSLAM_OK;
  while (t->count > 0) {
    const PoolTracker::InFlight f = t->q[0];
    if (wait_all) SVO_HIP(hipEventSynchronize(t->ev[f.slot]));
    else {
      const hipError_t e = hipEventQuery(t->ev[f.slot]);
      if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
      SVO_HIP(e);
    }
    pool->size = t->h_size[f.slot];
    pool->pending_bound -= f.bound;
    if (pool->pending_bound < 0) pool->pending_bound = 0;
    if (pool->pending > 0) pool->pending -= 1;
    for (int i = 1; i < t->count; i++) t->q[i - 1] = t->q[i];
    t->count--;
  }
  return SVOSLAM_OK;
}

// before a commit is enqueued: its last kernel stores the new size into h_size[next], which must not be the slot of a
// readback the host has not retired yet (with 8 readbacks in flight the 9th commit overwrote the oldest
// un-polled entry and pool->size jumped ahead and stepped back until the next drain)
static int tracker_make_room(svoslam_pool *pool) {
  PoolTracker *t = tracker_of(pool);
  if (!t || t->count < PoolTracker::kSlots) return SVOSLAM_OK;
  SVO_HIP(hipEventSynchronize(t->ev[t->q[0].slot]));  // the oldest readback is long done in practice
  return tracker_poll(pool, false);
}

// after a commit has been enqueued on `stream`
static int tracker_push(svoslam_pool *pool, int64_t bound, hipStream_t stream) {
  PoolTracker *t = tracker_of(pool);
  if (!t) return SVOSLAM_OK;
  if (t->count == PoolTracker::kSlots) {  // ring full: the oldest readback is long done in practice
    SVO_HIP(hipEventSynchronize(t->ev[t->q[0].slot]));
    SVO_TRY(tracker_poll(pool, false));
  }
  const int slot = t->next;
  t->next = (t->next + 1) % PoolTracker::kSlots;
  (void)slot;  // h_size[slot] is stored by mip_straddle_kernel (no copy operation between the commit and the next render)
  SVO_HIP(hipEventRecord(t->ev[slot], stream));
  t->q[t->count].slot = slot;
  t->q[t->count].bound = bound;
  t->count++;
  return SVOSLAM_OK;
}

int pool_sync(svoslam_pool *pool, hipStream_t stream) {
  if (!pool) return SVOSLAM_ERR_INVALID_ARG;
  if (tracker_of(pool) && tracker_of(pool)->count > 0) {  // the last commit's readback is the exact size
    SVO_TRY(tracker_poll(pool, true));
  } else if (pool->pending > 0 && pool->d_size) {
    int32_t sz = 0;
    SVO_HIP(hipMemcpyAsync(&sz, pool->d_size, 4, hipMemcpyDeviceToHost, stream));
    SVO_HIP(hipStreamSynchronize(stream));
    pool->size = sz;
  }
  pool->pending = 0;
  // what tracker_poll left of pending_bound is the reservation of plans that have no commit yet: zero unless the
  // structure chain is ahead of its commits
  if (!tracker_of(pool) || tracker_of(pool)->planned_ahead == 0) pool->pending_bound = 0;
  return SVOSLAM_OK;
}

static int ensure_device_size(svoslam_pool *pool, hipStream_t stream) {
  if (pool->d_size) return SVOSLAM_OK;
  SVO_HIP(hipMalloc((void **)&pool->d_size, 4));
  SVO_HIP(hipMemcpyAsync(pool->d_size, &pool->size, 4, hipMemcpyHostToDevice, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  return tracker_create(pool);
}

static int grow_pool(svoslam_pool *pool, int64_t need_nodes, hipStream_t stream, int64_t live_nodes = 0) {
  if (need_nodes > (int64_t)kMask + 1) return SVOSLAM_ERR_POOL_LIMIT;
  if (need_nodes <= pool->capacity) return SVOSLAM_OK;
  SVO_TRY(pool_sync(pool, stream));  // the copy below needs the exact size
  // live_nodes > size: tiles that plans of the structure chain have written ahead of their commits move along
  const int64_t copy_nodes = live_nodes > pool->size ? (live_nodes < pool->capacity ? live_nodes : pool->capacity) : pool->size;
  int64_t cap = (int64_t)pool->capacity * 2;
  if (cap < need_nodes) cap = need_nodes;
  if (cap > (int64_t)kMask + 1) cap = (int64_t)kMask + 1;
  u32 *fresh = nullptr;
  SVO_HIP(hipMalloc((void **)&fresh, (size_t)cap * 8));
  if (pool->d_data && copy_nodes > 0)
    SVO_HIP(hipMemcpyAsync(fresh, pool->d_data, (size_t)copy_nodes * 8, hipMemcpyDeviceToDevice, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  pool_accel_rebind(pool->d_data, fresh);
  if (pool->d_data) SVO_HIP(hipFree(pool->d_data));
  pool->d_data = fresh;
  pool->capacity = (int32_t)cap;
  return SVOSLAM_OK;
}

int pool_init(svoslam_pool *pool, int32_t capacity_nodes, hipStream_t stream) {
  if (!pool) return SVOSLAM_ERR_INVALID_ARG;
  if (capacity_nodes < 8) capacity_nodes = 8;
  pool->d_data = nullptr; pool->size = 0; pool->capacity = 0;
  pool->d_size = nullptr; pool->pending = 0; pool->pending_bound = 0; pool->tracker = nullptr;
  SVO_TRY(grow_pool(pool, capacity_nodes, stream));
  pool_accel_register(pool);
  SVO_HIP(hipMemsetAsync(pool->d_data, 0, 64, stream));  // initOctree, svo.cu:24-31
  pool->size = 8;
  return ensure_device_size(pool, stream);
}

// ---- growing the root ("a correct expandBySize that re-roots the GPU pool") -------------
// The reference's Octree::expandBySize (octree.cpp:362-378) multiplies size_ and, for a GPU-backed root,
// moves nothing (Q16): the old nodes then describe the wrong region.  Re-rooting the linear tree is local:
// the 8 children of the old root (nodes 0..7) move to a fresh tile at the end of the pool, nodes 0..7
// become the children of the NEW root -- all empty except the octant that holds the old root, which gets
// the children flag, the index of that tile and the mean colour its mip pass would give it.  Every other
// node keeps its index, so nothing below has to change.
__global__ void reroot_kernel(u32 *__restrict__ pool, int *__restrict__ d_size, int n0, int octant) {
  __shared__ uint2 old[8];
  uint2 *nodes = reinterpret_cast<uint2 *>(pool);
  if (threadIdx.x < 8) {
    old[threadIdx.x] = nodes[threadIdx.x];
    nodes[n0 + threadIdx.x] = old[threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x < 8) nodes[threadIdx.x] = make_uint2(0u, 0u);  // initOctree, svo.cu:24-31
  __syncthreads();
  if (threadIdx.x == 0) {
    nodes[octant] = make_uint2(kFlag | ((u32)n0 & kMask), average_tile(pool, (u32)n0));
    if (d_size) *d_size = n0 + 8;
  }
}

// One doubling of the root cube towards `toward` (per axis: the side on which toward lies relative to the
// centre).  center / edge are updated to the new root; the caller fuses with max_depth + 1 from now on to keep
// its resolution.  Blocking.
int pool_expand(svoslam_pool *pool, float center[3], float *edge, const float toward[3], hipStream_t stream) {
  if (!pool || !pool->d_data || !center || !edge || !toward || !(*edge > 0.0f)) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipDeviceSynchronize());
  SVO_TRY(pool_sync(pool, stream));
  if ((int64_t)pool->size + 8 > (int64_t)kMask + 1) return SVOSLAM_ERR_POOL_LIMIT;
  SVO_TRY(grow_pool(pool, (int64_t)pool->size + 8, stream));
  // the old root becomes the child on the side AWAY from the growth: octant bit = old centre > new centre
  int octant = 0;
  float nc[3];
  for (int k = 0; k < 3; k++) {
    const bool grow_plus = toward[k] > center[k];
    nc[k] = center[k] + (grow_plus ? *edge : -*edge);
    if (center[k] > nc[k]) octant |= 1 << k;
  }
  reroot_kernel<<<1, 64, 0, stream>>>(pool->d_data, pool->d_size, pool->size, octant);
  SVO_LAUNCH_CHECK();
  pool_accel_invalidate(pool, 0, false);  // every cell of the level grid now lies one level deeper (the words are the same ones)
  SVO_HIP(hipStreamSynchronize(stream));
  pool->size += 8;
  for (int k = 0; k < 3; k++) center[k] = nc[k];
  *edge = *edge * 2.0f;
  return SVOSLAM_OK;
}

// back to initOctree (8 zeroed root children) keeping the allocation, so recorded launch graphs stay valid.  Blocking.
int pool_reset(svoslam_pool *pool, hipStream_t stream) {
  if (!pool || !pool->d_data) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipDeviceSynchronize());
  SVO_TRY(pool_sync(pool, stream));  // drains the size tracker
  SVO_HIP(memset_sync(pool->d_data, 0, 64));
  pool_accel_invalidate(pool, -1);
  pool->size = 8; pool->pending = 0; pool->pending_bound = 0;
  if (tracker_of(pool)) tracker_of(pool)->planned_ahead = 0;  // plans of the structure chain whose commit never came (an error mid-run) hold no reservation any more
  if (pool->d_size) SVO_HIP(hipMemcpy(pool->d_size, &pool->size, 4, hipMemcpyHostToDevice));
  return SVOSLAM_OK;
}

int pool_reserve(svoslam_pool *pool, int32_t capacity_nodes, hipStream_t stream) {
  if (!pool) return SVOSLAM_ERR_INVALID_ARG;
  return grow_pool(pool, capacity_nodes, stream);
}

// ---- checkpoint / resume -------------------------------------------------------------
// The pool IS the reference's linear tree (the layout OctreeNode::pushToGPU assembles, octree.cpp:41-79:
// 2-word nodes, 0x40000000 children flag, 30-bit child index), so a checkpoint is the node words behind
// a 64-byte header.  The reference's own (de)serialiser is unfinished (addToLinearTree never sets the
// flag, octree.cpp:138-160), so there is no reference byte stream to match.
struct PoolFileHeader {
  char magic[8];        // "SVOPOOL1"
  uint32_t version;     // 1
  int32_t num_nodes;
  float center[3];
  float edge_length;    // half edge of the root cube
  int32_t max_depth;
  uint32_t reserved;
  uint64_t checksum;    // FNV-1a over the node words
  uint8_t pad[16];
};
static_assert(sizeof(PoolFileHeader) == 64, "header is 64 bytes");

static uint64_t fnv1a_words(const u32 *w, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) h = (h ^ (uint64_t)w[i]) * 1099511628211ull;
  return h;
}

int pool_save(svoslam_pool *pool, const char *path, const float center[3], float edge, int depth, hipStream_t stream) {
  if (!pool || !path || !center || !pool->d_data) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipDeviceSynchronize());  // commits may be in flight on other streams
  SVO_TRY(pool_sync(pool, stream));
  const size_t words = 2 * (size_t)pool->size;
  std::vector<u32> host(words);
  SVO_HIP(hipMemcpy(host.data(), pool->d_data, words * 4, hipMemcpyDeviceToHost));
  PoolFileHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "SVOPOOL1", 8);
  h.version = 1; h.num_nodes = pool->size;
  h.center[0] = center[0]; h.center[1] = center[1]; h.center[2] = center[2];
  h.edge_length = edge; h.max_depth = depth;
  h.checksum = fnv1a_words(host.data(), words);
  FILE *f = fopen(path, "wb");
  if (!f) return SVOSLAM_ERR_IO;
  const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && (words == 0 || fwrite(host.data(), 4, words, f) == words);
  return (fclose(f) == 0 && ok) ? SVOSLAM_OK : SVOSLAM_ERR_IO;
}

// Replaces the pool's contents by num_nodes host nodes (a linear tree in the reference format): validates the child
// pointers, waits for everything in flight, and resets ALL size bookkeeping -- host size, device-resident size,
// reservations of asynchronous fusions.  Blocking.
int pool_set_nodes(svoslam_pool *pool, const uint32_t *h_words, int32_t num_nodes, hipStream_t stream) {
  if (!pool || !h_words || num_nodes < 8 || (num_nodes & 7) != 0) return SVOSLAM_ERR_INVALID_ARG;
  for (size_t i = 0; i < (size_t)num_nodes; i++) {  // every child tile must lie inside the pool
    const u32 w0 = h_words[2 * i];
    if ((w0 & kFlag) && ((w0 & kMask) + 8u > (u32)num_nodes || ((w0 & kMask) & 7u))) return SVOSLAM_ERR_FORMAT;
  }
  SVO_HIP(hipDeviceSynchronize());
  if (!pool->d_data) SVO_TRY(pool_init(pool, num_nodes, stream));
  SVO_TRY(pool_sync(pool, stream));
  SVO_TRY(grow_pool(pool, num_nodes, stream));
  SVO_HIP(hipMemcpy(pool->d_data, h_words, (size_t)num_nodes * 8, hipMemcpyHostToDevice));
  pool_accel_invalidate(pool);
  pool->size = num_nodes;
  pool->pending = 0; pool->pending_bound = 0;
  if (tracker_of(pool)) tracker_of(pool)->planned_ahead = 0;
  if (pool->d_size) SVO_HIP(hipMemcpy(pool->d_size, &pool->size, 4, hipMemcpyHostToDevice));
  return SVOSLAM_OK;
}

// dst becomes a byte-identical replica of src (same nodes, same size, at least the same capacity).  Blocking: waits
// for the device (either pool may have work in flight on any stream).
int pool_copy(svoslam_pool *dst, svoslam_pool *src, hipStream_t stream) {
  if (!dst || !src || dst == src || !src->d_data) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipDeviceSynchronize());
  SVO_TRY(pool_sync(src, stream));
  if (!dst->d_data) SVO_TRY(pool_init(dst, src->capacity, stream));
  SVO_TRY(pool_sync(dst, stream));
  if (dst->capacity < src->capacity) {
    dst->size = 0;  // nothing worth copying over to the larger allocation
    SVO_TRY(grow_pool(dst, src->capacity, stream));
  }
  SVO_HIP(hipMemcpy(dst->d_data, src->d_data, (size_t)src->size * 8, hipMemcpyDeviceToDevice));
  pool_accel_invalidate(dst);
  SVO_HIP(hipDeviceSynchronize());
  dst->size = src->size;
  dst->pending = 0; dst->pending_bound = 0;
  SVO_TRY(ensure_device_size(dst, stream));
  SVO_HIP(hipMemcpy(dst->d_size, &dst->size, 4, hipMemcpyHostToDevice));
  return SVOSLAM_OK;
}

int pool_load(svoslam_pool *pool, const char *path, float center[3], float *edge, int *depth, hipStream_t stream) {
  if (!pool || !path) return SVOSLAM_ERR_INVALID_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) return SVOSLAM_ERR_IO;
  PoolFileHeader h;
  if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SVOPOOL1", 8) != 0 || h.version != 1 || h.num_nodes < 8 ||
      (h.num_nodes & 7) != 0) {
    fclose(f);
    return SVOSLAM_ERR_FORMAT;
  }
  const size_t words = 2 * (size_t)h.num_nodes;
  std::vector<u32> host(words);
  const bool ok = fread(host.data(), 4, words, f) == words;
  fclose(f);
  if (!ok || fnv1a_words(host.data(), words) != h.checksum) return SVOSLAM_ERR_FORMAT;
  SVO_TRY(pool_set_nodes(pool, host.data(), h.num_nodes, stream));
  if (center) { center[0] = h.center[0]; center[1] = h.center[1]; center[2] = h.center[2]; }
  if (edge) *edge = h.edge_length;
  if (depth) *depth = h.max_depth;
  return SVOSLAM_OK;
}

static int reserve_common(svoslam_workspace *ws, int n, int depth) {
  const size_t nn = (size_t)(n > 0 ? n : 1);
  SVO_TRY(ws->keys_a.reserve(nn * 8));
  SVO_TRY(ws->keys_b.reserve(nn * 8));
  SVO_TRY(ws->vals_a.reserve(nn * 4));
  SVO_TRY(ws->vals_b.reserve(nn * 4));
  const size_t tiles = (size_t)cdiv(n, 256) + 1;
  SVO_TRY(ws->tile_hist.reserve(256 * tiles * 4));
  SVO_TRY(ws->reserve_small());  // (zeroed when created: any_valid and the plan's arrival ticket start at zero)
  SVO_TRY(ws->leaf_t.reserve(nn));
  SVO_TRY(ws->leaf_f.reserve(nn * 4));
  SVO_TRY(ws->path_nodes.reserve(nn * 4 * (size_t)(depth > 1 ? depth - 1 : 1)));
  if (!ws->h_counts) SVO_HIP(hipHostMalloc((void **)&ws->h_counts, sizeof(PlanCounts), hipHostMallocDefault));
  return SVOSLAM_OK;
}

// layout of ws->small (u32 words): [0,256) totals | [256,513) bucket_base | [520..) PlanCounts | [640] any_valid | [648] n0 | [656] plan ticket (any_valid and the ticket start at 0 and are left at 0 by every plan)
static inline u32 *small_totals(svoslam_workspace *ws) { return ws->small.as<u32>(); }
static inline u32 *small_bucket_base(svoslam_workspace *ws) { return ws->small.as<u32>() + 256; }
static inline PlanCounts *small_counts(svoslam_workspace *ws) { return reinterpret_cast<PlanCounts *>(ws->small.as<u32>() + 520); }
static inline int *small_any(svoslam_workspace *ws) { return reinterpret_cast<int *>(ws->small.as<u32>() + 640); }
static inline u32 *small_n0(svoslam_workspace *ws) { return ws->small.as<u32>() + 648; }  // deferred commit: first new tile
static inline unsigned *small_ticket(svoslam_workspace *ws) { return ws->small.as<u32>() + 656; }  // plan_scan_finish_kernel's arrival count
static inline unsigned *small_strad_ticket(svoslam_workspace *ws, int slot) { return ws->small.as<u32>() + 664 + 8 * slot; }  // mip_straddle2_kernel's (zero between launches)

// keys of the n inputs are in ws->keys_a
// The blocking insert's sort (round 5): one packed word per point where key and index fit 64 bits -- and the key ALONE on the
// voxel-grid path, whose colours are paired by position (Q20) -- instead of (8-byte key, 4-byte index) pairs through 8-bit digits:
// config 5's 318 M voxels moved 12 bytes x 2 x 7 passes, now 8 x 2 x 6.  -1: the pair sort (svoslam_config.sort_pairs, or no room).
static int blocking_sort_idx_bits(int n, int depth, bool keys_only);

static int svo_insert(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, const void *d_colors, bool vec4,
                      bool color_by_position, svoslam_fuse_stats *stats, hipStream_t stream, long long sort_token, int idx_bits) {
  pool_accel_invalidate(pool, depth, false);  // the blocking path does not track what it touches: the next render rebuilds the level grid
  SVO_TRY(pool_sync(pool, stream));
  u64 *skey = nullptr; u32 *sidx = nullptr;
  long long tk = -1;  // stage brackets of the blocking path (bench.py's mesh configurations; off by default).  The sort's bracket
  // was opened by the caller in front of its key kernel (sort_token)
  if (idx_bits >= 0)
    SVO_TRY(radix_sort_packed_ex(ws, n, 3 * depth + 1, idx_bits, radix_packed_digit_bits_for(n), false, !color_by_position, stream, &skey, &sidx));
  else
    SVO_TRY(radix_sort_pairs(ws, n, 3 * depth + 1, stream, &skey, &sidx));
  (void)stage_end(kStageFuseSort, sort_token, stream);
  const int tiles = (int)cdiv(n, 256);
  unsigned char *leaf_t = ws->leaf_t.as<unsigned char>();
  u32 *leaf_f = ws->leaf_f.as<u32>();
  u32 *tile_hist = ws->tile_hist.as<u32>();
  const int ptiles = (int)cdiv(n, kPlanThreads);
  (void)stage_begin(kStageFusePlan, stream, &tk);
  plan_count_kernel<<<xcd_grid(ptiles), kPlanThreads, 0, stream>>>(skey, n, depth, pool->d_data, leaf_t, leaf_f, nullptr, tile_hist, ptiles, small_any(ws));
  plan_scan_finish_kernel<<<256, 256, 0, stream>>>(tile_hist, ptiles, small_totals(ws), small_ticket(ws), small_bucket_base(ws),
                                                   small_counts(ws), small_any(ws), nullptr, nullptr);
  SVO_LAUNCH_CHECK();
  SVO_HIP(hipMemcpyAsync(ws->h_counts, small_counts(ws), sizeof(PlanCounts), hipMemcpyDeviceToHost, stream));
  SVO_HIP(hipStreamSynchronize(stream));  // the one host round trip of a fused frame
  const PlanCounts hc = *ws->h_counts;
  const int total = hc.total_records;
  const int32_t size0 = pool->size;
  if (stats) {
    stats->num_points = n; stats->num_split = total; stats->pool_size_before = size0;
    for (int p = 0; p <= SVOSLAM_MAX_DEPTH; p++) stats->pass_sizes[p] = hc.pass_start[p + 1] - hc.pass_start[p];
  }
  if (total == 0) (void)stage_end(kStageFusePlan, tk, stream);
  if (total > 0) {
    SVO_TRY(grow_pool(pool, (int64_t)size0 + 8ll * total, stream));
    SVO_TRY(ws->rec_key.reserve((size_t)total * 8));
    SVO_TRY(ws->rec_front.reserve((size_t)total * 4));
    u64 *rec_key = ws->rec_key.as<u64>();
    u32 *rec_front = ws->rec_front.as<u32>();
    plan_emit_kernel<<<xcd_grid(ptiles), kPlanThreads, 0, stream>>>(skey, n, depth, leaf_t, leaf_f, small_bucket_base(ws), tile_hist, ptiles, rec_key, rec_front, nullptr, nullptr);
    (void)stage_end(kStageFusePlan, tk, stream);
    (void)stage_begin(kStageFuseCommit, stream, &tk);
    for (int p = 0; p <= SVOSLAM_MAX_DEPTH; p++) {  // expandTreeAtKeys, svo.cu:278-289
      const int begin = hc.pass_start[p], end = hc.pass_start[p + 1];
      if (end > begin)
        split_pass_kernel<<<cdiv(end - begin, 256), 256, 0, stream>>>(rec_key, rec_front, begin, end, p, pool->d_data, (u32)size0);
    }
    if (pool->d_size) pool_size_update_kernel<<<1, 64, 0, stream>>>(pool->d_size, small_counts(ws));
    SVO_LAUNCH_CHECK();
    (void)stage_end(kStageFuseCommit, tk, stream);
    pool->size = size0 + 8 * total;
  }
  if (stats) stats->pool_size_after = pool->size;
  if (hc.any_valid) {
    (void)stage_begin(kStageFuseCommit, stream, &tk);  // (a second entry of the stage: the reader sums them)
    u32 *path_nodes = ws->path_nodes.as<u32>();
    SVO_TRY(ws->leaf_start.reserve((size_t)n));  // (the async path's buffer, free in the blocking one: one byte per key here)
    unsigned char *leaf_c = ws->leaf_start.as<unsigned char>();
    if (vec4)
      fill_kernel<true><<<tiles, 256, 0, stream>>>(skey, sidx, n, depth, leaf_t, d_colors, color_by_position, pool->d_data, path_nodes, leaf_c);
    else
      fill_kernel<false><<<tiles, 256, 0, stream>>>(skey, sidx, n, depth, leaf_t, d_colors, color_by_position, pool->d_data, path_nodes, leaf_c);
    for (int d = depth - 1; d >= 1; d--)  // mipmapNodes, svo.cu:450-465
      mip_level_kernel<<<cdiv(n, 1024), 256, 0, stream>>>(n, d, leaf_c, path_nodes, pool->d_data);
    mip_root_kernel<<<1, 64, 0, stream>>>(pool->d_data, small_counts(ws));
    SVO_LAUNCH_CHECK();
    (void)stage_end(kStageFuseCommit, tk, stream);
  }
  return SVOSLAM_OK;
}

// worst-case number of split records of one call: at depth d at most min(8^d, n) distinct prefixes can be
// split (d < D), plus at most n octant-7 leaves (Q4)
static int64_t max_records(int n, int depth) {
  int64_t r = n;
  for (int d = 1; d < depth; d++) {
    const int64_t cells = d >= 11 ? (int64_t)1 << 62 : (int64_t)1 << (3 * d);
    r += cells < n ? cells : n;
  }
  return r;
}

// Asynchronous fusion: same kernels up to the plan, then every split in one launch (split_all_kernel), no
// readback.  The host only knows an upper bound of the pool size; capacity is kept ahead of it.
// ---- asynchronous fusion in three phases --------------------------------------------------------
// sort:   keys + radix sort of the new points.  Touches only the workspace.
// plan:   per sorted leaf, where the pool's current tree ends and which nodes must be split (records in
//         reference order).  READS the pool's structure words; must follow the previous call's commit.
// commit: all splits in one launch, leaf blend, mip levels.  WRITES the pool.
// A caller that renders between fusions can run sort + plan of frame k+1 on another stream while frame
// k is ray-marched (the pool is only read), and commit when the render is done; results are those of
// the one-call form.  Phases of one fusion share one workspace; concurrent fusions need their own.
static int packed_idx_bits(int n) {
  int b = 1;
  while ((1ll << b) < (long long)n) b++;
  return b;
}
// svoslam_config.sort_pairs = 1: the (key, index) pair sort of round 1 for every fusion (tests)
static bool sort_pairs_forced() { return config().sort_pairs != 0; }
static int blocking_sort_idx_bits(int n, int depth, bool keys_only) {
  if (sort_pairs_forced()) return -1;
  if (keys_only) return 0;
  const int b = packed_idx_bits(n);
  return 3 * depth + 1 + b <= 64 ? b : -1;
}

// buffers of the asynchronous phases (plan + commit) for a batch of n sorted keys
static int reserve_async(svoslam_workspace *ws, int n, int depth) {
  const int64_t rmax = max_records(n, depth);
  if (rmax > 0x7FFFFFFFll / 8) return SVOSLAM_ERR_POOL_LIMIT;
  SVO_TRY(reserve_common(ws, n, depth));
  SVO_TRY(ws->rec_key.reserve((size_t)rmax * 8));
  SVO_TRY(ws->rec_front.reserve((size_t)rmax * 4));
  SVO_TRY(ws->rec_pass.reserve((size_t)rmax));
  SVO_TRY(ws->leaf_rec0.reserve((size_t)n * 4));
  SVO_TRY(ws->leaf_start.reserve((size_t)n * 4));
  return SVOSLAM_OK;
}

// Sorted keys from ELSEWHERE (round 3: another rank of a frame-sharded session sorted this frame and all-gathered the
// result): the workspace adopts d_keys[n] (ascending depth-D Morton keys with their leading 1, invalid points = 1) and
// d_idx[n] (the point index of each key: the colour svo_fuse_commit looks up; equal keys in ascending index order, as the
// stable sort leaves them) as the outcome of its sort phase.  svo_fuse_plan / _commit follow as after svo_fuse_sort.  The
// arrays stay the caller's and must stay valid until the commit has run.
int svo_fuse_adopt_sorted(svoslam_workspace *ws, const unsigned long long *d_keys, const uint32_t *d_idx, int n, int depth) {
  if (!ws || n < 0 || (n > 0 && (!d_keys || !d_idx))) return SVOSLAM_ERR_INVALID_ARG;
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  ws->sorted_keys = nullptr; ws->sorted_idx = nullptr; ws->planned_n = -1;
  if (n == 0) return SVOSLAM_OK;
  SVO_TRY(reserve_async(ws, n, depth));
  SVO_TRY(ws->tile_hist.reserve((size_t)256 * ((size_t)cdiv(n, kPlanThreads) + 1) * 4));
  ws->sorted_keys = d_keys; ws->sorted_idx = d_idx;
  return SVOSLAM_OK;
}

// the outcome of this workspace's sort phase, copied out (device to device) for an exchange: d_keys_out[n], d_idx_out[n]
int svo_fuse_export_sorted(svoslam_workspace *ws, int n, unsigned long long *d_keys_out, uint32_t *d_idx_out, hipStream_t stream) {
  if (!ws || n < 0 || (n > 0 && (!d_keys_out || !d_idx_out))) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  if (!ws->sorted_keys || !ws->sorted_idx) return SVOSLAM_ERR_INVALID_ARG;  // svo_fuse_sort has not run on this workspace
  SVO_HIP(hipMemcpyAsync(d_keys_out, ws->sorted_keys, (size_t)n * 8, hipMemcpyDeviceToDevice, stream));
  SVO_HIP(hipMemcpyAsync(d_idx_out, ws->sorted_idx, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  return SVOSLAM_OK;
}

// keys + sort of one batch: from a point array (fs == nullptr) or straight from a depth image and a device-resident pose
static int fuse_sort_impl(svoslam_workspace *ws, const float *d_points, const FrameSource *fs, int n, int depth, const float center[3],
                          float edge, float *d_bbox7, hipStream_t stream) {
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  ws->sorted_keys = nullptr; ws->sorted_idx = nullptr; ws->planned_n = -1;
  if (n == 0) return SVOSLAM_OK;
  SVO_TRY(reserve_async(ws, n, depth));
  const int key_bits = 3 * depth + 1, idx_bits = packed_idx_bits(fs ? fs->w * fs->h : n);  // (a row band carries whole-image pixel indices)
  const bool packed = key_bits + idx_bits <= 64 && !sort_pairs_forced();
  if (!packed && fs) return SVOSLAM_ERR_INVALID_ARG;  // (callers fall back to the stand-alone kernels + svo_fuse_sort)
  const int tiles = radix_packed_tiles(n);
  if (radix_packed_tile() != kKeysThreads * kKeysIPT) return SVOSLAM_ERR_INVALID_ARG;  // the first histogram is per sort tile
  unsigned *ticket = nullptr;
  float *partial = nullptr;
  if (packed) {
    SVO_TRY(ws->tile_hist.reserve(radix_packed_hist_words(n) * 4));
    if (fs && d_bbox7) {
      const size_t need = (size_t)(8 + 7 * (size_t)tiles) * 4;
      if (need > ws->frame_bbox.bytes) {
        SVO_TRY(ws->frame_bbox.reserve(need));
        SVO_HIP(memset_sync(ws->frame_bbox.ptr, 0, ws->frame_bbox.bytes));  // the ticket starts at zero (blocking; once per size)
      }
      ticket = ws->frame_bbox.as<unsigned>();
      partial = ws->frame_bbox.as<float>() + 8;
    }
  }
  u64 *skey = nullptr; u32 *sidx = nullptr;
  auto enqueue = [&]() -> int {
    if (packed) {
      const int bits0 = radix_packed_first_bits(key_bits);
      if (fs)
        keys_packed_kernel<true><<<tiles, kKeysThreads, 0, stream>>>(nullptr, *fs, n, depth, center[0], center[1], center[2], edge, idx_bits, bits0,
                                                            ws->keys_a.as<u64>(), ws->tile_hist.as<u32>(), partial, ticket, d_bbox7);
      else
        keys_packed_kernel<false><<<tiles, kKeysThreads, 0, stream>>>(d_points, FrameSource{}, n, depth, center[0], center[1], center[2], edge,
                                                             idx_bits, bits0, ws->keys_a.as<u64>(), ws->tile_hist.as<u32>(), nullptr,
                                                             nullptr, nullptr);
      SVO_TRY(radix_sort_packed(ws, n, key_bits, idx_bits, stream, &skey, &sidx));
    } else {
      compute_keys_kernel<3><<<cdiv(n, 256), 256, 0, stream>>>(d_points, n, depth, center[0], center[1], center[2], edge, ws->keys_a.as<u64>());
      SVO_TRY(radix_sort_pairs(ws, n, key_bits, stream, &skey, &sidx));
    }
    SVO_LAUNCH_CHECK();
    return SVOSLAM_OK;
  };
  GraphKey key;
  key.add(fs ? (const void *)fs->depth : (const void *)d_points).add((unsigned long long)n).add((unsigned long long)depth)
     .addf(center[0]).addf(center[1]).addf(center[2]).addf(edge).add(ws->layout_hash()).add(fs ? fs->pose : nullptr).add(d_bbox7)
     .add(ws->frame_bbox.ptr).add((unsigned long long)(fs ? fs->first : 0));
  {
    StageScope timed(kStageFuseSort, stream);
    SVO_TRY(ws->g_sort.run(key, stream, enqueue));
  }
  if (!skey) {  // replayed: where the recorded sort leaves its result
    if (packed) SVO_TRY(radix_sort_packed_output(ws, key_bits, &skey, &sidx));
    else SVO_TRY(radix_sort_output(ws, n, key_bits, &skey, &sidx));
  }
  ws->sorted_keys = skey; ws->sorted_idx = sidx;
  return SVOSLAM_OK;
}

int svo_fuse_sort(svoslam_workspace *ws, const float *d_points, int n, int depth, const float center[3], float edge,
                  hipStream_t stream) {
  if (!ws || n < 0 || (n > 0 && !d_points)) return SVOSLAM_ERR_INVALID_ARG;
  return fuse_sort_impl(ws, d_points, nullptr, n, depth, center, edge, nullptr, stream);
}

// The sort phase fed by a raw depth image: generateVertexMap + transformVertexMap(d_pose) + computePointCloudBoundingBox
// + computeKeys (main.cpp:39-44, svo.cu:33-66) in one launch, no point cloud in memory, then the sort.  d_bbox7 (optional):
// {min xyz, max xyz, any} as svoslam_point_cloud_bbox_device writes it.  Needs 3 depth + 1 + ceil(log2(w h)) <= 64
// (SVOSLAM_ERR_INVALID_ARG otherwise: use the stand-alone calls).
int svo_fuse_sort_frame(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int w, int h, float fx, float fy, int depth,
                        const float center[3], float edge, float *d_bbox7, hipStream_t stream) {
  if (!ws || !d_depth || !d_pose || w <= 0 || h <= 0 || (long long)w * h > 0x7FFFFFFFll) return SVOSLAM_ERR_INVALID_ARG;
  FrameSource fs;
  fs.depth = d_depth; fs.pose = d_pose; fs.w = w; fs.h = h; fs.fx = fx; fs.fy = fy;
  return fuse_sort_impl(ws, nullptr, &fs, w * h, depth, center, edge, d_bbox7, stream);
}

// The same for ONE ROW BAND of the frame ("each GPU computes keys for its band"): rows [first_row, first_row + rows);
// the sorted point indices are those of the WHOLE image, so that bands merge into the order the one-GPU sort produces.
int svo_fuse_sort_frame_band(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int w, int h, float fx, float fy, int depth,
                             const float center[3], float edge, int first_row, int rows, hipStream_t stream) {
  if (!ws || !d_depth || !d_pose || w <= 0 || h <= 0 || (long long)w * h > 0x7FFFFFFFll) return SVOSLAM_ERR_INVALID_ARG;
  if (first_row < 0 || rows < 0 || first_row + rows > h) return SVOSLAM_ERR_INVALID_ARG;
  FrameSource fs;
  fs.depth = d_depth; fs.pose = d_pose; fs.w = w; fs.h = h; fs.fx = fx; fs.fy = fy; fs.first = first_row * w;
  return fuse_sort_impl(ws, nullptr, &fs, rows * w, depth, center, edge, nullptr, stream);
}

// k-way merge of sorted (key, point index) lists whose index ranges are disjoint and ascending with the list number (row
// bands): element e of list a lands at (its place in a) + sum over b < a of #{keys of b <= e.key} + sum over b > a of
// #{keys of b < e.key} -- equal keys stay in ascending index order, i.e. exactly the order one stable sort of the whole
// image leaves them in (R1: the lowest point index of a run of equal keys is its head).  One thread per element, binary
// searches in L2-resident lists.
constexpr int kMergeMaxLists = 16;
struct MergeArgs { const u64 *keys[kMergeMaxLists]; const u32 *idx[kMergeMaxLists]; int offset[kMergeMaxLists + 1]; int lists; };
__global__ __launch_bounds__(256) void merge_sorted_kernel(MergeArgs A, u64 *__restrict__ out_keys, u32 *__restrict__ out_idx) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= A.offset[A.lists]) return;
  int a = 0;
  while (g >= A.offset[a + 1]) a++;
  const int p = g - A.offset[a];
  const u64 key = A.keys[a][p];
  int rank = p;
  for (int b = 0; b < A.lists; b++) {
    if (b == a) continue;
    const u64 *kb = A.keys[b];
    int lo = 0, hi = A.offset[b + 1] - A.offset[b];
    if (b < a) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (kb[mid] <= key) lo = mid + 1; else hi = mid; } }  // upper bound
    else       { while (lo < hi) { const int mid = (lo + hi) >> 1; if (kb[mid] < key) lo = mid + 1; else hi = mid; } }   // lower bound
    rank += lo;
  }
  out_keys[rank] = key;
  out_idx[rank] = A.idx[a][p];
}

int svo_fuse_merge_sorted(const unsigned long long *const *d_keys, const uint32_t *const *d_idx, const int32_t *counts, int lists,
                          unsigned long long *d_keys_out, uint32_t *d_idx_out, hipStream_t stream) {
  if (!d_keys || !d_idx || !counts || lists < 1 || lists > kMergeMaxLists || !d_keys_out || !d_idx_out) return SVOSLAM_ERR_INVALID_ARG;
  MergeArgs A;
  A.lists = lists; A.offset[0] = 0;
  for (int b = 0; b < lists; b++) {
    if (counts[b] < 0 || (counts[b] > 0 && (!d_keys[b] || !d_idx[b]))) return SVOSLAM_ERR_INVALID_ARG;
    A.keys[b] = d_keys[b]; A.idx[b] = d_idx[b];
    if ((long long)A.offset[b] + counts[b] > 0x7FFFFFFFll) return SVOSLAM_ERR_INVALID_ARG;
    A.offset[b + 1] = A.offset[b] + counts[b];
  }
  if (A.offset[lists] == 0) return SVOSLAM_OK;
  merge_sorted_kernel<<<cdiv(A.offset[lists], 256), 256, 0, stream>>>(A, d_keys_out, d_idx_out);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

__global__ void pool_structure_begin_kernel(int *__restrict__ d_struct, const int *__restrict__ d_size) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *d_struct = *d_size;
}

// The structure chain (for callers that do not render every frame: one rank of a frame-sharded session).  A plan reads
// only the pool's STRUCTURE words (children flags and links), and those are final once the splits of the previous frame
// are in -- its leaf blend and mip levels write colour words only.  svo_fuse_plan_structure = plan + split_all (links
// included) in one go, numbering its tiles from a size that follows the PLANS (d_struct); it needs no commit to have
// finished, only the previous svo_fuse_plan_structure (same stream, or ordered by the caller).  The commits -- leaf kernel +
// straddlers, as after svo_fuse_split_early -- follow in order on another stream.  Frames then cost the longer of the two
// chains instead of their sum.  The caller keeps a ray march of frame k away from the structure of frame k+1: no
// svo_fuse_plan_structure(k+1) before the march of frame k is done.  pool_structure_begin: once, after everything
// earlier on the pool has completed in stream order (d_struct := the pool's size).
int pool_structure_begin(svoslam_pool *pool, hipStream_t stream) {
  if (!pool) return SVOSLAM_ERR_INVALID_ARG;
  if (pool->size == 0) SVO_TRY(pool_init(pool, 8, stream));
  SVO_TRY(ensure_device_size(pool, stream));
  pool_structure_begin_kernel<<<1, 64, 0, stream>>>(tracker_of(pool)->d_struct, pool->d_size);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

static int fuse_plan_impl(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, bool structure, hipStream_t stream) {
  if (!ws || !pool || n < 0) return SVOSLAM_ERR_INVALID_ARG;
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  if (pool->size == 0) SVO_TRY(pool_init(pool, 8, stream));
  ws->planned_n = -1;
  ws->early_split_pool = nullptr;
  if (structure && pool_shadow_pending(pool)) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) { ws->planned_n = 0; return SVOSLAM_OK; }
  if (!ws->sorted_keys) return SVOSLAM_ERR_INVALID_ARG;  // svo_fuse_sort has not run on this workspace
  SVO_TRY(ensure_device_size(pool, stream));
  SVO_TRY(tracker_poll(pool, false));  // sizes the device has reported since the last call
  const int64_t rmax = max_records(n, depth);
  int64_t bound = (int64_t)pool->size + pool->pending_bound + 8 * rmax;
  if (bound > pool->capacity) {
    // the bound is very loose (surface data splits ~n, not ~6n): first let the commits in flight report their sizes
    SVO_TRY(tracker_poll(pool, true));
    bound = (int64_t)pool->size + pool->pending_bound + 8 * rmax;
  }
  if (bound > pool->capacity) {
    // growing moves the pool: nothing on ANY stream may still be using it (a render of the previous frame)
    SVO_HIP(hipDeviceSynchronize());
    SVO_TRY(pool_sync(pool, stream));
    bound = (int64_t)pool->size + pool->pending_bound + 8 * rmax;  // (pending_bound: plans of the structure chain without a commit yet; else 0)
    if (bound > pool->capacity) {
      int64_t want = (int64_t)pool->size + pool->pending_bound + 16 * 8 * rmax;  // room for ~16 worst-case calls before the next sync
      if (want > (int64_t)kMask + 1) want = (int64_t)kMask + 1;
      if (want < bound) return SVOSLAM_ERR_POOL_LIMIT;
      int32_t live = 0;  // structure chain: plans may be ahead of their commits (the device is idle here: read their size)
      if (structure) SVO_HIP(hipMemcpy(&live, tracker_of(pool)->d_struct, 4, hipMemcpyDeviceToHost));
      SVO_TRY(grow_pool(pool, want, stream, live));
    }
  }
  const u64 *skey = ws->sorted_keys;
  const int tiles = (int)cdiv(n, kPlanThreads);
  unsigned char *leaf_t = ws->leaf_t.as<unsigned char>();
  u32 *leaf_f = ws->leaf_f.as<u32>();
  u32 *tile_hist = ws->tile_hist.as<u32>();
  int *d_struct = structure ? tracker_of(pool)->d_struct : nullptr;
  int split_blocks = (int)cdiv(rmax, 256);
  if (split_blocks > 2048) split_blocks = 2048;
  auto enqueue = [&]() -> int {  // three launches (round 1: a memset and five)
    plan_count_kernel<<<xcd_grid(tiles), kPlanThreads, 0, stream>>>(skey, n, depth, pool->d_data, leaf_t, leaf_f, ws->leaf_start.as<u32>(), tile_hist, tiles, small_any(ws));
    plan_scan_finish_kernel<<<256, 256, 0, stream>>>(tile_hist, tiles, small_totals(ws), small_ticket(ws), small_bucket_base(ws),
                                                     small_counts(ws), small_any(ws), d_struct, small_n0(ws));
    plan_emit_kernel<<<xcd_grid(tiles), kPlanThreads, 0, stream>>>(skey, n, depth, leaf_t, leaf_f, small_bucket_base(ws), tile_hist, tiles,
                                                         ws->rec_key.as<u64>(), ws->rec_front.as<u32>(), ws->rec_pass.as<unsigned char>(),
                                                         ws->leaf_rec0.as<u32>());
    if (structure)  // tiles AND links, numbered from the plan's own size; the level-grid marks come from the commit's leaf kernel
      split_all_kernel<<<split_blocks, 256, 0, stream>>>(ws->rec_key.as<u64>(), ws->rec_front.as<u32>(), ws->rec_pass.as<unsigned char>(),
                                                         small_bucket_base(ws), small_counts(ws), pool->d_data, pool->d_size, depth, nullptr,
                                                         small_n0(ws), 1);
    SVO_LAUNCH_CHECK();
    return SVOSLAM_OK;
  };
  GraphKey key;
  key.add(skey).add((unsigned long long)n).add((unsigned long long)depth).add(pool->d_data).add(ws->layout_hash())
     .add((unsigned long long)structure).add(d_struct);
  {
    StageScope timed(kStageFusePlan, stream);
    SVO_TRY(ws->g_plan.run(key, stream, enqueue));
  }
  ws->planned_n = n;
  ws->planned_pool = pool;
  if (structure) { ws->early_split_pool = pool; tracker_of(pool)->planned_ahead++; ws->structure_planned = true; }  // the commit: leaf kernel (links again, same values; marks) + straddlers
  pool->pending_bound += 8 * rmax;  // reserved from now on
  return SVOSLAM_OK;
}

int svo_fuse_plan(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, hipStream_t stream) {
  return fuse_plan_impl(ws, n, depth, pool, false, stream);
}
int svo_fuse_plan_structure(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, hipStream_t stream) {
  return fuse_plan_impl(ws, n, depth, pool, true, stream);
}

// Applies the planned commit to `pool`.  slot / keep_plan serve callers that keep several byte-identical replicas of
// one map (the frame scheduler ray-marches one replica while the next frame is committed to the other): the same plan
// -- made against ANY of the replicas in the state before this commit -- is applied to each of them, every
// application with its own slot (0 or 1: the scratch list of the mip pass) and all but the last with keep_plan.
// The part of the next commit that no ray march can see, ahead of the commit: the child tiles of the planned splits are
// initialised beyond the pool's present size (split_all_kernel without its links and without its level-grid marks) while the
// previous frame is still being ray-marched.  The commit that follows on this workspace and pool then runs two launches
// instead of three -- its leaf kernel writes the links -- which takes ~20 us off the stream that bounds the frame
// (commit + march).  Same pool contents as the plain commit.  After svo_fuse_plan, before svo_fuse_commit, same pool.
int svo_fuse_split_early(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, hipStream_t stream) {
  if (!ws || !pool || n < 0) return SVOSLAM_ERR_INVALID_ARG;
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  if (ws->planned_n != n || ws->planned_pool != pool || pool_shadow_pending(pool)) return SVOSLAM_ERR_INVALID_ARG;
  ws->early_split_pool = nullptr;
  if (n == 0) return SVOSLAM_OK;
  int split_blocks = (int)cdiv(max_records(n, depth), 256);
  if (split_blocks > 2048) split_blocks = 2048;
  {
    StageScope timed(kStageFusePlan, stream);
    split_all_kernel<<<split_blocks, 256, 0, stream>>>(ws->rec_key.as<u64>(), ws->rec_front.as<u32>(), ws->rec_pass.as<unsigned char>(),
                                                       small_bucket_base(ws), small_counts(ws), pool->d_data, pool->d_size, depth, nullptr,
                                                       small_n0(ws), 0);
  }
  SVO_LAUNCH_CHECK();
  ws->early_split_pool = pool;
  return SVOSLAM_OK;
}

// keyrange (key-range sharded commit, below): a deferred commit of this rank's slice of the frame -- the workspace's sorted arrays hold the
// slice followed by padding, *n_live its length --, without marks for the ray march's grid / bricks (keyrange_mark_kernel makes them, from
// ALL keys), without the pool's new size and without the size readback (svo_fuse_keyrange_apply: keyrange_finish_kernel, tracker_push)
static int commit_impl(svoslam_workspace *ws, const uint8_t *d_colors, int n, int depth, svoslam_pool *pool, int slot,
                       bool keep_plan, bool deferred, hipStream_t stream, const int *n_live = nullptr) {
  const bool keyrange = n_live != nullptr;
  if (!ws || !pool || n < 0 || (n > 0 && !d_colors) || slot < 0 || slot > 1) return SVOSLAM_ERR_INVALID_ARG;
  if (pool_shadow_pending(pool)) return SVOSLAM_ERR_INVALID_ARG;  // a deferred commit of this pool has not been applied
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  if (ws->planned_n != n) return SVOSLAM_ERR_INVALID_ARG;  // svo_fuse_plan has not run for this batch
  if (!keep_plan) ws->planned_n = -1;
  ws->deferred_pool = nullptr;
  const bool early = ws->early_split_pool != nullptr;
  if (early && (ws->early_split_pool != pool || deferred || keep_plan)) return SVOSLAM_ERR_INVALID_ARG;  // one pool, direct commit
  ws->early_split_pool = nullptr;
  if (ws->structure_planned) {
    ws->structure_planned = false;
    if (tracker_of(pool) && tracker_of(pool)->planned_ahead > 0) tracker_of(pool)->planned_ahead--;
  }
  if (n == 0) {
    if (deferred) { ws->deferred_pool = pool; ws->deferred_n = 0; }
    return SVOSLAM_OK;
  }
  const int64_t rmax = max_records(n, depth);
  if (pool != ws->planned_pool) {  // a replica the plan did not look at: same tree, same worst case, its own bookkeeping
    SVO_TRY(ensure_device_size(pool, stream));
    if ((int64_t)pool->size + pool->pending_bound + 8 * rmax > pool->capacity) {
      SVO_TRY(tracker_poll(pool, true));
      if ((int64_t)pool->size + pool->pending_bound + 8 * rmax > pool->capacity) return SVOSLAM_ERR_POOL_LIMIT;  // replicas must be reserved alike
    }
    pool->pending_bound += 8 * rmax;
  }
  const u64 *skey = ws->sorted_keys;
  const u32 *sidx = ws->sorted_idx;
  const unsigned char *leaf_t = ws->leaf_t.as<unsigned char>();
  int split_blocks = (int)cdiv(rmax, 256);
  if (split_blocks > 2048) split_blocks = 2048;
  const int fill_tiles = (int)cdiv(n, kFillThreads);
  DeviceBuffer &sb = slot == 0 ? ws->strad : ws->strad_b;
  // [level][tile] straddler entries | per-tile boundary records | [level][group] super straddlers (mip_straddle2_kernel)
  const int strad_groups = (int)cdiv(fill_tiles, kStradGroup);
  const size_t strad_words = (size_t)(SVOSLAM_MAX_DEPTH + 1) * (size_t)fill_tiles * 2, bc_words = 2 * (size_t)fill_tiles;
  SVO_TRY(sb.reserve((strad_words + bc_words + (size_t)(SVOSLAM_MAX_DEPTH + 1) * (size_t)strad_groups * 2) * 4));
  u32 *strad = sb.as<u32>();
  int *strad_bc = reinterpret_cast<int *>(strad + strad_words);
  u32 *sstrad = strad + strad_words + bc_words;
  const bool two_tier = !deferred;  // (deferred commits keep the single-workgroup straddler pass: its list doubles as the apply list)
  const u32 *leaf_start = ws->leaf_start.as<u32>();  // the leaf kernel resumes the plan's walk (round 3: -8 us)
  SVO_TRY(ensure_device_size(pool, stream));         // (creates the size tracker)
  PoolTracker *trk = tracker_of(pool);
  unsigned long long *shadow = nullptr;
  u32 epoch = 0, *apply_nodes = nullptr;
  if (deferred) {
    SVO_TRY(ws->apply_nodes.reserve(((size_t)fill_tiles * kFillThreads * (size_t)depth + (size_t)fill_tiles) * 4));  // lists + counts
    apply_nodes = ws->apply_nodes.as<u32>();
    SVO_TRY(pool_shadow_begin(pool, stream, &shadow, &epoch));
    ws->deferred_pool = pool; ws->deferred_n = n; ws->deferred_depth = depth; ws->deferred_tiles = fill_tiles;
  }
  SVO_TRY(tracker_make_room(pool));
  int brick_shift = -1;
  u32 *grid_dirty = keyrange ? nullptr : pool_accel_dirty_bitmap(pool, deferred ? (int)(epoch & 1u) : 0, depth, &brick_shift);  // nullptr: not a registered pool
  auto enqueue = [&]() -> int {
    if (!early)
      split_all_kernel<<<split_blocks, 256, 0, stream>>>(ws->rec_key.as<u64>(), ws->rec_front.as<u32>(),
                                                         ws->rec_pass.as<unsigned char>(), small_bucket_base(ws), small_counts(ws),
                                                         pool->d_data, pool->d_size, depth, grid_dirty, deferred ? small_n0(ws) : nullptr, 0);
    if (depth <= 12) fill_mip_local_kernel<12><<<xcd_grid(fill_tiles), kFillThreads, 0, stream>>>(skey, sidx, n, depth, leaf_t, d_colors, pool->d_data, strad, fill_tiles,
                                                                   grid_dirty, shadow, epoch, apply_nodes, ws->rec_key.as<u64>(),
                                                                   small_bucket_base(ws), small_n0(ws), early ? ws->leaf_rec0.as<u32>() : nullptr, leaf_start,
                                                                   two_tier ? strad_bc : nullptr, brick_shift, n_live);
    else fill_mip_local_kernel<16><<<xcd_grid(fill_tiles), kFillThreads, 0, stream>>>(skey, sidx, n, depth, leaf_t, d_colors, pool->d_data, strad, fill_tiles,
                                                                   grid_dirty, shadow, epoch, apply_nodes, ws->rec_key.as<u64>(),
                                                                   small_bucket_base(ws), small_n0(ws), early ? ws->leaf_rec0.as<u32>() : nullptr, leaf_start,
                                                                   two_tier ? strad_bc : nullptr, brick_shift, n_live);
    if (two_tier)
      mip_straddle2_kernel<<<strad_groups, kStrad2Threads, 0, stream>>>(pool->d_data, strad, strad_bc, sstrad, small_strad_ticket(ws, slot), fill_tiles,
                                                                        depth, small_counts(ws), pool->d_size, grid_dirty,
                                                                        trk ? trk->h_size : nullptr, trk ? trk->d_slot : nullptr);
    else
      mip_straddle_kernel<<<1, kStradThreads, 0, stream>>>(pool->d_data, strad, fill_tiles, depth, small_counts(ws), pool->d_size, grid_dirty,
                                                           trk ? trk->h_size : nullptr, trk ? trk->d_slot : nullptr, shadow, epoch, keyrange ? 1 : 0, n_live);
    SVO_LAUNCH_CHECK();
    return SVOSLAM_OK;
  };
  if (deferred) {  // the epoch changes with every call: not a recorded sequence
    {
      StageScope timed(kStageFuseCommit, stream);
      SVO_TRY(enqueue());
    }
    if (keyrange) { ws->keyrange_bound = 8 * rmax; return SVOSLAM_OK; }
    pool->pending += 1;
    return tracker_push(pool, 8 * rmax, stream);
  }
  GraphKey key;
  key.add(skey).add(d_colors).add((unsigned long long)n).add((unsigned long long)depth).add(pool->d_data).add(pool->d_size)
     .add((unsigned long long)slot).add(grid_dirty).add(trk ? (const void *)trk->h_size : nullptr).add(ws->layout_hash())
     .add((unsigned long long)early).add((unsigned long long)(brick_shift + 1));
  {
    StageScope timed(kStageFuseCommit, stream);
    SVO_TRY(ws->g_commit.run(key, stream, enqueue));
  }
  pool->pending += 1;
  return tracker_push(pool, 8 * rmax, stream);
}

int svo_fuse_commit_to(svoslam_workspace *ws, const uint8_t *d_colors, int n, int depth, svoslam_pool *pool, int slot,
                       bool keep_plan, hipStream_t stream) {
  return commit_impl(ws, d_colors, n, depth, pool, slot, keep_plan, false, stream);
}

int svo_fuse_commit(svoslam_workspace *ws, const uint8_t *d_colors, int n, int depth, svoslam_pool *pool, hipStream_t stream) {
  return commit_impl(ws, d_colors, n, depth, pool, 0, false, false, stream);
}

// Deferred commit: the commit's whole computation (splits, leaf blends, mip levels) WITHOUT a single store a ray march
// of the pool in its present state could observe -- new tiles lie beyond the pool's size, colour words go to a shadow
// array, the links of the pass-0 records wait -- so it may run while the previous frame is still being rendered.
// svo_fuse_apply (same workspace, before the workspace is used again) then publishes it with one short launch; the
// pool must not be read by anything that expects the new state, nor written, in between.  Same final pool contents as
// svo_fuse_commit.  Used by the frame scheduler: its map stream carries apply + march instead of commit + march.
int svo_fuse_commit_deferred(svoslam_workspace *ws, const uint8_t *d_colors, int n, int depth, svoslam_pool *pool, hipStream_t stream) {
  return commit_impl(ws, d_colors, n, depth, pool, 0, false, true, stream);
}

int svo_fuse_apply(svoslam_workspace *ws, svoslam_pool *pool, hipStream_t stream) {
  if (!ws || !pool || ws->deferred_pool != pool) return SVOSLAM_ERR_INVALID_ARG;
  ws->deferred_pool = nullptr;
  const int n = ws->deferred_n, depth = ws->deferred_depth, tiles = ws->deferred_tiles;
  if (n == 0) return SVOSLAM_OK;
  unsigned long long *shadow = nullptr;
  u32 epoch = 0;
  SVO_TRY(pool_shadow_current(pool, &shadow, &epoch));
  int blocks = tiles;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  commit_apply_kernel<<<blocks, 256, 0, stream>>>(pool->d_data, shadow, ws->apply_nodes.as<u32>(), tiles, kFillThreads * depth, ws->strad.as<u32>(), tiles,
                                                  depth * tiles, ws->rec_front.as<u32>(), ws->rec_pass.as<unsigned char>(),
                                                  small_counts(ws), small_n0(ws));
  SVO_LAUNCH_CHECK();
  pool_shadow_end(pool);
  return SVOSLAM_OK;
}

int svo_from_point_cloud_async(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int n, int depth,
                               svoslam_pool *pool, const float center[3], float edge, hipStream_t stream) {
  if (!ws || !pool || n < 0 || (n > 0 && (!d_points || !d_colors))) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(svo_fuse_sort(ws, d_points, n, depth, center, edge, stream));
  SVO_TRY(svo_fuse_plan(ws, n, depth, pool, stream));
  return svo_fuse_commit(ws, d_colors, n, depth, pool, stream);
}

int svo_from_point_cloud(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int n, int depth,
                         svoslam_pool *pool, const float center[3], float edge, svoslam_fuse_stats *stats,
                         hipStream_t stream) {
  if (!ws || !pool || n < 0 || (n > 0 && (!d_points || !d_colors))) return SVOSLAM_ERR_INVALID_ARG;
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  if (pool->size == 0) SVO_TRY(pool_init(pool, 8, stream));  // svo.cu:646-649
  if (n == 0) {
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->pool_size_before = stats->pool_size_after = pool->size; }
    return SVOSLAM_OK;
  }
  SVO_TRY(reserve_common(ws, n, depth));
  long long tk = -1;
  (void)stage_begin(kStageFuseSort, stream, &tk);
  const int idx_bits = blocking_sort_idx_bits(n, depth, false);
  compute_keys_kernel<3><<<cdiv(n, 256), 256, 0, stream>>>(d_points, n, depth, center[0], center[1], center[2], edge, ws->keys_a.as<u64>(), idx_bits);
  return svo_insert(ws, n, depth, pool, d_colors, false, false, stats, stream, tk, idx_bits);
}

int svo_from_voxel_grid(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int n, int depth,
                        svoslam_pool *pool, const float center[3], float edge, svoslam_fuse_stats *stats,
                        hipStream_t stream) {
  if (!ws || !pool || n < 0 || (n > 0 && (!d_centers || !d_colors))) return SVOSLAM_ERR_INVALID_ARG;
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  if (pool->size == 0) SVO_TRY(pool_init(pool, 8, stream));
  if (n == 0) {
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->pool_size_before = stats->pool_size_after = pool->size; }
    return SVOSLAM_OK;
  }
  SVO_TRY(reserve_common(ws, n, depth));
  long long tk = -1;
  (void)stage_begin(kStageFuseSort, stream, &tk);
  const int idx_bits = blocking_sort_idx_bits(n, depth, true);
  compute_keys_kernel<4><<<cdiv(n, 256), 256, 0, stream>>>(d_centers, n, depth, center[0], center[1], center[2], edge, ws->keys_a.as<u64>(), idx_bits);
  // Q20 (svo.cu:601-602,629): the reference sorts the keys alone, so sorted key i
  // stays paired with colour i -> color_by_position
  return svo_insert(ws, n, depth, pool, d_colors, true, true, stats, stream, tk, idx_bits);
}

// ----------------------------------------------------------------------------
// extraction (svo.cu:498-582, 699-745): level-synchronous BFS with an
// order-preserving compaction per level
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bfs_count_kernel(const u32 *__restrict__ pool, const u64 *__restrict__ parents,
                                                        int num, unsigned char *__restrict__ mask8,
                                                        u32 *__restrict__ tile_cnt) {
  __shared__ u32 tmp[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  u32 cntv = 0;
  if (i < num) {
    const u64 key = parents[i];
    const int d = (63 - __clzll((long long)key)) / 3;
    bool has_children = true;
    u32 pointer = 0;
    for (int l = d - 1; l >= 0; l--) {  // getOccupiedChildren :515-520
      pointer += (u32)(key >> (3 * l)) & 7u;
      const u32 w0 = pool[2 * (size_t)pointer];
      has_children = (w0 & kFlag) != 0;
      pointer = w0 & kMask;
    }
    u32 m = 0;
    if (has_children) {
      const uint4 *tile = reinterpret_cast<const uint4 *>(pool + 2 * (size_t)pointer);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint4 v = tile[q];
        if ((v.y >> 24) > 127u) m |= 1u << (2 * q);
        if ((v.w >> 24) > 127u) m |= 1u << (2 * q + 1);
      }
    }
    mask8[i] = (unsigned char)m;
    cntv = __popc(m);
  }
  u32 total;
  (void)block256_exclusive_scan(cntv, tmp, total);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void bfs_emit_kernel(const u64 *__restrict__ parents, int num,
                                                       const unsigned char *__restrict__ mask8,
                                                       const u32 *__restrict__ tile_prefix, u64 *__restrict__ children) {
  __shared__ u32 tmp[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const u32 m = i < num ? mask8[i] : 0u;
  u32 total;
  u32 pos = tile_prefix[blockIdx.x] + block256_exclusive_scan(__popc(m), tmp, total);
  if (i < num) {
    const u64 key = parents[i];
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (m & (1u << k)) children[pos++] = (key << 3) + (u64)k;
  }
}

// voxelGridFromKeys, svo.cu:538-582
__global__ __launch_bounds__(256) void voxel_grid_from_keys_kernel(const u32 *__restrict__ pool, const u64 *__restrict__ keys,
                                                                   int num, float cx, float cy, float cz, float edge,
                                                                   float4 *__restrict__ centers, float4 *__restrict__ colors) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= num) return;
  const u64 key = keys[i];
  const int d = (63 - __clzll((long long)key)) / 3;
  u32 node = 0, child = 0;
  for (int l = d - 1; l >= 0; l--) {
    const u32 p = (u32)(key >> (3 * l)) & 7u;
    node = child + p;
    child = pool[2 * (size_t)node] & kMask;
    edge /= 2.0f;
    cx += edge * ((p & 1u) ? 1 : -1);
    cy += edge * ((p & 2u) ? 1 : -1);
    cz += edge * ((p & 4u) ? 1 : -1);
  }
  const u32 val = pool[2 * (size_t)node + 1];
  centers[i] = make_float4(cx, cy, cz, 1.0f);
  colors[i] = make_float4((float)(val & 0xFF) / 255.0f, (float)((val >> 8) & 0xFF) / 255.0f,
                          (float)((val >> 16) & 0xFF) / 255.0f, (float)((val >> 24) & 0xFF) / 255.0f);
}

int extract_voxel_grid(svoslam_workspace *ws, const svoslam_pool *pool, int depth, const float center[3], float edge,
                       float **d_centers, float **d_colors, int32_t *n_out, hipStream_t stream) {
  if (!ws || !pool || !d_centers || !d_colors || !n_out) return SVOSLAM_ERR_INVALID_ARG;
  if (depth < 1 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  *d_centers = nullptr; *d_colors = nullptr; *n_out = 0;
  if (pool->size == 0) return SVOSLAM_OK;
  if (pool->pending > 0) SVO_HIP(hipStreamSynchronize(stream));  // (size itself is not needed by the BFS)
  SVO_TRY(ws->reserve_small());  // (zeroed when created: any_valid and the plan's arrival ticket start at zero)
  SVO_TRY(ws->bfs_a.reserve(8));
  const u64 one = 1;
  SVO_HIP(hipMemcpyAsync(ws->bfs_a.ptr, &one, 8, hipMemcpyHostToDevice, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  svoslam::DeviceBuffer *cur = &ws->bfs_a, *nxt = &ws->bfs_b;
  int num = 1;
  for (int lvl = 0; lvl < depth && num > 0; lvl++) {
    const int tiles = (int)cdiv(num, 256);
    SVO_TRY(ws->bfs_mask.reserve((size_t)num));
    SVO_TRY(ws->tile_hist.reserve((size_t)(tiles + 1) * 4));
    bfs_count_kernel<<<tiles, 256, 0, stream>>>(pool->d_data, cur->as<u64>(), num, ws->bfs_mask.as<unsigned char>(), ws->tile_hist.as<u32>());
    row_scan_rows1(ws->tile_hist.as<u32>(), tiles, small_totals(ws), stream);
    unsigned next_num = 0;
    SVO_HIP(hipMemcpyAsync(&next_num, small_totals(ws), 4, hipMemcpyDeviceToHost, stream));
    SVO_HIP(hipStreamSynchronize(stream));
    if (next_num > 0) {
      SVO_TRY(nxt->reserve((size_t)next_num * 8));
      bfs_emit_kernel<<<tiles, 256, 0, stream>>>(cur->as<u64>(), num, ws->bfs_mask.as<unsigned char>(), ws->tile_hist.as<u32>(), nxt->as<u64>());
      SVO_LAUNCH_CHECK();
    }
    svoslam::DeviceBuffer *t = cur; cur = nxt; nxt = t;
    num = (int)next_num;
  }
  if (num <= 0) return SVOSLAM_OK;
  float *ce = nullptr, *co = nullptr;
  SVO_HIP(hipMalloc((void **)&ce, (size_t)num * 16));
  SVO_HIP(hipMalloc((void **)&co, (size_t)num * 16));
  voxel_grid_from_keys_kernel<<<cdiv(num, 256), 256, 0, stream>>>(pool->d_data, cur->as<u64>(), num, center[0], center[1], center[2], edge,
                                                                  reinterpret_cast<float4 *>(ce), reinterpret_cast<float4 *>(co));
  SVO_LAUNCH_CHECK();
  SVO_HIP(hipStreamSynchronize(stream));
  *d_centers = ce; *d_colors = co; *n_out = num;
  return SVOSLAM_OK;
}


// ----------------------------------------------------------------------------
// key-range sharded commit (protocol pinned on the CPU by tests/test_keyrange_gloo.py)
// ----------------------------------------------------------------------------
// The plan + commit of ONE frame cut across `world` ranks by key range instead of being replicated on every rank.  Every rank holds a
// byte-identical replica of the pool and the frame's sorted keys (sorted by their owners and all-gathered: svo_fuse_export_sorted /
// svo_fuse_merge_sorted).  Two calls per frame and rank, ONE all-gather between them:
//   svo_fuse_keyrange_commit  the rank's slice of the sorted keys -- the keys under a contiguous run of level-3 prefixes holding about
//                             n / world keys (keyrange_bounds_kernel: every rank computes the same cuts) -- is planned (svo.cu:179-237)
//                             and committed (:239-465) by the unchanged kernels as a DEFERRED commit: new tiles beyond the pool's size
//                             in the rank's own numbering, colour words in the shadow array, nothing a replica could not still
//                             discard.  keyrange_pack_* then writes the rank's DELTA: its (pass, depth) bucket sizes, its new tiles
//                             (16 words each, links still in local numbering), the frontier nodes its pass-0 records link from,
//                             {node, colour word} of every existing node it changed, the bricks whose siblings its splits created;
//   [all-gather of the deltas -- the caller's: RCCL, or a table of precomputed deltas for an emulated rank]
//   svo_fuse_keyrange_apply   numbering: the reference numbers the new tiles of a pass by the rank of their key among the pass's sorted
//                             unique keys = bucket-major, key order inside a bucket; slices are key ranges, so rank s's records of
//                             bucket b follow those of ranks < s: global index = bucket base + sum of the lower ranks' counts + local
//                             rank in the bucket -- one table of world x 256 offsets (keyrange_setup_kernel).  Every delta (the own one
//                             included) is written to its global place; the marks of the ray march's grid / bricks are made from ALL the
//                             frame's keys against this replica's own dirty state (ranks render different frames: their dirty states
//                             differ); the colour words of the nodes above the splitter level -- shared by several ranks' paths -- are
//                             recomputed from the merged children, level by level, then the root pass (Q6); size and size readback.
// Frames whose splits reach ABOVE the splitter level (a node of level 1 or 2 without children: the first frames of a map, new territory)
// make several ranks plan the SAME records (the prefix of such a record lies on paths of more than one slice) and create the same tiles:
// every delta lists its records above the splitter level by key, keyrange_setup_kernel ranks them in the ranks' UNION (the reference's
// order inside their buckets) and clears their tiles, and the apply writes of such a tile only the nodes a rank actually filled -- a
// level-3 node has one owner; the shallower ones get the same link from everybody and their colour words from the recomputation.
constexpr int kKrLevel = 3;
constexpr int kKrMaxWorld = 16;
constexpr int kKrHeader = 512;       // words: scalars, then the 256 bucket sizes at [256, 512)
constexpr int kKrSibCap = 8192;      // entries
constexpr int kKrShallowCap = 1024;  // keys (two words each)
constexpr int kKrTopCap = 128;       // records above the splitter level: {key (two words), local record, bucket}; a rank has at most 8 + 64
constexpr int kKrTop0 = kKrHeader + kKrSibCap + 2 * kKrShallowCap;
constexpr int kKrTiles0 = kKrTop0 + 4 * kKrTopCap;  // first word of the tiles
enum { kKrMagic = 0, kKrRecords = 1, kKrWords = 2, kKrLinks = 3, kKrSib = 4, kKrShallow = 5, kKrAnyValid = 6, kKrOverflow = 7, kKrSliceKeys = 8,
       kKrN0 = 9, kKrUsed = 10, kKrCapacity = 11, kKrDepth = 12, kKrTop = 13 };
enum { kKrOverflowed = 2, kKrMismatch = 4 };
constexpr u32 kKrEmpty1 = 127u << 24;  // word1 of a node splitNodes has just created (svo.cu:269-275)
__host__ __device__ inline size_t kr_bid0(u32 records) { return (size_t)kKrTiles0 + 16 * (size_t)records; }
__host__ __device__ inline size_t kr_links0(u32 records) { return kr_bid0(records) + (records + 3u) / 4u; }
__host__ __device__ inline size_t kr_words0(u32 records, u32 links) { return kr_links0(records) + links; }

// window of rank `rank`: win[0] = first, win[1] = end, win[2] = length of its slice of the sorted keys (invalid keys -- key 1 -- sort first
// and belong to nobody).  Cut r lies at the end of the level-L run that holds key number r x valid / world.
__global__ void keyrange_bounds_kernel(const u64 *__restrict__ skey, int n, int depth, int rank, int world, int *__restrict__ win) {
  const int r = (int)threadIdx.x;
  auto first_at_least = [&](int lo, int hi, u64 bound, int shift) {  // first j in [lo, hi) with (skey[j] >> shift) >= bound
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((skey[mid] >> shift) < bound) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const int v0 = first_at_least(0, n, 2ull, 0);
  const long long nv = n - v0;
  int b = v0;
  if (r >= world) b = n;
  else if (r > 0) {
    const int i = v0 + (int)((long long)r * nv / world);
    if (i > v0) { const int sh = 3 * (depth - kKrLevel); b = first_at_least(i, n, (skey[i - 1] >> sh) + 1ull, sh); }
  }
  if (r <= world) win[4 + r] = b;
  __syncthreads();
  if (r == 0) { const int lo = win[4 + rank], hi = win[4 + rank + 1]; win[0] = lo; win[1] = hi; win[2] = hi - lo; }
}

__global__ __launch_bounds__(256) void keyrange_slice_kernel(const u64 *__restrict__ skey, const u32 *__restrict__ sidx, int n,
                                                             const int *__restrict__ win, u64 *__restrict__ out_key, u32 *__restrict__ out_idx) {
  const int j = (int)(blockIdx.x * 256u + threadIdx.x);
  if (j >= n) return;
  const int lo = win[0], len = win[2];
  out_key[j] = j < len ? skey[lo + j] : 1ull;  // the slice at the front, padding (invalid keys) behind it
  out_idx[j] = j < len ? sidx[lo + j] : 0u;
}

__global__ __launch_bounds__(256) void keyrange_pack_header_kernel(u32 *__restrict__ delta, long long capacity_words, const u32 *__restrict__ bucket_base,
                                                                   const PlanCounts *__restrict__ counts, const u32 *__restrict__ n0_saved,
                                                                   const int *__restrict__ win, int depth) {
  const int t = (int)threadIdx.x;
  delta[256 + t] = bucket_base[t + 1] - bucket_base[t];
  if (t == 0) {
    const u32 R = (u32)counts->total_records, links = (u32)(counts->pass_start[1] - counts->pass_start[0]);
    delta[kKrMagic] = 0x4B52414Eu;
    delta[kKrRecords] = R; delta[kKrWords] = 0u; delta[kKrLinks] = links; delta[kKrSib] = 0u; delta[kKrShallow] = 0u; delta[kKrTop] = 0u;
    delta[kKrAnyValid] = (u32)counts->any_valid; delta[kKrSliceKeys] = (u32)win[2]; delta[kKrN0] = *n0_saved; delta[kKrDepth] = (u32)depth;
    delta[kKrCapacity] = capacity_words > 0xFFFFFFFFll ? 0xFFFFFFFFu : (u32)capacity_words;
    const bool fits = (long long)kr_words0(R, links) <= capacity_words;
    delta[kKrOverflow] = fits ? 0u : 1u;
    delta[kKrUsed] = (u32)kr_words0(R, links);
  }
}

// the new tiles (one lane per node), the records' buckets, the pass-0 records' frontier nodes, and what the receivers' brick / grid marks
// need from the records: the bricks whose node this commit created (their childless siblings get their lines: pool_grid.hip
// brick_siblings) and the keys of splits above the grid's block level (they re-label a whole cube)
__global__ __launch_bounds__(256) void keyrange_pack_tiles_kernel(u32 *__restrict__ delta, const u32 *__restrict__ pool,
                                                                  const unsigned long long *__restrict__ shadow, u32 epoch,
                                                                  const u64 *__restrict__ rec_key, const u32 *__restrict__ rec_front,
                                                                  const unsigned char *__restrict__ rec_pass, int brick_shift) {
  if (delta[kKrOverflow]) return;
  const u32 R = delta[kKrRecords], links = delta[kKrLinks], n0 = delta[kKrN0];
  unsigned char *bid = reinterpret_cast<unsigned char *>(delta + kr_bid0(R));
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < 8 * (size_t)R; i += (size_t)gridDim.x * 256u) {
    const u32 r = (u32)(i >> 3), q = (u32)(i & 7u);
    const u32 node = n0 + 8u * r + q;
    const unsigned long long sh = shadow[node];
    const u32 w0 = pool[2 * (size_t)node], w1 = (u32)(sh >> 32) == epoch ? (u32)sh : pool[2 * (size_t)node + 1];
    reinterpret_cast<uint2 *>(delta + kKrTiles0)[i] = make_uint2(w0, w1);
    if (q == 0u) {
      const u64 key = rec_key[r];
      const int d = (63 - __clzll((long long)key)) / 3, pass = rec_pass[r];
      bid[r] = (unsigned char)bucket_id(pass, d);
      if (r < links) delta[kr_links0(R) + r] = rec_front[r];
      if (d < kKrLevel) {  // a record several ranks may hold: listed by key for the union numbering
        const u32 pos = atomicAdd(&delta[kKrTop], 1u);
        if (pos < (u32)kKrTopCap) {
          u32 *e = delta + kKrTop0 + 4 * pos;
          e[0] = (u32)key; e[1] = (u32)(key >> 32); e[2] = r; e[3] = bucket_id(pass, d);
        }
      }
      if (d < kPoolGridBlockLevel) {
        const u32 pos = atomicAdd(&delta[kKrShallow], 1u);
        if (pos < (u32)kKrShallowCap) reinterpret_cast<u64 *>(delta + kKrHeader + kKrSibCap)[pos] = key;
      }
      if (brick_shift >= 0 && d == brick_node_level(brick_shift) && pass >= 1) {
        u32 x = 0, y = 0, z = 0;
        for (int k = 1; k <= d; k++) {
          const u32 oct = (u32)(key >> (3 * (d - k))) & 7u;
          x = (x << 1) | (oct & 1u); y = (y << 1) | ((oct >> 1) & 1u); z = (z << 1) | (oct >> 2);
        }
        const u32 org = brick_window_origin(brick_shift) >> 2;
        x -= org; y -= org; z -= org;
        if ((x | y | z) < (kBrickWindowCells >> 2)) {
          const u32 pos = atomicAdd(&delta[kKrSib], 1u);
          if (pos < (u32)kKrSibCap) delta[kKrHeader + pos] = brick_list_entry(x, y, z);
        }
      }
    }
  }
}

// {node, colour word} of the EXISTING nodes (below the pool's size) this commit changed: the leaf kernel's per-workgroup lists, then the
// straddler list (workgroups past the lists take 2048 entries each).  A workgroup counts, reserves with one atomic, writes.
__global__ __launch_bounds__(256) void keyrange_pack_words_kernel(u32 *__restrict__ delta, const unsigned long long *__restrict__ shadow,
                                                                  const u32 *__restrict__ apply_nodes, int fill_tiles, int list_cap,
                                                                  const u32 *__restrict__ strad, int strad_first, int strad_end) {
  if (delta[kKrOverflow]) return;
  __shared__ u32 wave_cnt[4], base_s;
  const u32 R = delta[kKrRecords], links = delta[kKrLinks], n0 = delta[kKrN0], cap = delta[kKrCapacity];
  const int t = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lists = t < fill_tiles;
  const u32 *src; u32 cnt, stride;
  if (lists) { src = apply_nodes + (size_t)t * list_cap; cnt = apply_nodes[(size_t)fill_tiles * list_cap + t]; stride = 1u; }
  else {
    const long long first = strad_first + (long long)(t - fill_tiles) * 2048;
    src = strad + 2 * first; stride = 2u;
    const long long left = (long long)strad_end - first;
    cnt = left <= 0 ? 0u : (left < 2048 ? (u32)left : 2048u);
  }
  auto wanted = [&](u32 i) { if (i >= cnt) return false; const u32 node = src[(size_t)i * stride]; return node != kNoStraddler && node < n0; };
  u32 mine = 0;
  for (u32 i = (u32)tid; i < cnt; i += 256u) mine += wanted(i) ? 1u : 0u;
  u32 incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o); if (lane >= o) incl += v; }
  if (lane == 63) wave_cnt[wave] = incl;
  __syncthreads();
  u32 before = 0, total = 0;
  for (int w = 0; w < 4; w++) { if (w < wave) before += wave_cnt[w]; total += wave_cnt[w]; }
  if (tid == 0) base_s = total ? atomicAdd(&delta[kKrWords], total) : 0u;
  __syncthreads();
  if (!total) return;
  const size_t w0 = kr_words0(R, links);
  if (w0 + 2 * ((size_t)base_s + total) > (size_t)cap) { if (tid == 0) delta[kKrOverflow] = 1u; return; }
  u32 pos = base_s + before + incl - mine;
  for (u32 i = (u32)tid; i < cnt; i += 256u)
    if (wanted(i)) {
      const u32 node = src[(size_t)i * stride];
      delta[w0 + 2 * (size_t)pos] = node;
      delta[w0 + 2 * (size_t)pos + 1] = (u32)shadow[node];
      pos++;
    }
  if (tid == 0) atomicMax(&delta[kKrUsed], (u32)(w0 + 2 * ((size_t)base_s + total)));
}

struct KrDeltas { const u32 *d[kKrMaxWorld]; };

// numbering: table[s][b] = what to add to rank s's local record rank in bucket b (depth >= 3) to get its place in the reference's order;
// topmap[s] = {count, then (local record, place) pairs} for rank s's records above the splitter level, ranked in the ranks' union; the
// union's tiles cleared; scal[0] = records of all ranks (shared ones once), [1] = status flags, [2] = first new tile, [3] = any valid key
constexpr int kKrTopMax = kKrMaxWorld * kKrTopCap;
__global__ __launch_bounds__(256) void keyrange_setup_kernel(KrDeltas D, int world, int *__restrict__ table, u32 *__restrict__ topmap,
                                                             u32 *__restrict__ scal, u32 *__restrict__ pool) {
  __shared__ unsigned tmp[4];
  __shared__ u64 top_key[kKrTopMax];
  __shared__ unsigned short top_bs[kKrTopMax];   // bucket | rank << 8
  __shared__ unsigned char top_first[kKrTopMax];
  __shared__ u32 top_base[kKrMaxWorld + 1], union_cnt[256];
  const int b = (int)threadIdx.x;
  u32 tot = 0, flags = 0, any = 0;
  union_cnt[b] = 0u;
  if (b == 0) {
    u32 run = 0;
    for (int s = 0; s < world; s++) { top_base[s] = run; const u32 c = D.d[s][kKrTop]; run += c < (u32)kKrTopCap ? c : (u32)kKrTopCap; }
    top_base[world] = run;
  }
  __syncthreads();
  const int M = (int)top_base[world];
  for (int s = 0; s < world; s++) {
    if (D.d[s][kKrOverflow] || D.d[s][kKrTop] > (u32)kKrTopCap) flags |= kKrOverflowed;
    if (D.d[s][kKrN0] != D.d[0][kKrN0] || D.d[s][kKrMagic] != 0x4B52414Eu) flags |= kKrMismatch;
    any |= D.d[s][kKrAnyValid];
    const int cnt = (int)(top_base[s + 1] - top_base[s]);
    for (int i = b; i < cnt; i += 256) {
      const u32 *e = D.d[s] + kKrTop0 + 4 * i;
      top_key[top_base[s] + i] = ((u64)e[1] << 32) | e[0];
      top_bs[top_base[s] + i] = (unsigned short)(e[3] | ((u32)s << 8));
    }
  }
  __syncthreads();
  // the union: an entry is its record's FIRST occurrence when no earlier entry holds the same (bucket, key)
  for (int e = b; e < M; e += 256) {
    bool first = true;
    for (int f = 0; f < e && first; f++) first = !(top_key[f] == top_key[e] && (top_bs[f] & 255u) == (top_bs[e] & 255u));
    top_first[e] = first ? 1 : 0;
    if (first) atomicAdd(&union_cnt[top_bs[e] & 255u], 1u);
  }
  __syncthreads();
  const int d = (b & 15) + 1;  // bucket_id(p, d) = 16 p + d - 1
  if (d < kKrLevel) tot = union_cnt[b];
  else for (int s = 0; s < world; s++) tot += D.d[s][256 + b];
  unsigned total;
  const u32 gbase = block256_exclusive_scan(tot, tmp, total);
  __shared__ u32 gbase_s[256];
  gbase_s[b] = gbase;
  u32 lower = 0;
  for (int s = 0; s < world; s++) {
    const u32 c = D.d[s][256 + b];
    unsigned ltot;
    const u32 lbase = block256_exclusive_scan(c, tmp, ltot);
    table[s * 256 + b] = (int)(gbase + lower) - (int)lbase;
    lower += c;
  }
  __syncthreads();
  const u32 n0 = D.d[0][kKrN0];
  for (int e = b; e < M; e += 256) {
    const u32 bk = top_bs[e] & 255u, s = top_bs[e] >> 8;
    u32 rank = 0;  // first occurrences of the bucket with a smaller key
    for (int f = 0; f < M; f++) rank += (top_first[f] && (top_bs[f] & 255u) == bk && top_key[f] < top_key[e]) ? 1u : 0u;
    const u32 place = gbase_s[bk] + rank;
    const u32 slot = (u32)e - top_base[s];
    topmap[s * (1 + 2 * kKrTopCap) + 1 + 2 * slot] = D.d[s][kKrTop0 + 4 * slot + 2];
    topmap[s * (1 + 2 * kKrTopCap) + 2 + 2 * slot] = place;
    if (top_first[e] && !flags) {  // the shared tile starts as eight empty children; the ranks then write what they filled
      uint4 *tile = reinterpret_cast<uint4 *>(pool + 2 * ((size_t)n0 + 8 * (size_t)place));
      const uint4 init = make_uint4(0u, kKrEmpty1, 0u, kKrEmpty1);
      tile[0] = init; tile[1] = init; tile[2] = init; tile[3] = init;
    }
  }
  if (b < world) topmap[b * (1 + 2 * kKrTopCap)] = top_base[b + 1] - top_base[b];
  if (flags) atomicOr(&scal[1], flags);
  if (b == 0) { scal[0] = total; scal[2] = n0; scal[3] = any; }
}

// every delta to its global place (blockIdx.y = the delta's rank): tiles with their links renumbered, the pass-0 links, the colour words
// of existing nodes, and the record-borne marks (sibling ring, cubes of shallow splits) into this replica's dirty state
__global__ __launch_bounds__(256) void keyrange_apply_kernel(KrDeltas D, const int *__restrict__ table, const u32 *__restrict__ topmap,
                                                             const u32 *__restrict__ scal, u32 *__restrict__ pool, u32 *__restrict__ dirty) {
  if (scal[1]) return;  // overflowed / mismatching deltas: nothing is applied (svo_fuse_keyrange_status reports it)
  const int s = (int)blockIdx.y;
  const u32 *delta = D.d[s];
  const int *T = table + s * 256;
  const u32 *tm = topmap + s * (1 + 2 * kKrTopCap);
  const u32 R = delta[kKrRecords], links = delta[kKrLinks], words = delta[kKrWords], n0 = delta[kKrN0];
  const unsigned char *bid = reinterpret_cast<const unsigned char *>(delta + kr_bid0(R));
  auto shared_record = [&](u32 r) { return (int)(bid[r] & 15u) + 1 < kKrLevel; };
  auto place = [&](u32 r) {
    if (shared_record(r)) {  // ranked in the ranks' union (a handful per frame, in the first frames of a map)
      const u32 cnt = tm[0];
      for (u32 i = 0; i < cnt; i++) if (tm[1 + 2 * i] == r) return tm[2 + 2 * i];
      return 0u;
    }
    return (u32)((int)r + T[bid[r]]);
  };
  const size_t stride = (size_t)gridDim.x * 256u, t0 = (size_t)blockIdx.x * 256u + threadIdx.x;
  for (size_t i = t0; i < 8 * (size_t)R; i += stride) {
    const u32 r = (u32)(i >> 3), q = (u32)(i & 7u);
    uint2 w = reinterpret_cast<const uint2 *>(delta + kKrTiles0)[i];
    if (shared_record(r) && w.x == 0u && w.y == kKrEmpty1) continue;  // a node of a shared tile this rank did not fill
    if (w.x & kFlag) w.x = kFlag | ((n0 + 8u * place(((w.x & kMask) - n0) >> 3)) & kMask);
    reinterpret_cast<uint2 *>(pool)[(size_t)n0 + 8 * (size_t)place(r) + q] = w;
  }
  for (size_t r = t0; r < links; r += stride) pool[2 * (size_t)delta[kr_links0(R) + r]] = kFlag | ((n0 + 8u * place((u32)r)) & kMask);
  const size_t w0 = kr_words0(R, links);
  for (size_t i = t0; i < words; i += stride) pool[2 * (size_t)delta[w0 + 2 * i] + 1] = delta[w0 + 2 * i + 1];
  if (dirty) {
    const u32 sib = delta[kKrSib] < (u32)kKrSibCap ? delta[kKrSib] : (u32)kKrSibCap;
    for (size_t i = t0; i < sib; i += stride) brick_sibling_list(dirty, delta[kKrHeader + i]);
    const u32 sh = delta[kKrShallow];
    if (sh > (u32)kKrShallowCap) {  // more shallow splits than the list holds: every block is stale
      for (size_t i = t0; i < (size_t)kPoolGridDirtyWords; i += stride) dirty[i] = 0xFFFFFFFFu;
    } else {
      for (size_t i = t0; i < sh; i += stride) {
        const u64 key = reinterpret_cast<const u64 *>(delta + kKrHeader + kKrSibCap)[i];
        pool_grid_mark(dirty, key, (63 - __clzll((long long)key)) / 3);
      }
    }
  }
}

// the marks of the ray march's level grid and occupancy bricks from ALL keys of the frame (as the leaf kernel makes them for the keys it
// commits: pool_grid.hpp), and the level-2 prefixes that occur (top[0..1]: a 64-bit mask) for the shared nodes' colour words
__global__ __launch_bounds__(256) void keyrange_mark_kernel(const u64 *__restrict__ skey, int n, int depth, u32 *__restrict__ dirty, int brick_shift,
                                                            unsigned long long *__restrict__ top) {
  __shared__ u32 brick_cnt, brick_base;
  __shared__ unsigned long long mask_s;
  const int tid = (int)threadIdx.x, j = (int)(blockIdx.x * 256u + threadIdx.x);
  if (tid == 0) { brick_cnt = 0u; mask_s = 0ull; }
  __syncthreads();
  u64 key = 1; int c = 0;
  const bool head = j < n && is_head(skey, j, key, c, depth);
  if (head && c < 2) atomicOr(&mask_s, 1ull << ((key >> (3 * (depth - 2))) & 63ull));
  if (dirty && head && c < kPoolGridBlockLevel) pool_grid_mark(dirty, key, depth);
  const bool bricks_on = dirty != nullptr && brick_shift >= 0 && depth >= brick_node_level(brick_shift);
  u32 entry = 0, off = 0;
  const bool mine = bricks_on && brick_mark_test(dirty, head && c < brick_node_level(brick_shift), key, depth, brick_shift, entry);
  const unsigned long long bm = __ballot(mine);
  if (bm) {
    const int leader = __ffsll((long long)bm) - 1;
    u32 woff = 0;
    if ((tid & 63) == leader) woff = atomicAdd(&brick_cnt, (u32)__popcll(bm));
    off = (u32)__shfl((int)woff, leader) + (u32)__popcll(bm & ((1ull << (tid & 63)) - 1ull));
  }
  __syncthreads();
  if (tid == 0) {
    if (brick_cnt) brick_base = brick_ring_reserve(dirty, brick_cnt);
    if (mask_s) atomicOr(top, mask_s);
  }
  __syncthreads();
  if (mine) brick_ring_store(dirty, brick_base + off, entry);
}

// one workgroup, behind everything else: the colour words of the nodes above the splitter level on the frame's paths from their merged
// children (mipmapNodes restricted to levels 2 and 1, svo.cu:450-465), the root pass (Q6), the pool's size and its readback, the list of
// the marked grid blocks
__global__ __launch_bounds__(256) void keyrange_finish_kernel(u32 *__restrict__ pool, u32 *__restrict__ scal, unsigned long long *__restrict__ top,
                                                              int *__restrict__ d_size, int32_t *__restrict__ h_sizes, int *__restrict__ d_slot,
                                                              u32 *__restrict__ dirty) {
  const int tid = (int)threadIdx.x;
  const bool ok = scal[1] == 0u;
  const unsigned long long m2 = *top;
  if (ok) {
    if (tid < 64 && ((m2 >> tid) & 1ull)) {
      const u32 node1 = (u32)tid >> 3, base1 = pool[2 * (size_t)node1] & kMask;
      const u32 node2 = base1 + ((u32)tid & 7u);
      pool[2 * (size_t)node2 + 1] = average_tile(pool, pool[2 * (size_t)node2] & kMask);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 8 && ((m2 >> (8 * tid)) & 0xFFull)) pool[2 * (size_t)tid + 1] = average_tile(pool, pool[2 * (size_t)tid] & kMask);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (tid == 0) {
    if (ok && scal[3]) pool[1] = average_tile(pool, 0u);  // Q6
    const int size_now = ok ? (int)(scal[2] + 8u * scal[0]) : *d_size;
    *d_size = size_now;
    if (h_sizes) { const int sl = *d_slot; h_sizes[sl] = size_now; *d_slot = (sl + 1) % 8; }
    *top = 0ull;
  }
  if (dirty) pool_grid_compact(dirty, 256);
}

static int kr_scratch(svoslam_workspace *ws, int n) {  // slice arrays + window / table / scalars (zeroed once)
  SVO_TRY(ws->kr_keys.reserve((size_t)n * 8));
  SVO_TRY(ws->kr_idx.reserve((size_t)n * 4));
  if (ws->kr_small.bytes < 65536) {
    SVO_TRY(ws->kr_small.reserve(65536));
    SVO_HIP(memset_sync(ws->kr_small.ptr, 0, ws->kr_small.bytes));
  }
  return SVOSLAM_OK;
}
static inline int *kr_win(svoslam_workspace *ws) { return ws->kr_small.as<int>(); }                       // [0..3] window, [4..4+world] cuts
static inline u32 *kr_scal(svoslam_workspace *ws) { return ws->kr_small.as<u32>() + 64; }                 // setup scalars
static inline unsigned long long *kr_top(svoslam_workspace *ws) { return reinterpret_cast<unsigned long long *>(ws->kr_small.as<u32>() + 96); }
static inline int *kr_table(svoslam_workspace *ws) { return ws->kr_small.as<int>() + 128; }               // [world][256]
static inline u32 *kr_topmap(svoslam_workspace *ws) { return ws->kr_small.as<u32>() + 128 + kKrMaxWorld * 256; }  // [world][1 + 2 x kKrTopCap]

int svo_fuse_keyrange_commit(svoslam_workspace *ws, const unsigned long long *d_keys, const uint32_t *d_idx, const uint8_t *d_colors, int n,
                             int depth, svoslam_pool *pool, int rank, int world, uint32_t *d_delta, long long delta_bytes, hipStream_t stream) {
  if (!ws || !pool || !d_delta || n <= 0 || !d_keys || !d_idx || !d_colors) return SVOSLAM_ERR_INVALID_ARG;
  if (world < 1 || world > kKrMaxWorld || rank < 0 || rank >= world) return SVOSLAM_ERR_INVALID_ARG;
  if (depth < kKrLevel + 3 || depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  if (delta_bytes < (long long)(kKrTiles0 + 64) * 4) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(kr_scratch(ws, n));
  int *win = kr_win(ws);
  keyrange_bounds_kernel<<<1, 64, 0, stream>>>(d_keys, n, depth, rank, world, win);
  keyrange_slice_kernel<<<cdiv(n, 256), 256, 0, stream>>>(d_keys, d_idx, n, win, ws->kr_keys.as<u64>(), ws->kr_idx.as<u32>());
  SVO_LAUNCH_CHECK();
  SVO_TRY(svo_fuse_adopt_sorted(ws, ws->kr_keys.as<u64>(), ws->kr_idx.as<u32>(), n, depth));
  SVO_TRY(svo_fuse_plan(ws, n, depth, pool, stream));
  SVO_TRY(commit_impl(ws, d_colors, n, depth, pool, 0, false, true, stream, win + 2));
  // the delta
  unsigned long long *shadow = nullptr;
  u32 epoch = 0;
  SVO_TRY(pool_shadow_current(pool, &shadow, &epoch));
  int brick_shift = -1;
  (void)pool_accel_dirty_bitmap(pool, 0, depth, &brick_shift);  // (the shape this pool's bricks have, or will have, at this depth)
  if (depth < brick_node_level(brick_shift < 0 ? 0 : brick_shift)) brick_shift = -1;
  const int fill_tiles = ws->deferred_tiles;
  const long long rmax = max_records(n, depth);
  int tile_blocks = (int)cdiv(8 * rmax, 256);
  if (tile_blocks > 4096) tile_blocks = 4096;
  const int strad_first = fill_tiles, strad_end = depth * fill_tiles;
  const int strad_blocks = (int)cdiv((long long)strad_end - strad_first, 2048);
  keyrange_pack_header_kernel<<<1, 256, 0, stream>>>(d_delta, delta_bytes / 4, small_bucket_base(ws), small_counts(ws), small_n0(ws), win, depth);
  keyrange_pack_tiles_kernel<<<tile_blocks, 256, 0, stream>>>(d_delta, pool->d_data, shadow, epoch, ws->rec_key.as<u64>(), ws->rec_front.as<u32>(),
                                                              ws->rec_pass.as<unsigned char>(), brick_shift);
  keyrange_pack_words_kernel<<<fill_tiles + strad_blocks, 256, 0, stream>>>(d_delta, shadow, ws->apply_nodes.as<u32>(), fill_tiles, kFillThreads * depth,
                                                                            ws->strad.as<u32>(), strad_first, strad_end);
  SVO_LAUNCH_CHECK();
  pool_shadow_end(pool);
  ws->deferred_pool = nullptr;
  ws->keyrange_pool = pool;
  return SVOSLAM_OK;
}

int svo_fuse_keyrange_apply(svoslam_workspace *ws, const unsigned long long *d_keys, int n, int depth, svoslam_pool *pool,
                            const uint32_t *const *d_deltas, int world, hipStream_t stream) {
  if (!ws || !pool || !d_deltas || !d_keys || n <= 0 || world < 1 || world > kKrMaxWorld) return SVOSLAM_ERR_INVALID_ARG;
  if (ws->keyrange_pool != pool) return SVOSLAM_ERR_INVALID_ARG;  // svo_fuse_keyrange_commit of this frame has not run on this workspace
  ws->keyrange_pool = nullptr;
  KrDeltas D;
  for (int s = 0; s < kKrMaxWorld; s++) D.d[s] = s < world ? d_deltas[s] : nullptr;
  for (int s = 0; s < world; s++) if (!D.d[s]) return SVOSLAM_ERR_INVALID_ARG;
  PoolTracker *trk = tracker_of(pool);
  int brick_shift = -1;
  u32 *dirty = pool_accel_dirty_bitmap(pool, 0, depth, &brick_shift);  // direct-commit state; nullptr: not a registered pool
  const long long rmax = max_records(n, depth);
  int blocks = (int)cdiv(8 * rmax / (world > 1 ? world : 1) + 1, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 64) blocks = 64;
  keyrange_setup_kernel<<<1, 256, 0, stream>>>(D, world, kr_table(ws), kr_topmap(ws), kr_scal(ws), pool->d_data);
  keyrange_apply_kernel<<<dim3((unsigned)blocks, (unsigned)world), 256, 0, stream>>>(D, kr_table(ws), kr_topmap(ws), kr_scal(ws), pool->d_data, dirty);
  keyrange_mark_kernel<<<cdiv(n, 256), 256, 0, stream>>>(d_keys, n, depth, dirty, brick_shift, kr_top(ws));
  SVO_TRY(tracker_make_room(pool));
  keyrange_finish_kernel<<<1, 256, 0, stream>>>(pool->d_data, kr_scal(ws), kr_top(ws), pool->d_size, trk ? trk->h_size : nullptr,
                                                trk ? trk->d_slot : nullptr, dirty);
  SVO_LAUNCH_CHECK();
  pool->pending += 1;
  return tracker_push(pool, ws->keyrange_bound, stream);
}

// a svo_fuse_keyrange_commit whose delta is wanted but whose apply will not follow on this pool (the deltas of OTHER ranks, produced on one
// device for an emulated rank: bench.py --exchange keyrange --emulate-rank): the plan's reservation is released, the pool is as it was
int svo_fuse_keyrange_discard(svoslam_workspace *ws, svoslam_pool *pool) {
  if (!ws || !pool || ws->keyrange_pool != pool) return SVOSLAM_ERR_INVALID_ARG;
  ws->keyrange_pool = nullptr;
  pool->pending_bound -= ws->keyrange_bound;
  if (pool->pending_bound < 0) pool->pending_bound = 0;
  return SVOSLAM_OK;
}

// flags of the svo_fuse_keyrange_apply calls on this workspace since the last call of this function (blocking; the flags are sticky on
// the device and cleared here): 0 = every frame applied; kKrOverflowed / kKrMismatch = a frame was NOT applied (the replica
// is then behind the others)
int svo_fuse_keyrange_status(svoslam_workspace *ws, int *flags, hipStream_t stream) {
  if (!ws || !flags || ws->kr_small.bytes == 0) return SVOSLAM_ERR_INVALID_ARG;
  u32 f = 0;
  SVO_HIP(hipMemcpyAsync(&f, kr_scal(ws) + 1, 4, hipMemcpyDeviceToHost, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  *flags = (int)f;
  if (f) SVO_HIP(memset_sync(kr_scal(ws) + 1, 0, 4));
  return SVOSLAM_OK;
}

}  // namespace svoslam
