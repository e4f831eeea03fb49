// mesh.hip -- mesh -> voxel grid on gfx950: replaces voxelization::meshToVoxelGrid
// (include/octree_slam/world/voxelization/voxelization.h:21, src/world/voxelization/
// voxelization.cu:50-139,219-236,381-405) and the VoxelPipe instantiation it relies on
// (THIN_RASTER, NO_BLENDING, Float/FP32S; external/include/voxelpipe/coarse.h:59-102,
// utils.h:185-254, fine.h:130-152,239-365,936-959), plus the host-side OBJ/BMP loaders
// (external/src/objUtil/objloader.cpp:14-122, obj.cpp:33-135,227-238; src/world/scene.cpp:35-62).
//
// Organisation: VoxelPipe bins triangles into 8^3 tiles, radix-sorts the (tile, triangle) pairs,
// rasterises per tile into a dense N^3 framebuffer (64 MiB at N = 256) and the reference then
// scans all N^3 cells for occupied ones (with N^3 workgroups).  Here the same per-cell rule is
// evaluated per (triangle, scanline) straight into a SPARSE fragment list (tiled cell index,
// triangle id), which the library's radix sort orders; the last fragment of each run (highest
// triangle id = the deterministic reading of NO_BLENDING's last-writer-wins) becomes the voxel.
// No dense grid exists, so N = 2^10 .. 2^16 per axis cost what their surface costs.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "config.hpp"
#include "mesh.hpp"
#include "stage_timing.hpp"
#include "radix_sort.hpp"
#include "wave_rank.hpp"

namespace svoslam {

typedef unsigned long long u64;
typedef unsigned int u32;

struct GridParams {
  float bbox0[3], bbox1[3], delta[3], inv_delta[3];
  int log_N, log_T, N, T;
};

// per-triangle raster setup (coarse.h:59-102, utils.h:185-254), recomputed where needed
struct TriSetup {
  int lo[3], hi[3];
  int axis, U, V, W;
  float a[3], ndu[3], ndv[3], inv_du[3];
  float px, py, pz;
  float n[3], v0[3];
};

__device__ inline float pick(const float *a, int k) { return k == 0 ? a[0] : (k == 1 ? a[1] : a[2]); }

__device__ inline void setup_triangle(const float *__restrict__ vbo, int t, const GridParams &G, TriSetup &S) {
  float v0[3], v1[3], v2[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { v0[k] = vbo[9 * (size_t)t + k]; v1[k] = vbo[9 * (size_t)t + 3 + k]; v2[k] = vbo[9 * (size_t)t + 6 + k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float lo = (v0[k] - G.bbox0[k]) * G.inv_delta[k];
    lo = fminf((v1[k] - G.bbox0[k]) * G.inv_delta[k], lo);
    lo = fminf((v2[k] - G.bbox0[k]) * G.inv_delta[k], lo);
    float hi = (v0[k] - G.bbox0[k]) * G.inv_delta[k];
    hi = fmaxf((v1[k] - G.bbox0[k]) * G.inv_delta[k], hi);
    hi = fmaxf((v2[k] - G.bbox0[k]) * G.inv_delta[k], hi);
    int a = (int)lo; a = a < 0 ? 0 : a; a = a > G.N - 1 ? G.N - 1 : a;
    int b = (int)ceilf(hi); b = b < 0 ? 0 : b; b = b > G.N - 1 ? G.N - 1 : b;
    S.lo[k] = a; S.hi[k] = b;
    S.v0[k] = v0[k];
  }
  float e0[3], e1[3], e2[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { e0[k] = v1[k] - v0[k]; e1[k] = v2[k] - v1[k]; e2[k] = v0[k] - v2[k]; }
  // anti_cross(edge0, edge2), utils.h:97-103
  S.n[0] = e0[2] * e2[1] - e0[1] * e2[2];
  S.n[1] = e0[0] * e2[2] - e0[2] * e2[0];
  S.n[2] = e0[1] * e2[0] - e0[0] * e2[1];
  const bool byx = fabsf(S.n[1]) > fabsf(S.n[0]), byz = fabsf(S.n[1]) > fabsf(S.n[2]), bzx = fabsf(S.n[2]) > fabsf(S.n[0]);
  S.axis = byx ? (byz ? 1 : 2) : (bzx ? 2 : 0);
  // utils.h:114-176: (u, v, w) of each dominant axis and the winding sign
  S.U = S.axis == 0 ? 1 : 0;
  S.V = S.axis == 2 ? 1 : 2;
  S.W = S.axis;
  const float sgn = S.axis == 0 ? (S.n[0] > 0.0f ? 1.0f : -1.0f) : S.axis == 1 ? (S.n[1] < 0.0f ? 1.0f : -1.0f) : (S.n[2] > 0.0f ? 1.0f : -1.0f);
  const float *vv[3] = {v0, v1, v2};
  const float *ee[3] = {e0, e1, e2};
#pragma unroll
  for (int k = 0; k < 3; k++) {  // triangle_setup, utils.h:185-232
    const float nx = -pick(ee[k], S.V) * sgn, ny = pick(ee[k], S.U) * sgn;
    const float d = -(nx * pick(vv[k], S.U) + ny * pick(vv[k], S.V)) + fmaxf(0.0f, pick(G.delta, S.U) * nx) + fmaxf(0.0f, pick(G.delta, S.V) * ny);
    S.a[k] = (nx * pick(G.bbox0, S.U) + ny * pick(G.bbox0, S.V)) + d;
    S.ndu[k] = nx * pick(G.delta, S.U);
    S.ndv[k] = ny * pick(G.delta, S.V);
    S.inv_du[k] = 1.0f / S.ndu[k];
  }
  // plane_setup, utils.h:236-254 (__frcp_rn = correctly rounded reciprocal)
  const float inv_n = 1.0f / pick(S.n, S.W);
  S.px = pick(S.n, S.U) * inv_n;
  S.py = pick(S.n, S.V) * inv_n;
  S.pz = S.px * pick(v0, S.U) + S.py * pick(v0, S.V) + pick(v0, S.W) - pick(G.bbox0, S.W) - S.px * pick(G.bbox0, S.U) - S.py * pick(G.bbox0, S.V);
}

// number of scanlines (v range of the integer bbox) per triangle
__global__ __launch_bounds__(256) void tri_scanline_count_kernel(const float *__restrict__ vbo, int n_tris, GridParams G,
                                                                 u32 *__restrict__ counts) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tris) return;
  TriSetup S;
  setup_triangle(vbo, t, G, S);
  counts[t] = (u32)(S.hi[S.V] - S.lo[S.V] + 1);
}

// One workgroup per 256 consecutive (triangle, scanline) pairs.  EMIT = false: count the fragments of each
// scanline; EMIT = true: write them at the scanned offsets.  Per cell the rule is VoxelPipe's: conservative
// 2-D edge functions give a u range per scanline (compute_scanline_bounds, fine.h:130-152), the plane gives
// ONE w per (u, v) (rasterize_scanline, fine.h:318-341), kept if the tile that holds (u, v, w) overlaps the
// triangle's integer bbox and passes the tile/plane test.
// Each thread sets up its own scanline (u range, plane) into LDS; the cells of all 256 scanlines are then
// walked by the whole workgroup (cell c -> scanline by search in the LDS prefix of the u-range lengths), so a
// 65536-cell scanline of a long thin triangle costs 256 iterations, not 65536.  Fragments of one scanline have
// distinct cells, so their order inside the scanline's slot range does not matter (LDS atomic slot counter).
constexpr unsigned kPieceCells = 32768;  // candidate cells per raster workgroup (128 rounds of its 256 lanes)
struct ScanlineSetup {
  int t, v, min_u, axis, tw_lo, tw_hi;
  float px, py, pz, vf, n[3], r1, r2;
  u32 len;  // cells of the scan line's u range (0: none)
};
static_assert(sizeof(ScanlineSetup) == 64, "one 64-byte record per scan line");

// The set-up of every scan line ONCE, as a 64-byte record (round 5): the count and the emit pass used to repeat it (a binary
// search over the triangles' scan-line offsets and the triangle's raster set-up per lane), and so would every piece of a chunk.
__global__ __launch_bounds__(256) void scanline_setup_kernel(const float *__restrict__ vbo, int n_tris, const u32 *__restrict__ tri_start,
                                                             u32 total_scanlines, GridParams G, ScanlineSetup *__restrict__ out,
                                                             u32 *__restrict__ pieces) {
  // pieces[chunk] (chunk = this workgroup's 256 scan lines): how many workgroups the raster passes give the chunk -- one per
  // kPieceCells candidate cells.  Config 5's stand-in has 33.6 M scan lines in 131 130 chunks with a median of 512 cells, and 100
  // chunks (the faces whose scan lines run along the long axis) of 1-7 M cells: a uniform number of slices either leaves those
  // to a few workgroups or multiplies 131 000 nearly empty ones.
  __shared__ u32 tmp_red[4];
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  u32 my_len = 0;
  if (s < total_scanlines) {
  // triangle of this scanline: last t with tri_start[t] <= s
  int lo = 0, hi = n_tris - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tri_start[mid] <= s) lo = mid; else hi = mid - 1;
  }
  const int t = lo;
  TriSetup S;
  setup_triangle(vbo, t, G, S);
  const int v = S.lo[S.V] + (int)(s - tri_start[t]);
  const float b[3] = {S.a[0] + (float)v * S.ndv[0], S.a[1] + (float)v * S.ndv[1], S.a[2] + (float)v * S.ndv[2]};
  int min_u = S.lo[S.U], max_u = S.hi[S.U];
  bool invalid = false;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (S.ndu[k] > 0.0f) { const int c = (int)ceilf(-b[k] * S.inv_du[k]); min_u = c > min_u ? c : min_u; }
    else if (S.ndu[k] < 0.0f) { const int c = (int)(-b[k] * S.inv_du[k]); max_u = c < max_u ? c : max_u; }
    else if (b[k] < 0.0f) invalid = true;
  }
  ScanlineSetup Q;
  Q.len = (!invalid && max_u >= min_u) ? (u32)(max_u - min_u + 1) : 0u;
  Q.t = t; Q.v = v; Q.min_u = min_u; Q.axis = S.axis;
  Q.tw_lo = S.lo[S.W] >> G.log_T; Q.tw_hi = S.hi[S.W] >> G.log_T;
  Q.px = S.px; Q.py = S.py; Q.pz = S.pz;
  Q.vf = ((float)v + 0.5f) * pick(G.delta, S.V);
  Q.n[0] = S.n[0]; Q.n[1] = S.n[1]; Q.n[2] = S.n[2];
  // fine.h:936-959, the two tile-independent terms of the plane/tile test
  const float T = (float)G.T;
  const float c0 = S.n[0] > 0 ? G.delta[0] * T : 0.0f, c1 = S.n[1] > 0 ? G.delta[1] * T : 0.0f, c2 = S.n[2] > 0 ? G.delta[2] * T : 0.0f;
  Q.r1 = S.n[0] * (c0 - S.v0[0]) + S.n[1] * (c1 - S.v0[1]) + S.n[2] * (c2 - S.v0[2]);
  Q.r2 = S.n[0] * (G.delta[0] * T - c0 - S.v0[0]) + S.n[1] * (G.delta[1] * T - c1 - S.v0[1]) + S.n[2] * (G.delta[2] * T - c2 - S.v0[2]);
  out[s] = Q;
  my_len = Q.len;
  }
  u32 chunk_cells;
  (void)block256_exclusive_scan(my_len, tmp_red, chunk_cells);
  if (threadIdx.x == 0) {
    u32 p = (chunk_cells + kPieceCells - 1u) / kPieceCells;
    pieces[blockIdx.x] = p < 1u ? 1u : (p > 256u ? 256u : p);
  }
}

// piece -> chunk: chunk c owns pieces [piece_base[c], piece_base[c] + pieces)
__global__ __launch_bounds__(256) void piece_map_kernel(const u32 *__restrict__ piece_base, u32 chunks, u32 total_pieces, u32 *__restrict__ map) {
  const u32 c = blockIdx.x * 256u + threadIdx.x;
  if (c >= chunks) return;
  const u32 b = piece_base[c], e = c + 1 < chunks ? piece_base[c + 1] : total_pieces;
  for (u32 p = b; p < e; p++) map[p] = c;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void scanline_kernel(const ScanlineSetup *__restrict__ lines, u32 total_scanlines, GridParams G,
                                                       u32 *__restrict__ frag_count, const u32 *__restrict__ frag_start,
                                                       u64 *__restrict__ frag_key, u32 *__restrict__ frag_tri, int pack_shift,
                                                       const u32 *__restrict__ piece_base, const u32 *__restrict__ piece_map, u32 chunks,
                                                       u32 total_pieces) {
  // One workgroup per PIECE of a chunk of 256 scan lines (scanline_setup_kernel): piece y of the chunk's S pieces takes the cells
  // [total * y / S, total * (y + 1) / S) of the chunk and counts / emits into entry 256 * piece_base[chunk] + lane * S + y of
  // frag_count / frag_start -- scan-line-major, so that the fragments of scan line s precede those of s + 1 whatever S is.
  // pack_shift >= 0 (round 5): ONE word per fragment, framebuffer index << pack_shift | triangle id, for the packed sort
  // (radix_sort.hip) -- 8 bytes per fragment and pass instead of 8 + 4 in two arrays
  __shared__ ScanlineSetup setup[256];
  __shared__ u32 cell_prefix[257], slot[256], out_base[256], tmp[4];
  const u32 chunk = piece_map[blockIdx.x];
  const u32 pb = piece_base[chunk];
  const u32 slices = (chunk + 1 < chunks ? piece_base[chunk + 1] : total_pieces) - pb, slice = blockIdx.x - pb;
  const size_t entry = (size_t)256 * pb + (size_t)threadIdx.x * slices + slice;
  const u32 s = chunk * 256u + threadIdx.x;
  u32 len = 0;
  if (s < total_scanlines) {
    setup[threadIdx.x] = lines[s];  // (one 64-byte record: scanline_setup_kernel)
    len = setup[threadIdx.x].len;
  }
  u32 total_cells;
  const u32 ex = block256_exclusive_scan(len, tmp, total_cells);
  cell_prefix[threadIdx.x] = ex;
  if (threadIdx.x == 255) cell_prefix[256] = total_cells;
  slot[threadIdx.x] = 0;
  out_base[threadIdx.x] = (EMIT && s < total_scanlines) ? frag_start[entry] : 0u;
  __syncthreads();

  const int M = 1 << (G.log_N - G.log_T);
  u32 sl = 0;
  const u32 c_begin = (u32)((unsigned long long)total_cells * slice / (unsigned)slices);
  const u32 c_end = (u32)((unsigned long long)total_cells * (slice + 1u) / (unsigned)slices);
  for (u32 c = c_begin + threadIdx.x; c < c_end; c += 256u) {
    if (!(cell_prefix[sl] <= c && c < cell_prefix[sl + 1])) {  // last sl with cell_prefix[sl] <= c
      u32 lo = 0, hi = 255;
      while (lo < hi) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (cell_prefix[mid] <= c) lo = mid; else hi = mid - 1;
      }
      sl = lo;
    }
    const ScanlineSetup &Q = setup[sl];
    const int W = Q.axis, U = W == 0 ? 1 : 0, V = W == 2 ? 1 : 2;
    const int u = Q.min_u + (int)(c - cell_prefix[sl]);
    const float uf = ((float)u + 0.5f) * pick(G.delta, U);
    const float wf = Q.pz - (Q.px * uf + Q.py * Q.vf);
    const int w = (int)(wf * pick(G.inv_delta, W));
    // (selects, not xyz[U] = u: an array indexed by a runtime axis lives in scratch memory -- a store and three loads per cell)
    // axis 0: (U, V, W) = (1, 2, 0); axis 1: (0, 2, 1); axis 2: (0, 1, 2)
    const int xyz[3] = {W == 0 ? w : u, W == 0 ? u : (W == 1 ? w : Q.v), W == 2 ? w : Q.v};
    (void)V;
    // the tile holding (u, v, w) must be one of the tiles of the triangle's integer bbox ...
    const int tw = w >> G.log_T;
    bool keep = !(w < 0 || tw < Q.tw_lo || tw > Q.tw_hi);
    // ... and pass the plane test
    const int tx = (xyz[0] >> G.log_T) << G.log_T, ty = (xyz[1] >> G.log_T) << G.log_T, tz = (xyz[2] >> G.log_T) << G.log_T;
    const float np = Q.n[0] * (G.bbox0[0] + (float)tx * G.delta[0]) + Q.n[1] * (G.bbox0[1] + (float)ty * G.delta[1]) + Q.n[2] * (G.bbox0[2] + (float)tz * G.delta[2]);
    keep = keep && ((np + Q.r1) * (np + Q.r2) <= 0.0f);
    // The slot of the fragment inside its scan line's range.  The 64 cells of a wavefront almost always belong to ONE scan line (they
    // are up to 65536 cells long): one LDS atomic for the wavefront and a popcount rank instead of 64 returning atomics on one
    // address, which the LDS serialises -- at 2^16 cells per axis that was the kernel (round 5: 8 % VALU issue, 8.5 ms per pass).
    const unsigned long long km = __ballot(keep);
    if (!km) continue;
    const int leader = __ffsll((long long)km) - 1;
    const u32 sl0 = (u32)__shfl((int)sl, leader);
    u32 k = 0;
    if (__all(!keep || sl == sl0)) {
      u32 base = 0;
      if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&slot[sl0], (u32)__popcll(km));
      k = (u32)__shfl((int)base, leader) + (u32)__popcll(km & ((1ull << (threadIdx.x & 63u)) - 1ull));
    } else if (keep) {
      k = atomicAdd(&slot[sl], 1u);
    }
    if (!keep) continue;
    if (EMIT) {
      const u64 tile = (u64)(xyz[0] >> G.log_T) + (u64)M * (u64)(xyz[1] >> G.log_T) + (u64)M * M * (u64)(xyz[2] >> G.log_T);
      const u64 pix = (u64)(xyz[0] & (G.T - 1)) + (u64)G.T * (u64)(xyz[1] & (G.T - 1)) + (u64)G.T * G.T * (u64)(xyz[2] & (G.T - 1));
      const u32 pos = out_base[sl] + k;
      const u64 fb = tile * (u64)G.T * G.T * G.T + pix;  // fb index of voxelization.cu:141-164
      if (pack_shift >= 0) {
        frag_key[pos] = (fb << pack_shift) | (u64)(u32)Q.t;
      } else {
        frag_key[pos] = fb;
        frag_tri[pos] = (u32)Q.t;
      }
    }
  }
  if (!EMIT) {
    __syncthreads();
    frag_count[entry] = s < total_scanlines ? slot[threadIdx.x] : 0u;  // (every entry of the piece is written: the scan reads them all)
  }
}

// last fragment of each run of equal cell indices (stable sort: highest triangle id) -> flag + tile count
__global__ __launch_bounds__(256) void voxel_flag_kernel(const u64 *__restrict__ skey, u32 n, u32 *__restrict__ tile_cnt) {
  __shared__ u32 tmp[4];
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  const u32 is_last = (i < n && (i + 1 == n || skey[i + 1] != skey[i])) ? 1u : 0u;
  u32 total;
  (void)block256_exclusive_scan(is_last, tmp, total);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// ColorShader::shade, voxelization.cu:90-138 (vertex-0 texel, no interpolation; alpha 127)
__device__ inline u32 shade(u32 tri, const float *__restrict__ tex, int tw, int th, const float *__restrict__ tbo, int tbosize) {
  if (tw == 0) return (255u << 8) + (127u << 24);
  float cr, cg, cb;
  if (tbosize == 0) {
    const int r = (int)((double)tex[0] * 255.0), g = (int)((double)tex[1] * 255.0), b = (int)((double)tex[2] * 255.0);
    return (u32)(r + (g << 8) + (b << 16)) + (127u << 24);
  }
  const int tx = (int)(tbo[6 * (size_t)tri] * (float)tw);
  const int ty = (int)(tbo[6 * (size_t)tri + 1] * (float)th);
  long long idx = (long long)ty * tw + tx;
  // u or v == 1.0 indexes past the row/image in the reference (out-of-bounds read); clamp into the image
  idx = idx < 0 ? 0 : idx;
  idx = idx > (long long)tw * th - 1 ? (long long)tw * th - 1 : idx;
  cr = tex[3 * idx]; cg = tex[3 * idx + 1]; cb = tex[3 * idx + 2];
  const int r = (int)(fminf(fmaxf(cr, 0.0f), 1.0f) * 255.0f);
  const int g = (int)(fminf(fmaxf(cg, 0.0f), 1.0f) * 255.0f);
  const int b = (int)(fminf(fmaxf(cb, 0.0f), 1.0f) * 255.0f);
  return (u32)(r + (g << 8) + (b << 16)) + (127u << 24);
}

// createVoxelGrid + getCenterFromIndex (voxelization.cu:50-76,219-236), compacting as it goes
__global__ __launch_bounds__(256) void voxel_emit_kernel(const u64 *__restrict__ skey, const u32 *__restrict__ stri, u32 n,
                                                         const u32 *__restrict__ tile_prefix, GridParams G,
                                                         const float *__restrict__ tex, int tw, int th,
                                                         const float *__restrict__ tbo, int tbosize,
                                                         float4 *__restrict__ centers, float4 *__restrict__ colors,
                                                         u64 *__restrict__ indices) {
  __shared__ u32 tmp[4];
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  const u32 is_last = (i < n && (i + 1 == n || skey[i + 1] != skey[i])) ? 1u : 0u;
  u32 total;
  const u32 pos = tile_prefix[blockIdx.x] + block256_exclusive_scan(is_last, tmp, total);
  if (!is_last) return;
  const u64 idx = skey[i];
  const int M = 1 << (G.log_N - G.log_T), T = G.T;
  const u64 T3 = (u64)T * T * T;
  const u64 tile_num = idx / T3, pix_num = idx % T3;
  const int tz = (int)(tile_num / ((u64)M * M) % M), pz = (int)(pix_num / (u64)(T * T) % T);
  const int ty = (int)(tile_num / M % M), py = (int)(pix_num / T % T);
  const int tx = (int)(tile_num % M), px = (int)(pix_num % T);
  float t_d[3], p_d[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { t_d[k] = (G.bbox1[k] - G.bbox0[k]) / (float)M; p_d[k] = t_d[k] / (float)T; }
  const float cx = G.bbox0[0] + (float)tx * t_d[0] + (float)px * p_d[0] + p_d[0] / 2.0f;
  const float cy = G.bbox0[1] + (float)ty * t_d[1] + (float)py * p_d[1] + p_d[1] / 2.0f;
  const float cz = G.bbox0[2] + (float)tz * t_d[2] + (float)pz * p_d[2] + p_d[2] / 2.0f;
  centers[pos] = make_float4(cx, cy, cz, 1.0f);
  const int color = (int)shade(stri[i], tex, tw, th, tbo, tbosize);
  // Q21: the reference leaves colors[].a unwritten; 0 here
  colors[pos] = make_float4((float)((double)(color & 0xFF) / 255.0), (float)((double)((color >> 8) & 0xFF) / 255.0),
                            (float)((double)((color >> 16) & 0xFF) / 255.0), 0.0f);
  if (indices) indices[pos] = idx;
}

static GridParams make_grid(const float bbox0[3], const float bbox1[3], int log_N, int log_T) {
  GridParams G;
  G.log_N = log_N; G.log_T = log_T; G.N = 1 << log_N; G.T = 1 << log_T;
  for (int k = 0; k < 3; k++) {
    G.bbox0[k] = bbox0[k]; G.bbox1[k] = bbox1[k];
    G.delta[k] = (bbox1[k] - bbox0[k]) / (float)G.N;      // voxelpipe_inline.h:111-119
    G.inv_delta[k] = (float)G.N / (bbox1[k] - bbox0[k]);
  }
  return G;
}

int mesh_to_voxel_grid(svoslam_workspace *ws, const svoslam_mesh *mesh, const svoslam_texture *tex, int log_N, int log_T,
                       float **d_centers, float **d_colors, unsigned long long **d_indices, int32_t *n_out, float *scale_out,
                       hipStream_t stream) {
  if (!ws || !mesh || !d_centers || !d_colors || !n_out) return SVOSLAM_ERR_INVALID_ARG;
  if (log_T < 1 || log_N < log_T || log_N > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  *d_centers = nullptr; *d_colors = nullptr; *n_out = 0;
  if (d_indices) *d_indices = nullptr;
  if (scale_out) *scale_out = (mesh->bbox1[0] - mesh->bbox0[0]) / (float)(1 << log_N) / 2.0f;  // computeScale, :78-80
  const int n_tris = mesh->n_tris;
  ws->mesh_fragments = 0;
  if (n_tris <= 0) return SVOSLAM_OK;
  const GridParams G = make_grid(mesh->bbox0, mesh->bbox1, log_N, log_T);
  // upload mesh + texture
  DeviceBuffer &dv = ws->bfs_a, &dt = ws->bfs_b, &dx = ws->misc;
  SVO_TRY(dv.reserve((size_t)n_tris * 9 * 4));
  SVO_HIP(hipMemcpyAsync(dv.ptr, mesh->vbo, (size_t)n_tris * 9 * 4, hipMemcpyHostToDevice, stream));
  const int tbosize = mesh->tbo ? mesh->tbosize : 0;
  SVO_TRY(dt.reserve((size_t)(tbosize > 0 ? tbosize : 1) * 4));
  if (tbosize > 0) SVO_HIP(hipMemcpyAsync(dt.ptr, mesh->tbo, (size_t)tbosize * 4, hipMemcpyHostToDevice, stream));
  const int tw = tex && tex->data ? tex->width : 0, th = tex && tex->data ? tex->height : 0;
  SVO_TRY(dx.reserve((size_t)(tw * th > 0 ? tw * th : 1) * 12));
  if (tw * th > 0) SVO_HIP(hipMemcpyAsync(dx.ptr, tex->data, (size_t)tw * th * 12, hipMemcpyHostToDevice, stream));
  // scanlines per triangle -> exclusive scan
  SVO_TRY(ws->reserve_small());
  SVO_TRY(ws->leaf_f.reserve((size_t)n_tris * 4));
  u32 *tri_start = ws->leaf_f.as<u32>();
  u32 *d_total = ws->small.as<u32>();
  long long tk_raster = -1, tk_sort = -1, tk_emit = -1;  // per-stage event brackets (svoslam_stage_timing; off by default)
  (void)stage_begin(kStageMeshRaster, stream, &tk_raster);
  tri_scanline_count_kernel<<<cdiv(n_tris, 256), 256, 0, stream>>>(dv.as<float>(), n_tris, G, tri_start);
  SVO_TRY(exclusive_scan_u32(ws, tri_start, (u32)n_tris, d_total, stream));
  u32 total_scan = 0;
  SVO_HIP(hipMemcpyAsync(&total_scan, d_total, 4, hipMemcpyDeviceToHost, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  if (total_scan == 0) { (void)stage_end(kStageMeshRaster, tk_raster, stream); return SVOSLAM_OK; }
  // every scan line's set-up, once (64 bytes each) + the pieces of every chunk of 256 scan lines; then fragments per (scan line,
  // piece) -> exclusive scan -> emit
  const u32 chunks = cdiv(total_scan, 256);
  SVO_TRY(ws->path_nodes.reserve((size_t)total_scan * sizeof(ScanlineSetup)));
  SVO_TRY(ws->leaf_t.reserve((size_t)(chunks + 1) * 4));
  ScanlineSetup *lines = ws->path_nodes.as<ScanlineSetup>();
  u32 *piece_base = ws->leaf_t.as<u32>();
  scanline_setup_kernel<<<chunks, 256, 0, stream>>>(dv.as<float>(), n_tris, tri_start, total_scan, G, lines, piece_base);
  SVO_LAUNCH_CHECK();
  SVO_TRY(exclusive_scan_u32(ws, piece_base, chunks, d_total, stream));
  u32 total_pieces = 0;
  SVO_HIP(hipMemcpyAsync(&total_pieces, d_total, 4, hipMemcpyDeviceToHost, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  if ((unsigned long long)total_pieces * 256ull > 0x7FFFFFFFull) {
    set_last_error_text("mesh_to_voxel_grid: %u raster pieces (of 32768 candidate cells) x 256 count entries exceed 2^31 (%u scan lines)", total_pieces, total_scan);
    (void)stage_end(kStageMeshRaster, tk_raster, stream);
    return SVOSLAM_ERR_OOM;
  }
  SVO_TRY(ws->leaf_rec0.reserve((size_t)total_pieces * 4));
  u32 *piece_map = ws->leaf_rec0.as<u32>();
  piece_map_kernel<<<cdiv(chunks, 256), 256, 0, stream>>>(piece_base, chunks, total_pieces, piece_map);
  SVO_LAUNCH_CHECK();
  const u32 count_entries = total_pieces * 256u;
  SVO_TRY(ws->rec_front.reserve((size_t)count_entries * 4));
  u32 *frag_start = ws->rec_front.as<u32>();
  const unsigned raster_grid = total_pieces;
  scanline_kernel<false><<<raster_grid, 256, 0, stream>>>(lines, total_scan, G, frag_start, nullptr, nullptr, nullptr, -1, piece_base, piece_map,
                                                         chunks, total_pieces);
  SVO_TRY(exclusive_scan_u32(ws, frag_start, count_entries, d_total, stream));
  u32 total_frag = 0;
  SVO_HIP(hipMemcpyAsync(&total_frag, d_total, 4, hipMemcpyDeviceToHost, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  if (total_frag == 0) { (void)stage_end(kStageMeshRaster, tk_raster, stream); return SVOSLAM_OK; }
  if (total_frag > 0x7FFFFFFFu) {
    set_last_error_text("mesh_to_voxel_grid: %u (cell, triangle) fragments exceed 2^31", total_frag);
    (void)stage_end(kStageMeshRaster, tk_raster, stream);
    return SVOSLAM_ERR_OOM;
  }
  const int nf = (int)total_frag;
  ws->mesh_fragments = nf;
  SVO_TRY(ws->keys_a.reserve((size_t)nf * 8));
  SVO_TRY(ws->keys_b.reserve((size_t)nf * 8));
  SVO_TRY(ws->vals_a.reserve((size_t)nf * 4));
  SVO_TRY(ws->vals_b.reserve((size_t)nf * 4));
  SVO_TRY(ws->tile_hist.reserve(256 * ((size_t)cdiv(nf, 256) + 1) * 4));
  // one packed word per fragment where the framebuffer index (3 log_N bits) and the triangle id fit 64 bits
  int tri_bits = 1;
  while ((1ll << tri_bits) < (long long)n_tris) tri_bits++;
  const int pack_shift = (3 * log_N + tri_bits <= 64 && config().sort_pairs == 0) ? tri_bits : -1;
  scanline_kernel<true><<<raster_grid, 256, 0, stream>>>(lines, total_scan, G, nullptr, frag_start, ws->keys_a.as<u64>(), ws->vals_a.as<u32>(),
                                                        pack_shift, piece_base, piece_map, chunks, total_pieces);
  SVO_LAUNCH_CHECK();
  (void)stage_end(kStageMeshRaster, tk_raster, stream);
  // order by framebuffer index (stable: equal cells keep ascending triangle id -- fragments are emitted triangle by triangle)
  u64 *skey = nullptr; u32 *stri = nullptr;
  (void)stage_begin(kStageMeshSort, stream, &tk_sort);
  if (pack_shift >= 0)
    SVO_TRY(radix_sort_packed_ex(ws, nf, 3 * log_N, pack_shift, radix_packed_digit_bits_for(nf), false, true, stream, &skey, &stri));
  else
    SVO_TRY(radix_sort_pairs(ws, nf, 3 * log_N, stream, &skey, &stri, false));
  (void)stage_end(kStageMeshSort, tk_sort, stream);
  const int tiles = (int)cdiv(nf, 256);
  u32 *tile_cnt = ws->tile_hist.as<u32>();
  (void)stage_begin(kStageMeshEmit, stream, &tk_emit);
  voxel_flag_kernel<<<tiles, 256, 0, stream>>>(skey, (u32)nf, tile_cnt);
  SVO_TRY(exclusive_scan_u32(ws, tile_cnt, (u32)tiles, d_total, stream));
  u32 n_vox = 0;
  SVO_HIP(hipMemcpyAsync(&n_vox, d_total, 4, hipMemcpyDeviceToHost, stream));
  SVO_HIP(hipStreamSynchronize(stream));
  float *ce = nullptr, *co = nullptr;
  u64 *ix = nullptr;
  SVO_HIP(hipMalloc((void **)&ce, (size_t)n_vox * 16));
  SVO_HIP(hipMalloc((void **)&co, (size_t)n_vox * 16));
  if (d_indices) SVO_HIP(hipMalloc((void **)&ix, (size_t)n_vox * 8));
  voxel_emit_kernel<<<tiles, 256, 0, stream>>>(skey, stri, (u32)nf, tile_cnt, G, dx.as<float>(), tw, th, dt.as<float>(), tbosize,
                                               reinterpret_cast<float4 *>(ce), reinterpret_cast<float4 *>(co), ix);
  SVO_LAUNCH_CHECK();
  (void)stage_end(kStageMeshEmit, tk_emit, stream);
  SVO_HIP(hipStreamSynchronize(stream));
  // the scan lines' set-up records (64 bytes each: 2.1 GB at 2^16 cells per axis) do not stay with the workspace
  if (ws->path_nodes.bytes > ((size_t)256 << 20)) ws->path_nodes.release();
  *d_centers = ce; *d_colors = co; *n_out = (int32_t)n_vox;
  if (d_indices) *d_indices = ix;
  return SVOSLAM_OK;
}

// voxelization::voxelGridToMesh (voxelization.cu:325-379) / createCubeMesh (:184-217): one copy of the cube mesh per voxel.
// The reference runs one thread per voxel over all of its floats (stride-vbosize stores); here one thread per
// output element, so every store of a wavefront is contiguous.  The index offset is the reference's own
// (idx * cube_ibosize, not the vertex offset).
__global__ __launch_bounds__(256) void cube_mesh_kernel(const float4 *__restrict__ centers, const float4 *__restrict__ colors,
                                                       float scale_factor, long long total_v, long long total_i,
                                                       const float *__restrict__ cube_vbo, int cube_vbosize,
                                                       const int *__restrict__ cube_ibo, int cube_ibosize,
                                                       const float *__restrict__ cube_nbo, float *__restrict__ out_vbo,
                                                       int *__restrict__ out_ibo, float *__restrict__ out_nbo,
                                                       float *__restrict__ out_cbo) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < total_v) {
    const int idx = (int)(t / cube_vbosize), i = (int)(t - (long long)idx * cube_vbosize), a = i % 3;
    const float4 c = centers[idx], k = colors[idx];
    out_vbo[t] = cube_vbo[i] * scale_factor + (a == 0 ? c.x : (a == 1 ? c.y : c.z));
    out_cbo[t] = a == 0 ? k.x : (a == 1 ? k.y : k.z);
    out_nbo[t] = cube_nbo[i];
  }
  if (t < total_i) {
    const int idx = (int)(t / cube_ibosize), i = (int)(t - (long long)idx * cube_ibosize);
    out_ibo[t] = cube_ibo[i] + idx * cube_ibosize;
  }
}

int voxel_grid_to_mesh(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int n, float scale_factor,
                       const float *cube_vbo, int cube_vbosize, const int *cube_ibo, int cube_ibosize, const float *cube_nbo,
                       float *d_vbo, int *d_ibo, float *d_nbo, float *d_cbo, hipStream_t stream) {
  if (!ws || n < 0 || cube_vbosize <= 0 || cube_ibosize <= 0 || !cube_vbo || !cube_ibo || !cube_nbo) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  if (!d_centers || !d_colors || !d_vbo || !d_ibo || !d_nbo || !d_cbo) return SVOSLAM_ERR_INVALID_ARG;
  if ((long long)n * cube_ibosize > 0x7FFFFFFFll) return SVOSLAM_ERR_INVALID_ARG;  // the reference's int offsets
  DeviceBuffer &dc = ws->misc;
  SVO_TRY(dc.reserve((size_t)(2 * cube_vbosize + cube_ibosize) * 4));
  float *dv = dc.as<float>(), *dn = dv + cube_vbosize;
  int *di = reinterpret_cast<int *>(dn + cube_vbosize);
  SVO_HIP(hipMemcpyAsync(dv, cube_vbo, (size_t)cube_vbosize * 4, hipMemcpyHostToDevice, stream));
  SVO_HIP(hipMemcpyAsync(dn, cube_nbo, (size_t)cube_vbosize * 4, hipMemcpyHostToDevice, stream));
  SVO_HIP(hipMemcpyAsync(di, cube_ibo, (size_t)cube_ibosize * 4, hipMemcpyHostToDevice, stream));
  const long long total_v = (long long)n * cube_vbosize, total_i = (long long)n * cube_ibosize;
  const long long total = total_v > total_i ? total_v : total_i;
  cube_mesh_kernel<<<(unsigned)cdiv(total, 256), 256, 0, stream>>>(reinterpret_cast<const float4 *>(d_centers),
                                                                  reinterpret_cast<const float4 *>(d_colors), scale_factor, total_v,
                                                                  total_i, dv, cube_vbosize, di, cube_ibosize, dn, d_vbo, d_ibo, d_nbo, d_cbo);
  SVO_LAUNCH_CHECK();
  SVO_HIP(hipStreamSynchronize(stream));  // the staged cube arrays are pageable host memory of the caller
  return SVOSLAM_OK;
}

// ----------------------------------------------------------------------------
// host loaders
// ----------------------------------------------------------------------------
static bool next_tok(const char *&p, std::string &out) {  // std::getline(ss, tok, ' ')
  out.clear();
  if (*p == 0) return false;
  while (*p && *p != ' ') out.push_back(*p++);
  if (*p == ' ') p++;
  return true;
}

static void face_normal(const std::vector<float> &pts, const int *f, int i0, int i1, int i2, int i3, float out[3]) {
  float a[3], b[3];
  for (int k = 0; k < 3; k++) { a[k] = pts[4 * f[i0] + k] - pts[4 * f[i1] + k]; b[k] = pts[4 * f[i2] + k] - pts[4 * f[i3] + k]; }
  const float c[3] = {a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]};
  const float inv = 1.0f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  out[0] = c[0] * inv; out[1] = c[1] * inv; out[2] = c[2] * inv;
}

// obj::isConvex, obj.cpp:137-164
static bool is_convex(const std::vector<float> &pts, const int *f, int n) {
  if (n <= 3) return true;
  const double EPS = 2.220446049250313e-16;
  const int k = n - 1;
  float nn[3], m[3];
  face_normal(pts, f, 0, k, 0, 1, nn);
  for (int i = 2; i < n; i++) {
    face_normal(pts, f, i - 1, i - 2, i - 1, i, m);
    if (fabs(m[0] - nn[0]) > EPS || fabs(m[1] - nn[1]) > EPS || fabs(m[2] - nn[2]) > EPS) return false;
  }
  face_normal(pts, f, k, k - 1, k, 0, m);
  return !(fabs(m[0] - nn[0]) > EPS || fabs(m[1] - nn[1]) > EPS || fabs(m[2] - nn[2]) > EPS);
}

// Scene::loadObjFile = objLoader + obj::buildVBOs (recenter, fan triangulation, non-indexed VBO) + objToMesh
int mesh_load_obj(const char *path, svoslam_mesh *out) {
  if (!path || !out) return SVOSLAM_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  FILE *fp = fopen(path, "r");
  if (!fp) return SVOSLAM_ERR_INVALID_ARG;
  std::vector<float> pts, tcs;
  std::vector<int> fidx, fstart, tidx, tstart;
  bool have_tex_faces = false, maxmin = false;
  float xmax = 0, xmin = 0, ymax = 0, ymin = 0, zmax = 0, zmin = 0;
  char *line = nullptr; size_t cap = 0; ssize_t len;
  std::string tok;
  while ((len = getline(&line, &cap, fp)) >= 0) {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    if (len == 0) continue;
    const char *p = line;
    if (line[0] == 'v' && line[1] == 't') {
      float c[3] = {0, 0, 0};
      next_tok(p, tok);
      for (int k = 0; k < 3; k++) { tok.clear(); next_tok(p, tok); c[k] = (float)atof(tok.c_str()); }
      tcs.insert(tcs.end(), {c[0], c[1], c[2], 1.0f});
    } else if (line[0] == 'v' && line[1] == 'n') {
      // normals are not consumed by the voxelizer
    } else if (line[0] == 'v') {
      float c[3] = {0, 0, 0};
      next_tok(p, tok);
      for (int k = 0; k < 3; k++) { tok.clear(); next_tok(p, tok); c[k] = (float)atof(tok.c_str()); }
      pts.insert(pts.end(), {c[0], c[1], c[2], 1.0f});
      if (maxmin) {  // obj::compareMaxMin, obj.cpp:112-135
        xmax = c[0] > xmax ? c[0] : xmax; xmin = c[0] < xmin ? c[0] : xmin;
        ymax = c[1] > ymax ? c[1] : ymax; ymin = c[1] < ymin ? c[1] : ymin;
        zmax = c[2] > zmax ? c[2] : zmax; zmin = c[2] < zmin ? c[2] : zmin;
      } else { xmax = xmin = c[0]; ymax = ymin = c[1]; zmax = zmin = c[2]; maxmin = true; }
    } else if (line[0] == 'f') {
      next_tok(p, tok);
      fstart.push_back((int)fidx.size());
      const bool has_slash = strchr(line, '/') != nullptr, has_dslash = strstr(line, "//") != nullptr;
      if (has_slash && !has_dslash) { tstart.push_back((int)tidx.size()); have_tex_faces = true; }
      while (next_tok(p, tok)) {
        const size_t s1 = tok.find('/');
        fidx.push_back((int)(atof(tok.substr(0, s1).c_str()) - 1));
        if (has_slash && !has_dslash && s1 != std::string::npos) {
          const size_t s2 = tok.find('/', s1 + 1);
          tidx.push_back((int)(atof(tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1).c_str()) - 1));
        }
      }
    }
  }
  free(line);
  fclose(fp);
  fstart.push_back((int)fidx.size());
  if (have_tex_faces) tstart.push_back((int)tidx.size());
  const int npts = (int)pts.size() / 4, nfaces = (int)fstart.size() - 1;
  if (npts > 0) {  // obj::recenter, obj.cpp:227-238: x/z centred, y min to 0
    const float center[3] = {(xmax + xmin) / 2, ymin, (zmax + zmin) / 2};
    xmax = xmin = pts[0] - center[0];
    ymax = ymin = pts[1] - center[1];
    zmax = zmin = pts[2] - center[2];
    for (int i = 0; i < npts; i++) {
      float *q = &pts[4 * i];
      q[0] = q[0] - center[0]; q[1] = q[1] - center[1]; q[2] = q[2] - center[2];
      xmax = q[0] > xmax ? q[0] : xmax; xmin = q[0] < xmin ? q[0] : xmin;
      ymax = q[1] > ymax ? q[1] : ymax; ymin = q[1] < ymin ? q[1] : ymin;
      zmax = q[2] > zmax ? q[2] : zmax; zmin = q[2] < zmin ? q[2] : zmin;
    }
  }
  // The reference's loader indexes points / texture coordinates with whatever the file says (objloader.cpp: negative,
  // zero or too large indices and faces of fewer than three corners read out of bounds).  Behind a public C ABI that
  // takes a caller-supplied path such a file is refused instead: every index must name an existing element, a face
  // needs three corners, and a file that textures some faces must texture all of them (tstart is per face).
  const int ntcs = (int)tcs.size() / 4;
  for (int k = 0; k < nfaces; k++)
    if (fstart[k + 1] - fstart[k] < 3) return SVOSLAM_ERR_FORMAT;
  for (int v : fidx)
    if (v < 0 || v >= npts) return SVOSLAM_ERR_FORMAT;
  if (have_tex_faces) {
    if ((int)tstart.size() != nfaces + 1) return SVOSLAM_ERR_FORMAT;
    for (int k = 0; k < nfaces; k++)
      if (tstart[k + 1] - tstart[k] < 3) return SVOSLAM_ERR_FORMAT;
    for (int v : tidx)
      if (v < 0 || v >= ntcs) return SVOSLAM_ERR_FORMAT;
  }
  std::vector<float> V, T;
  const bool has_texture = have_tex_faces && tstart.size() > 1;
  for (int k = 0; k < nfaces; k++) {  // obj::buildVBOs, obj.cpp:33-110
    const int *f = &fidx[fstart[k]];
    const int n = fstart[k + 1] - fstart[k];
    if (!is_convex(pts, f, n)) continue;
    for (int i = 2; i < n; i++) {
      const int tri[3] = {f[0], f[i - 1], f[i]};
      for (int c = 0; c < 3; c++)
        for (int d = 0; d < 3; d++) V.push_back(pts[4 * tri[c] + d]);
      if (has_texture) {  // facetexture[0], [1], [2] for every fan triangle (obj.cpp:75-80)
        const int *ft = &tidx[tstart[k]];
        for (int c = 0; c < 3; c++) { T.push_back(tcs[4 * ft[c]]); T.push_back(tcs[4 * ft[c] + 1]); }
      }
    }
  }
  out->n_tris = (int32_t)(V.size() / 9);
  out->vbo = (float *)malloc(sizeof(float) * (V.size() ? V.size() : 1));
  memcpy(out->vbo, V.data(), sizeof(float) * V.size());
  out->tbosize = (int32_t)T.size();
  out->tbo = nullptr;
  if (!T.empty()) { out->tbo = (float *)malloc(sizeof(float) * T.size()); memcpy(out->tbo, T.data(), sizeof(float) * T.size()); }
  out->bbox0[0] = xmin; out->bbox0[1] = ymin; out->bbox0[2] = zmin;  // scene.cpp:129-130
  out->bbox1[0] = xmax; out->bbox1[1] = ymax; out->bbox1[2] = zmax;
  return SVOSLAM_OK;
}

int mesh_free(svoslam_mesh *m) {
  if (!m) return SVOSLAM_OK;
  free(m->vbo); free(m->tbo);
  memset(m, 0, sizeof(*m));
  return SVOSLAM_OK;
}

// Scene::loadBMP, scene.cpp:35-62 (54-byte header, 24-bit BGR -> RGB/255, rows as stored)
int texture_load_bmp(const char *path, svoslam_texture *out) {
  if (!path || !out) return SVOSLAM_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  FILE *f = fopen(path, "rb");
  if (!f) return SVOSLAM_ERR_INVALID_ARG;
  unsigned char info[54];
  if (fread(info, 1, 54, f) != 54) { fclose(f); return SVOSLAM_ERR_INVALID_ARG; }
  int w, h;
  memcpy(&w, info + 18, 4);
  memcpy(&h, info + 22, 4);
  if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) { fclose(f); return SVOSLAM_ERR_INVALID_ARG; }
  const size_t size = (size_t)3 * w * h;
  std::vector<unsigned char> raw(size, 0);
  const size_t got = fread(raw.data(), 1, size, f);
  fclose(f);
  if (got != size) return SVOSLAM_ERR_FORMAT;  // truncated file (the reference would use a partly uninitialised texture)
  out->data = (float *)malloc(sizeof(float) * size);
  for (size_t i = 0; i < size; i += 3) {
    out->data[i] = (int)raw[i + 2] / 255.0f;
    out->data[i + 1] = (int)raw[i + 1] / 255.0f;
    out->data[i + 2] = (int)raw[i] / 255.0f;
  }
  out->width = w; out->height = h;
  return SVOSLAM_OK;
}

int texture_free(svoslam_texture *t) {
  if (!t) return SVOSLAM_OK;
  free(t->data);
  memset(t, 0, sizeof(*t));
  return SVOSLAM_OK;
}

}  // namespace svoslam
