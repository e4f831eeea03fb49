// icp.hip -- point-to-plane ICP normal equations + the RGBDCamera tracker loop,
// resident on the device.
//
// Contract = sensor::computeICPCost2 (src/sensor/localization_kernels.cu:154-229,
// 303-326) and sensor::RGBDCamera::update (src/sensor/rgbd_camera.cpp:53-222).
//
// Organisation (vs the reference):
//  * accumulation: the reference gives each of 16-thread blocks' threads 20..60
//    consecutive pixels, writes a 168-byte Mat6x7 partial per thread to HBM and
//    thrust::reduces them in an unspecified float order.  Here every lane keeps
//    the 27 distinct terms (A is symmetric) in registers as EXACT fixed-point
//    values carried in float64 (q = rint(fl32(product) * 2^20 | 2^30)), reduces
//    them across the wave with shuffles, across the workgroup through LDS and
//    adds them to 27 global doubles.  Integer-valued sums are associative, so the
//    result does not depend on lane/workgroup/GPU partitioning -- row-band
//    partials from several GPUs all-reduce (RCCL, float64 sum) to the same bits.
//  * the per-iteration rigid update is applied on the fly: the kernel replays
//    the short chain of 4x4 transforms on each current-frame vertex/normal
//    instead of rewriting both maps in HBM after every iteration
//    (rgbd_camera.cpp:116-120,163-167); same operation order, same floats.
//  * the 6x6 Cholesky (rgbd_camera.cpp:194-222) and the pose composition
//    (:154-160,172-173) run in a one-lane kernel: 19 iterations per frame with
//    no host round trip (the reference copies 168 B back and solves on the host
//    19 times per frame).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "graph_cache.hpp"
#include "config.hpp"
#include "icp.hpp"
#include "icp_device.hpp"
#include "image_kernels.hpp"
#include "track_persistent.hpp"
#include "stage_timing.hpp"
#include "wave_rank.hpp"

namespace svoslam {

// (tracker state, wave reductions, Cholesky and pose composition: icp_device.hpp)


// ----------------------------------------------------------------------------
// accumulate
// ----------------------------------------------------------------------------
constexpr int kIcpThreads = 512;   // 8 wavefronts per workgroup
constexpr int kIcpWaves = kIcpThreads / kWave;
// workgroups of an accumulate launch (cap).  cfg4 (1080p, launch chain with work maps), frames/s on one box, means of 3:
// 256 -> 768, 384 -> 790, 512 -> 771 (before the work maps: 256 -> 721, 512 -> 735, 1024 -> 696, 2048 -> 637)
constexpr int kMaxIcpBlocks = 384;


// Body of the accumulate kernel.  The 27 sums of a lane are reduced across the wavefront in registers
// (DPP) and across the 8 wavefronts through a 1.7 KB LDS array.  (Earlier forms: 27 x 6 ds_bpermute
// shuffles per lane were LDS-issue bound; a [27][512] LDS transpose was fast but its 108 KB kept the
// kernel off every CU that still held raycast workgroups with their 48 KB tables.)
template <bool WORK, bool COHERENT = false>
__device__ inline void accumulate_block(const float *__restrict__ last_v, const float *__restrict__ last_n,
                                        const float *cur_v, const float *cur_n, int first, int end,
                                        const CamState *state, int flags, int chain_len, double *__restrict__ partial,
                                        double (*wsum)[27], float *chain_s, int chain_first_arg = 0, float *work_v = nullptr,
                                        float *work_n = nullptr) {
  const int chain_first = WORK ? chain_first_arg : 0;
  // chain_first > 0 (work maps): cur_v / cur_n already carry the level-start transform and chain[0 .. chain_first), as
  // the reference's maps do after transformVertexMap / transformNormalMap (rgbd_camera.cpp:163-167); only the rest is
  // applied here.  work_v / work_n != nullptr: the transformed values are stored for the next iteration.  Same
  // matrices in the same order on the same floats as the replay from the raw maps, so the same bits.
  int nchain = 0;
  bool lost = false;
  bool corrected = false;  // (stateless calls -- computeICPCost2 of the ABI -- are the reference's)
  if (state) {
    corrected = state->corrected != 0;
    // iteration 0 of a level: the level-start transform is update_trans as the previous level left it
    // and the "tracking lost" flag of the previous level no longer applies
    lost = !(flags & kFlagFirstIter) && state->lost != 0;
    if ((flags & kFlagLevelStart) && chain_first == 0) {
      const float *src = (flags & kFlagFirstIter) ? state->update_trans : state->level_start;
      if (threadIdx.x < 16) chain_s[threadIdx.x] = src[threadIdx.x];
      nchain = 1;
    }
    const int skip = chain_first > 0 ? chain_first - 1 : 0;  // chain_first = 1 + number of chain entries already applied
    for (int i = threadIdx.x; i < (chain_len - skip) * 16; i += kIcpThreads) chain_s[nchain * 16 + i] = (&state->chain[0][0])[skip * 16 + i];
    nchain += chain_len - skip;
  }
  __syncthreads();
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0.0;
  if (!lost) {
    for (int p = first + blockIdx.x * kIcpThreads + threadIdx.x; p < end; p += gridDim.x * kIcpThreads) {
      float v2x = cur_v[3 * (size_t)p], v2y = cur_v[3 * (size_t)p + 1], v2z = cur_v[3 * (size_t)p + 2];
      float n2x = cur_n[3 * (size_t)p], n2y = cur_n[3 * (size_t)p + 1], n2z = cur_n[3 * (size_t)p + 2];
      const float v1x = last_v[3 * (size_t)p], v1y = last_v[3 * (size_t)p + 1], v1z = last_v[3 * (size_t)p + 2];
      const float n1x = last_n[3 * (size_t)p], n1y = last_n[3 * (size_t)p + 1], n1z = last_n[3 * (size_t)p + 2];
      for (int k = 0; k < nchain; k++) {  // transformVertexMap / transformNormalMap replayed
        float ox, oy, oz;
        mat4_mul_point(chain_s + 16 * k, v2x, v2y, v2z, 1.0f, ox, oy, oz);
        v2x = ox; v2y = oy; v2z = oz;
        mat4_mul_point(chain_s + 16 * k, n2x, n2y, n2z, 0.0f, ox, oy, oz);
        n2x = ox; n2y = oy; n2z = oz;
      }
      if (WORK && work_v) {
        work_v[3 * (size_t)p] = v2x; work_v[3 * (size_t)p + 1] = v2y; work_v[3 * (size_t)p + 2] = v2z;
        work_n[3 * (size_t)p] = n2x; work_n[3 * (size_t)p + 1] = n2y; work_n[3 * (size_t)p + 2] = n2z;
      }
      icp_pixel_terms(v1x, v1y, v1z, n1x, n1y, n1z, v2x, v2y, v2z, n2x, n2y, n2z, acc, corrected);
    }
  }
  // workgroup -> one 27-double row; every partial is an integer-valued double (exact, order-free).
  // Plain stores only: the rows are summed in the NEXT launch, so visibility rests on the kernel
  // boundary alone (no cross-XCD atomics on data).
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
  for (int i = 0; i < 27; i++) {
    const double t = wave_sum_to_lane63(acc[i]);
    if (lane == 63u) wsum[wave][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < kIcpWaves; w++) v += wsum[w][threadIdx.x];
    // COHERENT: the rows are summed by the LAST workgroup of this very launch (icp_accumulate_work_solve_kernel): written
    // through to where another XCD's CU will find them
    if (COHERENT) __hip_atomic_store(&partial[(size_t)blockIdx.x * 27 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partial[(size_t)blockIdx.x * 27 + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(kIcpThreads) void icp_accumulate_kernel(
    const float *__restrict__ last_v, const float *__restrict__ last_n, const float *__restrict__ cur_v,
    const float *__restrict__ cur_n, int first, int end, const CamState *__restrict__ state, int flags, int chain_len,
    double *__restrict__ partial) {
  SVO_HIGH_PRIO();
  __shared__ double wsum[kIcpWaves][27];
  __shared__ float chain_s[(kMaxChain + 1) * 16];
  accumulate_block<false>(last_v, last_n, cur_v, cur_n, first, end, state, flags, chain_len, partial, wsum, chain_s);
}

// the same over work maps (launch-chain tracker of one whole camera: camera_track): iteration `it` reads what iteration
// it - 1 stored and applies one matrix instead of replaying it + 1 (at 1920x1080 the replay is most of the kernel)
__global__ __launch_bounds__(kIcpThreads) void icp_accumulate_work_kernel(
    const float *__restrict__ last_v, const float *__restrict__ last_n, const float *cur_v, const float *cur_n, int first, int end,
    const CamState *__restrict__ state, int flags, int chain_len, int chain_first, float *work_v, float *work_n,
    double *__restrict__ partial) {
  SVO_HIGH_PRIO();
  __shared__ double wsum[kIcpWaves][27];
  __shared__ float chain_s[(kMaxChain + 1) * 16];
  accumulate_block<true>(last_v, last_n, cur_v, cur_n, first, end, state, flags, chain_len, partial, wsum, chain_s, chain_first, work_v, work_n);
}

// photometric terms of the same pixels (own specification, icp_device.hpp rgbd_pixel_terms): the current vertex goes
// through the same replayed transform chain as in accumulate_block, so both systems linearise at the same estimate
__global__ __launch_bounds__(kIcpThreads) void rgbd_accumulate_kernel(
    const float *__restrict__ last_i, const float *__restrict__ last_g, const float *__restrict__ last_v,
    const float *__restrict__ cur_i, const float *__restrict__ cur_v, int first, int end, float fx, float fy, float sx, float sy,
    const CamState *__restrict__ state, int flags, int chain_len, double *__restrict__ partial) {
  SVO_HIGH_PRIO();
  __shared__ double wsum[kIcpWaves][27];
  __shared__ float chain_s[(kMaxChain + 1) * 16];
  int nchain = 0;
  bool lost = false;
  if (state) {
    lost = !(flags & kFlagFirstIter) && state->lost != 0;
    if (flags & kFlagLevelStart) {
      const float *src = (flags & kFlagFirstIter) ? state->update_trans : state->level_start;
      if (threadIdx.x < 16) chain_s[threadIdx.x] = src[threadIdx.x];
      nchain = 1;
    }
    for (int i = threadIdx.x; i < chain_len * 16; i += kIcpThreads) chain_s[nchain * 16 + i] = (&state->chain[0][0])[i];
    nchain += chain_len;
  }
  __syncthreads();
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0.0;
  if (!lost) {
    for (int p = first + blockIdx.x * kIcpThreads + threadIdx.x; p < end; p += gridDim.x * kIcpThreads) {
      float v2x = cur_v[3 * (size_t)p], v2y = cur_v[3 * (size_t)p + 1], v2z = cur_v[3 * (size_t)p + 2];
      for (int k = 0; k < nchain; k++) {
        float ox, oy, oz;
        mat4_mul_point(chain_s + 16 * k, v2x, v2y, v2z, 1.0f, ox, oy, oz);
        v2x = ox; v2y = oy; v2z = oz;
      }
      rgbd_pixel_terms(last_v[3 * (size_t)p], last_v[3 * (size_t)p + 1], last_v[3 * (size_t)p + 2], v2x, v2y, v2z,
                       last_g[2 * (size_t)p], last_g[2 * (size_t)p + 1], last_i[p] - cur_i[p], fx, fy, sx, sy, acc);
    }
  }
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
  for (int i = 0; i < 27; i++) {
    const double t = wave_sum_to_lane63(acc[i]);
    if (lane == 63u) wsum[wave][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < kIcpWaves; w++) v += wsum[w][threadIdx.x];
    partial[(size_t)blockIdx.x * 27 + threadIdx.x] = v;
  }
}

// column sums of partial[rows][27] into LDS totals[27] (exact integer-valued sums);
// blockDim.x / 32 row groups x 32 columns, all loads of a thread independent (<= 8 rows each)
constexpr int kReduceThreads = 1024;  // (256, a workgroup that fits wherever one march workgroup has left: cfg4 769 -> 745 frames/s)
__device__ inline void reduce_rows(const double *__restrict__ partial, int rows, double (*red)[27], double *totals) {
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5, ngrp = blockDim.x >> 5;
  double s = 0.0;
  if (col < 27)
    for (int r = grp; r < rows; r += ngrp) s += partial[(size_t)r * 27 + col];
  if (col < 27) red[grp][col] = s;
  __syncthreads();
  if (threadIdx.x < 27) {
    double t = 0.0;
    for (int g = 0; g < ngrp; g++) t += red[g][threadIdx.x];
    totals[threadIdx.x] = t;
  }
  __syncthreads();
}

// acc[27] += column sums (used by the stateless ABI call and the multi-GPU path)
__global__ __launch_bounds__(kReduceThreads) void icp_reduce_kernel(const double *__restrict__ partial, int rows,
                                                                    double *__restrict__ acc) {
  SVO_HIGH_PRIO();
  __shared__ double red[kReduceThreads / 32][27];
  __shared__ double totals[27];
  reduce_rows(partial, rows, red, totals);
  if (threadIdx.x < 27) acc[threadIdx.x] += totals[threadIdx.x];
}

static int accumulate_range(int w, int h, int &first, int num, int &end) {
  const int n = w * h;
  // Q15: load_size = 20*w/640; the reference reduces floor(n/load) partials, the tail is dropped
  const int load_size = 20 * w / 640;
  int limit = n;
  if (load_size > 0) limit = (n / load_size) * load_size;
  end = first + num;
  if (end > limit) end = limit;
  if (first < 0) first = 0;
  int blocks = (int)cdiv(end > first ? end - first : 1, kIcpThreads);
  if (blocks > kMaxIcpBlocks) blocks = kMaxIcpBlocks;
  if (blocks < 1) blocks = 1;
  return blocks;
}

// accumulate pixels [first, first+num) and ADD their 27 sums into d_acc (in-stream, no atomics)
static int launch_accumulate(const float *lv, const float *ln, const float *cv, const float *cn, int w, int h, int first,
                             int num, const CamState *state, int flags, int chain_len, double *d_partial,
                             double *d_acc, hipStream_t s) {
  int end;
  const int blocks = accumulate_range(w, h, first, num, end);
  icp_accumulate_kernel<<<blocks, kIcpThreads, 0, s>>>(lv, ln, cv, cn, first, end, state, flags, chain_len, d_partial);
  icp_reduce_kernel<<<1, kReduceThreads, 0, s>>>(d_partial, blocks, d_acc);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int icp_accumulate(svoslam::DeviceBuffer &scratch, const float *lv, const float *ln, const float *cv, const float *cn,
                   int w, int h, int first, int num, double *d_acc, hipStream_t s) {
  if (!lv || !ln || !cv || !cn || !d_acc || w <= 0 || h <= 0 || first < 0 || num < 0) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(scratch.reserve((size_t)(kMaxIcpBlocks + 1) * 27 * sizeof(double)));
  return launch_accumulate(lv, ln, cv, cn, w, h, first, num, nullptr, 0, 0, scratch.as<double>() + 27, d_acc, s);
}

static void icp_finish_host(const double acc[27], float A[36], float b[6]) {
  int k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      const float v = (float)(acc[k++] * (1.0 / kScaleA));
      A[6 * i + j] = v;
      A[6 * j + i] = v;
    }
  for (int i = 0; i < 6; i++) b[i] = (float)(acc[21 + i] * (1.0 / kScaleB));
}

int icp_cost2(svoslam::DeviceBuffer &scratch, const float *lv, const float *ln, const float *cv, const float *cn, int w,
              int h, float A[36], float b[6], hipStream_t s) {
  if (!A || !b) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(scratch.reserve((size_t)(kMaxIcpBlocks + 1) * 27 * sizeof(double)));
  double *d_acc = scratch.as<double>();  // [0,27) totals, then the per-workgroup rows
  SVO_HIP(hipMemsetAsync(d_acc, 0, 27 * sizeof(double), s));
  SVO_TRY(icp_accumulate(scratch, lv, ln, cv, cn, w, h, 0, w * h, d_acc, s));
  double acc[27];
  SVO_HIP(hipMemcpyAsync(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  icp_finish_host(acc, A, b);
  return SVOSLAM_OK;
}

// computeRGBDCost (declared localization_kernels.h:42; empty in the reference, :328-331): own specification, see
// icp_device.hpp.  last_g = gradient() of last_i.  Blocking (A, b to the host).
int rgbd_cost(svoslam::DeviceBuffer &scratch, const float *last_i, const float *last_g, const float *last_v, const float *cur_i,
              const float *cur_v, int w, int h, float fx, float fy, int img_w, int img_h, float A[36], float b[6], hipStream_t s) {
  if (!last_i || !last_g || !last_v || !cur_i || !cur_v || !A || !b || w <= 0 || h <= 0 || img_w < w || img_h < h)
    return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(scratch.reserve((size_t)(kMaxIcpBlocks + 1) * 27 * sizeof(double)));
  double *d_acc = scratch.as<double>();
  SVO_HIP(hipMemsetAsync(d_acc, 0, 27 * sizeof(double), s));
  const int n = w * h;
  int blocks = (int)cdiv(n, kIcpThreads);
  if (blocks > kMaxIcpBlocks) blocks = kMaxIcpBlocks;
  rgbd_accumulate_kernel<<<blocks, kIcpThreads, 0, s>>>(last_i, last_g, last_v, cur_i, cur_v, 0, n, fx, fy, (float)(img_w / w),
                                                        (float)(img_h / h), nullptr, 0, 0, d_acc + 27);
  icp_reduce_kernel<<<1, kReduceThreads, 0, s>>>(d_acc + 27, blocks, d_acc);
  SVO_LAUNCH_CHECK();
  double acc[27];
  SVO_HIP(hipMemcpyAsync(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  int k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      const float v = (float)(acc[k++] * (1.0 / kScaleRgbdA));
      A[6 * i + j] = v;
      A[6 * j + i] = v;
    }
  for (int i = 0; i < 6; i++) b[i] = (float)(acc[21 + i] * (1.0 / kScaleRgbdB));
  return SVOSLAM_OK;
}

// ----------------------------------------------------------------------------
// computeICPCost (localization_kernels.cu:59-152,231-301): the variant with a correspondence stencil.
// The reference compacts the matching pixels (three thrust::copy_if), gives each thread 10 consecutive
// matches and reduces floor(M/10) partials, so exactly the first floor(M/10)*10 matches (in pixel order)
// contribute.  Here: count per 256-pixel tile -> one scan -> every match knows its rank in the compacted
// order without compacting anything, and adds its 27 exact terms if the rank is below that limit.
// ----------------------------------------------------------------------------
__device__ inline bool icp_corr_match(const float *__restrict__ lv, const float *__restrict__ ln, const float *__restrict__ cv,
                                      const float *__restrict__ cn, size_t p) {
  const float v2x = cv[3 * p], v2y = cv[3 * p + 1], v2z = cv[3 * p + 2];
  const float v1x = lv[3 * p], v1y = lv[3 * p + 1], v1z = lv[3 * p + 2];
  if (!finitef_(v2x) || !finitef_(v2y) || !finitef_(v2z) || !finitef_(v1x) || !finitef_(v1y) || !finitef_(v1z)) return false;
  const float n2x = cn[3 * p], n2y = cn[3 * p + 1], n2z = cn[3 * p + 2];
  const float n1x = ln[3 * p], n1y = ln[3 * p + 1], n1z = ln[3 * p + 2];
  if (!finitef_(n2x) || !finitef_(n2y) || !finitef_(n2z) || !finitef_(n1x) || !finitef_(n1y) || !finitef_(n1z)) return false;
  const float dx = v2x - v1x, dy = v2y - v1y, dz = v2z - v1z;
  if (sqrtf(dot3(dx, dy, dz, dx, dy, dz)) > kDistThresh) return false;  // :84
  if (dot3(n2x, n2y, n2z, n1x, n1y, n1z) < kNormThresh) return false;   // :89
  return true;
}

__global__ __launch_bounds__(256) void icp_corr_count_kernel(const float *__restrict__ lv, const float *__restrict__ ln,
                                                             const float *__restrict__ cv, const float *__restrict__ cn, int n,
                                                             unsigned *__restrict__ tile_count) {
  __shared__ unsigned tmp[4];
  const int p = blockIdx.x * 256 + threadIdx.x;
  const unsigned m = (p < n && icp_corr_match(lv, ln, cv, cn, (size_t)p)) ? 1u : 0u;
  unsigned total;
  (void)block256_exclusive_scan(m, tmp, total);
  if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

// exclusive scan of the tile counts in place; counts[num_tiles] = total number of correspondences
__global__ __launch_bounds__(256) void icp_corr_scan_kernel(unsigned *__restrict__ counts, int num_tiles) {
  __shared__ unsigned tmp[4];
  unsigned carry = 0;
  for (int base = 0; base < num_tiles; base += 256) {
    const int t = base + (int)threadIdx.x;
    const unsigned v = t < num_tiles ? counts[t] : 0u;
    unsigned total;
    const unsigned ex = block256_exclusive_scan(v, tmp, total);
    if (t < num_tiles) counts[t] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) counts[num_tiles] = carry;
}

__global__ __launch_bounds__(256) void icp_corr_cost_kernel(const float *__restrict__ lv, const float *__restrict__ ln,
                                                            const float *__restrict__ cv, const float *__restrict__ cn, int n,
                                                            const unsigned *__restrict__ tile_offset, int num_tiles,
                                                            double *__restrict__ partial) {
  __shared__ unsigned tmp[4];
  __shared__ double wsum[4][27];
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool match = p < n && icp_corr_match(lv, ln, cv, cn, (size_t)p);
  unsigned total;
  const unsigned rank = tile_offset[blockIdx.x] + block256_exclusive_scan(match ? 1u : 0u, tmp, total);
  const unsigned limit = (tile_offset[num_tiles] / 10u) * 10u;  // :289-293
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0.0;
  if (match && rank < limit) {
    const size_t q = (size_t)p;
    const float v2x = cv[3 * q], v2y = cv[3 * q + 1], v2z = cv[3 * q + 2];
    const float v1x = lv[3 * q], v1y = lv[3 * q + 1], v1z = lv[3 * q + 2];
    const float n1x = ln[3 * q], n1y = ln[3 * q + 1], n1z = ln[3 * q + 2];
    float J[6];  // A_T = G_T * n with n the LAST frame's normal (:123-131)
    J[0] = (0.0f * n1x + (-v2x) * n1y) + (-v2y) * n1z;
    J[1] = ((-v2z) * n1x + 0.0f * n1y) + v2x * n1z;
    J[2] = (v2y * n1x + v2z * n1y) + 0.0f * n1z;
    J[3] = (1.0f * n1x + 0.0f * n1y) + 0.0f * n1z;
    J[4] = (0.0f * n1x + 1.0f * n1y) + 0.0f * n1z;
    J[5] = (0.0f * n1x + 0.0f * n1y) + 1.0f * n1z;
    const float bb = dot3(n1x, n1y, n1z, v1x - v2x, v1y - v2y, v1z - v2z);
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = i; j < 6; j++) acc[k++] = (double)rintf((J[i] * J[j]) * 1048576.0f);
#pragma unroll
    for (int i = 0; i < 6; i++) acc[21 + i] = (double)rintf((bb * J[i]) * 1073741824.0f);
  }
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
  for (int i = 0; i < 27; i++) {
    const double t = wave_sum_to_lane63(acc[i]);
    if (lane == 63u) wsum[wave][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < 27) partial[(size_t)blockIdx.x * 27 + threadIdx.x] = ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
}

int icp_cost(svoslam::DeviceBuffer &scratch, const float *lv, const float *ln, const float *cv, const float *cn, int w, int h,
             float A[36], float b[6], int *num_corr, hipStream_t s) {
  if (!lv || !ln || !cv || !cn || !A || !b || w <= 0 || h <= 0) return SVOSLAM_ERR_INVALID_ARG;
  const int n = w * h, tiles = (int)cdiv(n, 256);
  // layout: [0,27) totals | rows tiles x 27 doubles | tile counts (tiles + 1 words)
  const size_t rows_off = 27, counts_off_bytes = (size_t)(27 + (size_t)tiles * 27) * sizeof(double);
  SVO_TRY(scratch.reserve(counts_off_bytes + (size_t)(tiles + 1) * 4 + 64));
  double *d_acc = scratch.as<double>();
  double *d_rows = d_acc + rows_off;
  unsigned *d_counts = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(scratch.ptr) + counts_off_bytes);
  SVO_HIP(hipMemsetAsync(d_acc, 0, 27 * sizeof(double), s));
  icp_corr_count_kernel<<<tiles, 256, 0, s>>>(lv, ln, cv, cn, n, d_counts);
  icp_corr_scan_kernel<<<1, 256, 0, s>>>(d_counts, tiles);
  icp_corr_cost_kernel<<<tiles, 256, 0, s>>>(lv, ln, cv, cn, n, d_counts, tiles, d_rows);
  icp_reduce_kernel<<<1, kReduceThreads, 0, s>>>(d_rows, tiles, d_acc);
  SVO_LAUNCH_CHECK();
  double acc[27];
  unsigned m = 0;
  SVO_HIP(hipMemcpyAsync(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipMemcpyAsync(&m, d_counts + tiles, 4, hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  if (num_corr) *num_corr = (int)m;
  if (m > 0) icp_finish_host(acc, A, b);  // :251-253: without correspondences A and b are left as they are
  return SVOSLAM_OK;
}


// single-GPU iteration tail: sum the workgroup rows, solve, compose -- ONE launch
__global__ __launch_bounds__(kReduceThreads) void cam_reduce_solve_kernel(CamState *st, const double *__restrict__ partial,
                                                                          int rows, int slot, int flags,
                                                                          const double *__restrict__ partial2, int rows2) {
  SVO_HIGH_PRIO();
  __shared__ double red[kReduceThreads / 32][27];
  __shared__ double totals[27], totals2[27];
  __shared__ float tail_sm[kTailScratch];
  const TailPrefetch pre = tail_prefetch(st, flags);  // written by the previous launch
  reduce_rows(partial, rows, red, totals);
  if (partial2) reduce_rows(partial2, rows2, red, totals2);  // photometric system (W_RGBD x it is added in the solve)
  if (threadIdx.x >= 64) return;  // the tail runs on the first wavefront
  iteration_tail_wave(st, totals, slot, flags, tail_sm, pre, partial2 ? totals2 : nullptr);
}

// multi-GPU iteration tail: acc[] holds the all-reduced sums
__global__ void cam_solve_kernel(CamState *st, double *acc, int slot, int flags) {
  SVO_HIGH_PRIO();
  // Read-and-reset, one element per LANE.  (A single thread doing "sums[i] = acc[i]; acc[i] = 0" gets
  // uniform-address SCALAR loads followed by vector stores of a constant: nothing orders the two memory
  // paths, and the zero was observed to overtake the load -- six of the 27 sums read back as 0 once in a
  // few hundred frames.  A lane's vector load and store of one address stay in order.)
  __shared__ double sums[27];
  __shared__ float tail_sm[kTailScratch];
  if (blockIdx.x) return;
  const TailPrefetch pre = tail_prefetch(st, flags);
  if (threadIdx.x < 27) {
    sums[threadIdx.x] = acc[threadIdx.x];
    acc[threadIdx.x] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  iteration_tail_wave(st, sums, slot, flags, tail_sm, pre);
}

__global__ void cam_frame_end_kernel(CamState *st, int apply_update) {
  SVO_HIGH_PRIO();
  if (threadIdx.x || blockIdx.x) return;
  frame_end_step(st, apply_update, st->update_trans);
}

// ---- frame-parallel tracking (multi-GPU) ---------------------------------------
// RGBDCamera::update starts update_trans at the identity for every frame (rgbd_camera.cpp:100) and iterates on the
// vertex / normal maps of frames k-1 and k only: the 19 ICP iterations of a frame are a pure function of TWO DEPTH
// IMAGES, independent of every earlier pose.  Only :172-173 (position, orientation *= update_trans) chain the frames.
// So frames can be tracked in any order, on any GPU: delta_export_kernel hands out a frame's update_trans (+ the number of
// pyramid levels it abandoned), cam_apply_delta_kernel is :172-173 + main.cpp:40 for a matrix from anywhere.
constexpr int kDeltaFloats = 20;  // update_trans[16], levels lost (int bits), give-up code of the one-launch tracker (int bits; 0 = none), 2 pad: 80 bytes per frame
__global__ void delta_export_kernel(CamState *st, const TrackSync *__restrict__ sy, float *__restrict__ out) {
  const int e = (int)threadIdx.x;
  if (blockIdx.x || e >= kDeltaFloats) return;
  float v = 0.0f;
  if (e < 16) v = st->update_trans[e];
  else if (e == 16) v = __int_as_float(st->tracking_lost_count);
  else if (e == 17) v = __int_as_float((int)sy->fail);  // a timed-out hand-off travels with the record: the pose camera reports it
  out[e] = v;
  if (e == 16) st->tracking_lost_count = 0;  // a delta camera counts per frame
}

__global__ void cam_apply_delta_kernel(CamState *st, TrackSync *__restrict__ sy, const float *__restrict__ delta) {
  SVO_HIGH_PRIO();
  if (threadIdx.x || blockIdx.x) return;
  if (delta) {
    const int fail = __float_as_int(delta[17]);
    if (fail != 0) sy->fail = (unsigned)fail;  // surfaces from this camera's next readback (check_tracker_health)
    float m[16];
    for (int i = 0; i < 16; i++) { m[i] = delta[i]; st->update_trans[i] = m[i]; }
    st->tracking_lost_count += __float_as_int(delta[16]);
    frame_end_step(st, 1, m);
  } else {
    frame_end_step(st, 0, st->update_trans);  // a camera's first frame: no ICP (rgbd_camera.cpp:99 `pass >= 1`)
  }
}

}  // namespace svoslam

// ----------------------------------------------------------------------------
// host mirror of sensor::RGBDCamera
// ----------------------------------------------------------------------------
using svoslam::CamState;

struct svoslam_camera {
  int width = 0, height = 0;
  float fx = 0, fy = 0;
  bool have_stamp = false;
  long long latest_stamp = 0;
  int band_first = 0, band_rows = 0;
  uint16_t *filt[3] = {nullptr, nullptr, nullptr};
  // pyramid maps of frame f live in set f % 3: the tracker reads sets f and f-1 while the maps of frame f+1
  // can already be generated on another stream (rgbd_camera.cpp:181-189 swaps two sets)
  float *vert[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  float *norm[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  unsigned prepared = 0;  // frames whose maps have been generated
  unsigned tracked = 0;   // frames whose pose has been estimated (rgbd_camera.h:75 `pass` saturates this at 2)
  CamState *d_state = nullptr;
  double *d_acc = nullptr;  // defaults to d_state->acc; may be redirected for multi-GPU all-reduce
  double *d_partial = nullptr;  // per-workgroup rows of the accumulate kernel
  bool frame_has_icp = false;
  int ring_slot = 0;       // fusion_ring slot of the frame being / last tracked
  svoslam::GraphCache g_prep, g_track;  // recorded launch sequences (graph_cache.hpp)
  // photometric RGB-D term (off by default = the reference, which ships it commented out): intensity and Sobel
  // gradient pyramids per map set, rows of the photometric accumulate kernel
  bool rgbd = false;
  float *inten[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  float *grad[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  float *tmp_inten = nullptr, *tmp_inten2 = nullptr;
  double *d_partial2 = nullptr;
  float *work_v = nullptr, *work_n = nullptr;  // launch-chain tracker: the current level's maps as transformed so far
  // one-launch tracker (track_persistent.hip)
  svoslam::TrackSync *d_sync = nullptr;
  unsigned *d_tickets = nullptr;
  hipStream_t cap_stream = nullptr;  // stream whose resident-workgroup capacity is cached below
  int capacity = 0;
  bool delta_fed = false;  // poses come from camera_apply_delta: there are no maps of the previous frame to track against
  // frame-to-model tracking (off by default): a map set of its own, filled by camera_set_model_depth, that the
  // ICP associates the incoming frame with instead of the previous frame's maps (the TODO of rgbd_camera.cpp:185)
  bool to_model = false, have_model = false;
  bool corrected = false;  // svoslam_camera_set_strict_reference(cam, 0): the corrected tracker (icp_device.hpp icp_rot_rows); a setting, survives reset
  uint16_t *model_filt[3] = {nullptr, nullptr, nullptr};
  float *model_v[3] = {nullptr, nullptr, nullptr}, *model_n[3] = {nullptr, nullptr, nullptr};
};

namespace svoslam {

static const int kPyramidIters[3] = {10, 5, 4};  // rgbd_camera.cpp:19

int camera_icp_iters(int level) { return (level >= 0 && level < 3) ? kPyramidIters[level] : 0; }

int camera_reset(svoslam_camera *c);
int camera_destroy(svoslam_camera *c);

int camera_create(svoslam_camera **out, int w, int h, float fx, float fy) {
  if (!out || w < 8 || h < 8) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(ensure_device());
  svoslam_camera *c = new svoslam_camera();
  c->width = w; c->height = h; c->fx = fx; c->fy = fy;
  c->band_first = 0; c->band_rows = h;
  for (int i = 0; i < 3; i++) {
    const size_t n = (size_t)(w >> i) * (size_t)(h >> i);
    SVO_HIP(hipMalloc((void **)&c->filt[i], n * 2));
    for (int s = 0; s < 3; s++) {
      SVO_HIP(hipMalloc((void **)&c->vert[s][i], n * 12));
      SVO_HIP(hipMalloc((void **)&c->norm[s][i], n * 12));
    }
  }
  SVO_HIP(hipMalloc((void **)&c->d_state, sizeof(CamState)));
  SVO_HIP(hipMalloc((void **)&c->d_partial, (size_t)kMaxIcpBlocks * 27 * sizeof(double)));
  SVO_HIP(hipMalloc((void **)&c->d_sync, sizeof(TrackSync)));
  SVO_HIP(hipMalloc((void **)&c->d_tickets, track_persistent_ticket_bytes()));
  c->d_acc = c->d_state->acc;
  const int rc = camera_reset(c);
  if (rc != SVOSLAM_OK) { camera_destroy(c); return rc; }
  *out = c;
  return SVOSLAM_OK;
}

// back to the state of a new RGBDCamera (pose identity, no frame seen); device buffers and the recorded
// launch graphs are kept.  Blocking (waits for the device).
int camera_reset(svoslam_camera *c) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipDeviceSynchronize());
  CamState init;
  memset(&init, 0, sizeof(init));
  init.orientation[0] = init.orientation[4] = init.orientation[8] = 1.0f;  // glm::mat3() = identity, vec3() = 0
  for (int i = 0; i < 16; i += 5) {
    init.update_trans[i] = 1.0f; init.fusion[i] = 1.0f; init.level_start[i] = 1.0f;
    for (int r = 0; r < 4; r++) init.fusion_ring[r][i] = 1.0f;
  }
  init.corrected = c->corrected ? 1 : 0;
  SVO_HIP(hipMemcpy(c->d_state, &init, sizeof(init), hipMemcpyHostToDevice));
  SVO_HIP(memset_sync(c->d_sync, 0, sizeof(TrackSync)));
  SVO_HIP(memset_sync(c->d_tickets, 0, track_persistent_ticket_bytes()));
  c->have_stamp = false; c->latest_stamp = 0;
  c->prepared = 0; c->tracked = 0;
  c->frame_has_icp = false;
  c->ring_slot = 0;
  c->delta_fed = false;
  c->have_model = false;  // (the mode itself is a setting and stays)
  return SVOSLAM_OK;
}

int camera_destroy(svoslam_camera *c) {
  if (!c) return SVOSLAM_OK;
  for (int i = 0; i < 3; i++) {
    if (c->filt[i]) (void)hipFree(c->filt[i]);
    for (int s = 0; s < 3; s++) {
      if (c->vert[s][i]) (void)hipFree(c->vert[s][i]);
      if (c->norm[s][i]) (void)hipFree(c->norm[s][i]);
    }
  }
  if (c->d_state) (void)hipFree(c->d_state);
  if (c->d_partial) (void)hipFree(c->d_partial);
  for (int st = 0; st < 3; st++)
    for (int i = 0; i < 3; i++) {
      if (c->inten[st][i]) (void)hipFree(c->inten[st][i]);
      if (c->grad[st][i]) (void)hipFree(c->grad[st][i]);
    }
  if (c->tmp_inten) (void)hipFree(c->tmp_inten);
  if (c->tmp_inten2) (void)hipFree(c->tmp_inten2);
  if (c->d_partial2) (void)hipFree(c->d_partial2);
  if (c->d_sync) (void)hipFree(c->d_sync);
  if (c->d_tickets) (void)hipFree(c->d_tickets);
  if (c->work_v) (void)hipFree(c->work_v);
  if (c->work_n) (void)hipFree(c->work_n);
  for (int i = 0; i < 3; i++) {
    if (c->model_filt[i]) (void)hipFree(c->model_filt[i]);
    if (c->model_v[i]) (void)hipFree(c->model_v[i]);
    if (c->model_n[i]) (void)hipFree(c->model_n[i]);
  }
  delete c;
  return SVOSLAM_OK;
}

// bilateral filter + the three pyramid levels of the incoming frame (rgbd_camera.cpp:62-93) into map set `set`
static int enqueue_preprocess(const svoslam_camera *c, const uint16_t *d_depth, const uint8_t *d_rgb, int set, hipStream_t s) {
  const int W = c->width, H = c->height;
  if (c->rgbd) {  // :66-69, 85, 90: intensity pyramid by plain 2x2 subsampling; Sobel gradient of every level
    SVO_TRY(color_to_intensity(d_rgb, c->tmp_inten, W * H, s));
    for (int i = 0; i < 3; i++) {
      const int w = W >> i, h = H >> i;
      SVO_HIP(hipMemcpyAsync(c->inten[set][i], c->tmp_inten, (size_t)w * h * 4, hipMemcpyDeviceToDevice, s));
      SVO_TRY(gradient(c->inten[set][i], c->grad[set][i], w, h, s));
      if (i != 2) SVO_TRY(subsample_f32(c->tmp_inten, c->tmp_inten2, w, h, s));
    }
  }
  SVO_TRY(bilateral_filter(d_depth, c->filt[0], W, H, s));  // :62-64
  for (int i = 0; i < 3; i++) {                             // :72-93
    const int w = W >> i, h = H >> i;
    SVO_TRY(generate_vertex_normal_maps(c->filt[i], c->vert[set][i], c->norm[set][i], w, h, c->fx, c->fy, W, H, s));
    if (i != 2) SVO_TRY(subsample_depth_u16_to(c->filt[i], c->filt[i + 1], w, h, s));
  }
  return SVOSLAM_OK;
}

// Maps of the next frame.  Independent of the pose estimation of earlier frames, so it may run on
// another stream while they are tracked -- the caller orders it after the camera_track of the frame
// three before it (whose "last" set it overwrites) and before the camera_track of its own frame.
int camera_prepare(svoslam_camera *c, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp, int32_t *processed,
                   hipStream_t s) {
  if (!c || !d_depth) return SVOSLAM_ERR_INVALID_ARG;
  if (c->rgbd && !d_rgb) return SVOSLAM_ERR_INVALID_ARG;  // (without the photometric term the colours are not looked at)
  if (c->delta_fed) return SVOSLAM_ERR_INVALID_ARG;  // fed by camera_apply_delta: no maps of the previous frame here
  if (c->have_stamp && timestamp <= c->latest_stamp) {  // :55-59
    if (processed) *processed = 0;
    return SVOSLAM_OK;
  }
  if (c->prepared - c->tracked >= 2) return SVOSLAM_ERR_INVALID_ARG;  // the third set still holds a frame to be tracked against
  c->have_stamp = true;
  c->latest_stamp = timestamp;
  if (processed) *processed = 1;
  const int set = (int)(c->prepared % 3u);
  GraphKey key;
  key.add(d_depth).add((unsigned long long)set).add(c->rgbd ? d_rgb : nullptr);
  {
    StageScope timed(kStageMaps, s);
    SVO_TRY(c->g_prep.run(key, s, [&]() -> int { return enqueue_preprocess(c, d_depth, d_rgb, set, s); }));
  }
  c->prepared++;
  return SVOSLAM_OK;
}

int camera_begin(svoslam_camera *c, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp, int32_t *processed,
                 hipStream_t s) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  int32_t used = 0;
  if (c->prepared != c->tracked) return SVOSLAM_ERR_INVALID_ARG;  // a prepared frame is waiting for camera_track
  if (c->rgbd) return SVOSLAM_ERR_INVALID_ARG;  // the stepping API carries the geometric system only
  SVO_TRY(camera_prepare(c, d_depth, d_rgb, timestamp, &used, s));
  if (processed) *processed = used;
  c->frame_has_icp = used && c->tracked >= 1;
  if (used) c->ring_slot = (int)(c->tracked & 3u);
  return SVOSLAM_OK;
}

static int iter_flags(int level, int iter) {
  int f = 0;
  if (level < 2) f |= kFlagLevelStart;
  if (iter == 0) f |= kFlagFirstIter;
  if (level == 2 && iter == 0) f |= kFlagFirstOfFrame;
  if (level == 0 && iter == kPyramidIters[0] - 1) f |= kFlagLastOfFrame;
  return f;
}

struct LevelArgs { const float *lv, *ln, *cv, *cn; int w, h, first, num; };
static LevelArgs level_args(const svoslam_camera *c, int level) {
  LevelArgs a;
  a.w = c->width >> level; a.h = c->height >> level;
  const int r0 = c->band_first >> level, r1 = (c->band_first + c->band_rows) >> level;
  const int cur = (int)(c->tracked % 3u), last = (int)((c->tracked + 2u) % 3u);  // frames f and f-1
  a.lv = c->vert[last][level]; a.ln = c->norm[last][level];
  if (c->to_model && c->have_model) { a.lv = c->model_v[level]; a.ln = c->model_n[level]; }
  a.cv = c->vert[cur][level]; a.cn = c->norm[cur][level];
  a.first = r0 * a.w; a.num = (r1 - r0) * a.w;
  return a;
}

// stepping API (multi-GPU): this band's sums are ADDED to d_acc; the caller all-reduces d_acc
int camera_icp_accumulate(svoslam_camera *c, int level, int iter, hipStream_t s) {
  if (!c || level < 0 || level > 2 || iter < 0 || iter >= kPyramidIters[level]) return SVOSLAM_ERR_INVALID_ARG;
  if (!c->frame_has_icp) return SVOSLAM_OK;
  const LevelArgs a = level_args(c, level);
  return launch_accumulate(a.lv, a.ln, a.cv, a.cn, a.w, a.h, a.first, a.num, c->d_state, iter_flags(level, iter), iter,
                           c->d_partial, c->d_acc, s);
}

int camera_icp_solve(svoslam_camera *c, int level, int iter, hipStream_t s) {
  if (!c || level < 0 || level > 2 || iter < 0 || iter >= kPyramidIters[level]) return SVOSLAM_ERR_INVALID_ARG;
  if (!c->frame_has_icp) return SVOSLAM_OK;
  cam_solve_kernel<<<1, 64, 0, s>>>(c->d_state, c->d_acc, iter, iter_flags(level, iter));
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int camera_end(svoslam_camera *c, hipStream_t s) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (c->tracked >= c->prepared) return SVOSLAM_ERR_INVALID_ARG;
  if (!c->frame_has_icp) {  // first frame: no ICP, only the fusion transform (the last solve did it otherwise)
    cam_frame_end_kernel<<<1, 64, 0, s>>>(c->d_state, 0);
    SVO_LAUNCH_CHECK();
  }
  c->tracked++;  // "swap current/last", :176-189
  c->frame_has_icp = false;
  return SVOSLAM_OK;
}

// Pose of the oldest prepared frame.  Default: ONE launch for the 19 iterations (track_persistent.hip): the register-resident
// form up to 640x480-class images, the STREAMING form (coarsest level in registers, the finer ones through the work maps) for
// large ones -- the default since round 4: with the march over bricks the frame no longer loses what this form's residency
// takes (cfg4 811 -> 893 frames/s; round 3, beside the tree march: 810 -> 690).  svoslam_config.track_mode = 1 selects the
// launch chain -- two launches per ICP iteration (accumulate over work maps; reduce + solve + compose); same bits, ~2x the
// time at 640x480 -- for several processes sharing one device; track_stream = 0: large images run their coarsest level in the
// one launch and the finer ones by the chain (round 3's hybrid).  (Built, bit-exact, measured and removed: accumulate + reduce
// + solve of a chain iteration in one launch by its last-arriving workgroup -- cfg4 815-819 against 809-811 frames/s, the
// tracker alone 0.666 against 0.620 ms; the two coarsest levels in the one launch, 1.29-1.36 against 1.16 ms; the chain
// replaying the whole transform chain from the raw maps instead of work maps, 714 against 768 frames/s.)
static bool track_chain_forced() { return config().track_mode == 1; }
static bool track_one_launch_forced() { return config().track_mode == 2; }
static bool track_stream_enabled() { return config().track_stream != 0; }

// returns 0: the whole frame was enqueued; 1: nothing was (caller: launch chain from level 2); 2 + l: levels 2 .. l + 1 were
// enqueued in the one launch, the launch chain continues at level l
static int track_one_launch(svoslam_camera *c, hipStream_t s) {
  TrackArgs A;
  for (int level = 0; level < 3; level++) {
    LevelArgs a = level_args(c, level);
    int end;
    (void)accumulate_range(a.w, a.h, a.first, a.num, end);
    A.level[level].lv = a.lv; A.level[level].ln = a.ln; A.level[level].cv = a.cv; A.level[level].cn = a.cn;
    A.level[level].first = a.first; A.level[level].end = end > a.first ? end : a.first;
    A.iters[level] = kPyramidIters[level];
  }
  A.corrected = c->corrected ? 1 : 0;
  if (c->capacity == 0 || c->cap_stream != s) {
    SVO_TRY(track_persistent_capacity(s, &c->capacity));
    c->cap_stream = s;
  }
  if (c->capacity < 2) return 1;  // no room for a solver and a worker workgroup on this stream's CUs: the launch chain
  SVO_TRY(track_persistent_plan(A, c->capacity));
  if (A.slots[0] > kTrkSlots && !track_one_launch_forced() && track_stream_enabled()) {
    // the streaming one-launch form (track_persistent.hip): work maps allocated on first use (before any capture)
    if (!c->work_v) {
      const size_t n = (size_t)c->width * (size_t)c->height;
      SVO_HIP(hipMalloc((void **)&c->work_v, n * 12));
      SVO_HIP(hipMalloc((void **)&c->work_n, n * 12));
    }
    int cap1 = 0;
    SVO_TRY(track_persistent_capacity(s, &cap1, 1));
    if (cap1 >= 2) {
      A.work_v = c->work_v; A.work_n = c->work_n;
      SVO_TRY(track_persistent_plan_stream(A, cap1));
      return track_persistent_launch(c->d_state, c->d_sync, c->d_tickets, A, s);
    }
  }
  if (A.slots[0] > kTrkSlots && !track_one_launch_forced()) {
    SVO_TRY(track_persistent_plan_coarse(A, c->capacity, 1));
    SVO_TRY(track_persistent_launch(c->d_state, c->d_sync, c->d_tickets, A, s));
    return 2 + 1;                // the chain continues at level 1
  }
  return track_persistent_launch(c->d_state, c->d_sync, c->d_tickets, A, s);
}

int camera_track(svoslam_camera *c, hipStream_t s) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (c->tracked >= c->prepared) return SVOSLAM_ERR_INVALID_ARG;  // nothing prepared
  const bool has_icp = c->tracked >= 1;
  const int ring_slot = (int)(c->tracked & 3u);
  StageScope timed(has_icp ? kStageTracker : -1, s);  // (the first frame has no ICP: not a tracker sample)
  int top_level = 2;  // coarsest level the launch chain handles
  if (has_icp && !track_chain_forced() && !c->rgbd) {
    const int rc = track_one_launch(c, s);
    if (rc < 0) return rc;
    if (rc == 0) {
      c->ring_slot = ring_slot;
      c->tracked++;
      c->frame_has_icp = false;
      return SVOSLAM_OK;
    }
    if (rc >= 2) top_level = rc - 2;  // hybrid: the coarser levels are already enqueued
  }
  // Work maps: allocated on the first chain-tracked frame (cameras served by the one-launch tracker never pay for them),
  // before any capture
  const bool work_maps = true;
  if (has_icp && !c->work_v) {
    const size_t n = (size_t)c->width * (size_t)c->height;
    SVO_HIP(hipMalloc((void **)&c->work_v, n * 12));
    SVO_HIP(hipMalloc((void **)&c->work_n, n * 12));
  }
  GraphKey key;
  key.add((unsigned long long)(c->tracked % 3u)).add((unsigned long long)has_icp)
     .add((unsigned long long)c->band_first).add((unsigned long long)c->band_rows).add((unsigned long long)c->rgbd)
     .add((unsigned long long)work_maps).add((unsigned long long)top_level).add((unsigned long long)(c->to_model && c->have_model))
     .add((unsigned long long)c->corrected);
  auto enqueue = [&]() -> int {
    if (has_icp) {
      for (int level = top_level; level >= 0; level--) {  // coarse to fine, :103
        LevelArgs a = level_args(c, level);
        int end;
        const int blocks = accumulate_range(a.w, a.h, a.first, a.num, end);
        const int cur = (int)(c->tracked % 3u), last = (int)((c->tracked + 2u) % 3u);
        int blocks2 = (int)cdiv(a.w * a.h, kIcpThreads);
        if (blocks2 > kMaxIcpBlocks) blocks2 = kMaxIcpBlocks;
        for (int it = 0; it < kPyramidIters[level]; it++) {
          const int flags = iter_flags(level, it);
          if (work_maps) {
            // iteration it > 0 reads what iteration it - 1 stored (level start and chain[0 .. it - 1) applied) and applies
            // chain[it - 1]; the last iteration of a level stores nothing
            const bool store = it + 1 < kPyramidIters[level];
            icp_accumulate_work_kernel<<<blocks, kIcpThreads, 0, s>>>(a.lv, a.ln, it ? c->work_v : a.cv, it ? c->work_n : a.cn, a.first, end,
                                                                      c->d_state, flags, it, it, store ? c->work_v : nullptr,
                                                                      store ? c->work_n : nullptr, c->d_partial);
          } else {
            icp_accumulate_kernel<<<blocks, kIcpThreads, 0, s>>>(a.lv, a.ln, a.cv, a.cn, a.first, end, c->d_state, flags, it,
                                                                 c->d_partial);
          }
          if (c->rgbd)  // rgbd_camera.cpp:126-128 (there commented out): the photometric system of the same estimate
            rgbd_accumulate_kernel<<<blocks2, kIcpThreads, 0, s>>>(c->inten[last][level], c->grad[last][level], a.lv, c->inten[cur][level],
                                                                   a.cv, 0, a.w * a.h, c->fx, c->fy, (float)(c->width / a.w),
                                                                   (float)(c->height / a.h), c->d_state, flags, it, c->d_partial2);
          cam_reduce_solve_kernel<<<1, kReduceThreads, 0, s>>>(c->d_state, c->d_partial, blocks, it, flags, c->rgbd ? c->d_partial2 : nullptr, blocks2);
        }
      }
    } else {  // first frame: no ICP, only the fusion transform (the last solve does it otherwise)
      cam_frame_end_kernel<<<1, 64, 0, s>>>(c->d_state, 0);
    }
    SVO_LAUNCH_CHECK();
    return SVOSLAM_OK;
  };
  SVO_TRY(c->g_track.run(key, s, enqueue));
  c->ring_slot = ring_slot;
  c->tracked++;
  c->frame_has_icp = false;
  return SVOSLAM_OK;
}

// RGBDCamera::update (rgbd_camera.cpp:53): maps, then pose, on one stream
int camera_update(svoslam_camera *c, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp, int32_t *processed,
                  hipStream_t s) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (c->prepared != c->tracked) return SVOSLAM_ERR_INVALID_ARG;  // a prepared frame is waiting for camera_track
  int32_t used = 0;
  SVO_TRY(camera_prepare(c, d_depth, d_rgb, timestamp, &used, s));
  if (processed) *processed = used;
  if (!used) return SVOSLAM_OK;
  return camera_track(c, s);
}

// update_trans of the frame `cur` against the frame `prev` (rgbd_camera.cpp:62-168 for that pair of images), written to
// d_delta[20] (layout: delta_export_kernel).  The camera serves as a "delta camera": its map sets 0 / 1 and its tracker
// are used, its pose afterwards means nothing, its frame counters are left at zero -- keep a second camera for the pose
// (camera_apply_delta).  Non-blocking.
int camera_pair_delta(svoslam_camera *c, const uint16_t *d_depth_prev, const uint8_t *d_rgb_prev, const uint16_t *d_depth_cur,
                      const uint8_t *d_rgb_cur, float *d_delta, hipStream_t s) {
  if (!c || !d_depth_prev || !d_depth_cur || !d_delta) return SVOSLAM_ERR_INVALID_ARG;
  if (c->rgbd && (!d_rgb_prev || !d_rgb_cur)) return SVOSLAM_ERR_INVALID_ARG;
  if (c->prepared != 0 || c->tracked != 0) return SVOSLAM_ERR_INVALID_ARG;  // a camera that has seen update() / apply_delta()
  for (int set = 0; set < 2; set++) {
    const uint16_t *d = set ? d_depth_cur : d_depth_prev;
    const uint8_t *rgb = set ? d_rgb_cur : d_rgb_prev;
    GraphKey key;
    key.add(d).add((unsigned long long)set).add(c->rgbd ? rgb : nullptr);
    SVO_TRY(c->g_prep.run(key, s, [&]() -> int { return enqueue_preprocess(c, d, rgb, set, s); }));
  }
  c->prepared = 2; c->tracked = 1;  // level_args(): current maps = set 1, last maps = set 0
  const int rc = camera_track(c, s);
  c->prepared = 0; c->tracked = 0; c->frame_has_icp = false;
  SVO_TRY(rc);
  delta_export_kernel<<<1, 64, 0, s>>>(c->d_state, c->d_sync, d_delta);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

// The pose step of RGBDCamera::update for a frame tracked elsewhere: position / orientation *= update_trans
// (rgbd_camera.cpp:172-173), fusion transform into the pose ring (main.cpp:40), lost-level count.  d_delta = nullptr (or a
// camera's first frame, which has no ICP: `pass >= 1`, :99) leaves the pose as it is.  Same device code as the tracker's
// own frame end (frame_end_step), so a stream of apply_delta(pair_delta(k-1, k)) equals a stream of update(k) bit for bit.
int camera_apply_delta(svoslam_camera *c, const float *d_delta, long long timestamp, int32_t *processed, hipStream_t s) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (c->prepared != c->tracked) return SVOSLAM_ERR_INVALID_ARG;  // a prepared frame is waiting for camera_track
  if (c->tracked > 0 && !c->delta_fed) return SVOSLAM_ERR_INVALID_ARG;  // either update() or apply_delta() feeds a camera
  if (c->have_stamp && timestamp <= c->latest_stamp) {  // :55-59
    if (processed) *processed = 0;
    return SVOSLAM_OK;
  }
  c->delta_fed = true;
  c->have_stamp = true;
  c->latest_stamp = timestamp;
  if (processed) *processed = 1;
  cam_apply_delta_kernel<<<1, 64, 0, s>>>(c->d_state, c->d_sync, c->tracked >= 1 ? d_delta : nullptr);
  SVO_LAUNCH_CHECK();
  c->ring_slot = (int)(c->tracked & 3u);
  c->tracked++;
  c->prepared++;
  c->frame_has_icp = false;
  return SVOSLAM_OK;
}

// strict = 1 (default): RGBDCamera::update as the reference has it (Q14, Q17 included).  strict = 0: this build's corrected
// tracker (own specification: icp_device.hpp icp_rot_rows; oracle ora_camera_set_strict_reference).  Before the first frame only.
int camera_set_strict_reference(svoslam_camera *c, int strict) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (c->prepared != 0 && (strict == 0) != c->corrected) return SVOSLAM_ERR_INVALID_ARG;
  if (!strict && c->rgbd) return SVOSLAM_ERR_INVALID_ARG;  // (the photometric term shares the reference's rows: not combined)
  if ((strict == 0) == c->corrected) return SVOSLAM_OK;   // unchanged value: nothing to do at ANY time (as camera_set_rgbd; ADVICE r04:
  // a defensive set_strict_reference(cam, 1) after frames had been processed used to wipe pose, frame count and sync words)
  c->corrected = strict == 0;
  return camera_reset(c);  // (prepared == 0 here: the mode flag lives in the device state the reset rewrites)
}

// RGBDCamera with the photometric term of rgbd_camera.cpp:126-141 switched on (W_RGBD = 0.1); before the first frame only
int camera_set_rgbd(svoslam_camera *c, int enable) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (c->prepared != 0 && (enable != 0) != c->rgbd) return SVOSLAM_ERR_INVALID_ARG;
  if (enable && c->to_model) return SVOSLAM_ERR_INVALID_ARG;  // (not combined with frame-to-model tracking)
  if (enable && c->corrected) return SVOSLAM_ERR_INVALID_ARG;  // (nor with the corrected tracker)
  if (enable && !c->tmp_inten) {
    const size_t n0 = (size_t)c->width * c->height;
    SVO_HIP(hipMalloc((void **)&c->tmp_inten, n0 * 4));
    SVO_HIP(hipMalloc((void **)&c->tmp_inten2, n0));
    SVO_HIP(hipMalloc((void **)&c->d_partial2, (size_t)kMaxIcpBlocks * 27 * sizeof(double)));
    for (int i = 0; i < 3; i++) {
      const size_t n = (size_t)(c->width >> i) * (size_t)(c->height >> i);
      for (int st = 0; st < 3; st++) {
        SVO_HIP(hipMalloc((void **)&c->inten[st][i], n * 4));
        SVO_HIP(hipMalloc((void **)&c->grad[st][i], n * 8));
      }
    }
  }
  c->rgbd = enable != 0;
  return SVOSLAM_OK;
}

// Frame-to-model tracking (own specification; oracle: ora_camera_set_model_depth / ora_camera_set_frame_to_model).  d_depth
// -- a depth image in the sensor's pixel grid and unit, e.g. raycast_model_depth from the pose of the frame just tracked --
// goes through the front end of a sensor frame (bilateral filter, three pyramid levels, vertex and normal maps:
// rgbd_camera.cpp:62-93) into a map set of its own.  Stream-ordered: the caller orders it behind the tracker launch that
// still reads the previous model and ahead of the next one.
int camera_set_model_depth(svoslam_camera *c, const uint16_t *d_depth, hipStream_t s) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (!d_depth) { c->have_model = false; return SVOSLAM_OK; }  // no model: the following frames are tracked against the previous frame again
  const int W = c->width, H = c->height;
  if (!c->model_n[2]) {  // first model: the set's nine buffers, all or none (a failure frees what it got: ADVICE r03)
    bool ok = true;
    for (int i = 0; i < 3 && ok; i++) {
      const size_t n = (size_t)(W >> i) * (size_t)(H >> i);
      ok = hipMalloc((void **)&c->model_filt[i], n * 2) == hipSuccess && hipMalloc((void **)&c->model_v[i], n * 12) == hipSuccess &&
           hipMalloc((void **)&c->model_n[i], n * 12) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      for (int i = 0; i < 3; i++) {
        if (c->model_filt[i]) (void)hipFree(c->model_filt[i]);
        if (c->model_v[i]) (void)hipFree(c->model_v[i]);
        if (c->model_n[i]) (void)hipFree(c->model_n[i]);
        c->model_filt[i] = nullptr; c->model_v[i] = nullptr; c->model_n[i] = nullptr;
      }
      return SVOSLAM_ERR_OOM;
    }
  }
  SVO_TRY(bilateral_filter(d_depth, c->model_filt[0], W, H, s));
  for (int i = 0; i < 3; i++) {
    const int w = W >> i, h = H >> i;
    SVO_TRY(generate_vertex_normal_maps(c->model_filt[i], c->model_v[i], c->model_n[i], w, h, c->fx, c->fy, W, H, s));
    if (i != 2) SVO_TRY(subsample_depth_u16_to(c->model_filt[i], c->model_filt[i + 1], w, h, s));
  }
  c->have_model = true;
  return SVOSLAM_OK;
}

// every ICP iteration associates the incoming frame with the model set (once one has been given) instead of the
// previous frame's maps; not combined with the photometric term (the model has no intensity image)
int camera_set_frame_to_model(svoslam_camera *c, int enable) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  if (enable && c->rgbd) return SVOSLAM_ERR_INVALID_ARG;
  c->to_model = enable != 0;
  return SVOSLAM_OK;
}

int camera_set_band(svoslam_camera *c, int first_row, int rows) {
  if (!c || first_row < 0 || rows < 0 || first_row + rows > c->height) return SVOSLAM_ERR_INVALID_ARG;
  c->band_first = first_row; c->band_rows = rows;
  return SVOSLAM_OK;
}

int camera_set_acc(svoslam_camera *c, double *d_acc) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  c->d_acc = d_acc ? d_acc : c->d_state->acc;
  return SVOSLAM_OK;
}

double *camera_acc(svoslam_camera *c) { return c ? c->d_acc : nullptr; }

// a bounded spin of the one-launch tracker gave up (should never happen: report it instead of a garbage pose)
static int check_tracker_health(svoslam_camera *c, hipStream_t s) {
  TrackSync sy;
  SVO_HIP(hipMemcpyAsync(&sy, c->d_sync, sizeof(sy), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  if (sy.fail != 0u) {
    set_last_error("one-launch tracker: a workgroup hand-off timed out (camera_reset clears it)", hipErrorLaunchTimeOut);
    return SVOSLAM_ERR_HIP;
  }
  return SVOSLAM_OK;
}

int camera_pose(svoslam_camera *c, float pos[3], float ori[9], hipStream_t s) {
  if (!c || !pos || !ori) return SVOSLAM_ERR_INVALID_ARG;
  CamState st;
  SVO_HIP(hipMemcpyAsync(&st, c->d_state, sizeof(st), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  SVO_TRY(check_tracker_health(c, s));
  memcpy(pos, st.position, sizeof(st.position));
  memcpy(ori, st.orientation, sizeof(st.orientation));
  return SVOSLAM_OK;
}

int camera_last_system(svoslam_camera *c, float A[36], float b[6], float x[6], hipStream_t s) {
  if (!c) return SVOSLAM_ERR_INVALID_ARG;
  CamState st;
  SVO_HIP(hipMemcpyAsync(&st, c->d_state, sizeof(st), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  if (A) memcpy(A, st.lastA, sizeof(st.lastA));
  if (b) memcpy(b, st.lastb, sizeof(st.lastb));
  if (x) memcpy(x, st.lastx, sizeof(st.lastx));
  return SVOSLAM_OK;
}

// pose of the frame most recently handed to update()/begin(); the slot stays untouched until three more
// frames have been started, so another stream may read it while the tracker runs ahead
const float *camera_fusion_transform_device(svoslam_camera *c) { return c ? c->d_state->fusion_ring[c->ring_slot] : nullptr; }
const float *camera_last_vertex(svoslam_camera *c, int level) {
  return (c && level >= 0 && level < 3) ? c->vert[(c->tracked + 2u) % 3u][level] : nullptr;
}
const float *camera_last_normal(svoslam_camera *c, int level) {
  return (c && level >= 0 && level < 3) ? c->norm[(c->tracked + 2u) % 3u][level] : nullptr;
}
int camera_latest_timestamp(svoslam_camera *c, int32_t *have, long long *timestamp) {
  if (!c || !have || !timestamp) return SVOSLAM_ERR_INVALID_ARG;
  *have = c->have_stamp ? 1 : 0;
  *timestamp = c->latest_stamp;
  return SVOSLAM_OK;
}

int camera_track_profile(svoslam_camera *c, unsigned long long *h_stamps, hipStream_t s) {
  if (!c || !h_stamps) return SVOSLAM_ERR_INVALID_ARG;
  return track_persistent_profile(c->d_sync, h_stamps, s);
}

int camera_tracking_lost_count(svoslam_camera *c, int *count, hipStream_t s) {
  if (!c || !count) return SVOSLAM_ERR_INVALID_ARG;
  CamState st;
  SVO_HIP(hipMemcpyAsync(&st, c->d_state, sizeof(st), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  SVO_TRY(check_tracker_health(c, s));
  *count = st.tracking_lost_count;
  return SVOSLAM_OK;
}

}  // namespace svoslam
