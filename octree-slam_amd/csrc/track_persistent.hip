// track_persistent.hip -- the 19 ICP iterations of RGBDCamera::update (src/sensor/rgbd_camera.cpp:103-168) in ONE launch.
//
// The launch-chain tracker (icp.hip) spends a frame in 38 dependent launches of 5-13 us each (profiles/
// r02_tracker_chain_timeline_before.txt): every iteration re-reads 48 B per pixel, replays the growing chain of
// rigid updates on it, and pays two kernel boundaries.  Here the pixels of a pyramid level stay in REGISTERS for all
// iterations of the level (the reference rewrites both maps in HBM after every iteration, :163-167; a lane applies
// the same 4x4 product to the values it holds -- same operands, same order, same floats), and the iterations are
// separated by one fan-in / one broadcast inside the launch instead of two kernel boundaries:
//
//   workgroup 0 ("solver")        polls the epoch's 8 x 27 fan-in accumulators (acc_of below: each holds the exact
//                                 integer sum of the arrived workers' term AND their number), adds the eight parts,
//                                 runs the 6x6 Cholesky + pose composition on one wavefront (icp_device.hpp) and
//                                 publishes this_trans / update_trans / lost as 33 eight-byte {tag = epoch, value}
//                                 granules -- in both directions the data is its own flag;
//   workgroups 1..W ("workers")   accumulate their pixels' terms (transposed: a lane owns one of the 27), reduce them in
//                                 LDS, add each term to its accumulator with one 64-bit integer atomic, then sweep
//                                 the granules until every tag matches and apply this_trans to their registers.
//
// Every shared word is accessed with agent-scope loads / stores / atomics only, so the exchange does not depend on which
// XCD a workgroup runs on (cdna_hip_programming.md Guideline 16, form R2 in both directions).
// Nothing is zeroed per launch: tags carry a device-resident generation (replay-safe), the two banks of accumulators
// alternate and the solver clears the idle one.  The solver is the only writer of CamState, from one CU.
// All spins are bounded; a give-up code in TrackSync::fail makes every later call of the camera return an error.
//
// Images too large for registers (1920x1080 at level 0: 23 pixels per lane) stream, per level, through work maps inside the
// one launch (the <2, 3, true> form below).
#include <stdlib.h>

#include "icp_device.hpp"
#include <map>
#include <mutex>

#include "config.hpp"
#include "track_persistent.hpp"

namespace svoslam {

constexpr int kTrkWaves = kTrkThreads / kWave;
constexpr int kGranules = 33;  // 16 this_trans + 16 update_trans + flags
constexpr unsigned kSpinLimit = 1u << 22;  // ~ seconds; the tracker itself takes ~0.2 ms

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_agent32(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent32(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// accumulator line of (bank, epoch, x = blockIdx & 7): 512 bytes apart so that the eight lines of an epoch live in
// different memory channels (device-scope atomics on one address serialise at ~12 ns each)
constexpr int kTicketStride = 64;  // 64-bit words
__device__ __host__ inline size_t ticket_index(unsigned bank, int e, int x) { return ((size_t)(bank * 32u + (unsigned)e) * 8u + (unsigned)x) * kTicketStride; }
// The fan-in: accumulator (bank, epoch, x, term) is a 64-bit INTEGER that every worker with (blockIdx & 7) == x adds
// (its term sum + kArriveBias) to with one agent-scope atomic.  The term sums are integer-valued (R3) and far below the bias,
// so the word carries both the exact sum of the arrived workers and their NUMBER: n = (word + bias / 2) / bias, sum = word -
// n x bias.  The data is its own flag: no row store, no drain before an arrival counter, no pass over 150 rows -- the solver
// polls 8 x 27 words until every count is there (640x480: the tracker alone 0.213 -> 0.188 ms, 1080p 0.69 -> 0.65).  <= 31 workers per x
// (kTrkMaxWorkers / 8) x 2^57 stays below 2^62; |sum| < 2^56 is implied by the exactness of the doubles it comes from (< 2^53).
constexpr unsigned long long kArriveBias = 1ull << 57;
static_assert((kTrkMaxWorkers + 7) / 8 <= 31, "arrival counts and sums share a 64-bit word");
__device__ inline unsigned long long *acc_of(unsigned long long *acc, unsigned bank, int e, int x, int term) { return acc + ticket_index(bank, e, x) + term; }

// one wavefront: wait until all kGranules tags equal `tag`; values -> LDS out[0..33).  Returns false on give-up.
__device__ inline bool sweep_broadcast(TrackSync *sy, unsigned tag, float *out, int *out_flags) {
  const int lane = (int)(threadIdx.x & 63u);
  const int g = lane < kGranules ? lane : kGranules - 1;
  for (unsigned spins = 0;; spins++) {
    const unsigned long long x = ld_agent64(&sy->granule[g]);
    const bool ok = (unsigned)(x >> 32) == tag;
    if (__all(ok)) {
      if (lane < 32) out[lane] = __uint_as_float((unsigned)x);
      if (lane == 32) *out_flags = (int)(unsigned)x;
      return true;
    }
    if ((spins & 63u) == 63u) {
      if (ld_agent32(&sy->fail) != 0u) return false;
      if (spins > kSpinLimit) { if (lane == 0) st_agent32(&sy->fail, 2u); return false; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}


// ---- transposed accumulation --------------------------------------------------------------------------------------
// A lane does not own a pixel's 27 sums (54 VGPRs of doubles and a 27 x 6-step cross-lane reduction per iteration):
// it owns ONE term.  Every lane forms the gates, J[6] and b of its pixel (localization_kernels.cu:186-226) and puts
// them as one 32-byte row into LDS; then lane (t, half) walks the 32 rows of its half of the wavefront and adds
// rint(fl(J_i * J_j) * 2^20) (or rint(fl(b * J_i) * 2^30)) of every pixel to its single accumulator.  Same products,
// same exact integer-valued addends, any order (R3); what is left to combine per workgroup is 16 partials per term.
// LDS layout of a wavefront's rows: COLUMN-major, column c (J0..J5, b, and an all-zero column for the idle lanes) = the 64
// pixels' values in a row of kColStride floats.  A lane reads four pixels of its column per ds_read_b128 (the row-major
// layout needed a ds_read2_b32 per two pixels: 32 LDS instructions per operand pair and slot against 16, and the SQ counters
// of the 1080p tracker showed 59 % of its wavefront-cycles waiting).  kColStride = 68: 16-byte aligned columns that start four
// banks apart, so the seven columns (and the two halves, 32 banks apart) of one read never share a bank.
constexpr int kCols = 8, kColStride = 68, kWaveRowFloats = kCols * kColStride;

// The fixed-point scales live in the ROWS (round 6): column c holds J_c * 2^10, column 6 holds b * 2^20, so that a lane's product of two
// entries IS fl(J_i * J_j) * 2^20 (or fl(b * J_i) * 2^30): scaling by a power of two commutes with the rounding of the product unless the
// product is subnormal (|J_i * J_j| < 2^-126), where both forms round to 0 under rint; nothing here comes near overflow (|J| is metres).
// Seven multiplies per pixel where the rows are formed instead of one per pixel and TERM where they are read.
constexpr float kRowScaleJ = 1024.0f, kRowScaleB = 1048576.0f;
static_assert((double)kRowScaleJ * kRowScaleJ == kScaleA && (double)kRowScaleB * kRowScaleJ == kScaleB, "row scales multiply to the terms' fixed-point scales");
struct TermLane { int off_a, off_b; };  // byte offsets of the lane's two columns
__device__ inline TermLane term_of_lane(int t) {
  TermLane L;
  if (t < 21) {  // (i, j), i <= j, row-major upper triangle: the order of Mat6x7's A entries in the 27 sums
    int i = 0, r = t;
    for (; i < 6; i++) { const int len = 6 - i; if (r < len) break; r -= len; }
    L.off_a = 4 * kColStride * i; L.off_b = 4 * kColStride * (i + r);
  } else if (t < 27) {
    L.off_a = 4 * kColStride * 6; L.off_b = 4 * kColStride * (t - 21);  // b * J[i]
  } else {
    L.off_a = 4 * kColStride * 7; L.off_b = 4 * kColStride * 7;  // idle lanes read the zero column
  }
  return L;
}

// gates + Jacobian row of one pixel pair -> the lane's entry of columns 0..6 (zeros when a gate rejects the pair)
__device__ __forceinline__ void icp_pixel_row(float v1x, float v1y, float v1z, float n1x, float n1y, float n1z, float v2x,
                                              float v2y, float v2z, float n2x, float n2y, float n2z, float *row, bool corrected) {
  bool ok = finitef_(v2x) && finitef_(v2y) && finitef_(v2z) && finitef_(v1x) && finitef_(v1y) && finitef_(v1z) &&
            !(v1z < 0.1f) && !(v2z < 0.1f) && !(v1z > 10.0f) && !(v2z > 10.0f);
  ok = ok && finitef_(n2x) && finitef_(n2y) && finitef_(n2z) && finitef_(n1x) && finitef_(n1y) && finitef_(n1z);
  const float dx = v2x - v1x, dy = v2y - v1y, dz = v2z - v1z;
  ok = ok && !beyond_dist_thresh(dot3(dx, dy, dz, dx, dy, dz));  // !(length > DIST_THRESH), icp_device.hpp
  ok = ok && !(dot3(n2x, n2y, n2z, n1x, n1y, n1z) < kNormThresh);
  // A_T = G_T * n1 (icp_device.hpp icp_rot_rows / icp_pixel_terms spell out the reference's nine products per row group, zeros and ones
  // included).  Here the row is only USED when every gate passed, i.e. all twelve inputs are finite: then 0 * n is +-0, 1 * n is n, and
  // adding +-0 changes at most the sign of a zero -- which no product's rint'd integer can see.  So the zero / one terms are left out
  // (12 instructions per pixel and iteration), and the strict / corrected choice is one wavefront-uniform branch instead of six selects.
  float J[6];
  if (corrected) {
    J[0] = (-v2z) * n1y + v2y * n1z;
    J[1] = v2z * n1x + (-v2x) * n1z;
    J[2] = (-v2y) * n1x + v2x * n1y;
  } else {  // Q14: the reference's rows (0,-x,-y), (-z,0,x), (y,z,0)
    J[0] = (-v2x) * n1y + (-v2y) * n1z;
    J[1] = (-v2z) * n1x + v2x * n1z;
    J[2] = v2y * n1x + v2z * n1y;
  }
  J[3] = n1x; J[4] = n1y; J[5] = n1z;
  const float bb = dot3(n1x, n1y, n1z, v1x - v2x, v1y - v2y, v1z - v2z);
#pragma unroll
  for (int c = 0; c < 6; c++) row[c * kColStride] = ok ? J[c] * kRowScaleJ : 0.0f;  // (row = the wavefront's block + lane: consecutive banks)
  row[6 * kColStride] = ok ? bb * kRowScaleB : 0.0f;
}

// lane (t, half): add its term of the 32 pixels of its half (rows = this wavefront's columns) to acc0 / acc1: eight
// 128-bit reads per operand, the pixels of a read through the packed multiplier two at a time.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void accumulate_rows(const float *rows, const TermLane &T, int half, double &acc0, double &acc1) {
  const char *base = reinterpret_cast<const char *>(rows) + half * 32 * 4;
#pragma unroll
  for (int g = 0; g < 32; g += 8) {
    float4 a[2], b[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      a[u] = *reinterpret_cast<const float4 *>(base + T.off_a + (g + 4 * u) * 4);
      b[u] = *reinterpret_cast<const float4 *>(base + T.off_b + (g + 4 * u) * 4);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const v2f a01 = {a[u].x, a[u].y}, a23 = {a[u].z, a[u].w}, b01 = {b[u].x, b[u].y}, b23 = {b[u].z, b[u].w};
      const v2f q01 = a01 * b01, q23 = a23 * b23;  // fl(J_i * J_j) * 2^20 (the rows carry the scales)
      acc0 += (double)rintf(q01.x);
      acc1 += (double)rintf(q01.y);
      acc0 += (double)rintf(q23.x);
      acc1 += (double)rintf(q23.y);
    }
  }
}

#ifdef SVO_TRK_PROF
#define TRK_STAMP(cond, e, k) do { if (cond) sy->prof[(e) & 31][k] = (unsigned long long)clock64(); } while (0)
#else
#define TRK_STAMP(cond, e, k) do { } while (0)
#endif

// the solver's iteration tail as a real call: its ~70 VGPRs and the workers' register-resident pixels are then
// allocated independently (inlined, the allocator spills the tail's values around the 255-VGPR worker body)
__device__ __attribute__((noinline)) TailResult solver_tail(CamState *st, const double *totals, int it, int flags, float *tail_sm,
                                                            float pre_ut, int pre_lost, int corrected) {
  TailPrefetch pre;
  pre.ut = pre_ut; pre.lost = pre_lost; pre.corrected = corrected;
  // both arrays are the kernel's LDS: as generic pointers (this function is a real call) every access was a flat load / store
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  typedef __attribute__((address_space(3))) volatile float lds_vfloat;
  return iteration_tail_wave(st, (lds_cdouble *)totals, it, flags, (lds_vfloat *)tail_sm, pre);
}

template <int SLOTS>
struct PixelSet {  // one lane's pixels of the current level
  float v1[SLOTS][3], n1[SLOTS][3], v2[SLOTS][3], n2[SLOTS][3];
};

// 12 floats of one pixel pair, as loaded (streaming levels: software-pipelined one pixel ahead)
struct PixelRaw { float v1[3], n1[3], v2[3], n2[3]; };

constexpr int kTrkMinWaves = 2;  // (round 5, measured: 3 -- 168 VGPRs with 32 spilled, so that a march workgroup fits beside a tracker workgroup on its
// CU instead of finding 150 CUs taken -- cfg3 2416 / 2285 against 2449 / 2424 frames/s over 100 frames, 2055 / 2060 against 2020 / 2130 over 20)
// SLOTS = pixels of a level a lane may keep in registers; MINW = minimum wavefronts per SIMD the register budget allows.
// <kTrkSlots, 2, false>: images whose finest level fits the registers (up to 640x480-class: 4 pixels per lane on <= 247 workers,
// ~220 VGPRs: one workgroup per CU).  <2, kTrkStreamMinWaves, true> (round 3): LARGE images -- only the coarsest level is register-resident, the
// finer ones STREAM: every iteration a lane reads its pixels' work maps (the current frame's maps as transformed so far:
// what the reference rewrites in place, rgbd_camera.cpp:163-167), applies the one new this_trans, uses them and stores
// them back -- a lane only ever re-reads what it wrote itself -- with the next pixel's 12 floats requested before the
// current one is used.  <= 128 VGPRs, so its workgroups find room beside the march's instead of needing empty CUs.
template <int SLOTS, int MINW, bool STREAM>
__global__ __launch_bounds__(kTrkThreads, MINW) void track_persistent_kernel(CamState *st, TrackSync *sy, unsigned long long *acc, TrackArgs A) {
  SVO_HIGH_PRIO();
  __shared__ double wsum[16][27];          // workers: per-(wave, half) term sums; solver: row-group sums
  __shared__ __attribute__((aligned(16))) float rows_s[kTrkWaves * kWaveRowFloats];  // per wavefront: eight columns of 64 pixels in flight
  __shared__ double totals[27];
  __shared__ float bc[32];                 // this_trans[16], update_trans[16] of the current epoch
  __shared__ int bc_flags;                 // bit 0: level lost, bit 1: this_trans valid
  __shared__ float chain_s[(kMaxChain + 1) * 16];
  __shared__ float tail_sm[kTailScratch];
  __shared__ int s_fail;
  const unsigned gen = ld_agent32(&sy->gen);
  const unsigned bank = gen & 1u;
  const int tid = (int)threadIdx.x;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (tid < 16) bc[16 + tid] = (tid % 5 == 0) ? 1.0f : 0.0f;  // update_trans = identity at the start of a frame (:100)
  if (tid == 0) { bc_flags = 0; s_fail = 0; }
  rows_s[(tid >> 6) * kWaveRowFloats + 7 * kColStride + (tid & 63)] = 0.0f;  // the idle lanes' all-zero column
  __syncthreads();

  if (blockIdx.x == 0) {
    // ------------------------------------------------------------------ solver
    for (int i = tid; i < 32 * 8 * 32; i += kTrkThreads)  // the next launch's accumulators (kernel boundary): 27 (32) words of each (epoch, x)
      acc[ticket_index(bank ^ 1u, i >> 8, (i >> 5) & 7) + (i & 31)] = 0ull;
    int e = 0;
    for (int level = 2; level >= 0; level--) {
      const int P = A.participants[level];
      for (int it = 0; it < A.iters[level]; it++) {
        e++;
        TRK_STAMP(tid == 0, e, 0);
        {  // fan-in: thread (x, term) polls its accumulator until it holds the arrivals of every participant with that x
          const int x = tid / 27, term = tid - 27 * x;
          const int first_wid = (x + 7) & 7;  // smallest wid with ((wid + 1) & 7) == x
          const unsigned long long expect = x < 8 && P > first_wid ? (unsigned long long)((P - first_wid + 7) / 8) : 0ull;
          long long part = 0;
          if (expect) {
            bool ok = true;
            for (unsigned spins = 0;; spins++) {
              const unsigned long long w = ld_agent64(acc_of(acc, bank, e, x, term));
              const unsigned long long n = (w + (kArriveBias >> 1)) >> 57;
              if (n == expect) { part = (long long)(w - n * kArriveBias); break; }
              if ((spins & 63u) == 63u && (ld_agent32(&sy->fail) != 0u || spins > kSpinLimit)) { ok = false; break; }
              __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) { st_agent32(&sy->fail, 1u); s_fail = 1; }
          }
          if (tid < 8 * 27) wsum[x][term] = (double)part;  // (exact: |part| < 2^53)
          __syncthreads();
          if (s_fail) return;
          TRK_STAMP(tid == 0, e, 1);
          if (tid < 27) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < 8; g++) t += wsum[g][tid];
            totals[tid] = t;
          }
          __syncthreads();
        }
        TRK_STAMP(tid == 0, e, 2);
        if (wave == 0) {
          int flags = 0;
          if (level < 2) flags |= kFlagLevelStart;
          if (it == 0) flags |= kFlagFirstIter;
          if (level == 2 && it == 0) flags |= kFlagFirstOfFrame;
          if (level == 0 && it == A.iters[0] - 1) flags |= kFlagLastOfFrame;
          const TailResult res = solver_tail(st, totals, it, flags, tail_sm, bc[16 + (lane & 15u)], bc_flags & 1, A.corrected);
          const unsigned tag = gen * 32u + (unsigned)e;
          const unsigned fl = (unsigned)(res.lost ? 1 : 0) | (unsigned)(res.solved ? 2 : 0);
          const unsigned val = lane < 16 ? __float_as_uint(res.tt) : (lane < 32 ? __float_as_uint(res.ut) : fl);
          if (lane < (unsigned)kGranules) st_agent64(&sy->granule[lane], ((unsigned long long)tag << 32) | val);
          if (lane < 16) bc[16 + lane] = res.ut;
          if (lane == 0) bc_flags = (int)fl;
          TRK_STAMP(lane == 0, e, 3);
        }
        __syncthreads();
      }
    }
    if (tid == 0) sy->gen = gen + 1u;  // read by the next launch only
    return;
  }

  // -------------------------------------------------------------------- workers
  // Participation is a suffix of the levels (participants[] grows towards the finest level, which every worker
  // takes part in).  A worker sits out the coarser levels WITHOUT following their epochs -- the participants may be
  // several epochs ahead of a worker that has not even started -- and joins by waiting for the last epoch of the
  // level before its first one: that broadcast cannot be overwritten before this worker arrives at the next epoch.
  const int wid = (int)blockIdx.x - 1;
  const int term = (int)(lane & 31u), half = (int)(lane >> 5);
  const TermLane T = term_of_lane(term);
  int e = 0;
  bool joined = false;
  for (int level = 2; level >= 0; level--) {
    const TrackLevel L = A.level[level];
    const int P = A.participants[level], slots = A.slots[level];
    const bool part = wid < P;
    if (!part) { e += A.iters[level]; continue; }
    if (!joined && e > 0) {
      if (wave == 0) {
        const bool ok = sweep_broadcast(sy, gen * 32u + (unsigned)e, bc, &bc_flags);
        if (!ok && lane == 0) s_fail = 1;
      }
      __syncthreads();
      if (s_fail) return;
    }
    joined = true;
    const bool in_regs = slots <= SLOTS;
    PixelSet<SLOTS> px;
    int nchain = 0, applied = 0;  // chain_s entries so far; of those, already contained in this lane's work maps
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < SLOTS; k++) {
        const long long p = (long long)L.first + ((long long)k * P + wid) * kTrkThreads + tid;
        const bool have = k < slots && p < (long long)L.end;
        const size_t q = have ? (size_t)p : (size_t)L.first;
        if (k < slots) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            px.v1[k][c] = L.lv[3 * q + c]; px.n1[k][c] = L.ln[3 * q + c];
            px.v2[k][c] = L.cv[3 * q + c]; px.n2[k][c] = L.cn[3 * q + c];
          }
        } else {
#pragma unroll
          for (int c = 0; c < 3; c++) px.v1[k][c] = px.n1[k][c] = px.v2[k][c] = px.n2[k][c] = 0.0f;
        }
        if (!have) px.v1[k][0] = __builtin_nanf("");  // never passes the gates
        if (level < 2) {  // the level's copy is first transformed by update_trans as the coarser level left it (:116-120)
          float ox, oy, oz;
          mat4_mul_point(bc + 16, px.v2[k][0], px.v2[k][1], px.v2[k][2], 1.0f, ox, oy, oz);
          px.v2[k][0] = ox; px.v2[k][1] = oy; px.v2[k][2] = oz;
          mat4_mul_point(bc + 16, px.n2[k][0], px.n2[k][1], px.n2[k][2], 0.0f, ox, oy, oz);
          px.n2[k][0] = ox; px.n2[k][1] = oy; px.n2[k][2] = oz;
        }
      }
    } else if (level < 2) {
      if (tid < 16) chain_s[tid] = bc[16 + tid];
      nchain = 1;
    }
    __syncthreads();
    bool lost = false;
    for (int it = 0; it < A.iters[level]; it++) {
      e++;
      TRK_STAMP(wid == 0 && tid == 0, e, 4);
      {
        double acc0 = 0.0, acc1 = 0.0;
        float *my_rows = rows_s + wave * kWaveRowFloats;
        if (!lost) {
          if (in_regs) {
#pragma unroll
            for (int k = 0; k < SLOTS; k++) {
              if (k < slots) {
                icp_pixel_row(px.v1[k][0], px.v1[k][1], px.v1[k][2], px.n1[k][0], px.n1[k][1], px.n1[k][2], px.v2[k][0],
                              px.v2[k][1], px.v2[k][2], px.n2[k][0], px.n2[k][1], px.n2[k][2], my_rows + lane, A.corrected != 0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the rows of this wavefront are read by its own lanes only
                accumulate_rows(my_rows, T, half, acc0, acc1);
                __builtin_amdgcn_wave_barrier();
              }
            }
          } else if constexpr (STREAM) {
            // streaming level with work maps: chain_s[0 .. applied) is already in what this lane stored last iteration
            // (iteration 0 reads the raw maps: applied = 0); only chain_s[applied .. nchain) is applied now.  One pixel ahead.
            const bool store = it + 1 < A.iters[level];
            const bool raw = applied == 0;
            auto fetch = [&](int k, PixelRaw &r, bool &have) {
              const long long p = (long long)L.first + ((long long)k * P + wid) * kTrkThreads + tid;
              have = k < slots && p < (long long)L.end;
              const size_t q = have ? (size_t)p : (size_t)L.first;
              const float *cv = raw ? L.cv : A.work_v, *cn = raw ? L.cn : A.work_n;
#pragma unroll
              for (int c = 0; c < 3; c++) { r.v2[c] = cv[3 * q + c]; r.n2[c] = cn[3 * q + c]; r.v1[c] = L.lv[3 * q + c]; r.n1[c] = L.ln[3 * q + c]; }
            };
            PixelRaw cur, nxt;
            bool have_cur, have_nxt = false;
            fetch(0, cur, have_cur);
            // the usual case -- ONE matrix not yet in the work maps, this_trans of the previous iteration -- keeps its twelve used elements in
            // scalar registers for the whole pass (round 6) instead of reading them from LDS for every pixel (four ds_read_b96 and their wait
            // per pixel; a lane has up to 23 pixels of a 1080p level)
            const bool one = nchain - applied == 1;
            float m1[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
              m1[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(chain_s[16 * (one ? applied : 0) + i])));
            for (int k = 0; k < slots; k++) {  // uniform trip count: every lane writes a row (zeros past the end)
              if (k + 1 < slots) fetch(k + 1, nxt, have_nxt);
              float v2x = cur.v2[0], v2y = cur.v2[1], v2z = cur.v2[2], n2x = cur.n2[0], n2y = cur.n2[1], n2z = cur.n2[2];
              if (one) {
                float ox, oy, oz;
                mat4_mul_point(m1, v2x, v2y, v2z, 1.0f, ox, oy, oz);
                v2x = ox; v2y = oy; v2z = oz;
                mat4_mul_point(m1, n2x, n2y, n2z, 0.0f, ox, oy, oz);
                n2x = ox; n2y = oy; n2z = oz;
              } else {
                for (int c = applied; c < nchain; c++) {  // transformVertexMap / transformNormalMap: the matrices not yet in the work maps
                  float ox, oy, oz;
                  mat4_mul_point(chain_s + 16 * c, v2x, v2y, v2z, 1.0f, ox, oy, oz);
                  v2x = ox; v2y = oy; v2z = oz;
                  mat4_mul_point(chain_s + 16 * c, n2x, n2y, n2z, 0.0f, ox, oy, oz);
                  n2x = ox; n2y = oy; n2z = oz;
                }
              }
              if (store && have_cur) {
                const size_t q = (size_t)((long long)L.first + ((long long)k * P + wid) * kTrkThreads + tid);
                A.work_v[3 * q] = v2x; A.work_v[3 * q + 1] = v2y; A.work_v[3 * q + 2] = v2z;
                A.work_n[3 * q] = n2x; A.work_n[3 * q + 1] = n2y; A.work_n[3 * q + 2] = n2z;
              }
              icp_pixel_row(have_cur ? cur.v1[0] : __builtin_nanf(""), cur.v1[1], cur.v1[2], cur.n1[0], cur.n1[1], cur.n1[2], v2x, v2y, v2z, n2x, n2y,
                            n2z, my_rows + lane, A.corrected != 0);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              accumulate_rows(my_rows, T, half, acc0, acc1);
              __builtin_amdgcn_wave_barrier();
              cur = nxt; have_cur = have_nxt;
            }
            if (store) applied = nchain;  // (uniform: store and nchain are the same for every lane)
          } else {
            for (int k = 0; k < slots; k++) {  // uniform trip count: every lane writes a row (zeros past the end)
              const long long p = (long long)L.first + ((long long)k * P + wid) * kTrkThreads + tid;
              const bool have = p < (long long)L.end;
              const size_t q = have ? (size_t)p : (size_t)L.first;
              float v2x = L.cv[3 * q], v2y = L.cv[3 * q + 1], v2z = L.cv[3 * q + 2];
              float n2x = L.cn[3 * q], n2y = L.cn[3 * q + 1], n2z = L.cn[3 * q + 2];
              float v1x = L.lv[3 * q];
              const float v1y = L.lv[3 * q + 1], v1z = L.lv[3 * q + 2];
              const float n1x = L.ln[3 * q], n1y = L.ln[3 * q + 1], n1z = L.ln[3 * q + 2];
              if (!have) v1x = __builtin_nanf("");
              for (int c = 0; c < nchain; c++) {  // transformVertexMap / transformNormalMap replayed
                float ox, oy, oz;
                mat4_mul_point(chain_s + 16 * c, v2x, v2y, v2z, 1.0f, ox, oy, oz);
                v2x = ox; v2y = oy; v2z = oz;
                mat4_mul_point(chain_s + 16 * c, n2x, n2y, n2z, 0.0f, ox, oy, oz);
                n2x = ox; n2y = oy; n2z = oz;
              }
              icp_pixel_row(v1x, v1y, v1z, n1x, n1y, n1z, v2x, v2y, v2z, n2x, n2y, n2z, my_rows + lane, A.corrected != 0);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              accumulate_rows(my_rows, T, half, acc0, acc1);
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
        TRK_STAMP(wid == 0 && tid == 0, e, 5);
        if (term < 27) wsum[wave * 2 + half][term] = acc0 + acc1;
        __syncthreads();
        if (wave == 0) {
          if (tid < 27) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 2 * kTrkWaves; w++) v += wsum[w][tid];
            // sum and arrival in one atomic (see acc_of); v is integer-valued and |v| < 2^53
            (void)__hip_atomic_fetch_add(acc_of(acc, bank, e, (int)(blockIdx.x & 7u), tid), (unsigned long long)(long long)v + kArriveBias,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          TRK_STAMP(wid == 0 && tid == 0, e, 6);
        }
      }
      if (wave == 0) {
        const bool ok = sweep_broadcast(sy, gen * 32u + (unsigned)e, bc, &bc_flags);
        if (!ok && lane == 0) s_fail = 1;
      }
      __syncthreads();
      if (s_fail) return;
      TRK_STAMP(wid == 0 && tid == 0, e, 7);
      const int fl = bc_flags;
      lost = (fl & 1) != 0;
      if (fl & 2) {  // this_trans of the iteration: transformVertexMap / transformNormalMap (:163-167)
        if (in_regs) {
#pragma unroll
          for (int k = 0; k < SLOTS; k++)
            if (k < slots) {
              float ox, oy, oz;
              mat4_mul_point(bc, px.v2[k][0], px.v2[k][1], px.v2[k][2], 1.0f, ox, oy, oz);
              px.v2[k][0] = ox; px.v2[k][1] = oy; px.v2[k][2] = oz;
              mat4_mul_point(bc, px.n2[k][0], px.n2[k][1], px.n2[k][2], 0.0f, ox, oy, oz);
              px.n2[k][0] = ox; px.n2[k][1] = oy; px.n2[k][2] = oz;
            }
        } else if (nchain <= kMaxChain) {
          if (tid < 16) chain_s[16 * nchain + tid] = bc[tid];
          nchain++;
        }
      }
      __syncthreads();  // bc / chain_s are rewritten in the next epoch
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------
static int worker_cap(int dflt) {  // svoslam_config.track_workers: 0 = the form's own default
  const int v = config().track_workers;
  return v > 0 ? v : dflt;
}

int track_persistent_capacity(hipStream_t s, int *max_workgroups, int variant) {
  // resident workgroups the launch may count on: occupancy x the CUs the stream may use (per DEVICE: ADVICE r02)
  static std::mutex mu;
  static std::map<int, int> per_cu_tab[2];
  std::map<int, int> &per_cu_of = per_cu_tab[variant ? 1 : 0];
  int dev = 0, cus = 0;
  SVO_HIP(hipGetDevice(&dev));
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_cu_of.find(dev);
    if (it == per_cu_of.end()) {
      int n = 0;
      if (variant) SVO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, track_persistent_kernel<kTrkStreamSlots, kTrkStreamMinWaves, true>, kTrkThreads, 0));
      else SVO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, track_persistent_kernel<kTrkSlots, kTrkMinWaves, false>, kTrkThreads, 0));
      it = per_cu_of.emplace(dev, n).first;
    }
    per_cu = it->second;
  }
  SVO_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (s != nullptr && hipExtStreamGetCUMask(s, 8, mask) == hipSuccess) {
    int bits = 0;
    for (int i = 0; i < 8; i++) bits += __builtin_popcount(mask[i]);
    if (bits > 0 && bits < cus) cus = bits;
  } else {
    (void)hipGetLastError();
  }
  // the workers wait for each other: one workgroup per CU is all the launch counts on, however many would fit
  *max_workgroups = (per_cu > 0 ? 1 : 0) * cus;
  return SVOSLAM_OK;
}

// The hybrid of large images (round 3): only the `coarse_levels` coarsest pyramid levels run in the one launch (1: level 2;
// 2: levels 2 and 1), the finer ones stay with the launch chain (icp.hip), which picks update_trans / lost up from
// CamState exactly where the solver's iteration tail left them.  A coarse iteration costs one hand-off (~11 us) here
// against two launches (~45 us beside the march at 1920x1080) there.  Levels that stay with the chain get no participants
// and no iterations; the worker count follows the finest level that is included.
int track_persistent_plan_coarse(TrackArgs &A, int capacity, int coarse_levels) {
  if (coarse_levels < 1 || coarse_levels > 2) return SVOSLAM_ERR_INVALID_ARG;
  const int finest = 3 - coarse_levels;  // finest level handled here (2 or 1)
  int cap = worker_cap(kTrkMaxWorkers);
  if (cap > capacity - 1) cap = capacity - 1;
  if (cap < 1) return SVOSLAM_ERR_INVALID_ARG;
  const int slots_target = kTrkSlots;
  long long W = 1;
  for (int l = 2; l >= 0; l--) {
    if (l < finest) { A.participants[l] = 0; A.slots[l] = 0; A.iters[l] = 0; continue; }
    const long long n = A.level[l].end > A.level[l].first ? (long long)A.level[l].end - A.level[l].first : 0;
    long long P = l == finest ? (n + (long long)kTrkThreads * slots_target - 1) / ((long long)kTrkThreads * slots_target)
                              : (n + 2 * kTrkThreads - 1) / (2 * kTrkThreads);
    if (P < 1) P = 1;
    if (P > cap) P = cap;
    A.participants[l] = (int)P;
    A.slots[l] = (int)((n + P * kTrkThreads - 1) / (P * kTrkThreads));
    if (P > W) W = P;
  }
  // participation must be a suffix of the levels (a worker of a coarser level also works on every finer one)
  if (A.participants[2] > A.participants[finest]) {
    A.participants[2] = A.participants[finest];
    const long long n = (long long)A.level[2].end - A.level[2].first;
    A.slots[2] = (int)((n + (long long)A.participants[2] * kTrkThreads - 1) / ((long long)A.participants[2] * kTrkThreads));
  }
  A.workers = (int)W;
  return SVOSLAM_OK;
}

// Large images in ONE launch (round 3): every level runs here; the coarsest keeps its pixels in registers (kTrkStreamSlots
// per lane), the finer ones stream through the work maps (TrackArgs::work_v / work_n must be set by the caller).
int track_persistent_plan_stream(TrackArgs &A, int capacity) {
  // 176 workers: the form is bound by bandwidth, not by CUs, and what it leaves free the march and the fusion use -- cfg4 with
  // the brick march, frames/s by worker count (one box, medians of 3): 64 -> 542, 96 -> 707, 128 -> 835, 160 -> 881, 176 -> 893,
  // 192 -> 884, 208 -> 865, 224 -> 773, 255 -> 790; the launch chain 811
  int cap = worker_cap(176);
  if (cap > capacity - 1) cap = capacity - 1;
  if (cap < 1) return SVOSLAM_ERR_INVALID_ARG;
  long long W = 1;
  for (int l = 0; l < 3; l++) {
    const long long n = A.level[l].end > A.level[l].first ? (long long)A.level[l].end - A.level[l].first : 0;
    long long P = (n + (long long)kTrkThreads * kTrkStreamSlots - 1) / ((long long)kTrkThreads * kTrkStreamSlots);
    if (P < 1) P = 1;
    if (P > cap) P = cap;
    if (l > 0 && P > A.participants[l - 1]) P = A.participants[l - 1];  // participation is a suffix of the levels
    A.participants[l] = (int)P;
    A.slots[l] = (int)((n + P * kTrkThreads - 1) / (P * kTrkThreads));
    if (P > W) W = P;
  }
  A.workers = (int)W;
  A.variant = 1;
  return SVOSLAM_OK;
}

int track_persistent_plan(TrackArgs &A, int capacity) {
  // workers: the finest level decides (kTrkSlots pixels per lane); coarser levels use as many of them as give a lane
  // two pixels.  All workers take part in the finest (last) level.
  const int slots_target = kTrkSlots;
  int cap = worker_cap(kTrkMaxWorkers);
  if (cap > capacity - 1) cap = capacity - 1;
  if (cap < 1) return SVOSLAM_ERR_INVALID_ARG;
  const long long n0 = A.level[0].end > A.level[0].first ? (long long)A.level[0].end - A.level[0].first : 0;
  long long W = (n0 + (long long)kTrkThreads * slots_target - 1) / ((long long)kTrkThreads * slots_target);
  if (W < 1) W = 1;
  if (W > cap) W = cap;
  A.workers = (int)W;
  for (int l = 0; l < 3; l++) {
    const long long n = A.level[l].end > A.level[l].first ? (long long)A.level[l].end - A.level[l].first : 0;
    long long P = l == 0 ? W : (n + 2 * kTrkThreads - 1) / (2 * kTrkThreads);
    if (P < 1) P = 1;
    if (P > W) P = W;
    A.participants[l] = (int)P;
    A.slots[l] = (int)((n + P * kTrkThreads - 1) / (P * kTrkThreads));
  }
  return SVOSLAM_OK;
}

int track_persistent_profile(const TrackSync *d_sync, unsigned long long *out, hipStream_t s) {
  if (!d_sync || !out) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipMemcpyAsync(out, d_sync->prof, sizeof(d_sync->prof), hipMemcpyDeviceToHost, s));
  SVO_HIP(hipStreamSynchronize(s));
  return SVOSLAM_OK;
}

size_t track_persistent_ticket_bytes() { return ticket_index(2, 0, 0) * sizeof(unsigned long long); }

// The kernel's workgroups wait for each other, and its worker count assumes an otherwise idle device.  Two such launches
// dispatched at the same time (two cameras or sessions of one process on one device, on different streams) could each
// become partially resident and wait for the other until the bounded spins give up.  Launches of ONE
// process on ONE device are therefore chained: every launch is followed by an event record on its stream (~2.6 us of the
// tracker stream, which does not bound the frame), and a launch on a stream other than the previous one's first waits
// for that event.  (Recording on the previous stream at the time of the NEXT launch would cost nothing in the
// single-stream case, but that stream may have been destroyed by then.)  Several PROCESSES sharing a device (a test
// arrangement: bench.py's SVOSLAM_BENCH_ONE_DEVICE) use the launch chain (svoslam_config.track_mode = 1); a give-up still
// surfaces as an error from the camera's next readback, and travels with the delta record of a frame-sharded session.
int track_persistent_launch(CamState *st, TrackSync *sy, unsigned *tickets, const TrackArgs &A, hipStream_t s) {
  unsigned long long *acc = reinterpret_cast<unsigned long long *>(tickets);
  struct DevChain { hipStream_t last = nullptr; hipEvent_t ev = nullptr; bool used = false; };
  static std::mutex mu;
  static std::map<int, DevChain> chain_of;
  int dev = 0;
  SVO_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  DevChain &dc = chain_of[dev];
  if (!dc.ev) SVO_HIP(hipEventCreateWithFlags(&dc.ev, hipEventDisableTiming));
  if (dc.used && dc.last != s) SVO_HIP(hipStreamWaitEvent(s, dc.ev, 0));  // the previous launch (any stream) has finished
  if (A.variant) track_persistent_kernel<kTrkStreamSlots, kTrkStreamMinWaves, true><<<A.workers + 1, kTrkThreads, 0, s>>>(st, sy, acc, A);
  else track_persistent_kernel<kTrkSlots, kTrkMinWaves, false><<<A.workers + 1, kTrkThreads, 0, s>>>(st, sy, acc, A);
  SVO_LAUNCH_CHECK();
  SVO_HIP(hipEventRecord(dc.ev, s));
  dc.last = s; dc.used = true;
  return SVOSLAM_OK;
}

}  // namespace svoslam
