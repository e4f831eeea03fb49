// icp_device.hpp -- device-side pieces shared by icp.hip (launch-chain tracker, stateless ICP entry points) and
// track_persistent.hip (one-launch tracker): tracker state, exact wave reductions, the per-pixel normal-equation
// terms, the 6x6 Cholesky and the pose composition of RGBDCamera::update (src/sensor/rgbd_camera.cpp:53-222).
#pragma once
#include <math.h>

#include "common.hpp"

namespace svoslam {

__device__ constexpr float kDistThresh = 0.1f;   // localization_kernels.cu:17
__device__ constexpr float kNormThresh = 0.87f;  // :18
// The distance gate without the square root (round 6): sqrtf(d2) > kDistThresh  <=>  d2 > kDistThreshSq.  sqrtf is correctly rounded
// (IEEE 754; the build asks for it) and therefore monotonic, and kDistThreshSq = 0x3C23D70B is the largest binary32 whose root rounds to a
// value <= 0.1f (its successor's root rounds to the next float above 0.1f); NaN makes both comparisons false, +Inf both true
// (tests/test_icp_gate.py pins threshold and equivalence on the CPU).  The correctly rounded root was 16 instructions per pixel and iteration.
__device__ constexpr float kDistThreshSq = 0x1.47ae16p-7f;
__device__ __forceinline__ bool beyond_dist_thresh(float d2) { return d2 > kDistThreshSq; }
constexpr double kScaleA = 1048576.0;            // 2^20
constexpr double kScaleB = 1073741824.0;         // 2^30
constexpr int kMaxChain = 10;                    // max(PYRAMID_ITERS)
// photometric RGB-D term (own specification, oracle/svoslam_oracle.c "photometric RGB-D term")
__device__ constexpr float kWeightRgbd = 0.1f;   // W_RGBD, rgbd_camera.cpp:20
constexpr double kScaleRgbdA = 256.0;            // 2^8
constexpr double kScaleRgbdB = 1048576.0;        // 2^20

struct CamState {
  double acc[27];
  float update_trans[16];
  float level_start[16];
  float chain[kMaxChain][16];
  float position[3];
  float orientation[9];
  float fusion[16];
  float fusion_ring[4][16];  // fusion transform of the last 4 frames (slot = frame sequence & 3): lets the
                             // mapping stream read frame k's pose while the tracking stream is on frame k+1
  float lastA[36], lastb[6], lastx[6];
  int frames_done;          // frames whose pose is final; the next frame's pose goes to fusion_ring[frames_done & 3]
  int lost;                 // NaN seen at this pyramid level (rgbd_camera.cpp:148-151)
  int tracking_lost_count;  // levels abandoned so far
  int corrected;            // svoslam_camera_set_strict_reference(cam, 0): the corrected tracker (see icp_rot_rows)
};

// iteration flags (host-known)
constexpr int kFlagLevelStart = 1;  // level < 2: the level's copy is first transformed by update_trans (:116-120)
constexpr int kFlagFirstIter = 2;   // iteration 0 of its level
constexpr int kFlagFirstOfFrame = 4;
constexpr int kFlagLastOfFrame = 8;

// Sum of a double over the 64 lanes of a wavefront with DPP moves only (VALU; no LDS traffic):
// row_shr 1,2,4,8 leave each 16-lane row's sum in its last lane, row_bcast15 / row_bcast31 carry
// them on; the total ends up in lane 63.  Every addend is an integer-valued double, so the order
// of the additions does not matter (exact).
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
  return v + __hiloint2double(shi, slo);
}
__device__ inline double wave_sum_to_lane63(double v) {
  v = dpp_add<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}


// Rotational rows of A_T = G_T * n (localization_kernels.cu:207-213).  The reference's G_T rows are (0,-x,-y), (-z,0,x),
// (y,z,0) -- not the rows of [v]x (Q14) --, which makes its tracker turn by degrees per frame on clean data.  corrected (own
// specification, oracle: ora_camera_set_strict_reference(c, 0); include/svoslam.h svoslam_camera_set_strict_reference): the
// rows of [v2]x, i.e. A_T[0..2] = v2 x n1, the linearisation of n1 . (v1 - (R v2 + t)) in a small rotation vector; with it go
// this_trans = T(t) Rz Ry Rx with positive angles (iteration_tail_wave) and a position that takes the update's translation
// (frame_end_step).  Same products in the same order; `corrected` is wavefront-uniform.
__device__ __forceinline__ void icp_rot_rows(float x, float y, float z, float nx, float ny, float nz, bool corrected, float &j0, float &j1,
                                             float &j2) {
  const float a01 = corrected ? -z : -x, a02 = corrected ? y : -y;
  const float a10 = corrected ? z : -z, a12 = corrected ? -x : x;
  const float a20 = corrected ? -y : y, a21 = corrected ? x : z;
  j0 = (0.0f * nx + a01 * ny) + a02 * nz;
  j1 = (a10 * nx + 0.0f * ny) + a12 * nz;
  j2 = (a20 * nx + a21 * ny) + 0.0f * nz;
}

// The 27 exact normal-equation terms of one pixel pair (localization_kernels.cu:186-226) added to acc[27]:
// gates, A_T = G_T * n1 with the G_T rows of :208-213 (Q14), products in source order; fixed point:
// prod * 2^k is exact in binary32 (power-of-two scale), rintf gives the same integer as
// rint((double)prod * 2^k) of the specification (R3).
__device__ __forceinline__ void icp_pixel_terms(float v1x, float v1y, float v1z, float n1x, float n1y, float n1z, float v2x,
                                                float v2y, float v2z, float n2x, float n2y, float n2z, double (&acc)[27],
                                                bool corrected = false) {
  // Branch-free: the terms of a rejected pixel are formed (from whatever its floats hold) and replaced by +0
  // before they are added -- one set of accumulators, no per-pixel control flow, same sums.
  bool ok = finitef_(v2x) && finitef_(v2y) && finitef_(v2z) && finitef_(v1x) && finitef_(v1y) && finitef_(v1z) &&
            !(v1z < 0.1f) && !(v2z < 0.1f) && !(v1z > 10.0f) && !(v2z > 10.0f);
  ok = ok && finitef_(n2x) && finitef_(n2y) && finitef_(n2z) && finitef_(n1x) && finitef_(n1y) && finitef_(n1z);
  const float dx = v2x - v1x, dy = v2y - v1y, dz = v2z - v1z;
  ok = ok && !beyond_dist_thresh(dot3(dx, dy, dz, dx, dy, dz));  // !(length > DIST_THRESH)
  ok = ok && !(dot3(n2x, n2y, n2z, n1x, n1y, n1z) < kNormThresh);
  float J[6];
  icp_rot_rows(v2x, v2y, v2z, n1x, n1y, n1z, corrected, J[0], J[1], J[2]);
  J[3] = (1.0f * n1x + 0.0f * n1y) + 0.0f * n1z;
  J[4] = (0.0f * n1x + 1.0f * n1y) + 0.0f * n1z;
  J[5] = (0.0f * n1x + 0.0f * n1y) + 1.0f * n1z;
  const float bb = dot3(n1x, n1y, n1z, v1x - v2x, v1y - v2y, v1z - v2z);
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) {
      const float prod = J[i] * J[j];
      const float q = rintf(prod * 1048576.0f);
      acc[k++] += (double)(ok ? q : 0.0f);
    }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const float prod = bb * J[i];
    const float q = rintf(prod * 1073741824.0f);
    acc[21 + i] += (double)(ok ? q : 0.0f);
  }
}

// The 27 exact terms of the photometric system for one pixel (oracle: ora_rgbd_cost_raw): same-index association,
// residual r = I_last - I_cur, Jacobian through the pinhole derivative at the current vertex and the geometric term's
// own G_T rows.  ax = (fx / z) / sx, ay = (fy / z) / sy with sx, sy the level's pixel pitch in full-resolution pixels.
__device__ __forceinline__ void rgbd_pixel_terms(float v1x, float v1y, float v1z, float v2x, float v2y, float v2z, float gx, float gy,
                                                 float r, float fx, float fy, float sx, float sy, double (&acc)[27]) {
  bool ok = finitef_(v2x) && finitef_(v2y) && finitef_(v2z) && finitef_(v1x) && finitef_(v1y) && finitef_(v1z) &&
            !(v1z < 0.1f) && !(v2z < 0.1f) && !(v1z > 10.0f) && !(v2z > 10.0f);
  const float dx = v2x - v1x, dy = v2y - v1y, dz = v2z - v1z;
  ok = ok && !beyond_dist_thresh(dot3(dx, dy, dz, dx, dy, dz));  // !(length > DIST_THRESH)
  const float iz = 1.0f / v2z;
  const float ax = (fx * iz) / sx, ay = (fy * iz) / sy;
  const float wx = gx * ax;
  const float wy = -(gy * ay);
  const float wz = (gy * ay) * (v2y * iz) - (gx * ax) * (v2x * iz);
  float J[6];
  J[0] = (0.0f * wx + (-v2x) * wy) + (-v2y) * wz;
  J[1] = ((-v2z) * wx + 0.0f * wy) + v2x * wz;
  J[2] = (v2y * wx + v2z * wy) + 0.0f * wz;
  J[3] = (1.0f * wx + 0.0f * wy) + 0.0f * wz;
  J[4] = (0.0f * wx + 1.0f * wy) + 0.0f * wz;
  J[5] = (0.0f * wx + 0.0f * wy) + 1.0f * wz;
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) {
      const float prod = J[i] * J[j];
      const float q = rintf(prod * 256.0f);
      acc[k++] += (double)(ok ? q : 0.0f);
    }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const float prod = r * J[i];
    const float q = rintf(prod * 1048576.0f);
    acc[21 + i] += (double)(ok ? q : 0.0f);
  }
}

// ----------------------------------------------------------------------------
// device-resident solve + pose composition (one lane)
// ----------------------------------------------------------------------------
__device__ inline void d_identity(float *m) {
  for (int i = 0; i < 16; i++) m[i] = 0.0f;
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}
// glm operator*(mat4, mat4), type_mat4x4.inl:753-775
__device__ inline void d_mat4_mul(const float *a, const float *b, float *out) {
  float r[16];
  for (int c = 0; c < 4; c++)
    for (int row = 0; row < 4; row++)
      r[4 * c + row] = ((a[row] * b[4 * c] + a[4 + row] * b[4 * c + 1]) + a[8 + row] * b[4 * c + 2]) + a[12 + row] * b[4 * c + 3];
  for (int i = 0; i < 16; i++) out[i] = r[i];
}
// glm::translate, gtc/matrix_transform.inl:35-45
__device__ inline void d_translate(const float *m, const float *v, float *out) {
  float r[16];
  for (int i = 0; i < 16; i++) r[i] = m[i];
  for (int row = 0; row < 4; row++) r[12 + row] = ((m[row] * v[0] + m[4 + row] * v[1]) + m[8 + row] * v[2]) + m[12 + row];
  for (int i = 0; i < 16; i++) out[i] = r[i];
}
// Deterministic sin/cos in binary64 with explicit fma (Cody-Waite by pi/2 + fdlibm
// kernels), rounded once to binary32; the CPU oracle evaluates the same sequence.
// The reference calls the host libm through glm::rotate (matrix_transform.inl:60-61).
__device__ inline void d_sincos(float af, float &s_out, float &c_out) {
  const double x = (double)af;
  const double kd = rint(x * 0.63661977236758134308);
  double r = fma(kd, -1.57079632673412561417e+00, x);
  r = fma(kd, -6.07710050650619224932e-11, r);
  const double z = r * r;
  double sp = 1.58969099521155010221e-10;
  sp = fma(sp, z, -2.50507602534068634195e-08);
  sp = fma(sp, z, 2.75573137070700676789e-06);
  sp = fma(sp, z, -1.98412698298579493134e-04);
  sp = fma(sp, z, 8.33333333332248946124e-03);
  sp = fma(sp, z, -1.66666666666666324348e-01);
  const double sn = fma(r * z, sp, r);
  double cp = -1.13596475577881948265e-11;
  cp = fma(cp, z, 2.08757232129817482790e-09);
  cp = fma(cp, z, -2.75573143513906633035e-07);
  cp = fma(cp, z, 2.48015872894767294178e-05);
  cp = fma(cp, z, -1.38888888888741095749e-03);
  cp = fma(cp, z, 4.16666666666666019037e-02);
  const double cs = fma(z * z, cp, fma(z, -0.5, 1.0));
  const long long k = (long long)kd;
  double s, c;
  switch ((int)(k & 3)) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
  }
  s_out = (float)s;
  c_out = (float)c;
}
// glm::rotate (degrees API), gtc/matrix_transform.inl:47-86
__device__ inline void d_rotate_deg(const float *m, float angle, float vx, float vy, float vz, float *out) {
  const float a = angle * 0.01745329251994329576923690768489f;
  float c, s;
  d_sincos(a, s, c);
  const float inv = 1.0f / sqrtf((vx * vx + vy * vy) + vz * vz);
  const float axis[3] = {vx * inv, vy * inv, vz * inv};
  const float temp[3] = {(1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2]};
  float R[3][3];
  R[0][0] = c + temp[0] * axis[0];
  R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
  R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
  R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
  R[1][1] = c + temp[1] * axis[1];
  R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
  R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
  R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
  R[2][2] = c + temp[2] * axis[2];
  float r[16];
  for (int col = 0; col < 3; col++)
    for (int row = 0; row < 4; row++) r[4 * col + row] = (m[row] * R[col][0] + m[4 + row] * R[col][1]) + m[8 + row] * R[col][2];
  for (int row = 0; row < 4; row++) r[12 + row] = m[12 + row];
  for (int i = 0; i < 16; i++) out[i] = r[i];
}

// RGBDCamera::solveCholesky, rgbd_camera.cpp:194-222 (float storage, double inner sums)
__device__ inline void d_solve_cholesky(const float *A, const float *b, float *x) {
  float LU[36], y[6];
  for (int i = 0; i < 36; i++) LU[i] = 0.0f;
  for (int i = 0; i < 6; i++) y[i] = 0.0f;
  for (int k = 0; k < 6; ++k) {
    double sum = 0.;
    for (int p = 0; p < k; ++p) sum += LU[k * 6 + p] * LU[k * 6 + p];
    LU[k * 6 + k] = (float)sqrt(A[k * 6 + k] - sum);
    for (int i = k + 1; i < 6; ++i) {
      double sum2 = 0.;
      for (int p = 0; p < k; ++p) sum2 += LU[i * 6 + p] * LU[k * 6 + p];
      LU[i * 6 + k] = (float)((A[i * 6 + k] - sum2) / LU[k * 6 + k]);
    }
  }
  for (int i = 0; i < 6; ++i) {
    double sum = 0.;
    for (int k = 0; k < i; ++k) sum += LU[i * 6 + k] * y[k];
    y[i] = (float)((b[i] - sum) / LU[i * 6 + i]);
  }
  for (int i = 5; i >= 0; --i) {
    double sum = 0.;
    for (int k = i + 1; k < 6; ++k) sum += LU[k * 6 + i] * x[k];
    x[i] = (float)((y[i] - sum) / LU[i * 6 + i]);
  }
}

// ---- the same iteration tail spread over ONE wavefront ---------------------------------------------
// solveCholesky is a chain of 6 square roots and 27 divisions in binary64 (software sequences of ~30
// dependent instructions each): executed by one lane it costs ~5 us per ICP iteration, 19 times a frame.
// Here lane i (< 6) owns row i of A / LU: the 5 quotients of a column, and everything else that is
// independent in the reference's loops, run side by side; single values travel with v_readlane (the
// source lanes are compile-time constants).  Every value is produced by the reference's expression
// with its operand order (float products, double running sums, one rounding to float), so the bits
// are those of d_solve_cholesky.
__device__ inline float lane_bcast(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// must be called by all 64 lanes of a wavefront; sums = 27 doubles at a uniform address; x[6] on every lane
// sums2 (optional): the photometric system's 27 sums; A = A1 + W_RGBD * A2, b likewise (rgbd_camera.cpp:130-141)
// (SP: pointer to the sums -- generic, or LDS-qualified where the caller is not inlined into the kernel that owns the array)
template <class SP>
__device__ inline void wave_solve_cholesky(SP sums, const double *sums2, float *x, float &a_elem, float &b_elem) {
  const int lane = (int)(threadIdx.x & 63u);
  const int row = lane < 6 ? lane : 5;  // spare lanes shadow row 5
  // A is symmetric, sums hold its upper triangle row by row: index of (i <= j) = i*6 - i*(i-1)/2 + (j - i)
  float a[6], lu[6], diag[6];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const int i = row < c ? row : c, j = row < c ? c : row;
    const int idx = i * 6 - (i * (i - 1)) / 2 + (j - i);
    a[c] = (float)(sums[idx] * (1.0 / kScaleA));
    if (sums2) a[c] = a[c] + kWeightRgbd * (float)(sums2[idx] * (1.0 / kScaleRgbdA));
    lu[c] = 0.0f;
  }
  float b_own = (float)(sums[21 + row] * (1.0 / kScaleB));
  if (sums2) b_own = b_own + kWeightRgbd * (float)(sums2[21 + row] * (1.0 / kScaleRgbdB));
  {  // element `lane` of the row-major A (for the diagnostics copy), b likewise
    const int e = lane < 36 ? lane : 35, r = e / 6, c = e % 6;
    const int i = r < c ? r : c, j = r < c ? c : r;
    const int idx = i * 6 - (i * (i - 1)) / 2 + (j - i);
    a_elem = (float)(sums[idx] * (1.0 / kScaleA));
    if (sums2) a_elem = a_elem + kWeightRgbd * (float)(sums2[idx] * (1.0 / kScaleRgbdA));
    b_elem = b_own;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {  // rgbd_camera.cpp:198-209
    float rk[6];
#pragma unroll
    for (int p = 0; p < 6; p++) rk[p] = p < k ? lane_bcast(lu[p], k) : 0.0f;
    double sum = 0.;
#pragma unroll
    for (int p = 0; p < 6; p++) if (p < k) sum += lu[p] * lu[p];
    const float d_own = (float)sqrt(a[k] - sum);  // right on lane k
    diag[k] = lane_bcast(d_own, k);
    double sum2 = 0.;
#pragma unroll
    for (int p = 0; p < 6; p++) if (p < k) sum2 += lu[p] * rk[p];
    const float v = (float)((a[k] - sum2) / diag[k]);  // right on lanes > k
    lu[k] = lane == k ? diag[k] : (lane > k ? v : 0.0f);
  }
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {  // :210-215, row i on lane i
    double sum = 0.;
#pragma unroll
    for (int k = 0; k < 6; k++) if (k < i) sum += lu[k] * y[k];
    const float cand = (float)((b_own - sum) / diag[i]);
    y[i] = lane_bcast(cand, i);
  }
  float lt[6];  // column `lane` of LU: lt[k] = LU[k][lane]
#pragma unroll
  for (int k = 0; k < 6; k++) lt[k] = 0.0f;
#pragma unroll
  for (int c = 0; c < 6; c++)
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k > c) { const float t = lane_bcast(lu[c], k); lt[k] = lane == c ? t : lt[k]; }
#pragma unroll
  for (int i = 5; i >= 0; --i) {  // :216-221, column i on lane i
    double sum = 0.;
#pragma unroll
    for (int k = 0; k < 6; k++) if (k > i) sum += lt[k] * x[k];
    const float cand = (float)((y[i] - sum) / diag[i]);
    x[i] = lane_bcast(cand, i);
  }
}

// element e = 4 * col + row of glm operator*(mat4, mat4) (type_mat4x4.inl:753-775): the expression of d_mat4_mul
template <class MP>
__device__ inline float mat4_mul_elem(MP a, MP b, int e) {
  const int c = e >> 2, row = e & 3;
  return ((a[row] * b[4 * c] + a[4 + row] * b[4 * c + 1]) + a[8 + row] * b[4 * c + 2]) + a[12 + row] * b[4 * c + 3];
}

// :172-173 pose update (Q17: row-vector products) and the fusion transform of main.cpp:40; m = update_trans
__device__ inline void frame_end_step(CamState *st, int apply_update, const volatile float *m) {
  const int slot = st->frames_done;  // kept on the device so that the recorded launch sequence is the same for every frame
  if (apply_update) {
    float v[4] = {st->position[0], st->position[1], st->position[2], 1.0f};
    // corrected tracker: the row-vector product below gives R^T p and drops the update's translation (Q17); main.cpp:40 maps a
    // camera point x to orientation * (x + position), so composing with v_last = R v_cur + t needs R^T (p + t): add t first
    if (st->corrected) { v[0] = v[0] + m[12]; v[1] = v[1] + m[13]; v[2] = v[2] + m[14]; }
    float np[3];
    for (int i = 0; i < 3; i++) np[i] = ((m[4 * i] * v[0] + m[4 * i + 1] * v[1]) + m[4 * i + 2] * v[2]) + m[4 * i + 3] * v[3];
    st->position[0] = np[0]; st->position[1] = np[1]; st->position[2] = np[2];
    float o4[16], mm[16], no[16];
    d_identity(o4);
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) o4[4 * c + r] = st->orientation[3 * c + r];
    for (int i = 0; i < 16; i++) mm[i] = m[i];
    d_mat4_mul(o4, mm, no);
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) st->orientation[3 * c + r] = no[4 * c + r];
  }
  float o4[16], I[16], t[16];
  d_identity(o4);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) o4[4 * c + r] = st->orientation[3 * c + r];
  d_identity(I);
  d_translate(I, st->position, t);
  d_mat4_mul(o4, t, st->fusion);
  for (int i = 0; i < 16; i++) st->fusion_ring[slot & 3][i] = st->fusion[i];
  st->frames_done = slot + 1;
}

// One ICP iteration's host part (rgbd_camera.cpp:100, :116-120, :143-160, :172-173) on ONE wavefront; all 64 lanes
// call it.  sums = the 27 fixed-point sums (LDS), sm = 128 floats of LDS scratch.  The 4x4 matrices live one
// element per lane (lanes 0..15): the three rotations are built side by side on lanes 0..2, the four matrix
// products of :154-160 cost one LDS round trip each instead of 64 dependent multiply-adds on a single lane
// (the tail used to take ~3 of the launch's 8.6 us).  Every element is the reference's expression, unchanged.
constexpr int kTailScratch = 128;
// state words the tail needs, fetched by the caller BEFORE it waits for the sums (one round trip instead of two;
// a global access costs ~2 us while a raycast is running)
struct TailPrefetch { float ut; int lost; int corrected; };
// what an iteration leaves behind, element e = lane & 15 of each matrix on every lane (the one-launch tracker hands
// these to the other workgroups): update_trans, this_trans (valid when `solved`), the level's lost flag
struct TailResult { float ut, tt; int lost, solved; };
__device__ inline TailPrefetch tail_prefetch(const CamState *st, int flags) {
  TailPrefetch p;
  const int e = (int)(threadIdx.x & 15u);
  p.ut = st->update_trans[e];
  p.lost = st->lost;
  p.corrected = st->corrected;
  (void)flags;
  return p;
}

template <class SP, class MP>
__device__ inline TailResult iteration_tail_wave(CamState *st, SP sums, int slot, int flags, MP sm,
                                                 const TailPrefetch &pre, const double *sums2 = nullptr) {
  TailResult res;
  res.tt = 0.0f; res.solved = 0;
  const int lane = (int)(threadIdx.x & 63u), e = lane & 15;
  // level start (:100, :116-120): update_trans element e on lane e
  float ut = (flags & kFlagFirstOfFrame) ? ((e % 5 == 0) ? 1.0f : 0.0f) : pre.ut;
  int lost = 0;
  if (flags & kFlagFirstIter) {
    if (lane < 16) st->level_start[e] = ut;
    if (lane == 0) st->lost = 0;
  } else {
    lost = pre.lost;
  }
  if ((flags & kFlagFirstOfFrame) && lane < 16) st->update_trans[e] = ut;
  if (!lost) {
    float x[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, a_elem, b_elem;
    wave_solve_cholesky(sums, sums2, x, a_elem, b_elem);
    if (lane < 36) st->lastA[lane] = a_elem;
    if (lane < 6) { st->lastb[lane] = b_elem; st->lastx[lane] = x[lane < 6 ? lane : 0]; }
    if (x[0] != x[0] || x[1] != x[1] || x[2] != x[2] || x[3] != x[3] || x[4] != x[4] || x[5] != x[5]) {
      if (lane == 0) {
        st->lost = 1;  // "Camera tracking is lost." -> abandon this level (:148-151)
        st->tracking_lost_count++;
      }
      lost = 1;
    } else {
      // this_trans = Rz(-x2) * Ry(-x1) * Rx(-x0) * T(x3,x4,x5), glm degrees API (:154-158); corrected tracker: x is the rotation
      // vector and translation that carry a current-frame point to R v + t: T(x3,x4,x5) * Rz(x2) * Ry(x1) * Rx(x0)
      const bool cor = pre.corrected != 0;
      const int k = lane < 2 ? lane : 2;  // lane 0: Rz, lane 1: Ry, lanes 2..: Rx
      const float xk = k == 0 ? x[2] : (k == 1 ? x[1] : x[0]);
      float I[16], R[16], tr[16];
      d_identity(I);
      d_rotate_deg(I, (cor ? xk : -xk) * 180.0f / 3.14159f, k == 2 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 0 ? 1.0f : 0.0f, R);
      const float tv[3] = {x[3], x[4], x[5]};
      d_translate(I, tv, tr);
      if (lane < 3)
        for (int i = 0; i < 16; i++) sm[16 * lane + i] = R[i];
      if (lane == 3)
        for (int i = 0; i < 16; i++) sm[48 + i] = tr[i];
      if (lane < 16) sm[112 + e] = ut;
      __builtin_amdgcn_wave_barrier();
      const float t1 = cor ? mat4_mul_elem(sm + 48, sm, e) : mat4_mul_elem(sm, sm + 16, e);               // Rz * Ry      | T * Rz
      if (lane < 16) sm[64 + e] = t1;
      __builtin_amdgcn_wave_barrier();
      const float t2 = cor ? mat4_mul_elem(sm + 64, sm + 16, e) : mat4_mul_elem(sm + 64, sm + 32, e);     // * Rx         | * Ry
      if (lane < 16) sm[80 + e] = t2;
      __builtin_amdgcn_wave_barrier();
      const float tt = cor ? mat4_mul_elem(sm + 80, sm + 32, e) : mat4_mul_elem(sm + 80, sm + 48, e);     // * T          | * Rx  = this_trans
      if (lane < 16) sm[96 + e] = tt;
      __builtin_amdgcn_wave_barrier();
      ut = mat4_mul_elem(sm + 96, sm + 112, e);                // update_trans = this_trans * update_trans (:160)
      if (lane < 16) {
        st->update_trans[e] = ut;
        if (slot < kMaxChain) st->chain[slot][e] = tt;
      }
      res.tt = tt; res.solved = 1;
    }
  }
  if (flags & kFlagLastOfFrame) {
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) sm[112 + e] = ut;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) frame_end_step(st, 1, (const volatile float *)(sm + 112));
  }
  res.ut = ut; res.lost = lost;
  return res;
}

}  // namespace svoslam
