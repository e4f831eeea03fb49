// common.hpp -- shared host/device helpers for libsvoslam_hip (gfx950 only).
//
// All device arithmetic is compiled with -ffp-contract=off: every float
// expression is evaluated IEEE operation by operation in the order the
// reference source writes it; fused multiply-adds appear only as explicit
// fmaf()/fma() calls.  That is what makes the device results comparable bit for
// bit with the CPU oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svoslam.h"

namespace svoslam {

typedef long long octkey;  // src/world/svo/svo.cu:22

constexpr uint32_t kFlag = SVOSLAM_FLAG_CHILDREN;
constexpr uint32_t kMask = SVOSLAM_CHILD_MASK;
constexpr int kWave = 64;  // gfx950 wavefront

// ---- error plumbing -------------------------------------------------------
void set_last_error(const char *what, hipError_t e);
void set_last_error_text(const char *fmt, ...) __attribute__((format(printf, 1, 2)));  // a limit of the library, in words
int ensure_device();

// hipMemset of device memory may return before the fill has run, and the fill runs on the NULL stream, which the library's
// non-blocking streams do not wait for: a kernel enqueued right after it on one of them can see the old bytes, or have its
// own stores wiped.  Every one-off initialisation goes through this (fill, then wait for the null stream).
inline hipError_t memset_sync(void *p, int value, size_t bytes) {
  const hipError_t e = hipMemset(p, value, bytes);
  return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

#define SVO_HIP(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) {                            \
      ::svoslam::set_last_error(#expr, _e);            \
      return _e == hipErrorOutOfMemory ? SVOSLAM_ERR_OOM : SVOSLAM_ERR_HIP; \
    }                                                  \
  } while (0)

#define SVO_TRY(expr)               \
  do {                              \
    int _s = (expr);                \
    if (_s != SVOSLAM_OK) return _s; \
  } while (0)

#define SVO_LAUNCH_CHECK() SVO_HIP(hipGetLastError())

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// ---- POD vectors (glm layout) ---------------------------------------------
struct vec3 { float x, y, z; };
struct mat4 { float m[16]; };  // column-major: m[4*col + row]

// ---- device helpers ---------------------------------------------------------
__host__ __device__ inline uint32_t f2bits(float f) {
  union { float f; uint32_t u; } c; c.f = f; return c.u;
}
__host__ __device__ inline bool finitef_(float f) { return (f2bits(f) & 0x7F800000u) != 0x7F800000u; }

// glm operator*(mat4, vec4): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)   (type_mat4x4.inl:651-687)
__host__ __device__ inline void mat4_mul_point(const float *m, float x, float y, float z, float w, float &ox, float &oy,
                                               float &oz) {
  ox = (m[0] * x + m[4] * y) + (m[8] * z + m[12] * w);
  oy = (m[1] * x + m[5] * y) + (m[9] * z + m[13] * w);
  oz = (m[2] * x + m[6] * y) + (m[10] * z + m[14] * w);
}

__host__ __device__ inline float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}

}  // namespace svoslam

// Wavefront issue priority (s_setprio) for the tracker's kernels: its 38 dependent launches per frame are the
// longest chain of the pipeline and share the SIMDs with the raycast's ~20 wavefronts per CU, which is bound
// by its slowest rays anyway (+3 % frames/s); the three commit kernels of the map stream likewise (+2 %).
// Raising the sort / plan / map-generation kernels as well cancels the gain.
#ifdef __HIP_DEVICE_COMPILE__
#define SVO_HIGH_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define SVO_HIGH_PRIO()
#endif

