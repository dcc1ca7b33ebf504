// Internal helpers shared by the kernel translation units (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

#include "../../include/cirkit_hip.h"

namespace ck {

using Launch = std::function<hipError_t(hipStream_t)>;

// Records `msg` as the thread's last error and returns `st`.
int fail(ck_status st, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Runs `fn` on `stream` now, or appends it to the program being recorded on this thread.
int dispatch(Launch fn, void* stream);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ck_cp.hip: one dense / CP-T slot with contiguous (F, K, K) weights, K in {32, 64}.
// ck_gemm.hip: dense / CP-T layers with Ki, Ko multiples of 32 and up to 256 contracted inputs.
bool gemm_applies(int H, int Ki, int Ko, int mode);
int sum_lse_gemm(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int Ki,
                 int Ko, int mode, void* stream);
int cat_dense(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int K,
              void* stream);
int cp_single_slot(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int K,
                   void* stream);

constexpr int kWave = 64;  // gfx950 wavefront

#define CK_REQUIRE(cond, ...)                                     \
  do {                                                            \
    if (!(cond)) return ck::fail(CK_ERR_INVALID, __VA_ARGS__);    \
  } while (0)

// ---- device-side helpers ---------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// torch.clamp(amax, finfo.min, finfo.max) of semiring.py:392-399
__device__ __forceinline__ float clamp_finite(float m) {
  return fminf(fmaxf(m, -3.402823466e+38f), 3.402823466e+38f);
}

struct c32 {  // complex64 as two floats
  float re, im;
};
__device__ __forceinline__ c32 c_add(c32 a, c32 b) { return {a.re + b.re, a.im + b.im}; }
// exp(z - m) with real m
// (accurate expf/logf here, not the fast intrinsics: signed weights make the complex sums
// cancel, which amplifies the relative error of every term)
__device__ __forceinline__ c32 c_exp_shift(c32 z, float m) {
  float r = expf(z.re - m);
  float s, c;
  sincosf(z.im, &s, &c);
  return {r * c, r * s};
}
// complex log (forward of ComplexSafeLog is torch.log, utils.py:32-50) plus a real shift
__device__ __forceinline__ c32 c_log_shift(c32 z, float m) {
  // log|z| = log(hypot): scale-safe for the magnitudes the shifted sums can take
  float ax = fabsf(z.re), ay = fabsf(z.im);
  float big = fmaxf(ax, ay), small = fminf(ax, ay);
  float lr;
  if (big == 0.f) {
    lr = -INFINITY;
  } else {
    float q = small / big;
    lr = logf(big) + 0.5f * log1pf(q * q);
  }
  return {lr + m, atan2f(z.im, z.re)};
}

}  // namespace ck
