// Internal helpers shared by the kernel translation units (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

#include "../../include/cirkit_hip.h"
#include "../../include/cirkit_hip_internal.h"

namespace ck {

using Launch = std::function<hipError_t(hipStream_t)>;

// Records `msg` as the thread's last error and returns `st`.
int fail(ck_status st, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Runs `fn` on `stream` now, or appends it to the program being recorded on this thread.
int dispatch(Launch fn, void* stream);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ck_runtime.hip: the scratch buffer the caller lent to this thread's launches (ck_set_workspace), or {nullptr, 0}
struct Workspace {
  void* ptr;
  int64_t bytes;
};
Workspace workspace();
// ck_runtime.hip: per-launch inputs of a recorded program (ck_program_set_input).  While a program is being recorded on
// this thread, the address of its input cell `index` -- a recorded launch reads the pointer from it at every replay --
// otherwise nullptr.
constexpr int kProgramInputs = 4;
const void* const* program_input_slot(int index);
int num_cus();  // compute units of the current device (cached)

// Folds ride on grid.y (<= 65535): a layer with more folds -- a 256 x 256 image has 65536 leaves -- is launched in chunks
// of folds; `fn(first fold, folds)` issues one chunk with the per-fold pointers advanced by the caller.
constexpr int kMaxFoldsPerLaunch = 65535;
template <class Fn>
inline int chunk_folds(int F, Fn&& fn) {
  for (int f0 = 0; f0 < F; f0 += kMaxFoldsPerLaunch) {
    const int n = F - f0 < kMaxFoldsPerLaunch ? F - f0 : kMaxFoldsPerLaunch;
    if (int st = fn(f0, n)) return st;
  }
  return 0;
}

// ck_sum.hip: the test hook ck_debug_force_generic is on (A/B runs of the specialised kernels against the plain ones)
bool debug_force_generic();

// ck_cp.hip: one dense / CP-T slot with contiguous (F, K, K) weights, K in {32, 64}.
// ck_gemm.hip: dense / CP-T layers with Ki, Ko multiples of 32 and up to 256 contracted inputs.
bool gemm_applies(int H, int Ki, int Ko, int mode);
bool tucker_applies(int H, int Ki, int Ko, int mode);
int tucker_lse(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int B, int Ki, int Ko,
               void* stream, bool logits = false, int contraction = 0);
// logits: w holds logits theta and the weights are softmax(theta) over the last axis, normalised online by the launch;
// contraction: 0 exact fp32, 3 / 6 the bf16-split variants (stream-K launch only)
int sum_lse_gemm(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int Ki,
                 int Ko, int mode, void* stream, int contraction = 0);
int cat_dense(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int K,
              void* stream, int contraction = 0);
int cp_single_slot(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int K,
                   void* stream);

constexpr int kWave = 64;  // gfx950 wavefront

#define CK_REQUIRE(cond, ...)                                     \
  do {                                                            \
    if (!(cond)) return ck::fail(CK_ERR_INVALID, __VA_ARGS__);    \
  } while (0)

// ---- device-side helpers ---------------------------------------------------------------------
// maximum / sum over the 64 lanes of a wave, every lane gets the result, without LDS: DPP inside the 16-lane rows,
// v_permlane16_swap and v_permlane32_swap (gfx950) across them.  (A __shfl_xor compiles to ds_bpermute_b32: an LDS round
// trip per step, six of them in a chain per reduction.)
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x140, 0xF, 0xF, true)));  // row_mirror
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return op(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float wave_max(float v) { return wave_reduce<true>(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce<false>(v); }
// max of a value with its partner in the other 32-lane half of the wave (the two halves of a 32-row tile hold the
// two unit groups of a row).  v_permlane32_swap_b32 (gfx950) exchanges the halves inside the VALU; the __shfl_xor it
// replaces compiled to ds_bpermute_b32, an LDS round trip in the middle of every log-sum-exp step.
__device__ __forceinline__ float xhalf_max(float m) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// A pointer that came out of a descriptor table (LDS, kernel arguments by index) is a GENERIC pointer to the compiler: loads
// through it are FLAT instructions, which count on vmcnt AND lgkmcnt -- every wait behind one becomes "everything in flight,
// memory and LDS" and a prefetch stops overlapping anything.  as_global() says what the host guarantees: device memory.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* as_global(const T* p) {
  return (const __attribute__((address_space(1))) T*)p;
}
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T* as_global(T* p) {
  return (__attribute__((address_space(1))) T*)p;
}
// (float4 is a class in HIP: its copy operators do not take references into an address space -- 16-byte accesses to device
// memory through the compiler's own vector type)
typedef float gvec4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(const float* p) {
  const gvec4 t = *as_global(reinterpret_cast<const gvec4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void gstore4(float* p, float4 v) { *as_global(reinterpret_cast<gvec4*>(p)) = gvec4{v.x, v.y, v.z, v.w}; }
// address-space-qualified pointers for __builtin_amdgcn_global_load_lds (global memory -> LDS without registers)
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
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

// ---- cheaper complex exp / log for the K = 32 tile kernel -----------------------------------------------
// The OCML sincosf / atan2f / log1pf spend most of their ~300 instructions per element on argument ranges
// that cannot occur here: a phase is a sum of H values of atan2 (|im| <= H pi), and the magnitudes are
// sums of at most a few hundred terms <= |W|.  Polynomials fitted for this kernel (max error in fp32:
// sin 1.06 ulp, cos 0.86 ulp on [-pi/4, pi/4]; atan 1.63 ulp on [0, 1]), Cody-Waite reduction by pi/2.
__device__ __forceinline__ void sincos_small(float x, float& s, float& c) {
  if (__builtin_expect(fabsf(x) > 8192.f, 0)) {  // never for phases built as above
    sincosf(x, &s, &c);
    return;
  }
  const float n = rintf(x * 0.63661977236758134f);
  float r = fmaf(-n, 1.57079637050628662109375f, x);
  r = fmaf(-n, -4.371138828673793e-08f, r);
  const float z = r * r;
  float sp = fmaf(z, 2.717982852118439e-06f, -1.983928814297542e-04f);
  sp = fmaf(sp, z, 8.333329111337662e-03f);
  sp = fmaf(sp, z, -1.666666716337204e-01f);
  const float sn = fmaf(sp * z, r, r);
  float cp = fmaf(z, 2.438902811263688e-05f, -1.3886739034205675e-03f);
  cp = fmaf(cp, z, 4.166662320494652e-02f);
  cp = fmaf(cp, z, -0.5f);
  const float cs = fmaf(cp, z, 1.f);
  const int q = static_cast<int>(n);
  const float s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
// exp(d) on v_exp_f32 with the rounding error of d log2(e) put back (exp2(t + lo) = exp2(t) (1 + lo ln 2)): ~1e-7 relative, as
// expf, at a third of its instructions (the complex tile kernels are bound by issuing these: LAB_NOTES R5.2)
__device__ __forceinline__ float exp_comp(float d) {
  constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
  const float t = d * kLog2e;
  const float lo = fmaf(d, kLog2e, -t);
  const float e = __builtin_amdgcn_exp2f(t);
  return fabsf(t) < 1e30f ? fmaf(e, lo * kLn2, e) : e;  // (d = -inf: 0, not the NaN of inf - inf)
}
__device__ __forceinline__ c32 c_exp_shift_tile(c32 z, float m) {
#ifdef CK_CEXP_LIBM
  const float r = expf(z.re - m);
#else
  const float r = exp_comp(z.re - m);
#endif
  float s, c;
  sincos_small(z.im, s, c);
  return {r * c, r * s};
}
__device__ __forceinline__ c32 c_log_shift_tile(c32 z, float m) {
  const float ax = fabsf(z.re), ay = fabsf(z.im);
  const float big = fmaxf(ax, ay), small = fminf(ax, ay);
  if (__builtin_expect(!(big > 0.f) || !(big < 3.0e38f), 0)) return c_log_shift(z, m);  // zeros, infinities, NaN
  const float rc = __builtin_amdgcn_rcpf(big);
  float q = small * rc;
  q = fmaf(fmaf(-q, big, small), rc, q);  // one Newton step: q = small / big to ~1 ulp
  const float t = q * q;
  // log|z| = log big + log(1 + t) / 2   (1 + t in [1, 2]: v_log_f32 is accurate to an absolute 1e-7 there)
  const float lr = 0.69314718055994530942f * fmaf(0.5f, __builtin_amdgcn_logf(1.f + t), __builtin_amdgcn_logf(big)) + m;
  float a = fmaf(t, 2.8786147013306618e-03f, -1.6197500750422478e-02f);
  a = fmaf(a, t, 4.292982071638107e-02f);
  a = fmaf(a, t, -7.527676224708557e-02f);
  a = fmaf(a, t, 1.0653974115848541e-01f);
  a = fmaf(a, t, -1.4207743108272552e-01f);
  a = fmaf(a, t, 1.9993291795253754e-01f);
  a = fmaf(a, t, -3.333311975002289e-01f);
  a = fmaf(a * t, q, q);                               // atan(q), q in [0, 1]
  if (ay > ax) a = 1.57079632679489661923f - a;        // |im| > |re|
  if (z.re < 0.f) a = 3.14159265358979323846f - a;     // left half plane
  return {lr, copysignf(a, z.im)};
}

}  // namespace ck
