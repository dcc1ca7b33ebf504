// Sum-type inner layers fused with the log-sum-exp semiring reduction:
//   TorchSumLayer (dense / general arity), TorchCPTLayer (Hadamard -> dense), mixing layers,
//   TorchTensorDotLayer; real (lse-sum) and complex (complex-lse-sum).
//
// Numerics follow LSESumSemiring.apply_reduce (cirkit/backend/torch/semiring.py:383-408):
//   m = clamp(max_i v_i) over the INPUTS only, y_o = sum_i W[o,i] * exp(v_i - m) in linear space,
//   out_o = log(y_o) + m.
//
// Real-valued implementations:
//   * MFMA paths -- Ki = Ko in {32, 64}: one wavefront owns 32 batch rows of one fold and
//     evaluates the contraction with v_mfma_f32_32x32x2_f32 (exact fp32, an fmaf chain).  The lane
//     layout is chosen so that (a) the row maximum needs a single cross-lane exchange, (b) inputs
//     are read and outputs written as float4, and (c) the OUTPUT register layout equals the INPUT
//     register layout of the next layer (what cross-layer fusion needs).  K = 32: `sum_lse_tile32`
//     below (weights in registers, tiled layouts); K = 64: `cp_lse_kernel` (ck_cp.hip, weights in LDS).
//   * `sum_lse_generic` -- any shape; LDS-staged exp(v - m) rows and W chunks.
#include <algorithm>

#include "ck_internal.h"
#include "ck_tile.h"

namespace {

using ck::c32;

// ------------------------------------------------------------------------------------------------
// MFMA path
// ------------------------------------------------------------------------------------------------
// Lane l of a wave: b = l & 31 (batch row inside the 32-row tile), kh = l >> 5.
// Unit ownership: lane (b, kh) holds, for every 32-unit block q, units 32q + 8g + 4kh + t
// (g, t in 0..3) in register j = 4g + t  -> four float4 per block at float offset 32q + 8g + 4kh.
//
// v_mfma_f32_32x32x2_f32 computes D[i][j] += A[i][k] B[k][j], k in {0,1}, with
//   A: lane l supplies A[i = l&31][k = l>>5],  B: lane l supplies B[k = l>>5][j = l&31],
//   D: lane l holds D[i = (r&3) + 8(r>>2) + 4(l>>5)][j = l&31] in accumulator register r.
// We put the WEIGHTS in A (i = output unit) and the ACTIVATIONS in B (j = batch row): step
// s = 4g+t of block q contracts units {32q+8g+t (kh=0), 32q+8g+4+t (kh=1)} -- exactly what lane
// (., kh) holds in register (q, s) for both operands.  The result D[o][b] lands in lane
// (b, hi) register r with o = 8(r>>2) + 4hi + (r&3): the same ownership as the inputs.

// K = 32 on the shared register tile (ck_tile.h): supports the tiled weight layouts and the
// split-precision contraction; each wave keeps the fold's weights in registers across its tiles.
template <int LAYOUT>
__global__ void __launch_bounds__(256)
    sum_lse_tile32(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                   const float* __restrict__ w, float* __restrict__ out, int H, int B,
                   int tiles_per_wave) {
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  WRegs wr;
  load_w<LAYOUT>(w + static_cast<int64_t>(f) * kK * kK, lane, wr);
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const int b0 = (tile0 + tt) * 32;
    if (b0 >= B) break;
    const int b = b0 + b_in;
    const bool live = b < B;
    const int bl = live ? b : B - 1;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = 0.f;
    for (int h = 0; h < H; ++h) tile_load_add(arena + ro[h] + static_cast<int64_t>(bl) * kK + 4 * kh, v);
    sum_step<LAYOUT>(wr, v);
    if (live) tile_store(out + (static_cast<int64_t>(f) * B + b) * kK + 4 * kh, v);
  }
}

// Few outputs over 32 or 64 product-type inputs (the scalar sum folds at the top of a circuit: TorchCPTLayer with one output
// unit, optimized.py:171-178): a HALF-wave per batch row when Ki = 32, a wave when Ki = 64, lane = input unit; the row's maximum
// and every output's dot product are lane reductions, no LDS, no staging.  (These folds used to take the shape-generic kernel:
// 12 us per launch at BASELINE config 4 for 4 folds x 4096 rows.)
template <int KI>
__global__ void __launch_bounds__(256)
    sum_lse_few_outputs_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off, const float* __restrict__ w,
                               float* __restrict__ out, int H, int B, int Ko) {
  constexpr int kRowsPerWave = 64 / KI;
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane % KI, sub = lane / KI;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  for (int b0 = (blockIdx.x * 4 + wave) * kRowsPerWave; b0 < B; b0 += gridDim.x * 4 * kRowsPerWave) {
    const int b = b0 + sub;
    const bool live = b < B;
    const int64_t bl = live ? b : B - 1;
    float v = 0.f;
    for (int h = 0; h < H; ++h) v += arena[ro[h] + bl * KI + n];
    float m = v;
#pragma unroll
    for (int s2 = 1; s2 < KI; s2 <<= 1) m = fmaxf(m, __shfl_xor(m, s2, 64));
    m = ck::clamp_finite(m);
    const float e = __expf(v - m);
    for (int o = 0; o < Ko; ++o) {
      float y = w[(static_cast<int64_t>(f) * Ko + o) * KI + n] * e;
#pragma unroll
      for (int s2 = 1; s2 < KI; s2 <<= 1) y += __shfl_xor(y, s2, 64);
      if (live && n == 0) out[(static_cast<int64_t>(f) * B + b) * Ko + o] = __logf(y) + m;
    }
  }
}

// complex-lse-sum, K = 32, real weights (ComplexLSESumSemiring.apply_reduce, semiring.py:441-476, with
// `cast(weight)` real -> complex, :416-422): exp(z - m) = E_re + i E_im is split into two real 32 x 32
// register tiles, each goes through the SAME fp32 MFMA chain as the real layer (W . E_re, W . E_im),
// and the complex logarithm recombines them.  All 64 lanes work in every phase (the shape-generic
// kernel keeps half of them idle at K = 32), transcendental functions as in ck_internal.h.
__global__ void __launch_bounds__(256)
    sum_clse_tile32(const c32* __restrict__ arena, const int64_t* __restrict__ row_off,
                    const float* __restrict__ w, c32* __restrict__ out, int H, int B, int tiles_per_wave,
                    const float* __restrict__ table, const int32_t* __restrict__ child_fold,
                    const int32_t* __restrict__ child_var, const int32_t* __restrict__ xt, int C) {
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  WRegs wr;
  load_w<CK_W_ROWMAJOR>(w + static_cast<int64_t>(f) * kK * kK, lane, wr);
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const int b0 = (tile0 + tt) * 32;
    if (b0 >= B) break;
    const int b = b0 + b_in;
    const bool live = b < B;
    const int bl = live ? b : B - 1;
    float zr[16], zi[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) zr[j] = zi[j] = 0.f;
    for (int h = 0; h < H; ++h) {  // product of the children: complex addition in log space
      if (table != nullptr) {
        // the children are Embedding folds: rows of the REAL weight table (F0, C+1, 32) gathered by the batch
        // values and mapped to the complex-log semiring on the fly, z = (log|w|, pi if w < 0)
        // (TorchEmbeddingLayer.forward input.py:258-266, csafelog utils.py:32-50) -- as ck_embedding_clog_fwd
        const int64_t e = static_cast<int64_t>(f) * H + h;
        const int xv = xt[static_cast<int64_t>(child_var[e]) * B + bl];
        const int c = xv < 0 ? C : min(xv, C - 1);
        const float* src = table + (static_cast<int64_t>(child_fold[e]) * (C + 1) + c) * kK + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(src + 8 * g);
          const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            zr[4 * g + t] += __builtin_amdgcn_logf(fabsf(wv[t])) * kLN2;
            zi[4 * g + t] += wv[t] < 0.f ? 3.14159265358979323846f : 0.f;
          }
        }
        continue;
      }
      const float* src = reinterpret_cast<const float*>(arena + ro[h] + static_cast<int64_t>(bl) * kK + 4 * kh);
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // units 8g + 4kh + t: four (re, im) pairs = two float4
        const float4 a4 = *reinterpret_cast<const float4*>(src + 16 * g);
        const float4 b4 = *reinterpret_cast<const float4*>(src + 16 * g + 4);
        zr[4 * g + 0] += a4.x; zi[4 * g + 0] += a4.y;
        zr[4 * g + 1] += a4.z; zi[4 * g + 1] += a4.w;
        zr[4 * g + 2] += b4.x; zi[4 * g + 2] += b4.y;
        zr[4 * g + 3] += b4.z; zi[4 * g + 3] += b4.w;
      }
    }
    const float m = row_max16(zr);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const c32 e = ck::c_exp_shift_tile(c32{zr[j], zi[j]}, m);
      zr[j] = e.re;
      zi[j] = e.im;
    }
    f32x16 yr, yi;
#pragma unroll
    for (int r = 0; r < 16; ++r) yr[r] = yi[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float wq[4] = {wr.q[g].x, wr.q[g].y, wr.q[g].z, wr.q[g].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        yr = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[t], zr[4 * g + t], yr, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[t], zi[4 * g + t], yi, 0, 0, 0);
      }
    }
    if (live) {
      float* dst = reinterpret_cast<float*>(out + (static_cast<int64_t>(f) * B + b) * kK + 4 * kh);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const c32 o0 = ck::c_log_shift_tile(c32{yr[4 * g + 0], yi[4 * g + 0]}, m);
        const c32 o1 = ck::c_log_shift_tile(c32{yr[4 * g + 1], yi[4 * g + 1]}, m);
        const c32 o2 = ck::c_log_shift_tile(c32{yr[4 * g + 2], yi[4 * g + 2]}, m);
        const c32 o3 = ck::c_log_shift_tile(c32{yr[4 * g + 3], yi[4 * g + 3]}, m);
        *reinterpret_cast<float4*>(dst + 16 * g) = make_float4(o0.re, o0.im, o1.re, o1.im);
        *reinterpret_cast<float4*>(dst + 16 * g + 4) = make_float4(o2.re, o2.im, o3.re, o3.im);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Generic path (any H, Ki, Ko; real or complex activations; real or complex weights)
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Num;
template <>
struct Num<float> {
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ float re(float a) { return a; }
  static __device__ __forceinline__ float add(float a, float b) { return a + b; }
  static __device__ __forceinline__ float mul(float a, float b) { return a * b; }
  static __device__ __forceinline__ float exp_shift(float a, float m) { return __expf(a - m); }
  static __device__ __forceinline__ float log_shift(float a, float m) { return __logf(a) + m; }
};
template <>
struct Num<c32> {
  static __device__ __forceinline__ c32 zero() { return {0.f, 0.f}; }
  static __device__ __forceinline__ float re(c32 a) { return a.re; }
  static __device__ __forceinline__ c32 add(c32 a, c32 b) { return ck::c_add(a, b); }
  static __device__ __forceinline__ c32 mul(c32 a, c32 b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
  }
  static __device__ __forceinline__ c32 exp_shift(c32 a, float m) { return ck::c_exp_shift(a, m); }
  static __device__ __forceinline__ c32 log_shift(c32 a, float m) { return ck::c_log_shift(a, m); }
};
__device__ __forceinline__ float fma_w(float w, float e, float acc) { return fmaf(w, e, acc); }
__device__ __forceinline__ c32 fma_w(float w, c32 e, c32 acc) {
  return {fmaf(w, e.re, acc.re), fmaf(w, e.im, acc.im)};
}
__device__ __forceinline__ c32 fma_w(c32 w, c32 e, c32 acc) {
  return {fmaf(w.re, e.re, fmaf(-w.im, e.im, acc.re)), fmaf(w.re, e.im, fmaf(w.im, e.re, acc.im))};
}

constexpr int kGenNC = 32;  // W chunk width staged in LDS
constexpr int kGenOT = 64;  // output units per pass

// AT: activation type (float | c32), WT: weight type (float | c32).  RPT: rows per thread.
// Block = 256 threads = 64 output lanes x 4 row groups; tile = 4*RPT batch rows of one fold.
template <typename AT, typename WT, int RPT>
__global__ void __launch_bounds__(256)
    sum_lse_generic(const AT* __restrict__ arena, const int64_t* __restrict__ row_off,
                    const WT* __restrict__ w, AT* __restrict__ out, int H, int B, int Ki, int Ko,
                    int mode, int N) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TB = 4 * RPT;  // N: Ki (product), H*Ki (concatenation) or Ki**H (Kronecker)
  AT* e_s = reinterpret_cast<AT*>(smem);                          // [TB][N]
  WT* w_s = reinterpret_cast<WT*>(e_s + static_cast<size_t>(TB) * N);  // [kGenOT][kGenNC+1]
  float* m_s = reinterpret_cast<float*>(w_s + kGenOT * (kGenNC + 1));  // [TB]

  const int f = blockIdx.y;
  const int b0 = blockIdx.x * TB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;

  // phase A: gather v (cat or product of children), row maximum, e = exp(v - m) into LDS
  if (mode == CK_SUM_KRON) {
    // Tucker (optimized.py:89-103): one maximum PER INPUT; e[(i_0, .., i_{H-1})] = prod_h exp(x_h[i_h]-m_h)
    // with i_0 the slowest index.  The outer product is formed in LDS and never reaches memory.
    for (int r = wave; r < TB; r += 4) {
      const int b = min(b0 + r, B - 1);
      AT* er = e_s + static_cast<size_t>(r) * N;
      AT* xr = reinterpret_cast<AT*>(m_s + TB) + static_cast<size_t>(r) * H * Ki;  // [H][Ki] shifted inputs
      float msum = 0.f;
      for (int h = 0; h < H; ++h) {
        float mx = -INFINITY;
        for (int k = lane; k < Ki; k += 64) mx = fmaxf(mx, Num<AT>::re(arena[ro[h] + static_cast<int64_t>(b) * Ki + k]));
        mx = ck::clamp_finite(ck::wave_max(mx));
        for (int k = lane; k < Ki; k += 64)
          xr[h * Ki + k] = Num<AT>::exp_shift(arena[ro[h] + static_cast<int64_t>(b) * Ki + k], mx);
        msum += mx;
      }
      __builtin_amdgcn_wave_barrier();
      for (int n = lane; n < N; n += 64) {
        int rem = n;
        AT v = xr[(H - 1) * Ki + rem % Ki];
        rem /= Ki;
        for (int h = H - 2; h >= 0; --h) {
          v = Num<AT>::mul(xr[h * Ki + rem % Ki], v);
          rem /= Ki;
        }
        er[n] = v;
      }
      if (lane == 0) m_s[r] = msum;
    }
  } else
  for (int r = wave; r < TB; r += 4) {
    const int b = min(b0 + r, B - 1);
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) {
      AT v;
      if (mode == CK_SUM_PROD) {
        v = arena[ro[0] + static_cast<int64_t>(b) * Ki + n];
        for (int h = 1; h < H; ++h) v = Num<AT>::add(v, arena[ro[h] + static_cast<int64_t>(b) * Ki + n]);
      } else {
        const int h = n / Ki, k = n - h * Ki;
        v = arena[ro[h] + static_cast<int64_t>(b) * Ki + k];
      }
      e_s[static_cast<size_t>(r) * N + n] = v;
      mx = fmaxf(mx, Num<AT>::re(v));
    }
    mx = ck::clamp_finite(ck::wave_max(mx));
    for (int n = lane; n < N; n += 64)
      e_s[static_cast<size_t>(r) * N + n] = Num<AT>::exp_shift(e_s[static_cast<size_t>(r) * N + n], mx);
    if (lane == 0) m_s[r] = mx;
  }
  __syncthreads();

  // phase B: y[r][o] = sum_n W[o][n] e[r][n]
  const int o_l = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const WT* wf = w + static_cast<int64_t>(f) * Ko * N;
  for (int o_base = 0; o_base < Ko; o_base += kGenOT) {
    AT acc[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) acc[j] = Num<AT>::zero();
    for (int n0 = 0; n0 < N; n0 += kGenNC) {
      // stage W[o_base .. +64][n0 .. +32] (coalesced along n)
      for (int i = threadIdx.x; i < kGenOT * kGenNC; i += 256) {
        const int oo = i / kGenNC, nn = i - oo * kGenNC;
        WT val{};
        if (o_base + oo < Ko && n0 + nn < N) val = wf[static_cast<int64_t>(o_base + oo) * N + n0 + nn];
        w_s[oo * (kGenNC + 1) + nn] = val;
      }
      __syncthreads();
      const int nmax = min(kGenNC, N - n0);
      for (int nn = 0; nn < nmax; ++nn) {
        const WT wv = w_s[o_l * (kGenNC + 1) + nn];
#pragma unroll
        for (int j = 0; j < RPT; ++j)
          acc[j] = fma_w(wv, e_s[static_cast<size_t>(rg * RPT + j) * N + n0 + nn], acc[j]);
      }
      __syncthreads();
    }
    if (o_base + o_l < Ko) {
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const int r = rg * RPT + j, b = b0 + r;
        if (b < B)
          out[(static_cast<int64_t>(f) * B + b) * Ko + o_base + o_l] = Num<AT>::log_shift(acc[j], m_s[r]);
      }
    }
  }
}

template <typename AT, typename WT>
int launch_generic(const AT* arena, const int64_t* row_off, const WT* w, AT* out, int F, int H,
                   int B, int Ki, int Ko, int mode, void* stream) {
  int64_t n64 = mode == CK_SUM_PROD ? Ki : static_cast<int64_t>(H) * Ki;
  if (mode == CK_SUM_KRON) {
    n64 = 1;
    for (int h = 0; h < H && n64 <= (1 << 20); ++h) n64 *= Ki;
    if (n64 > (1 << 20))
      return ck::fail(CK_ERR_UNSUPPORTED, "ck_sum_lse_fwd: Kronecker product of %d inputs with %d units is too large", H, Ki);
  }
  const int N = static_cast<int>(n64);
  auto lds_bytes = [&](int tb) {
    return static_cast<size_t>(tb) * N * sizeof(AT) + kGenOT * (kGenNC + 1) * sizeof(WT) + tb * sizeof(float) +
           (mode == CK_SUM_KRON ? static_cast<size_t>(tb) * H * Ki * sizeof(AT) : 0);
  };
  int rpt = 4;
  while (rpt > 1 && lds_bytes(4 * rpt) > 64 * 1024) rpt >>= 1;
  const size_t lds = lds_bytes(4 * rpt);
  if (lds > 160 * 1024)
    return ck::fail(CK_ERR_UNSUPPORTED, "ck_sum_lse_fwd: N=%d inputs per row do not fit in LDS", N);
  const int tb = 4 * rpt;
  dim3 grid((B + tb - 1) / tb, F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern) {
          if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
          }
          hipLaunchKernelGGL(kern, grid, block, lds, s, arena, row_off, w, out, H, B, Ki, Ko, mode, N);
          return hipGetLastError();
        };
        if (rpt == 4) return go(sum_lse_generic<AT, WT, 4>);
        if (rpt == 2) return go(sum_lse_generic<AT, WT, 2>);
        return go(sum_lse_generic<AT, WT, 1>);
      },
      stream);
}

int check_sum_args(const void* arena, const void* row_off, const void* w, const void* out, int F,
                   int H, int B, int Ki, int Ko, int mode, const char* who) {
  CK_REQUIRE(arena && row_off && w && out, "%s: null pointer", who);
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && Ki > 0 && Ko > 0, "%s: non-positive size F=%d H=%d B=%d Ki=%d Ko=%d",
             who, F, H, B, Ki, Ko);
  CK_REQUIRE(mode == CK_SUM_CAT || mode == CK_SUM_PROD || mode == CK_SUM_KRON, "%s: unknown mode %d", who, mode);
  CK_REQUIRE(mode != CK_SUM_KRON || H >= 2, "%s: CK_SUM_KRON (Tucker) needs arity >= 2, found %d", who, H);
  return CK_OK;
}

// ------------------------------------------------------------------------------------------------
// Mixing layer: out[k] = log(sum_h mw[k,h] exp(x[h,k] - m)) + m, m over all (h,k)
// ------------------------------------------------------------------------------------------------
// One wave per batch row; lanes stride over k.  Two passes over the H*K inputs (the second one
// hits L1/L2).
__global__ void __launch_bounds__(256)
    mixing_lse_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                      const float* __restrict__ mw, float* __restrict__ out, int H, int B, int K,
                      int rows_per_block) {
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* mwf = mw + static_cast<int64_t>(f) * K * H;
  const int b_begin = blockIdx.x * rows_per_block;
  const int b_end = min(B, b_begin + rows_per_block);
  for (int b = b_begin + wave; b < b_end; b += 4) {
    float mx = -INFINITY;
    for (int h = 0; h < H; ++h) {
      const float* src = arena + ro[h] + static_cast<int64_t>(b) * K;
      for (int k = lane; k < K; k += 64) mx = fmaxf(mx, src[k]);
    }
    mx = ck::clamp_finite(ck::wave_max(mx));
    for (int k = lane; k < K; k += 64) {
      float acc = 0.f;
      for (int h = 0; h < H; ++h)
        acc = fmaf(mwf[static_cast<int64_t>(k) * H + h], __expf(arena[ro[h] + static_cast<int64_t>(b) * K + k] - mx), acc);
      out[(static_cast<int64_t>(f) * B + b) * K + k] = __logf(acc) + mx;
    }
  }
}

// Few units (the scalar root of a circuit with mixing weights: K = 1): one THREAD per (row, unit), every input of the
// element requested before the first use.  (The wave-per-row kernel above keeps one lane of 64 busy and walks its rows
// one after the other: 30 us for a 4096 x 12 x 1 layer.)
template <int HMAX>
__global__ void __launch_bounds__(256)
    mixing_lse_small(const float* __restrict__ arena, const int64_t* __restrict__ row_off, const float* __restrict__ mw,
                     float* __restrict__ out, int H, int B, int K) {
  const int f = blockIdx.y;
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;  // element (b, k) of the fold
  if (i >= static_cast<int64_t>(B) * K) return;
  const int b = static_cast<int>(i / K), k = static_cast<int>(i - static_cast<int64_t>(b) * K);
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* mwf = mw + static_cast<int64_t>(f) * K * H;
  // the maximum is over ALL (h, k) of the row (one max per row, as the reference's amax over the reduced axis)
  float x[HMAX];
  float mx = -INFINITY;
  for (int kk = 0; kk < K; ++kk) {
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
      const float v = h < H ? arena[ro[h] + static_cast<int64_t>(b) * K + kk] : -INFINITY;
      if (kk == k) x[h] = v;
      mx = fmaxf(mx, v);
    }
  }
  mx = ck::clamp_finite(mx);
  float acc = 0.f;
#pragma unroll
  for (int h = 0; h < HMAX; ++h)
    if (h < H) acc = fmaf(mwf[static_cast<int64_t>(k) * H + h], __expf(x[h] - mx), acc);
  out[(static_cast<int64_t>(f) * B + b) * K + k] = __logf(acc) + mx;
}

// float4 variant: K/4 lanes per row (K/4 a power of two <= 64), 64/(K/4) rows per wave pass.
// All H inputs of a row are loaded ONCE into registers (H <= HMAX; every load of a lane is in flight
// before the first use) and serve both the maximum and the weighted sum; the (K, H) coefficients are
// transposed into LDS so that a lane reads its four per input as one 16-byte word.  HMAX = 0: any H,
// two passes over the inputs (the second one hits L1/L2).
template <int HMAX>
__global__ void __launch_bounds__(256)
    mixing_lse_vec(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                   const float* __restrict__ mw, float* __restrict__ out, int H, int B, int K,
                   int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float mw_s[];  // [H][K]
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = K >> 2, rpw = 64 / lpr;
  const int r_in = lane / lpr, q = lane - r_in * lpr;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* mwf = mw + static_cast<int64_t>(f) * K * H;
  for (int i = threadIdx.x; i < K * H; i += blockDim.x) {
    const int k = i / H, h = i - k * H;
    mw_s[h * K + k] = mwf[i];
  }
  __syncthreads();
  const int b_begin = blockIdx.x * rows_per_block;
  const int b_end = min(B, b_begin + rows_per_block);
  for (int b0 = b_begin + wave * rpw; b0 < b_end; b0 += 4 * rpw) {
    const int b = min(b0 + r_in, B - 1);
    const bool live = b0 + r_in < b_end;
    float mx = -INFINITY;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (HMAX > 0) {
      float4 v[HMAX];
#pragma unroll
      for (int h = 0; h < HMAX; ++h)
        if (h < H) v[h] = reinterpret_cast<const float4*>(arena + ro[h] + static_cast<int64_t>(b) * K)[q];
#pragma unroll
      for (int h = 0; h < HMAX; ++h)
        if (h < H) mx = fmaxf(mx, fmaxf(fmaxf(v[h].x, v[h].y), fmaxf(v[h].z, v[h].w)));
      for (int o = lpr >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      mx = ck::clamp_finite(mx);
      const float nml = exp_offset(mx, 0.f);
#pragma unroll
      for (int h = 0; h < HMAX; ++h)
        if (h < H) {
          const float4 w4 = *reinterpret_cast<const float4*>(mw_s + h * K + 4 * q);
          acc.x = fmaf(w4.x, __builtin_amdgcn_exp2f(fmaf(v[h].x, kL2E, nml)), acc.x);
          acc.y = fmaf(w4.y, __builtin_amdgcn_exp2f(fmaf(v[h].y, kL2E, nml)), acc.y);
          acc.z = fmaf(w4.z, __builtin_amdgcn_exp2f(fmaf(v[h].z, kL2E, nml)), acc.z);
          acc.w = fmaf(w4.w, __builtin_amdgcn_exp2f(fmaf(v[h].w, kL2E, nml)), acc.w);
        }
    } else {
      for (int h = 0; h < H; ++h) {
        const float4 v = reinterpret_cast<const float4*>(arena + ro[h] + static_cast<int64_t>(b) * K)[q];
        mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
      }
      for (int o = lpr >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      mx = ck::clamp_finite(mx);
      const float nml = exp_offset(mx, 0.f);
      for (int h = 0; h < H; ++h) {
        const float4 v = reinterpret_cast<const float4*>(arena + ro[h] + static_cast<int64_t>(b) * K)[q];
        const float4 w4 = *reinterpret_cast<const float4*>(mw_s + h * K + 4 * q);
        acc.x = fmaf(w4.x, __builtin_amdgcn_exp2f(fmaf(v.x, kL2E, nml)), acc.x);
        acc.y = fmaf(w4.y, __builtin_amdgcn_exp2f(fmaf(v.y, kL2E, nml)), acc.y);
        acc.z = fmaf(w4.z, __builtin_amdgcn_exp2f(fmaf(v.z, kL2E, nml)), acc.z);
        acc.w = fmaf(w4.w, __builtin_amdgcn_exp2f(fmaf(v.w, kL2E, nml)), acc.w);
      }
    }
    if (live) {
      float4 o4;
      o4.x = fmaf(__builtin_amdgcn_logf(acc.x), kLN2, mx);
      o4.y = fmaf(__builtin_amdgcn_logf(acc.y), kLN2, mx);
      o4.z = fmaf(__builtin_amdgcn_logf(acc.z), kLN2, mx);
      o4.w = fmaf(__builtin_amdgcn_logf(acc.w), kLN2, mx);
      reinterpret_cast<float4*>(out + (static_cast<int64_t>(f) * B + b) * K)[q] = o4;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TensorDot: x (B, Kj*Kq) viewed (Kj, Kq); for each q: m_q = max_j x[j,q];
//   out[q*Kk + k] = log(sum_j W[k,j] exp(x[j,q] - m_q)) + m_q
// ------------------------------------------------------------------------------------------------
// One workgroup per (fold, batch row): e[q][j] staged in LDS (transposed), W[k][j] in LDS;
// thread t -> outputs (q, k) strided.  Kj, Kk are small (the rank K of the squared circuit).
template <typename AT, typename WT>
__global__ void __launch_bounds__(256)
    tensordot_lse_kernel(const AT* __restrict__ arena, const int64_t* __restrict__ row_off,
                         const WT* __restrict__ w, AT* __restrict__ out, int B, int Kj, int Kq,
                         int Kk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AT* e_s = reinterpret_cast<AT*>(smem);                                 // [Kq][Kj+1]
  WT* w_s = reinterpret_cast<WT*>(e_s + static_cast<size_t>(Kq) * (Kj + 1));  // [Kk][Kj+1]
  float* m_s = reinterpret_cast<float*>(w_s + static_cast<size_t>(Kk) * (Kj + 1));  // [Kq]
  const int f = blockIdx.y, b = blockIdx.x;
  const AT* src = arena + row_off[f] + static_cast<int64_t>(b) * Kj * Kq;
  const WT* wf = w + static_cast<int64_t>(f) * Kk * Kj;
  for (int i = threadIdx.x; i < Kk * Kj; i += blockDim.x) {
    const int k = i / Kj, j = i - k * Kj;
    w_s[k * (Kj + 1) + j] = wf[i];
  }
  for (int i = threadIdx.x; i < Kj * Kq; i += blockDim.x) {  // coalesced read of x[j][q]
    const int j = i / Kq, q = i - j * Kq;
    e_s[q * (Kj + 1) + j] = src[i];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < Kq; q += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < Kj; ++j) mx = fmaxf(mx, Num<AT>::re(e_s[q * (Kj + 1) + j]));
    m_s[q] = ck::clamp_finite(mx);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Kj * Kq; i += blockDim.x) {
    const int q = i / Kj, j = i - q * Kj;
    e_s[q * (Kj + 1) + j] = Num<AT>::exp_shift(e_s[q * (Kj + 1) + j], m_s[q]);
  }
  __syncthreads();
  AT* dst = out + (static_cast<int64_t>(f) * B + b) * Kq * Kk;
  for (int i = threadIdx.x; i < Kq * Kk; i += blockDim.x) {
    const int q = i / Kk, k = i - q * Kk;
    AT acc = Num<AT>::zero();
    for (int j = 0; j < Kj; ++j) acc = fma_w(w_s[k * (Kj + 1) + j], e_s[q * (Kj + 1) + j], acc);
    dst[i] = Num<AT>::log_shift(acc, m_s[q]);
  }
}

template <typename AT, typename WT>
int launch_tensordot(const AT* arena, const int64_t* row_off, const WT* w, AT* out, int F, int B,
                     int Kj, int Kq, int Kk, void* stream, const char* who) {
  CK_REQUIRE(arena && row_off && w && out, "%s: null pointer", who);
  CK_REQUIRE(F > 0 && B > 0 && Kj > 0 && Kq > 0 && Kk > 0, "%s: non-positive size", who);
  CK_REQUIRE(F <= 65535, "%s: F=%d exceeds grid.y", who, F);
  const size_t lds = static_cast<size_t>(Kq) * (Kj + 1) * sizeof(AT) + static_cast<size_t>(Kk) * (Kj + 1) * sizeof(WT) +
                     static_cast<size_t>(Kq) * sizeof(float);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "%s: Kj*Kq=%d does not fit in LDS", who, Kj * Kq);
  dim3 grid(B, F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        auto kern = tensordot_lse_kernel<AT, WT>;
        if (lds > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
          if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, block, lds, s, arena, row_off, w, out, B, Kj, Kq, Kk);
        return hipGetLastError();
      },
      stream);
}

bool g_force_generic = false;

}  // namespace

namespace ck {
bool debug_force_generic() { return g_force_generic; }
}  // namespace ck

extern "C" {

// Test hook: route every ck_sum_lse_fwd call through the generic kernel (A/B the MFMA path).
int ck_debug_force_generic(int on) {
  g_force_generic = on != 0;
  return CK_OK;
}

int ck_sum_lse_fwd(const float* arena, const int64_t* row_off, const float* w, float* out, int F,
                   int H, int B, int Ki, int Ko, int mode, int w_layout, void* stream) {
  return ck_sum_lse_fwd_v(arena, row_off, w, out, F, H, B, Ki, Ko, mode, w_layout, 0, stream);
}

int ck_sum_lse_fwd_v(const float* arena, const int64_t* row_off, const float* w, float* out, int F,
                     int H, int B, int Ki, int Ko, int mode, int w_layout, int contraction, void* stream) {
  if (int st = check_sum_args(arena, row_off, w, out, F, H, B, Ki, Ko, mode, "ck_sum_lse_fwd")) return st;
  CK_REQUIRE(w_layout == CK_W_ROWMAJOR || w_layout == CK_W_TILED_F32, "ck_sum_lse_fwd: unknown w_layout %d", w_layout);
  CK_REQUIRE(contraction == 0 || contraction == 3 || contraction == 6, "ck_sum_lse_fwd_v: contraction %d (0, 3 or 6)", contraction);
  if (F > ck::kMaxFoldsPerLaunch) {  // (words of weights per fold: Ko x the contracted inputs, whatever the layout)
    int64_t nin = Ki;
    if (mode == CK_SUM_CAT) nin = static_cast<int64_t>(H) * Ki;
    if (mode == CK_SUM_KRON) for (int h = 1; h < H; ++h) nin *= Ki;
    return ck::chunk_folds(F, [&](int f0, int n) {
      return ck_sum_lse_fwd_v(arena, row_off + static_cast<int64_t>(f0) * H, w + static_cast<int64_t>(f0) * Ko * nin, out + static_cast<int64_t>(f0) * B * Ko, n,
                              H, B, Ki, Ko, mode, w_layout, contraction, stream);
    });
  }
  const bool prod_like = mode == CK_SUM_PROD || H == 1;
  if (w_layout != CK_W_ROWMAJOR) {
    CK_REQUIRE(prod_like && Ki == kK && Ko == kK, "ck_sum_lse_fwd: tiled weight layouts need Ki = Ko = 32 and a product-type input");
    CK_REQUIRE(ck::aligned16(arena) && ck::aligned16(w) && ck::aligned16(out), "ck_sum_lse_fwd: buffers must be 16-byte aligned");
  }
  if (w_layout != CK_W_ROWMAJOR || (!g_force_generic && prod_like && Ki == kK && Ko == kK && ck::aligned16(arena) &&
                                    ck::aligned16(w) && ck::aligned16(out))) {
    const int tiles = (B + 31) / 32;
    int tpw = 1;
    while (tpw < 4 && static_cast<int64_t>(F) * ((tiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) >= 2048) tpw *= 2;
    dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          if (w_layout == CK_W_ROWMAJOR)
            hipLaunchKernelGGL(sum_lse_tile32<CK_W_ROWMAJOR>, grid, block, 0, s, arena, row_off, w, out, H, B, tpw);
          else
            hipLaunchKernelGGL(sum_lse_tile32<CK_W_TILED_F32>, grid, block, 0, s, arena, row_off, w, out, H, B, tpw);
          return hipGetLastError();
        },
        stream);
  }
  const bool mfma_ok = !g_force_generic && prod_like && Ki == Ko && Ki == 64 &&
                       ck::aligned16(arena) && ck::aligned16(w) && ck::aligned16(out);
  if (mfma_ok) {
    return ck::cp_single_slot(arena, row_off, w, out, F, H, B, Ki, stream);  // ck_cp.hip
  }
  if (!g_force_generic && prod_like && Ko <= 4 && (Ki == 32 || Ki == 64)) {
    const int rows_per_block = 4 * (64 / Ki);
    dim3 grid(static_cast<unsigned>(std::min((B + rows_per_block - 1) / rows_per_block, 1024)), F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          if (Ki == 32) hipLaunchKernelGGL(sum_lse_few_outputs_kernel<32>, grid, block, 0, s, arena, row_off, w, out, H, B, Ko);
          else hipLaunchKernelGGL(sum_lse_few_outputs_kernel<64>, grid, block, 0, s, arena, row_off, w, out, H, B, Ko);
          return hipGetLastError();
        },
        stream);
  }
  if (!g_force_generic && mode == CK_SUM_CAT && H > 1 && Ki == Ko && (Ki == 32 || Ki == 64) && ck::aligned16(arena) &&
      ck::aligned16(w) && ck::aligned16(out))
    return ck::cat_dense(arena, row_off, w, out, F, H, B, Ki, stream, contraction);  // ck_cp.hip
  if (!g_force_generic && ck::gemm_applies(H, Ki, Ko, mode) && ck::aligned16(arena) && ck::aligned16(w) && ck::aligned16(out))
    return ck::sum_lse_gemm(arena, row_off, w, out, F, H, B, Ki, Ko, mode, stream, contraction);  // ck_gemm.hip
  if (!g_force_generic && ck::tucker_applies(H, Ki, Ko, mode) && ck::aligned16(arena) && ck::aligned16(w) && ck::aligned16(out)) {
    if (contraction != 0) {  // (the variants exist in the stream-K launch: where it does not apply, exact fp32)
      const int st = ck::tucker_lse(arena, row_off, w, out, F, B, Ki, Ko, stream, false, contraction);
      if (st != CK_ERR_UNSUPPORTED) return st;
    }
    return ck::tucker_lse(arena, row_off, w, out, F, B, Ki, Ko, stream);  // ck_gemm.hip
  }
  return launch_generic<float, float>(arena, row_off, w, out, F, H, B, Ki, Ko, mode, stream);
}

int ck_tucker_fwd(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int B, int Ki, int Ko,
                  int w_is_logits, int contraction, void* stream) {
  CK_REQUIRE(arena && row_off && w && out, "ck_tucker_fwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && Ko > 0, "ck_tucker_fwd: non-positive size F=%d B=%d Ko=%d", F, B, Ko);
  CK_REQUIRE(Ki == 32 || Ki == 64, "ck_tucker_fwd: Ki must be 32 or 64, found %d", Ki);
  CK_REQUIRE(contraction == 0 || contraction == 3 || contraction == 6, "ck_tucker_fwd: contraction %d (0, 3 or 6)", contraction);
  CK_REQUIRE(ck::aligned16(arena) && ck::aligned16(w) && ck::aligned16(out), "ck_tucker_fwd: buffers must be 16-byte aligned");
  if (F > ck::kMaxFoldsPerLaunch)
    return ck::chunk_folds(F, [&](int f0, int n) {
      return ck_tucker_fwd(arena, row_off + static_cast<int64_t>(f0) * 2, w + static_cast<int64_t>(f0) * Ko * Ki * Ki,
                           out + static_cast<int64_t>(f0) * B * Ko, n, B, Ki, Ko, w_is_logits, contraction, stream);
    });
  return ck::tucker_lse(arena, row_off, w, out, F, B, Ki, Ko, stream, w_is_logits != 0, contraction);
}

int ck_tucker_logits_fwd(const float* arena, const int64_t* row_off, const float* theta, float* out, int F, int B, int Ki,
                         int Ko, void* stream) {
  return ck_tucker_fwd(arena, row_off, theta, out, F, B, Ki, Ko, 1, 0, stream);
}

int ck_sum_lse_fwd_c(const float* arena_c, const int64_t* row_off, const float* w, float* out_c,
                     int F, int H, int B, int Ki, int Ko, int mode, int w_is_complex, void* stream) {
  if (int st = check_sum_args(arena_c, row_off, w, out_c, F, H, B, Ki, Ko, mode, "ck_sum_lse_fwd_c")) return st;
  if (F > ck::kMaxFoldsPerLaunch) {
    int64_t nin = Ki;
    if (mode == CK_SUM_CAT) nin = static_cast<int64_t>(H) * Ki;
    if (mode == CK_SUM_KRON) for (int h = 1; h < H; ++h) nin *= Ki;
    return ck::chunk_folds(F, [&](int f0, int n) {
      return ck_sum_lse_fwd_c(arena_c, row_off + static_cast<int64_t>(f0) * H, w + static_cast<int64_t>(f0) * Ko * nin * (w_is_complex ? 2 : 1),
                              out_c + static_cast<int64_t>(f0) * B * Ko * 2, n, H, B, Ki, Ko, mode, w_is_complex, stream);
    });
  }
  const c32* a = reinterpret_cast<const c32*>(arena_c);
  c32* o = reinterpret_cast<c32*>(out_c);
  if (w_is_complex)
    return launch_generic<c32, c32>(a, row_off, reinterpret_cast<const c32*>(w), o, F, H, B, Ki, Ko, mode, stream);
  if (!g_force_generic && (mode == CK_SUM_PROD || H == 1) && Ki == kK && Ko == kK && ck::aligned16(arena_c) &&
      ck::aligned16(w) && ck::aligned16(out_c)) {
    const int tiles = (B + 31) / 32;
    int tpw = 1;
    while (tpw < 4 && static_cast<int64_t>(F) * ((tiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) >= 2048) tpw *= 2;
    dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(sum_clse_tile32, grid, block, 0, s, a, row_off, w, o, H, B, tpw, static_cast<const float*>(nullptr),
                             static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr),
                             static_cast<const int32_t*>(nullptr), 0);
          return hipGetLastError();
        },
        stream);
  }
  return launch_generic<c32, float>(a, row_off, w, o, F, H, B, Ki, Ko, mode, stream);
}

int ck_sum_clse_gather_fwd(const float* table, const int32_t* xt, const int32_t* child_fold, const int32_t* child_var,
                           const float* w, float* out_c, int F, int H, int B, int C, void* stream) {
  CK_REQUIRE(table && xt && child_fold && child_var && w && out_c, "ck_sum_clse_gather_fwd: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && C > 0, "ck_sum_clse_gather_fwd: non-positive size");
  CK_REQUIRE(F <= 65535, "ck_sum_clse_gather_fwd: F=%d exceeds grid.y", F);
  CK_REQUIRE(ck::aligned16(table) && ck::aligned16(w) && ck::aligned16(out_c), "ck_sum_clse_gather_fwd: buffers must be 16-byte aligned");
  const int tiles = (B + 31) / 32;
  int tpw = 1;
  while (tpw < 4 && static_cast<int64_t>(F) * ((tiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) >= 2048) tpw *= 2;
  dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), F), block(256);
  c32* o = reinterpret_cast<c32*>(out_c);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(sum_clse_tile32, grid, block, 0, s, static_cast<const c32*>(nullptr),
                           static_cast<const int64_t*>(nullptr), w, o, H, B, tpw, table, child_fold, child_var, xt, C);
        return hipGetLastError();
      },
      stream);
}

int ck_mixing_lse_fwd(const float* arena, const int64_t* row_off, const float* mw, float* out,
                      int F, int H, int B, int K, void* stream) {
  CK_REQUIRE(arena && row_off && mw && out, "ck_mixing_lse_fwd: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && K > 0, "ck_mixing_lse_fwd: non-positive size");
  if (F > ck::kMaxFoldsPerLaunch)
    return ck::chunk_folds(F, [&](int f0, int n) {
      return ck_mixing_lse_fwd(arena, row_off + static_cast<int64_t>(f0) * H, mw + static_cast<int64_t>(f0) * K * H,
                               out + static_cast<int64_t>(f0) * B * K, n, H, B, K, stream);
    });
  if (K <= 3 && H <= 16) {  // (K % 4 != 0 and tiny: the scalar root)
    dim3 grid(static_cast<unsigned>((static_cast<int64_t>(B) * K + 255) / 256), F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(mixing_lse_small<16>, grid, block, 0, s, arena, row_off, mw, out, H, B, K);
          return hipGetLastError();
        },
        stream);
  }
  const int lpr = K / 4;
  const bool vec = (K % 4 == 0) && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && ck::aligned16(arena) &&
                   ck::aligned16(out);
  // 128 rows per workgroup when that fills the chip; small launches (a few folds at batch 128) are one dependent load
  // chain per 4 x (256 / K) rows long: fewer rows per workgroup, down to one pass of its four waves
  int rows_per_block = vec ? 128 : 32;
  if (vec) {
    const int one_pass = 4 * (64 / lpr);
    while (rows_per_block > one_pass && static_cast<int64_t>(F) * ((B + rows_per_block - 1) / rows_per_block) < 4 * ck::num_cus())
      rows_per_block /= 2;
    rows_per_block = std::max(rows_per_block, one_pass);
  }
  dim3 grid((B + rows_per_block - 1) / rows_per_block, F), block(256);
  const size_t lds = static_cast<size_t>(K) * H * sizeof(float);
  CK_REQUIRE(!vec || lds <= 64 * 1024, "ck_mixing_lse_fwd: K*H=%d coefficients do not fit in LDS", K * H);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (vec && H <= 4)
          hipLaunchKernelGGL(mixing_lse_vec<4>, grid, block, lds, s, arena, row_off, mw, out, H, B, K, rows_per_block);
        else if (vec && H <= 8)
          hipLaunchKernelGGL(mixing_lse_vec<8>, grid, block, lds, s, arena, row_off, mw, out, H, B, K, rows_per_block);
        else if (vec && H <= 16)
          hipLaunchKernelGGL(mixing_lse_vec<16>, grid, block, lds, s, arena, row_off, mw, out, H, B, K, rows_per_block);
        else if (vec)
          hipLaunchKernelGGL(mixing_lse_vec<0>, grid, block, lds, s, arena, row_off, mw, out, H, B, K, rows_per_block);
        else
          hipLaunchKernelGGL(mixing_lse_kernel, grid, block, 0, s, arena, row_off, mw, out, H, B, K, rows_per_block);
        return hipGetLastError();
      },
      stream);
}

int ck_tensordot_lse_fwd(const float* arena, const int64_t* row_off, const float* w, float* out,
                         int F, int B, int Kj, int Kq, int Kk, void* stream) {
  return launch_tensordot<float, float>(arena, row_off, w, out, F, B, Kj, Kq, Kk, stream, "ck_tensordot_lse_fwd");
}

int ck_tensordot_lse_fwd_c(const float* arena_c, const int64_t* row_off, const float* w,
                           float* out_c, int F, int B, int Kj, int Kq, int Kk, int w_is_complex,
                           void* stream) {
  const c32* a = reinterpret_cast<const c32*>(arena_c);
  c32* o = reinterpret_cast<c32*>(out_c);
  if (w_is_complex)
    return launch_tensordot<c32, c32>(a, row_off, reinterpret_cast<const c32*>(w), o, F, B, Kj, Kq, Kk, stream,
                                      "ck_tensordot_lse_fwd_c");
  return launch_tensordot<c32, float>(a, row_off, w, o, F, B, Kj, Kq, Kk, stream, "ck_tensordot_lse_fwd_c");
}

}  // extern "C"
