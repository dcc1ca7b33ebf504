// Backward of the sum layers under the complex-lse-sum semiring -- what autograd gives the reference through
// ComplexLSESumSemiring.apply_reduce (semiring.py:441-476: y = log(sum_n w_n exp(v_n - m)) + m with a real shift m) and
// ComplexSafeLog (utils.py:22-50).  y is holomorphic in the inputs v and in the weights w:
//     dy_o / dv_n = w_on exp(v_n - y_o) =: p_on          dy_o / dw_on = exp(v_n - y_o)
// and torch's convention for a holomorphic function is  grad_in = conj(dy/din) * grad_y:
//     gv_n  = sum_o conj(p_on) gy_o
//     gw_on = sum_b conj(exp(v_n - y_o)) gy_o            (real weights: the real part)
// v is what the layer contracts: the concatenation of the children (CK_SUM_CAT), their sum (CK_SUM_PROD: a product in
// log space) or, for Tucker layers (CK_SUM_KRON, optimized.py:89-103), v_(i0..iH-1) = sum_h x_h[i_h]; gv goes back to the
// children accordingly.  A shape-generic kernel on the vector lanes: one workgroup per (fold, TB batch rows); the TB rows'
// contributions to dW are added up in the workgroup before ONE atomic per weight entry.
#include <cstdlib>
#include <type_traits>

#include "ck_internal.h"
#include "ck_bwd_tile.h"

namespace {

using ck::c32;

__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ c32 cconj(c32 a) { return {a.re, -a.im}; }
__device__ __forceinline__ c32 cexp(c32 z) {
  const float r = expf(z.re);
  float s, c;
  sincosf(z.im, &s, &c);
  return {r * c, r * s};
}

template <bool WC>
__global__ void __launch_bounds__(256)
    sum_clse_bwd_kernel(const c32* __restrict__ arena, c32* __restrict__ garena, const int64_t* __restrict__ row_off,
                        const float* __restrict__ w, const c32* __restrict__ out, const c32* __restrict__ gout,
                        float* __restrict__ dw, int H, int B, int Ki, int Ko, int mode, int N, int TB) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c32* v_s = reinterpret_cast<c32*>(smem);       // [TB][N] contracted inputs, then their gradients
  c32* y_s = v_s + static_cast<size_t>(TB) * N;  // [TB][Ko]
  c32* g_s = y_s + static_cast<size_t>(TB) * Ko;  // [TB][Ko]
  c32* gv_s = g_s + static_cast<size_t>(TB) * Ko;  // [TB][N]
  float* m_s = reinterpret_cast<float*>(gv_s + static_cast<size_t>(TB) * N);  // [TB] the rows' shifts
  const int f = blockIdx.y, b0 = blockIdx.x * TB;
  const int rows = min(TB, B - b0);
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int tid = threadIdx.x;
  for (int i = tid; i < rows * N; i += 256) {
    const int r = i / N, n = i - r * N;
    const int64_t b = b0 + r;
    c32 v;
    if (mode == CK_SUM_CAT) {
      v = arena[ro[n / Ki] + b * Ki + n % Ki];
    } else if (mode == CK_SUM_PROD) {
      v = arena[ro[0] + b * Ki + n];
      for (int h = 1; h < H; ++h) v = ck::c_add(v, arena[ro[h] + b * Ki + n]);
    } else {
      int rem = n;
      v = {0.f, 0.f};
      for (int h = H - 1; h >= 0; --h) {
        v = ck::c_add(v, arena[ro[h] + b * Ki + rem % Ki]);
        rem /= Ki;
      }
    }
    v_s[i] = v;
  }
  for (int i = tid; i < rows * Ko; i += 256) {
    const int r = i / Ko, o = i - r * Ko;
    y_s[i] = out[(static_cast<int64_t>(f) * B + b0 + r) * Ko + o];
    g_s[i] = gout[(static_cast<int64_t>(f) * B + b0 + r) * Ko + o];
  }
  __syncthreads();
  const float* wf = w + static_cast<int64_t>(f) * Ko * N * (WC ? 2 : 1);
  auto weight = [&](int o, int n) -> c32 {
    if constexpr (WC) return {wf[2 * (static_cast<int64_t>(o) * N + n)], wf[2 * (static_cast<int64_t>(o) * N + n) + 1]};
    else return {wf[static_cast<int64_t>(o) * N + n], 0.f};
  };
  // exp(v_n - y_o) = a_n b_o with a_n = exp(v_n - m), b_o = exp(m - y_o) and m the row's largest real part (the shift of the
  // forward, semiring.py:441-476): N + Ko complex exponentials per row instead of two per (n, o) pair.  With
  // t_o = conj(b_o) gy_o:   gv_n = conj(a_n) sum_o conj(w_on) t_o,   dW_on = sum_r conj(a_n) t_o.
  for (int r = tid; r < rows; r += 256) {
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, v_s[r * N + n].re);
    m_s[r] = ck::clamp_finite(m);
  }
  __syncthreads();
  for (int i = tid; i < rows * N; i += 256) {
    const c32 v = v_s[i];
    v_s[i] = cexp({v.re - m_s[i / N], v.im});  // a
  }
  for (int i = tid; i < rows * Ko; i += 256) {
    const c32 g = g_s[i], y = y_s[i];
    // (an output that receives no gradient contributes nothing -- also where y = -inf and b would be infinite)
    g_s[i] = (g.re == 0.f && g.im == 0.f) ? c32{0.f, 0.f} : cmul(cconj(cexp({m_s[i / Ko] - y.re, -y.im})), g);  // t
  }
  __syncthreads();
  for (int i = tid; i < rows * N; i += 256) {
    const int r = i / N, n = i - r * N;
    c32 acc{0.f, 0.f};
    for (int o = 0; o < Ko; ++o) acc = ck::c_add(acc, cmul(cconj(weight(o, n)), g_s[r * Ko + o]));
    gv_s[i] = cmul(cconj(v_s[i]), acc);
  }
  // dW[o][n] += sum_r conj(a_n) t_o
  for (int i = tid; i < Ko * N; i += 256) {
    const int o = i / N, n = i - o * N;
    c32 acc{0.f, 0.f};
    for (int r = 0; r < rows; ++r) acc = ck::c_add(acc, cmul(cconj(v_s[r * N + n]), g_s[r * Ko + o]));
    float* d = dw + (static_cast<int64_t>(f) * Ko * N + i) * (WC ? 2 : 1);
    atomicAdd(d, acc.re);
    if constexpr (WC) atomicAdd(d + 1, acc.im);
  }
  __syncthreads();
  // back to the children (stored: every entry of the children's gradient blocks is written exactly once)
  if (mode == CK_SUM_CAT) {
    for (int i = tid; i < rows * N; i += 256) {
      const int r = i / N, n = i - r * N;
      garena[ro[n / Ki] + static_cast<int64_t>(b0 + r) * Ki + n % Ki] = gv_s[i];
    }
  } else if (mode == CK_SUM_PROD) {
    for (int i = tid; i < rows * N; i += 256) {
      const int r = i / N, n = i - r * N;
      for (int h = 0; h < H; ++h) garena[ro[h] + static_cast<int64_t>(b0 + r) * Ki + n] = gv_s[i];
    }
  } else {
    for (int i = tid; i < rows * H * Ki; i += 256) {
      const int r = i / (H * Ki), h = (i / Ki) % H, k = i % Ki;
      int stride = 1;  // digit h of n (child 0 most significant) has stride Ki^(H-1-h)
      for (int j = h + 1; j < H; ++j) stride *= Ki;
      c32 acc{0.f, 0.f};
      for (int n = 0; n < N; ++n)
        if ((n / stride) % Ki == k) acc = ck::c_add(acc, gv_s[r * N + n]);
      garena[ro[h] + static_cast<int64_t>(b0 + r) * Ki + k] = acc;
    }
  }
}

// The same for the shape the squared circuits are made of -- REAL weights, 32 -> 32 units, a product of H children or one child --
// on the matrix instructions.  With a = exp(v - m) and t = conj(exp(m - y)) gy (both complex, m the row's shift):
//     gv = conj(a) (W^T t_re + i W^T t_im)          dW = t_re^T a_re + t_im^T a_im
// i.e. four real 32 x 32 contractions per 32-row tile, the ones of the real backward (ck_bwd_tile.h: child_gradient,
// dw_accumulate).  A workgroup = four waves walking the row tiles of one fold with dW in registers; one float atomic per weight
// entry and workgroup at the end.
__global__ void __launch_bounds__(256)
    sum_clse_bwd_tile32(const c32* __restrict__ arena, c32* __restrict__ garena, const int64_t* __restrict__ row_off,
                        const float* __restrict__ w, const c32* __restrict__ out, const c32* __restrict__ gout,
                        float* __restrict__ dw, int H, int B) {
  __shared__ __attribute__((aligned(16))) float wt_s[1024];        // W^T, "transposed tiled" (child_gradient)
  __shared__ __attribute__((aligned(16))) float scr_s[4][2][1024];  // per wave: the two operands of dw_accumulate
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* wf = w + static_cast<int64_t>(f) * 1024;
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int q = idx >> 8, ln = (idx >> 2) & 63, t = idx & 3;
    wt_s[idx] = wf[(8 * q + 4 * (ln >> 5) + t) * 32 + (ln & 31)];
  }
  __syncthreads();
  // 4 complex units = 8 floats: units 8 g + 4 kh + t of a row of 32 complex numbers
  auto load_c = [&](const c32* row, float (&re)[16], float (&im)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4* p = reinterpret_cast<const float4*>(row + 8 * g + 4 * kh);
      const float4 x = p[0], y = p[1];
      re[4 * g + 0] = x.x; im[4 * g + 0] = x.y; re[4 * g + 1] = x.z; im[4 * g + 1] = x.w;
      re[4 * g + 2] = y.x; im[4 * g + 2] = y.y; re[4 * g + 3] = y.z; im[4 * g + 3] = y.w;
    }
  };
  f32x16 dacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
  float ones[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ones[r] = 1.f;
  const int tiles = (B + 31) / 32;
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int b = tile * 32 + b_in;
    const bool live = b < B;
    const int64_t bl = live ? b : B - 1;
    float vre[16], vim[16];
    load_c(arena + ro[0] + bl * 32, vre, vim);
    for (int h = 1; h < H; ++h) {
      float xr[16], xi[16];
      load_c(arena + ro[h] + bl * 32, xr, xi);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        vre[j] += xr[j];
        vim[j] += xi[j];
      }
    }
    float m = vre[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) m = fmaxf(m, vre[j]);
    m = ck::clamp_finite(ck::xhalf_max(m));
    float are[16], aim[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const c32 a = ck::c_exp_shift_tile({vre[j], vim[j]}, m);
      are[j] = a.re;
      aim[j] = a.im;
    }
    float tre[16], tim[16];
    {
      float yre[16], yim[16], gre[16], gim[16];
      load_c(out + (static_cast<int64_t>(f) * B + bl) * 32, yre, yim);
      load_c(gout + (static_cast<int64_t>(f) * B + bl) * 32, gre, gim);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        // conj(exp(m - y)) = exp(m - y_re) (cos y_im + i sin y_im); an output without gradient contributes nothing (also
        // where y = -inf and the factor would be infinite)
        const bool on = live && !(gre[j] == 0.f && gim[j] == 0.f);
        const c32 cb = ck::c_exp_shift_tile({-yre[j], yim[j]}, -m);
        tre[j] = on ? cb.re * gre[j] - cb.im * gim[j] : 0.f;
        tim[j] = on ? cb.re * gim[j] + cb.im * gre[j] : 0.f;
      }
    }
    float Gre[16], Gim[16];
    child_gradient(wt_s, lane, tre, ones, Gre);
    child_gradient(wt_s, lane, tim, ones, Gim);
    if (live) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 x, y;  // conj(a) G
        x.x = are[4 * g + 0] * Gre[4 * g + 0] + aim[4 * g + 0] * Gim[4 * g + 0];
        x.y = are[4 * g + 0] * Gim[4 * g + 0] - aim[4 * g + 0] * Gre[4 * g + 0];
        x.z = are[4 * g + 1] * Gre[4 * g + 1] + aim[4 * g + 1] * Gim[4 * g + 1];
        x.w = are[4 * g + 1] * Gim[4 * g + 1] - aim[4 * g + 1] * Gre[4 * g + 1];
        y.x = are[4 * g + 2] * Gre[4 * g + 2] + aim[4 * g + 2] * Gim[4 * g + 2];
        y.y = are[4 * g + 2] * Gim[4 * g + 2] - aim[4 * g + 2] * Gre[4 * g + 2];
        y.z = are[4 * g + 3] * Gre[4 * g + 3] + aim[4 * g + 3] * Gim[4 * g + 3];
        y.w = are[4 * g + 3] * Gim[4 * g + 3] - aim[4 * g + 3] * Gre[4 * g + 3];
        for (int h = 0; h < H; ++h) {
          float4* p = reinterpret_cast<float4*>(garena + ro[h] + bl * 32 + 8 * g + 4 * kh);
          p[0] = x;
          p[1] = y;
        }
      }
    }
    dw_accumulate(dacc, scr_s[wave][0], scr_s[wave][1], b_in, kh, tre, are);
    dw_accumulate(dacc, scr_s[wave][0], scr_s[wave][1], b_in, kh, tim, aim);
  }
  // D[o][i] sits in lane (i, hi) register r with o = 8 (r >> 2) + 4 hi + (r & 3): the four waves' sums through LDS
  __syncthreads();
  float* red = &scr_s[0][0][0];  // [4][1024]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + (8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + b_in] = dacc[r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const float v = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
    if (v != 0.f) atomicAdd(dw + static_cast<int64_t>(f) * 1024 + idx, v);
  }
}

// TensorDot layers on their own layout (TorchTensorDotLayer, optimized.py:289-296: x (B, Kj, Kq), per (b, q) a dense sum over j
// with weights W (Kk, Kj), out (B, Kq, Kk)) for the partition function of a squared circuit -- one row, a chain of ~30 such
// layers, every launch a fixed cost: one workgroup per (fold, batch row), everything in LDS, and
//   * the Hadamard layer beneath (inner.py:126-127: a sum of H children in log space) read as a list of H blocks, its gradient
//     written to all of them (the layer itself is never launched);
//   * the PAIR of TensorDot layers a squared sum layer becomes (one over W, one over conj W: M' = W M W^T) in ONE launch, the
//     block between them handed over through memory inside the workgroup;
//   * the backward without transposed copies (the dense backward on rows (b, q) needs x permuted and permutes the gradient back).
// S: the arithmetic -- TdC: complex-lse-sum (the conventions of sum_clse_bwd_kernel above, REAL weights), TdR: lse-sum.
#ifdef CK_TD_STAMPS  // lab build (scripts/ubench/td_stamps.hip): wall-clock stamps of workgroup (0, 0)
__device__ long long* g_td_stamps = nullptr;
__device__ int g_td_n = 0;
#define TD_STAMP()                                                                                             \
  do {                                                                                                         \
    if (g_td_stamps != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && g_td_n < 32)        \
      g_td_stamps[g_td_n++] = static_cast<long long>(wall_clock64());                                          \
  } while (0)
#else
#define TD_STAMP() do {} while (0)
#endif

struct TdC {
  using T = c32;
  static __device__ __forceinline__ float re(c32 a) { return a.re; }
  static __device__ __forceinline__ c32 add(c32 a, c32 b) { return ck::c_add(a, b); }
  static __device__ __forceinline__ c32 exp_shift(c32 v, float m) { return cexp({v.re - m, v.im}); }                 // a
  static __device__ __forceinline__ c32 log_shift(c32 y, float m) { return ck::c_log_shift(y, m); }
  static __device__ __forceinline__ c32 tee(c32 y, c32 g, float m) {                                                 // t
    return (g.re == 0.f && g.im == 0.f) ? c32{0.f, 0.f} : cmul(cconj(cexp({m - y.re, -y.im})), g);
  }
  static __device__ __forceinline__ c32 zero() { return {0.f, 0.f}; }
  static __device__ __forceinline__ c32 fma_w(float w, c32 t, c32 acc) { return {fmaf(w, t.re, acc.re), fmaf(w, t.im, acc.im)}; }
  static __device__ __forceinline__ c32 child(c32 a, c32 acc) { return cmul(cconj(a), acc); }
  static __device__ __forceinline__ float dw(c32 a, c32 t) { return a.re * t.re + a.im * t.im; }  // Re(conj(a) t)
};
struct TdR {
  using T = float;
  static __device__ __forceinline__ float re(float a) { return a; }
  static __device__ __forceinline__ float add(float a, float b) { return a + b; }
  static __device__ __forceinline__ float exp_shift(float v, float m) { return expf(v - m); }
  static __device__ __forceinline__ float log_shift(float y, float m) { return logf(y) + m; }
  static __device__ __forceinline__ float tee(float y, float g, float m) { return g == 0.f ? 0.f : expf(m - y) * g; }
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ float fma_w(float w, float t, float acc) { return fmaf(w, t, acc); }
  static __device__ __forceinline__ float child(float a, float acc) { return a * acc; }
  static __device__ __forceinline__ float dw(float a, float t) { return a * t; }
};

// LDS of a stage: a [Kq][Kj + 1] values, (backward) t [Kq][Kk + 1] values, W [Kk][Kj + 1] floats, m [Kq] floats
template <class S>
__host__ __device__ constexpr size_t td_lds(int Kj, int Kq, int Kk, bool bwd) {
  return (static_cast<size_t>(Kq) * (Kj + 1) + (bwd ? static_cast<size_t>(Kq) * (Kk + 1) : 0)) * sizeof(typename S::T) +
         (static_cast<size_t>(Kk) * (Kj + 1) + Kq) * sizeof(float);
}

// a <- exp(x - m_q) with x[j][q] = load(j Kq + q), W staged: the common first half of both directions
template <class S, class Load>
__device__ __forceinline__ void td_stage(typename S::T* a_s, float* w_s, float* m_s, Load&& load, const float* __restrict__ wf, int Kj, int Kq,
                                         int Kk) {
  for (int i = threadIdx.x; i < Kk * Kj; i += 256) w_s[(i / Kj) * (Kj + 1) + i % Kj] = wf[i];
  for (int i = threadIdx.x; i < Kj * Kq; i += 256) {  // coalesced read of x[j][q]
    const int j = i / Kq, q = i - j * Kq;
    a_s[q * (Kj + 1) + j] = load(i);
  }
  __syncthreads();
  TD_STAMP();
  for (int q = threadIdx.x; q < Kq; q += 256) {
    float mx = -INFINITY;
    for (int j = 0; j < Kj; ++j) mx = fmaxf(mx, S::re(a_s[q * (Kj + 1) + j]));
    m_s[q] = ck::clamp_finite(mx);
  }
  __syncthreads();
  TD_STAMP();
  for (int i = threadIdx.x; i < Kq * Kj; i += 256) {
    const int q = i / Kj, j = i - q * Kj;
    a_s[q * (Kj + 1) + j] = S::exp_shift(a_s[q * (Kj + 1) + j], m_s[q]);
  }
  TD_STAMP();
}

template <class S, class Load>
__device__ __forceinline__ void td_fwd_body(char* smem, Load&& load, const float* __restrict__ wf, typename S::T* __restrict__ dst, int Kj, int Kq,
                                            int Kk) {
  using T = typename S::T;
  T* a_s = reinterpret_cast<T*>(smem);
  float* w_s = reinterpret_cast<float*>(a_s + static_cast<size_t>(Kq) * (Kj + 1));
  float* m_s = w_s + static_cast<size_t>(Kk) * (Kj + 1);
  td_stage<S>(a_s, w_s, m_s, load, wf, Kj, Kq, Kk);
  __syncthreads();
  TD_STAMP();
  for (int i = threadIdx.x; i < Kq * Kk; i += 256) {
    const int q = i / Kk, k = i - q * Kk;
    T acc = S::zero();
    for (int j = 0; j < Kj; ++j) acc = S::fma_w(w_s[k * (Kj + 1) + j], a_s[q * (Kj + 1) + j], acc);
    dst[i] = S::log_shift(acc, m_s[q]);
  }
  TD_STAMP();
}

// store(i, g): the gradient of x[i]; dwf: the fold's (Kk, Kj) weight gradient, added with float atomics
template <class S, class Load, class Store>
__device__ __forceinline__ void td_bwd_body(char* smem, Load&& load, Store&& store, const float* __restrict__ wf,
                                            const typename S::T* __restrict__ out, const typename S::T* __restrict__ gout, float* __restrict__ dwf,
                                            int Kj, int Kq, int Kk) {
  using T = typename S::T;
  T* a_s = reinterpret_cast<T*>(smem);
  T* t_s = a_s + static_cast<size_t>(Kq) * (Kj + 1);
  float* w_s = reinterpret_cast<float*>(t_s + static_cast<size_t>(Kq) * (Kk + 1));
  float* m_s = w_s + static_cast<size_t>(Kk) * (Kj + 1);
  td_stage<S>(a_s, w_s, m_s, load, wf, Kj, Kq, Kk);
  for (int i = threadIdx.x; i < Kq * Kk; i += 256) {
    const int q = i / Kk, k = i - q * Kk;
    t_s[q * (Kk + 1) + k] = S::tee(out[i], gout[i], m_s[q]);
  }
  __syncthreads();
  TD_STAMP();
  for (int i = threadIdx.x; i < Kj * Kq; i += 256) {  // the gradient of x[j][q], in x's layout
    const int j = i / Kq, q = i - j * Kq;
    T acc = S::zero();
    for (int k = 0; k < Kk; ++k) acc = S::fma_w(w_s[k * (Kj + 1) + j], t_s[q * (Kk + 1) + k], acc);
    store(i, S::child(a_s[q * (Kj + 1) + j], acc));
  }
  TD_STAMP();
  for (int i = threadIdx.x; i < Kk * Kj; i += 256) {
    const int k = i / Kj, j = i - k * Kj;
    float acc = 0.f;
    for (int q = 0; q < Kq; ++q) acc += S::dw(a_s[q * (Kj + 1) + j], t_s[q * (Kk + 1) + k]);
    if (acc != 0.f) atomicAdd(dwf + i, acc);
  }
  TD_STAMP();
}

// One or two stages.  Stage 1: x = the sum of the H blocks arena + row_off[f, h] (+ b Kj Kq), weights w1 (F, Kk1, Kj), output
// (Kq, Kk1) -- to `out`, or with TWO to `mid`; stage 2 reads it as (Kj2 = Kq, Kq2 = Kk1), weights w2 (F, Kk2, Kq), output (Kk1, Kk2).
template <class S, bool TWO>
__global__ void __launch_bounds__(256)
    td_fwd_kernel(const typename S::T* __restrict__ arena, const int64_t* __restrict__ row_off, int H, const float* __restrict__ w1,
                  typename S::T* __restrict__ mid, const float* __restrict__ w2, typename S::T* __restrict__ out, int B, int Kj, int Kq, int Kk1,
                  int Kk2) {
  using T = typename S::T;
  extern __shared__ __attribute__((aligned(16))) char td_smem[];
  const int f = blockIdx.y, b = blockIdx.x;
  TD_STAMP();
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t boff = static_cast<int64_t>(b) * Kj * Kq;
  auto load1 = [&](int i) {
    T v = arena[ro[0] + boff + i];
    for (int h = 1; h < H; ++h) v = S::add(v, arena[ro[h] + boff + i]);
    return v;
  };
  T* o1 = (TWO ? mid : out) + (static_cast<int64_t>(f) * B + b) * Kq * Kk1;
  td_fwd_body<S>(td_smem, load1, w1 + static_cast<int64_t>(f) * Kk1 * Kj, o1, Kj, Kq, Kk1);
  if constexpr (TWO) {
    __syncthreads();  // (the block between the stages: written and read by this workgroup only)
    const T* m1 = o1;
    td_fwd_body<S>(td_smem, [&](int i) { return m1[i]; }, w2 + static_cast<int64_t>(f) * Kk2 * Kq,
                   out + (static_cast<int64_t>(f) * B + b) * Kk1 * Kk2, Kq, Kk1, Kk2);
  }
}

template <class S, bool TWO>
__global__ void __launch_bounds__(256)
    td_bwd_kernel(const typename S::T* __restrict__ arena, typename S::T* __restrict__ garena, const int64_t* __restrict__ row_off, int H,
                  const float* __restrict__ w1, const typename S::T* __restrict__ mid, typename S::T* __restrict__ gmid, const float* __restrict__ w2,
                  const typename S::T* __restrict__ out, const typename S::T* __restrict__ gout, float* __restrict__ dw1, float* __restrict__ dw2,
                  int B, int Kj, int Kq, int Kk1, int Kk2) {
  using T = typename S::T;
  extern __shared__ __attribute__((aligned(16))) char td_smem[];
  const int f = blockIdx.y, b = blockIdx.x;
  TD_STAMP();
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t boff = static_cast<int64_t>(b) * Kj * Kq;
  const int64_t o1 = (static_cast<int64_t>(f) * B + b) * Kq * Kk1;
  const T* y1 = (TWO ? mid : out) + o1;   // stage 1's output and its gradient
  const T* g1 = (TWO ? gmid : gout) + o1;
  if constexpr (TWO) {
    const int64_t o2 = (static_cast<int64_t>(f) * B + b) * Kk1 * Kk2;
    T* gm = gmid + o1;
    td_bwd_body<S>(td_smem, [&](int i) { return y1[i]; }, [&](int i, T g) { gm[i] = g; }, w2 + static_cast<int64_t>(f) * Kk2 * Kq,
                   out + o2, gout + o2, dw2 + static_cast<int64_t>(f) * Kk2 * Kq, Kq, Kk1, Kk2);
    __syncthreads();
  }
  auto load1 = [&](int i) {
    T v = arena[ro[0] + boff + i];
    for (int h = 1; h < H; ++h) v = S::add(v, arena[ro[h] + boff + i]);
    return v;
  };
  auto store1 = [&](int i, T g) {
    for (int h = 0; h < H; ++h) garena[ro[h] + boff + i] = g;  // (every factor of the product receives the same gradient)
  };
  td_bwd_body<S>(td_smem, load1, store1, w1 + static_cast<int64_t>(f) * Kk1 * Kj, y1, g1, dw1 + static_cast<int64_t>(f) * Kk1 * Kj, Kj, Kq, Kk1);
}

// ---- 32 units everywhere (the partition function of BASELINE config 5: M' = W M W^T per fold, (32, 32) blocks) ---------------------
// The shape-generic bodies above are chains of phases of 1 - 4 us each on ONE row (scripts/ubench/td_stamps.hip: per stage 1.7 us of
// loads, 1.8 a serial maximum per column, 1.2 library exponentials, 3.9 a contraction whose 4 x 32 dependent LDS reads the
// compiler cannot unroll for run-time sizes + library logarithms; the second stage reloads from memory what the workgroup has
// just stored: 17 us forward, 28 us backward for ONE fold).  With the sizes fixed thread t owns the elements i = t + 256 r of every
// (32, 32) block in play -- the same four for the stage's input x[j][q] (i = 32 j + q), its output / the tee (i = 32 q + k), the
// child gradient (i = 32 j + q) and the weight gradient (i = 32 k + j) -- so the column maxima are four registers + one shuffle +
// four LDS words, the second stage's input and the gradient between the stages never leave the registers, and every LDS loop is
// unrolled over 16-byte reads.  The order of every sum is the generic kernels' (bit-identical contractions); exp / log / sincos /
// atan are the polynomial forms of the complex tile kernels (ck_internal.h, <= 2 ulp).
struct TdC32 : TdC {
  static constexpr int kA = 34;  // values per LDS row of a (32, 32) block: rows start on 16 bytes
  static __device__ __forceinline__ c32 exp_shift(c32 v, float m) { return ck::c_exp_shift_tile(v, m); }
  static __device__ __forceinline__ c32 log_shift(c32 y, float m) { return ck::c_log_shift_tile(y, m); }
  static __device__ __forceinline__ c32 tee(c32 y, c32 g, float m) {
    return (g.re == 0.f && g.im == 0.f) ? c32{0.f, 0.f} : cmul(cconj(ck::c_exp_shift_tile({-y.re, -y.im}, -m)), g);
  }
  static __device__ __forceinline__ void read4(const c32* p, c32 (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 2);
    v[0] = {a.x, a.y};
    v[1] = {a.z, a.w};
    v[2] = {b.x, b.y};
    v[3] = {b.z, b.w};
  }
};
struct TdR32 : TdR {
  static constexpr int kA = 36;
  static __device__ __forceinline__ void read4(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x;
    v[1] = a.y;
    v[2] = a.z;
    v[3] = a.w;
  }
};
constexpr int kTdW = 36;  // floats per LDS row of a (32, 32) weight matrix

template <class S>
struct Td32Lds {
  typename S::T a[32 * S::kA];   // a[q][j] = exp(x[j][q] - m_q)
  typename S::T t[32 * S::kA];   // (backward) t[q][k]
  typename S::T tt[32 * S::kA];  // (backward) t[q][k] at [k][q]
  float w[2][32 * kTdW];         // W[k][j] of the two stages
  float wt[2][32 * kTdW];        // (backward) W[k][j] at [j][k]
  float m[32];
  float red[4][32];
};

// the weights of a stage into LDS (visible after the stage's first barrier)
template <bool BWD>
__device__ __forceinline__ void td32_stage_w(const float* __restrict__ wf, float* w_s, float* wt_s, int t) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = t + 256 * r;
    const float v = wf[i];
    w_s[(i >> 5) * kTdW + (i & 31)] = v;
    if (BWD) wt_s[(i & 31) * kTdW + (i >> 5)] = v;
  }
}

// m_q = the clamped maximum of column q = t & 31 (this thread's four values are rows j = (t >> 5) + 8 r of it), a = exp(x - m_q):
// into `a` and, transposed, into LDS; m into LDS.  Ends with a barrier.
template <class S, class L>
__device__ __forceinline__ void td32_exp(L& l, const typename S::T (&x)[4], typename S::T (&a)[4], int t) {
  float pm = fmaxf(fmaxf(S::re(x[0]), S::re(x[1])), fmaxf(S::re(x[2]), S::re(x[3])));
  pm = fmaxf(pm, __shfl_xor(pm, 32, 64));
  if ((t & 63) < 32) l.red[t >> 6][t & 31] = pm;
  __syncthreads();
  const int q = t & 31;
  const float m = ck::clamp_finite(fmaxf(fmaxf(l.red[0][q], l.red[1][q]), fmaxf(l.red[2][q], l.red[3][q])));
  if (t < 32) l.m[t] = m;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a[r] = S::exp_shift(x[r], m);
    l.a[q * S::kA + (t >> 5) + 8 * r] = a[r];
  }
  __syncthreads();
}

// y[q'][k] = log(sum_j W[k][j] a[q'][j]) + m_q' for this thread's elements i = 32 q' + k
template <class S, class L>
__device__ __forceinline__ void td32_fwd_stage(L& l, const float* w_s, const typename S::T (&x)[4], typename S::T (&y)[4], int t) {
  using T = typename S::T;
  T a[4];
  td32_exp<S>(l, x, a, t);
  const int k = t & 31, q0 = t >> 5;
  T acc[4] = {S::zero(), S::zero(), S::zero(), S::zero()};
#pragma unroll 2
  for (int j = 0; j < 32; j += 4) {
    const float4 w4 = *reinterpret_cast<const float4*>(w_s + k * kTdW + j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T av[4];
      S::read4(l.a + (q0 + 8 * r) * S::kA + j, av);
      acc[r] = S::fma_w(w4.x, av[0], acc[r]);
      acc[r] = S::fma_w(w4.y, av[1], acc[r]);
      acc[r] = S::fma_w(w4.z, av[2], acc[r]);
      acc[r] = S::fma_w(w4.w, av[3], acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = S::log_shift(acc[r], l.m[q0 + 8 * r]);
}

// x, y, g: the stage's input, output and output gradient at this thread's elements; gx: the gradient of x; dwf += the fold's dW
template <class S, class L>
__device__ __forceinline__ void td32_bwd_stage(L& l, const float* wt_s, const typename S::T (&x)[4], const typename S::T (&y)[4],
                                               const typename S::T (&g)[4], typename S::T (&gx)[4], float* __restrict__ dwf, int t) {
  using T = typename S::T;
  T a[4];
  td32_exp<S>(l, x, a, t);
  const int c = t & 31, r0 = t >> 5;  // element i = 32 (r0 + 8 r) + c
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // i = 32 q' + k
    const int qp = r0 + 8 * r;
    const T tv = S::tee(y[r], g[r], l.m[qp]);
    l.t[qp * S::kA + c] = tv;
    l.tt[c * S::kA + qp] = tv;
  }
  __syncthreads();
  {  // i = 32 j + q: gx[j][q] = child(a[q][j], sum_k W[k][j] t[q][k])
    T acc[4] = {S::zero(), S::zero(), S::zero(), S::zero()};
#pragma unroll 2
    for (int k = 0; k < 32; k += 4) {
      T tv[4];
      S::read4(l.t + c * S::kA + k, tv);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 w4 = *reinterpret_cast<const float4*>(wt_s + (r0 + 8 * r) * kTdW + k);
        acc[r] = S::fma_w(w4.x, tv[0], acc[r]);
        acc[r] = S::fma_w(w4.y, tv[1], acc[r]);
        acc[r] = S::fma_w(w4.z, tv[2], acc[r]);
        acc[r] = S::fma_w(w4.w, tv[3], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) gx[r] = S::child(a[r], acc[r]);
  }
  {  // i = 32 k + j: dW[k][j] = sum_q dw(a[q][j], t[q][k])
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int q = 0; q < 32; q += 4) {
      T av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) av[u] = l.a[(q + u) * S::kA + c];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T tv[4];
        S::read4(l.tt + (r0 + 8 * r) * S::kA + q, tv);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[r] += S::dw(av[u], tv[u]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (acc[r] != 0.f) atomicAdd(dwf + t + 256 * r, acc[r]);
  }
}

template <class S, bool TWO>
__global__ void __launch_bounds__(256, 2)
    td32_fwd_kernel(const typename S::T* __restrict__ arena, const int64_t* __restrict__ row_off, int H, const float* __restrict__ w1,
                    typename S::T* __restrict__ mid, const float* __restrict__ w2, typename S::T* __restrict__ out, int B) {
  using T = typename S::T;
  __shared__ __attribute__((aligned(16))) Td32Lds<S> l;
  const int t = threadIdx.x, f = blockIdx.y, b = blockIdx.x;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t boff = static_cast<int64_t>(b) * 1024, fb = (static_cast<int64_t>(f) * B + b) * 1024;
  T x[4], y[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = arena[ro[0] + boff + t + 256 * r];
  for (int h = 1; h < H; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = S::add(x[r], arena[ro[h] + boff + t + 256 * r]);
  td32_stage_w<false>(w1 + static_cast<int64_t>(f) * 1024, l.w[0], nullptr, t);
  if (TWO) td32_stage_w<false>(w2 + static_cast<int64_t>(f) * 1024, l.w[1], nullptr, t);
  td32_fwd_stage<S>(l, l.w[0], x, y, t);
  T* o1 = (TWO ? mid : out) + fb;
#pragma unroll
  for (int r = 0; r < 4; ++r) o1[t + 256 * r] = y[r];
  if constexpr (TWO) {  // (stage 2 reads stage 1's (Kq, Kk1) block as its (Kj, Kq): element i of the one is element i of the other)
    td32_fwd_stage<S>(l, l.w[1], y, x, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[fb + t + 256 * r] = x[r];
  }
}

template <class S, bool TWO>
__global__ void __launch_bounds__(256, 2)
    td32_bwd_kernel(const typename S::T* __restrict__ arena, typename S::T* __restrict__ garena, const int64_t* __restrict__ row_off, int H,
                    const float* __restrict__ w1, const typename S::T* __restrict__ mid, typename S::T* __restrict__ gmid,
                    const float* __restrict__ w2, const typename S::T* __restrict__ out, const typename S::T* __restrict__ gout,
                    float* __restrict__ dw1, float* __restrict__ dw2, int B) {
  using T = typename S::T;
  __shared__ __attribute__((aligned(16))) Td32Lds<S> l;
  const int t = threadIdx.x, f = blockIdx.y, b = blockIdx.x;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t boff = static_cast<int64_t>(b) * 1024, fb = (static_cast<int64_t>(f) * B + b) * 1024;
  // everything the launch reads from memory is requested before the first barrier
  T x1[4], y1[4], g1[4], y2[4], g2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = t + 256 * r;
    x1[r] = arena[ro[0] + boff + i];
    y1[r] = (TWO ? mid : out)[fb + i];
    if (TWO) {
      y2[r] = out[fb + i];
      g2[r] = gout[fb + i];
    } else {
      g1[r] = gout[fb + i];
    }
  }
  for (int h = 1; h < H; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) x1[r] = S::add(x1[r], arena[ro[h] + boff + t + 256 * r]);
  td32_stage_w<true>(w1 + static_cast<int64_t>(f) * 1024, l.w[0], l.wt[0], t);
  if constexpr (TWO) {
    td32_stage_w<true>(w2 + static_cast<int64_t>(f) * 1024, l.w[1], l.wt[1], t);
    td32_bwd_stage<S>(l, l.wt[1], y1, y2, g2, g1, dw2 + static_cast<int64_t>(f) * 1024, t);  // (stage 2's input is stage 1's output)
#pragma unroll
    for (int r = 0; r < 4; ++r) gmid[fb + t + 256 * r] = g1[r];
  }
  T gx[4];
  td32_bwd_stage<S>(l, l.wt[0], x1, y1, g1, gx, dw1 + static_cast<int64_t>(f) * 1024, t);
  for (int h = 0; h < H; ++h)  // (every factor of the product receives the same gradient)
#pragma unroll
    for (int r = 0; r < 4; ++r) garena[ro[h] + boff + t + 256 * r] = gx[r];
}

// The ROOT of a squared circuit's partition function: the pair over a scalar sum layer (Kj = Kq = 32, ONE output unit in both
// stages: Z = w^T M w in log space).  Stage 1 is a (32, 32) block against one weight row, stage 2 thirty-two values against one:
// `td32_exp` + a row sum per lane, the sums in the shape-generic kernels' order (which spend 10 us on this fold alone, 30 beside
// other launches: phases sized for 1024 outputs).
template <class S>
__global__ void __launch_bounds__(256)
    td32_root_fwd_kernel(const typename S::T* __restrict__ arena, const int64_t* __restrict__ row_off, int H, const float* __restrict__ w1,
                         typename S::T* __restrict__ mid, const float* __restrict__ w2, typename S::T* __restrict__ out, int B) {
  using T = typename S::T;
  __shared__ __attribute__((aligned(16))) Td32Lds<S> l;
  const int t = threadIdx.x, f = blockIdx.y, b = blockIdx.x;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t boff = static_cast<int64_t>(b) * 1024, fb = static_cast<int64_t>(f) * B + b;
  T x[4], a[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = arena[ro[0] + boff + t + 256 * r];
  for (int h = 1; h < H; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = S::add(x[r], arena[ro[h] + boff + t + 256 * r]);
  if (t < 32) {
    l.w[0][t] = w1[static_cast<int64_t>(f) * 32 + t];
    l.w[1][t] = w2[static_cast<int64_t>(f) * 32 + t];
  }
  td32_exp<S>(l, x, a, t);
  if (t >= 32) return;  // (no barrier below: wave 0 alone)
  T acc = S::zero();
#pragma unroll 8
  for (int j = 0; j < 32; ++j) acc = S::fma_w(l.w[0][j], l.a[t * S::kA + j], acc);
  const T o1 = S::log_shift(acc, l.m[t]);
  mid[fb * 32 + t] = o1;
  float m2 = S::re(o1);
#pragma unroll
  for (int s2 = 1; s2 < 32; s2 <<= 1) m2 = fmaxf(m2, __shfl_xor(m2, s2, 64));
  m2 = ck::clamp_finite(m2);
  l.t[t] = S::exp_shift(o1, m2);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
  __builtin_amdgcn_wave_barrier();
  if (t == 0) {
    T y2 = S::zero();
    for (int q = 0; q < 32; ++q) y2 = S::fma_w(l.w[1][q], l.t[q], y2);
    out[fb] = S::log_shift(y2, m2);
  }
}

template <class S>
__global__ void __launch_bounds__(256)
    td32_root_bwd_kernel(const typename S::T* __restrict__ arena, typename S::T* __restrict__ garena, const int64_t* __restrict__ row_off, int H,
                         const float* __restrict__ w1, const typename S::T* __restrict__ mid, typename S::T* __restrict__ gmid,
                         const float* __restrict__ w2, const typename S::T* __restrict__ out, const typename S::T* __restrict__ gout,
                         float* __restrict__ dw1, float* __restrict__ dw2, int B) {
  using T = typename S::T;
  __shared__ __attribute__((aligned(16))) Td32Lds<S> l;
  const int t = threadIdx.x, f = blockIdx.y, b = blockIdx.x;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t boff = static_cast<int64_t>(b) * 1024, fb = static_cast<int64_t>(f) * B + b;
  T x[4], a[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = arena[ro[0] + boff + t + 256 * r];
  for (int h = 1; h < H; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = S::add(x[r], arena[ro[h] + boff + t + 256 * r]);
  if (t < 32) {
    l.w[0][t] = w1[static_cast<int64_t>(f) * 32 + t];
    // stage 2 backward, lane j: a2 = exp(y1[j] - m2), t2 = tee(out, gout, m2), g y1[j] = child(a2, W2[j] t2), dW2[j] = dw(a2, t2)
    const T y1 = mid[fb * 32 + t];
    float m2 = S::re(y1);
#pragma unroll
    for (int s2 = 1; s2 < 32; s2 <<= 1) m2 = fmaxf(m2, __shfl_xor(m2, s2, 64));
    m2 = ck::clamp_finite(m2);
    const T a2 = S::exp_shift(y1, m2);
    const T t2 = S::tee(out[fb], gout[fb], m2);
    const T g1 = S::child(a2, S::fma_w(w2[static_cast<int64_t>(f) * 32 + t], t2, S::zero()));
    gmid[fb * 32 + t] = g1;
    const float d2 = 0.f + S::dw(a2, t2);
    if (d2 != 0.f) atomicAdd(dw2 + static_cast<int64_t>(f) * 32 + t, d2);
    l.tt[t] = y1;  // (kept for the tee of stage 1 below, which needs m_q)
    l.tt[32 + t] = g1;
  }
  td32_exp<S>(l, x, a, t);  // (two barriers: the rows above are visible after them)
  if (t < 32) l.t[t] = S::tee(l.tt[t], l.tt[32 + t], l.m[t]);  // t1[q]
  __syncthreads();
  {  // the gradient of x[j][q] at this thread's elements i = 32 j + q
    const int q = t & 31, j0 = t >> 5;
    T gx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gx[r] = S::child(a[r], S::fma_w(l.w[0][j0 + 8 * r], l.t[q], S::zero()));
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) garena[ro[h] + boff + t + 256 * r] = gx[r];
  }
  if (t < 32) {  // dW1[j] = sum_q dw(a[q][j], t1[q])
    float acc = 0.f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) acc += S::dw(l.a[q * S::kA + t], l.t[q]);
    if (acc != 0.f) atomicAdd(dw1 + static_cast<int64_t>(f) * 32 + t, acc);
  }
}

template <class S, bool TWO, bool BWD>
int td_launch(const typename S::T* arena, typename S::T* garena, const int64_t* row_off, int H, const float* w1, typename S::T* mid,
              typename S::T* gmid, const float* w2, typename S::T* out, const typename S::T* gout, float* dw1, float* dw2, int F, int B, int Kj,
              int Kq, int Kk1, int Kk2, void* stream, const char* who) {
  CK_REQUIRE(arena && row_off && w1 && out, "%s: null pointer", who);
  CK_REQUIRE(!TWO || (mid && w2), "%s: null pointer", who);
  CK_REQUIRE(!BWD || (garena && gout && dw1 && (!TWO || (gmid && dw2))), "%s: null pointer", who);
  CK_REQUIRE(F > 0 && F <= 65535 && H > 0 && B > 0 && Kj > 0 && Kq > 0 && Kk1 > 0 && (!TWO || Kk2 > 0), "%s: bad sizes", who);
  size_t lds = td_lds<S>(Kj, Kq, Kk1, BWD);
  if (TWO) lds = std::max(lds, td_lds<S>(Kq, Kk1, Kk2, BWD));
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "%s: Kj=%d, Kq=%d, Kk=%d, %d do not fit in LDS", who, Kj, Kq, Kk1, Kk2);
  const dim3 grid(B, F), block(256);
  // (CK_TD_GENERIC=1: the shape-generic kernels also for 32 units -- tests compare the two)
  const char* env = getenv("CK_TD_GENERIC");
  const bool generic_only = env != nullptr && atoi(env) != 0;
  const bool all32 = !generic_only && Kj == 32 && Kq == 32 && Kk1 == 32 && (!TWO || Kk2 == 32);
  const bool root32 = !generic_only && TWO && Kj == 32 && Kq == 32 && Kk1 == 1 && Kk2 == 1;
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern, auto&&... args) {
          if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
          }
          hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
          return hipGetLastError();
        };
        if constexpr (TWO) {
          if (root32) {
            using S32 = std::conditional_t<std::is_same_v<S, TdC>, TdC32, TdR32>;
            if constexpr (BWD)
              hipLaunchKernelGGL((td32_root_bwd_kernel<S32>), grid, block, 0, s, arena, garena, row_off, H, w1, static_cast<const typename S::T*>(mid),
                                 gmid, w2, static_cast<const typename S::T*>(out), gout, dw1, dw2, B);
            else
              hipLaunchKernelGGL((td32_root_fwd_kernel<S32>), grid, block, 0, s, arena, row_off, H, w1, mid, w2, out, B);
            return hipGetLastError();
          }
        }
        if (all32) {  // (static LDS: 30 - 46 KB)
          using S32 = std::conditional_t<std::is_same_v<S, TdC>, TdC32, TdR32>;
          if constexpr (BWD)
            hipLaunchKernelGGL((td32_bwd_kernel<S32, TWO>), grid, block, 0, s, arena, garena, row_off, H, w1, static_cast<const typename S::T*>(mid),
                               gmid, w2, static_cast<const typename S::T*>(out), gout, dw1, dw2, B);
          else
            hipLaunchKernelGGL((td32_fwd_kernel<S32, TWO>), grid, block, 0, s, arena, row_off, H, w1, mid, w2, out, B);
          return hipGetLastError();
        }
        if constexpr (BWD)
          return go(td_bwd_kernel<S, TWO>, arena, garena, row_off, H, w1, static_cast<const typename S::T*>(mid), gmid, w2,
                    static_cast<const typename S::T*>(out), gout, dw1, dw2, B, Kj, Kq, Kk1, Kk2);
        else
          return go(td_fwd_kernel<S, TWO>, arena, row_off, H, w1, mid, w2, out, B, Kj, Kq, Kk1, Kk2);
      },
      stream);
}

}  // namespace

extern "C" int ck_tensordot_lse_fwd_h(const float* arena, const int64_t* row_off, int H, const float* w, float* out, int F, int B, int Kj,
                                      int Kq, int Kk, int complex_values, void* stream) {
  const char* who = "ck_tensordot_lse_fwd_h";
  if (complex_values)
    return td_launch<TdC, false, false>(reinterpret_cast<const ck::c32*>(arena), nullptr, row_off, H, w, nullptr, nullptr, nullptr,
                                        reinterpret_cast<ck::c32*>(out), nullptr, nullptr, nullptr, F, B, Kj, Kq, Kk, 0, stream, who);
  return td_launch<TdR, false, false>(arena, nullptr, row_off, H, w, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, F, B, Kj, Kq, Kk, 0,
                                      stream, who);
}

extern "C" int ck_tensordot2_lse_fwd(const float* arena, const int64_t* row_off, int H, const float* w1, float* mid, const float* w2, float* out,
                                     int F, int B, int Kj, int Kq, int Kk1, int Kk2, int complex_values, void* stream) {
  const char* who = "ck_tensordot2_lse_fwd";
  if (complex_values)
    return td_launch<TdC, true, false>(reinterpret_cast<const ck::c32*>(arena), nullptr, row_off, H, w1, reinterpret_cast<ck::c32*>(mid), nullptr,
                                       w2, reinterpret_cast<ck::c32*>(out), nullptr, nullptr, nullptr, F, B, Kj, Kq, Kk1, Kk2, stream, who);
  return td_launch<TdR, true, false>(arena, nullptr, row_off, H, w1, mid, nullptr, w2, out, nullptr, nullptr, nullptr, F, B, Kj, Kq, Kk1, Kk2, stream,
                                     who);
}

extern "C" int ck_tensordot_lse_bwd(const float* arena, float* garena, const int64_t* row_off, int H, const float* w, const float* out,
                                    const float* gout, float* dw, int F, int B, int Kj, int Kq, int Kk, int complex_values, void* stream) {
  const char* who = "ck_tensordot_lse_bwd";
  if (complex_values)
    return td_launch<TdC, false, true>(reinterpret_cast<const ck::c32*>(arena), reinterpret_cast<ck::c32*>(garena), row_off, H, w, nullptr, nullptr,
                                       nullptr, reinterpret_cast<ck::c32*>(const_cast<float*>(out)), reinterpret_cast<const ck::c32*>(gout), dw,
                                       nullptr, F, B, Kj, Kq, Kk, 0, stream, who);
  return td_launch<TdR, false, true>(arena, garena, row_off, H, w, nullptr, nullptr, nullptr, const_cast<float*>(out), gout, dw, nullptr, F, B, Kj, Kq,
                                     Kk, 0, stream, who);
}

extern "C" int ck_tensordot2_lse_bwd(const float* arena, float* garena, const int64_t* row_off, int H, const float* w1, const float* mid,
                                     float* gmid, const float* w2, const float* out, const float* gout, float* dw1, float* dw2, int F, int B,
                                     int Kj, int Kq, int Kk1, int Kk2, int complex_values, void* stream) {
  const char* who = "ck_tensordot2_lse_bwd";
  if (complex_values)
    return td_launch<TdC, true, true>(reinterpret_cast<const ck::c32*>(arena), reinterpret_cast<ck::c32*>(garena), row_off, H, w1,
                                      reinterpret_cast<ck::c32*>(const_cast<float*>(mid)), reinterpret_cast<ck::c32*>(gmid), w2,
                                      reinterpret_cast<ck::c32*>(const_cast<float*>(out)), reinterpret_cast<const ck::c32*>(gout), dw1, dw2, F, B,
                                      Kj, Kq, Kk1, Kk2, stream, who);
  return td_launch<TdR, true, true>(arena, garena, row_off, H, w1, const_cast<float*>(mid), gmid, w2, const_cast<float*>(out), gout, dw1, dw2, F, B, Kj,
                                    Kq, Kk1, Kk2, stream, who);
}

extern "C" int ck_sum_lse_bwd_c(const float* arena_c, float* garena_c, const int64_t* row_off, const float* w, const float* out_c,
                                const float* gout_c, float* dw, int F, int H, int B, int Ki, int Ko, int mode, int w_is_complex,
                                void* stream) {
  CK_REQUIRE(arena_c && garena_c && row_off && w && out_c && gout_c && dw, "ck_sum_lse_bwd_c: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && Ki > 0 && Ko > 0, "ck_sum_lse_bwd_c: non-positive size");
  CK_REQUIRE(mode == CK_SUM_CAT || mode == CK_SUM_PROD || mode == CK_SUM_KRON, "ck_sum_lse_bwd_c: unknown mode %d", mode);
  CK_REQUIRE(F <= 65535, "ck_sum_lse_bwd_c: F=%d exceeds grid.y", F);
  if (!w_is_complex && Ki == 32 && Ko == 32 && (mode == CK_SUM_PROD || H == 1) && ck::aligned16(arena_c) && ck::aligned16(garena_c) &&
      ck::aligned16(out_c) && ck::aligned16(gout_c)) {  // (offsets in row_off are multiples of 32 complex numbers per row)
    const int tiles = (B + 31) / 32;
    const dim3 grid(static_cast<unsigned>(std::max(1, std::min((tiles + 3) / 4, 16))), F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(sum_clse_bwd_tile32, grid, block, 0, s, reinterpret_cast<const ck::c32*>(arena_c), reinterpret_cast<ck::c32*>(garena_c),
                             row_off, w, reinterpret_cast<const ck::c32*>(out_c), reinterpret_cast<const ck::c32*>(gout_c), dw, H, B);
          return hipGetLastError();
        },
        stream);
  }
  int64_t N = Ki;
  if (mode == CK_SUM_CAT) N = static_cast<int64_t>(H) * Ki;
  if (mode == CK_SUM_KRON) for (int h = 1; h < H; ++h) N *= Ki;
  // rows per workgroup: as many as 64 KiB of LDS hold (two workgroups per CU) -- every workgroup ends with one float atomic per
  // weight entry, 64 rows instead of 8 are an eighth of them
  int TB = 64;
  while (TB > 1 && static_cast<int64_t>(TB) * ((2 * N + 2 * Ko) * 8 + 4) > 64 * 1024) TB /= 2;
  const size_t lds = static_cast<size_t>(TB) * ((2 * N + 2 * Ko) * 8 + 4);
  if (lds > 128 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_sum_lse_bwd_c: %lld contracted inputs do not fit the LDS", static_cast<long long>(N));
  const dim3 grid((B + TB - 1) / TB, F), block(256);
  const int Ni = static_cast<int>(N);
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern) {
          if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
          }
          hipLaunchKernelGGL(kern, grid, block, lds, s, reinterpret_cast<const ck::c32*>(arena_c), reinterpret_cast<ck::c32*>(garena_c), row_off, w,
                             reinterpret_cast<const ck::c32*>(out_c), reinterpret_cast<const ck::c32*>(gout_c), dw, H, B, Ki, Ko, mode, Ni, TB);
          return hipGetLastError();
        };
        return w_is_complex ? go(sum_clse_bwd_kernel<true>) : go(sum_clse_bwd_kernel<false>);
      },
      stream);
}
