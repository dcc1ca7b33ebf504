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
#include "ck_internal.h"

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

}  // namespace

extern "C" int ck_sum_lse_bwd_c(const float* arena_c, float* garena_c, const int64_t* row_off, const float* w, const float* out_c,
                                const float* gout_c, float* dw, int F, int H, int B, int Ki, int Ko, int mode, int w_is_complex,
                                void* stream) {
  CK_REQUIRE(arena_c && garena_c && row_off && w && out_c && gout_c && dw, "ck_sum_lse_bwd_c: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && Ki > 0 && Ko > 0, "ck_sum_lse_bwd_c: non-positive size");
  CK_REQUIRE(mode == CK_SUM_CAT || mode == CK_SUM_PROD || mode == CK_SUM_KRON, "ck_sum_lse_bwd_c: unknown mode %d", mode);
  CK_REQUIRE(F <= 65535, "ck_sum_lse_bwd_c: F=%d exceeds grid.y", F);
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
