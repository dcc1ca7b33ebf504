// Backward pass of the log-space forward: gradients of a scalar loss with
// respect to every layer input and every parameter, for the real lse-sum semiring.
//
// The reference obtains these from autograd through LSESumSemiring.apply_reduce
// (semiring.py:383-408): with v the (concatenated / multiplied) inputs, m = max v (a constant for
// differentiation purposes: the result does not depend on it), e = exp(v - m), y = W e and
// out = log y + m,
//     d out_o / d v_n  = W[o,n] e_n / y_o            d out_o / d W[o,n] = e_n / y_o
// so with gy_o = gout_o / y_o = gout_o exp(m - out_o):
//     gv_n = e_n * sum_o W[o,n] gy_o                 dW[o,n] += sum_b gy[b,o] e[b,n]
// and through the softmax parameterisation W = softmax(theta) (nodes.py:764-772):
//     dtheta[o,n] = W[o,n] (dW[o,n] - sum_n' W[o,n'] dW[o,n']).
// Children gradients are written (`accumulate` 0), added (1: a producer fold read by several
// layers, launches are ordered) or atomically added (2: read several times within this layer).
#include <algorithm>

#include "ck_internal.h"

namespace {

constexpr int kBwdNC = 32;

__device__ __forceinline__ void grad_store(float* p, float g, int accumulate) {
  if (accumulate == 0)
    *p = g;
  else if (accumulate == 1)
    *p += g;
  else
    atomicAdd(p, g);
}

// Generic sum-layer backward: any H, Ki, Ko; cat (mode 0) or product (mode 1) inputs.
// Block = 256 threads, tile = TB batch rows of one fold; LDS: e[TB][N], gy[TB][Ko], gv[TB][N],
// W chunk [64][kBwdNC+1].
template <int TB>
__global__ void __launch_bounds__(256)
    sum_lse_bwd_generic(const float* __restrict__ arena, float* __restrict__ garena,
                        const int64_t* __restrict__ row_off, const float* __restrict__ w,
                        const float* __restrict__ /*out: not needed, y is recomputed*/,
                        const float* __restrict__ gout, float* __restrict__ dw, int H, int B, int Ki,
                        int Ko, int mode, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = mode == CK_SUM_PROD ? Ki : H * Ki;
  float* e_s = reinterpret_cast<float*>(smem);   // [TB][N]
  float* gv_s = e_s + static_cast<size_t>(TB) * N;  // [TB][N]
  float* gy_s = gv_s + static_cast<size_t>(TB) * N;  // [TB][Ko]
  float* w_s = gy_s + static_cast<size_t>(TB) * Ko;  // [64][kBwdNC+1]
  float* m_s = w_s + 64 * (kBwdNC + 1);              // [TB]
  const int f = blockIdx.y;
  const int b0 = blockIdx.x * TB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* wf = w + static_cast<int64_t>(f) * Ko * N;
  float* dwf = dw + static_cast<int64_t>(f) * Ko * N;

  // phase A: v, m, e (rows beyond B are zeroed so they contribute nothing)
  for (int r = wave; r < TB; r += 4) {
    const int b = b0 + r;
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) {
      float v = -INFINITY;
      if (b < B) {
        if (mode == CK_SUM_PROD) {
          v = arena[ro[0] + static_cast<int64_t>(b) * Ki + n];
          for (int h = 1; h < H; ++h) v += arena[ro[h] + static_cast<int64_t>(b) * Ki + n];
        } else {
          const int h = n / Ki, k = n - h * Ki;
          v = arena[ro[h] + static_cast<int64_t>(b) * Ki + k];
        }
      }
      e_s[static_cast<size_t>(r) * N + n] = v;
      mx = fmaxf(mx, v);
    }
    mx = ck::clamp_finite(ck::wave_max(mx));
    for (int n = lane; n < N; n += 64) e_s[static_cast<size_t>(r) * N + n] = expf(e_s[static_cast<size_t>(r) * N + n] - mx);
    if (lane == 0) m_s[r] = mx;
  }
  __syncthreads();
  // phase Y: recompute y = W e in linear space.  (gy = gout * exp(m - out) would inherit the rounding
  // of `out` at ITS magnitude -- half an ulp of |out| ~ 4e3 is 2e-4 relative in y -- whereas autograd
  // in the reference divides by the fp32 y itself.)
  for (int i = threadIdx.x; i < TB * Ko; i += 256) gy_s[i] = 0.f;
  __syncthreads();
  for (int o0 = 0; o0 < Ko; o0 += 64) {
    for (int n0 = 0; n0 < N; n0 += kBwdNC) {
      for (int i = threadIdx.x; i < 64 * kBwdNC; i += 256) {
        const int oo = i / kBwdNC, nn = i - oo * kBwdNC;
        w_s[oo * (kBwdNC + 1) + nn] = (o0 + oo < Ko && n0 + nn < N) ? wf[static_cast<int64_t>(o0 + oo) * N + n0 + nn] : 0.f;
      }
      __syncthreads();
      const int omax = min(64, Ko - o0), nmax = min(kBwdNC, N - n0);
      for (int i = threadIdx.x; i < TB * 64; i += 256) {
        const int r = i >> 6, oo = i & 63;
        if (oo < omax) {
          float acc = 0.f;
          for (int nn = 0; nn < nmax; ++nn) acc = fmaf(w_s[oo * (kBwdNC + 1) + nn], e_s[static_cast<size_t>(r) * N + n0 + nn], acc);
          gy_s[r * Ko + o0 + oo] += acc;
        }
      }
      __syncthreads();
    }
  }
  // phase G: gy = gout / y   (0 where y = 0 or the row is padding)
  for (int i = threadIdx.x; i < TB * Ko; i += 256) {
    const int r = i / Ko, o = i - r * Ko, b = b0 + r;
    float g = 0.f;
    if (b < B) {
      const float y = gy_s[i];
      const float go = gout[(static_cast<int64_t>(f) * B + b) * Ko + o];
      if (y > 0.f && go != 0.f) g = go / y;
    }
    gy_s[i] = g;
  }
  for (int i = threadIdx.x; i < TB * N; i += 256) gv_s[i] = 0.f;
  __syncthreads();
  // phase B: stream W in [64 outputs][32 inputs] chunks
  for (int o0 = 0; o0 < Ko; o0 += 64) {
    for (int n0 = 0; n0 < N; n0 += kBwdNC) {
      for (int i = threadIdx.x; i < 64 * kBwdNC; i += 256) {
        const int oo = i / kBwdNC, nn = i - oo * kBwdNC;
        w_s[oo * (kBwdNC + 1) + nn] = (o0 + oo < Ko && n0 + nn < N) ? wf[static_cast<int64_t>(o0 + oo) * N + n0 + nn] : 0.f;
      }
      __syncthreads();
      const int omax = min(64, Ko - o0), nmax = min(kBwdNC, N - n0);
      // gv[r][n] += sum_o W[o][n] gy[r][o]
      for (int i = threadIdx.x; i < TB * kBwdNC; i += 256) {
        const int r = i / kBwdNC, nn = i - r * kBwdNC;
        if (nn < nmax) {
          float acc = 0.f;
          for (int oo = 0; oo < omax; ++oo) acc = fmaf(w_s[oo * (kBwdNC + 1) + nn], gy_s[r * Ko + o0 + oo], acc);
          gv_s[static_cast<size_t>(r) * N + n0 + nn] += acc;
        }
      }
      // dW[o][n] += sum_r gy[r][o] e[r][n]
      for (int i = threadIdx.x; i < 64 * kBwdNC; i += 256) {
        const int oo = i / kBwdNC, nn = i - oo * kBwdNC;
        if (oo < omax && nn < nmax) {
          float acc = 0.f;
          for (int r = 0; r < TB; ++r) acc = fmaf(gy_s[r * Ko + o0 + oo], e_s[static_cast<size_t>(r) * N + n0 + nn], acc);
          if (acc != 0.f) atomicAdd(&dwf[static_cast<int64_t>(o0 + oo) * N + n0 + nn], acc);
        }
      }
      __syncthreads();
    }
  }
  // phase C: gv *= e, scatter to the children
  for (int i = threadIdx.x; i < TB * N; i += 256) {
    const int r = i / N, n = i - r * N, b = b0 + r;
    if (b >= B) continue;
    const float g = gv_s[i] * e_s[i];
    if (mode == CK_SUM_PROD) {
      for (int h = 0; h < H; ++h) grad_store(garena + ro[h] + static_cast<int64_t>(b) * Ki + n, g, accumulate);
    } else {
      const int h = n / Ki, k = n - h * Ki;
      grad_store(garena + ro[h] + static_cast<int64_t>(b) * Ki + k, g, accumulate);
    }
  }
}

// Hadamard backward: every child receives the output gradient.
__global__ void __launch_bounds__(256)
    hadamard_bwd_kernel(float* __restrict__ garena, const int64_t* __restrict__ row_off,
                        const float* __restrict__ gout, int H, int64_t words_per_fold, int accumulate) {
  const int f = blockIdx.y;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* g = gout + static_cast<int64_t>(f) * words_per_fold;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < words_per_fold;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float v = g[i];
    for (int h = 0; h < H; ++h) grad_store(garena + ro[h] + i, v, accumulate);
  }
}

// Categorical backward: dtable[f, c, :] += sum over rows b with x[b] = c of gout[f, b, :].
// One workgroup per fold accumulates a (C, K) histogram in LDS (ds_add_f32), then writes it once.
__global__ void __launch_bounds__(256)
    categorical_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ xt,
                           const int64_t* __restrict__ scope, float* __restrict__ dtable, int B, int K,
                           int C) {
  extern __shared__ __attribute__((aligned(16))) float hist[];  // [C+1][K] (row C: marginalised rows)
  const int f = blockIdx.x;
  for (int i = threadIdx.x; i < (C + 1) * K; i += blockDim.x) hist[i] = 0.f;
  __syncthreads();
  const int32_t* xrow = xt + scope[f] * static_cast<int64_t>(B);
  const float* g = gout + static_cast<int64_t>(f) * B * K;
  for (int64_t i = threadIdx.x; i < static_cast<int64_t>(B) * K; i += blockDim.x) {
    const int b = static_cast<int>(i / K), k = static_cast<int>(i - static_cast<int64_t>(b) * K);
    int c = xrow[b];
    c = c < 0 ? C : min(c, C - 1);
    const float v = g[i];
    if (v != 0.f) atomicAdd(&hist[c * K + k], v);
  }
  __syncthreads();
  float* dst = dtable + static_cast<int64_t>(f) * (C + 1) * K;
  for (int i = threadIdx.x; i < (C + 1) * K; i += blockDim.x) dst[i] += hist[i];
}

// softmax backward over rows: dtheta = W * (dW - sum(W * dW));  W given row-major (rows, len).
__global__ void __launch_bounds__(256)
    softmax_bwd_rows_kernel(const float* __restrict__ w, const float* __restrict__ dw,
                            float* __restrict__ dtheta, int64_t rows, int len, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* wr = w + row * len;
  const float* dr = dw + row * len;
  float dot = 0.f;
  for (int i = lane; i < len; i += 64) dot = fmaf(wr[i], dr[i], dot);
  dot = ck::wave_sum(dot);
  for (int i = lane; i < len; i += 64) {
    const float g = wr[i] * (dr[i] - dot);
    if (accumulate)
      dtheta[row * len + i] += g;
    else
      dtheta[row * len + i] = g;
  }
}

// Categorical parameter backward: table (F, C, K) = log softmax_C(theta (F, K, C)) transposed.
// dtheta[f,k,c] = dT[f,c,k] - exp(T[f,c,k]) * sum_c' dT[f,c',k].   One block per fold.
__global__ void __launch_bounds__(256)
    log_table_bwd_kernel(const float* __restrict__ table, const float* __restrict__ dtable,
                         float* __restrict__ dtheta, int K, int C, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float colsum[];  // [K]
  const int f = blockIdx.x;
  const float* T = table + static_cast<int64_t>(f) * (C + 1) * K;  // rows 0..C-1; row C (integral, = 0) has no gradient
  const float* dT = dtable + static_cast<int64_t>(f) * (C + 1) * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dT[c * K + k];
    colsum[k] = s;
  }
  __syncthreads();
  float* dst = dtheta + static_cast<int64_t>(f) * K * C;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    const int k = i / C, c = i - k * C;
    const float g = dT[c * K + k] - expf(T[c * K + k]) * colsum[k];
    if (accumulate)
      dst[i] += g;
    else
      dst[i] = g;
  }
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ p, int64_t n, float v) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

// Adam step (torch.optim.Adam defaults semantics, no weight decay / amsgrad), fused over one tensor.
__global__ void __launch_bounds__(256)
    adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1,
                float* __restrict__ m2, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                float bc2, float gscale) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float gi = g[i] * gscale;
    const float a = b1 * m1[i] + (1.f - b1) * gi;
    const float v = b2 * m2[i] + (1.f - b2) * gi * gi;
    m1[i] = a;
    m2[i] = v;
    p[i] -= lr * (a / bc1) / (sqrtf(v / bc2) + eps);
  }
}

__global__ void __launch_bounds__(256)
    sgd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr, float gscale) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] -= lr * gscale * g[i];
}

unsigned grid1(int64_t n, int cap = 2048) { return static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, cap)); }

}  // namespace

extern "C" {

int ck_fill_f32(float* p, int64_t n, float value, void* stream) {
  CK_REQUIRE(p != nullptr && n > 0, "ck_fill_f32: bad arguments");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(fill_kernel, grid, block, 0, s, p, n, value);
        return hipGetLastError();
      },
      stream);
}

int ck_sum_lse_bwd(const float* arena, float* garena, const int64_t* row_off, const float* w,
                   const float* out, const float* gout, float* dw, int F, int H, int B, int Ki, int Ko,
                   int mode, int accumulate, void* stream) {
  CK_REQUIRE(arena && garena && row_off && w && out && gout && dw, "ck_sum_lse_bwd: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && Ki > 0 && Ko > 0, "ck_sum_lse_bwd: non-positive size");
  CK_REQUIRE(mode == CK_SUM_CAT || mode == CK_SUM_PROD, "ck_sum_lse_bwd: unsupported mode %d", mode);
  CK_REQUIRE(accumulate >= 0 && accumulate <= 2, "ck_sum_lse_bwd: accumulate must be 0, 1 or 2");
  CK_REQUIRE(F <= 65535, "ck_sum_lse_bwd: F=%d exceeds grid.y", F);
  const int N = mode == CK_SUM_PROD ? Ki : H * Ki;
  auto lds_bytes = [&](int tb) {
    return (static_cast<size_t>(2) * tb * N + static_cast<size_t>(tb) * Ko + 64 * (kBwdNC + 1) + tb) * sizeof(float);
  };
  int tb = 16;
  while (tb > 4 && lds_bytes(tb) > 64 * 1024) tb >>= 1;
  const size_t lds = lds_bytes(tb);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_sum_lse_bwd: N=%d does not fit in LDS", N);
  dim3 grid((B + tb - 1) / tb, F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern) {
          if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
          }
          hipLaunchKernelGGL(kern, grid, block, lds, s, arena, garena, row_off, w, out, gout, dw, H, B, Ki, Ko, mode,
                             accumulate);
          return hipGetLastError();
        };
        if (tb == 16) return go(sum_lse_bwd_generic<16>);
        if (tb == 8) return go(sum_lse_bwd_generic<8>);
        return go(sum_lse_bwd_generic<4>);
      },
      stream);
}

int ck_hadamard_bwd(float* garena, const int64_t* row_off, const float* gout, int F, int H, int B, int K,
                    int accumulate, void* stream) {
  CK_REQUIRE(garena && row_off && gout, "ck_hadamard_bwd: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && K > 0, "ck_hadamard_bwd: non-positive size");
  CK_REQUIRE(accumulate >= 0 && accumulate <= 2, "ck_hadamard_bwd: accumulate must be 0, 1 or 2");
  CK_REQUIRE(F <= 65535, "ck_hadamard_bwd: F=%d exceeds grid.y", F);
  const int64_t words = static_cast<int64_t>(B) * K;
  dim3 grid(grid1(words), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(hadamard_bwd_kernel, grid, block, 0, s, garena, row_off, gout, H, words, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_categorical_bwd(const float* gout, const int32_t* xt, const int64_t* scope, float* dtable, int F,
                       int B, int K, int C, void* stream) {
  CK_REQUIRE(gout && xt && scope && dtable, "ck_categorical_bwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && K > 0 && C > 0, "ck_categorical_bwd: non-positive size");
  const size_t lds = static_cast<size_t>(C + 1) * K * sizeof(float);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_categorical_bwd: C*K=%d does not fit in LDS", C * K);
  dim3 grid(F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (lds > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(categorical_bwd_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
          if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(categorical_bwd_kernel, grid, block, lds, s, gout, xt, scope, dtable, B, K, C);
        return hipGetLastError();
      },
      stream);
}

int ck_param_softmax_bwd(const float* w, const float* dw, float* dtheta, int64_t rows, int len,
                         int accumulate, void* stream) {
  CK_REQUIRE(w && dw && dtheta, "ck_param_softmax_bwd: null pointer");
  CK_REQUIRE(rows > 0 && len > 0, "ck_param_softmax_bwd: non-positive size");
  dim3 grid(static_cast<unsigned>((rows + 3) / 4)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(softmax_bwd_rows_kernel, grid, block, 0, s, w, dw, dtheta, rows, len, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_param_log_table_bwd(const float* table, const float* dtable, float* dtheta, int F, int K, int C,
                           int accumulate, void* stream) {
  CK_REQUIRE(table && dtable && dtheta, "ck_param_log_table_bwd: null pointer");
  CK_REQUIRE(F > 0 && K > 0 && C > 0, "ck_param_log_table_bwd: non-positive size");
  dim3 grid(F), block(256);
  const size_t lds = static_cast<size_t>(K) * sizeof(float);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(log_table_bwd_kernel, grid, block, lds, s, table, dtable, dtheta, K, C, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_adam_step(float* p, const float* g, float* m1, float* m2, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, void* stream) {
  CK_REQUIRE(p && g && m1 && m2, "ck_adam_step: null pointer");
  CK_REQUIRE(n > 0 && step > 0, "ck_adam_step: n and step must be positive");
  const float bc1 = 1.f - powf(beta1, static_cast<float>(step));
  const float bc2 = 1.f - powf(beta2, static_cast<float>(step));
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(adam_kernel, grid, block, 0, s, p, g, m1, m2, n, lr, beta1, beta2, eps, bc1, bc2, grad_scale);
        return hipGetLastError();
      },
      stream);
}

int ck_sgd_step(float* p, const float* g, int64_t n, float lr, float grad_scale, void* stream) {
  CK_REQUIRE(p && g, "ck_sgd_step: null pointer");
  CK_REQUIRE(n > 0, "ck_sgd_step: n must be positive");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(sgd_kernel, grid, block, 0, s, p, g, n, lr, grad_scale);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
