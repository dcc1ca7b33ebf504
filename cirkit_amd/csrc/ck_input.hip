// Input layers: batch staging (transpose), Categorical, Gaussian, Embedding, ConstantValue.
//
// All of these are pure HBM-write-bound streams: one (f, b) row of K outputs is produced from one
// scalar of the batch plus a K-row of per-fold parameters that stays in L2/Infinity Cache.  The
// kernels therefore (a) read the batch through a (D, B) transposed copy so the scalar loads are
// contiguous along B, (b) keep the parameter row contiguous in K (tables are (F, C, K)), and
// (c) write float4 per lane, a whole 256-row x K tile per workgroup, contiguous in HBM.
#include <algorithm>

#include "ck_internal.h"

namespace {

constexpr int kTile = 32;

template <typename TI, typename TO>
__global__ void transpose_kernel(const TI* __restrict__ x, TO* __restrict__ xt, int B, int D) {
  __shared__ TO tile[kTile][kTile + 1];
  const int d0 = blockIdx.x * kTile, b0 = blockIdx.y * kTile;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
#pragma unroll
  for (int j = 0; j < kTile; j += 8) {
    const int b = b0 + ty + j, d = d0 + tx;
    if (b < B && d < D) tile[ty + j][tx] = static_cast<TO>(x[static_cast<int64_t>(b) * D + d]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kTile; j += 8) {
    const int d = d0 + ty + j, b = b0 + tx;
    if (b < B && d < D) xt[static_cast<int64_t>(d) * B + b] = tile[tx][ty + j];
  }
}

// (B, D) int64 -> (D, B) int32 like transpose_kernel, and input validation on the way: a category >= num_states[d]
// (num_states[d] > 0) raises the sticky flag -- the reference's advanced indexing raises IndexError there
// (layers/input.py:399-412); the consumers clamp for memory safety, the circuit's last launch turns the flag into NaN outputs.
__global__ void stage_categories_kernel(const int64_t* __restrict__ x, int32_t* __restrict__ xt, int B, int D,
                                        const int32_t* __restrict__ num_states, int32_t* __restrict__ flag, int clamp,
                                        int64_t* __restrict__ x_copy) {
  __shared__ int32_t tile[kTile][kTile + 1];
  const int d0 = blockIdx.x * kTile, b0 = blockIdx.y * kTile;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
  bool bad = false;
#pragma unroll
  for (int j = 0; j < kTile; j += 8) {
    const int b = b0 + ty + j, d = d0 + tx;
    if (b < B && d < D) {
      const int64_t v = x[static_cast<int64_t>(b) * D + d];
      if (x_copy != nullptr) x_copy[static_cast<int64_t>(b) * D + d] = v;  // (the batch at an address of the caller's: recorded launches read it)
      const int ns = num_states[d];
      bad |= ns > 0 && v >= ns;
      // clamp: an out-of-range category is stored as the last one (what every consumer would make of it), a negative
      // value as -1: the 32-bit staging copy then holds -1 .. ns - 1 only
      tile[ty + j][tx] = (clamp && ns > 0) ? static_cast<int32_t>(v < 0 ? -1 : (v >= ns ? ns - 1 : v)) : static_cast<int32_t>(v);
    }
  }
  if (__any(bad) && ((threadIdx.y * 32 + threadIdx.x) & 63) == 0) atomicOr(flag, 1);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kTile; j += 8) {
    const int d = d0 + ty + j, b = b0 + tx;
    if (b < B && d < D) xt[static_cast<int64_t>(d) * B + b] = tile[tx][ty + j];
  }
}

__global__ void poison_kernel(float* __restrict__ out, int64_t n, const int32_t* __restrict__ flag) {
  if (*flag == 0) return;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    out[i] = __builtin_nanf("");
}

__global__ void zero_if_flag_kernel(float* __restrict__ p, int64_t n, const int32_t* __restrict__ flag) {
  if (*flag == 0) return;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = 0.f;
}

template <typename TI, typename TO>
int transpose_impl(const TI* x, TO* xt, int B, int D, void* stream, const char* who) {
  CK_REQUIRE(x != nullptr && xt != nullptr, "%s: null pointer", who);
  CK_REQUIRE(B > 0 && D > 0, "%s: B=%d D=%d must be positive", who, B, D);
  dim3 grid((D + kTile - 1) / kTile, (B + kTile - 1) / kTile), block(kTile, 8);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, block, 0, s, x, xt, B, D);
        return hipGetLastError();
      },
      stream);
}

// ---- gather-type input layers ------------------------------------------------------------------
// MODE 0: copy table row (categorical, lse-sum)
// MODE 1: log of table row (embedding, lse-sum)
// MODE 2: complex log of the real table row (embedding, complex-lse-sum): (log|w|, w<0 ? pi : 0)
// MODE 3: table row as a complex number with phase 0 (categorical, complex-lse-sum: the real log-likelihood mapped into
//         the complex semiring, layers/input.py:276-278 + semiring.py:512-514)
// MODE 4: complex log of a COMPLEX table row (embedding with complex weights under complex-lse-sum, input.py:258-266 +
//         utils.py:32-35: torch.log of a complex number = (log|w|, arg w)); the row is K = 2 x units floats (re, im) and so
//         is the output
__device__ __forceinline__ float4 log4(float4 v) {
  return make_float4(__logf(v.x), __logf(v.y), __logf(v.z), __logf(v.w));
}

// vector path: K % 4 == 0.  One lane = 4 consecutive units of one (f, b) row.
template <int MODE>
__global__ void __launch_bounds__(256)
    gather_rows_vec(const float* __restrict__ table, const int32_t* __restrict__ xt,
                    const int64_t* __restrict__ scope, float* __restrict__ out, int B, int K, int C,
                    int rows_per_block) {
  const int f = blockIdx.y;
  const int kv = K >> 2;                      // float4 per row
  const int lanes_rows = blockDim.x / kv;     // rows handled per pass
  const int r_in = threadIdx.x / kv, q = threadIdx.x - r_in * kv;
  if (r_in >= lanes_rows) return;
  const int64_t var = scope[f];
  const int32_t* xrow = xt + var * B;
  // tables carry C + 1 rows per fold: row C is the fold's integral (log-partition) row, selected by
  // a negative category = "this variable is marginalised" (IntegrateQuery, queries.py:103-150)
  const float4* tab = reinterpret_cast<const float4*>(table + static_cast<int64_t>(f) * (C + 1) * K);
  const int b_begin = blockIdx.x * rows_per_block;
  const int b_end = min(B, b_begin + rows_per_block);
  for (int b = b_begin + r_in; b < b_end; b += lanes_rows) {
    int c = xrow[b];
    c = c < 0 ? C : min(c, C - 1);  // memory safety; the reference raises on out-of-range categories
    float4 v = tab[static_cast<int64_t>(c) * kv + q];
    const int64_t o = (static_cast<int64_t>(f) * B + b) * K + 4 * q;
    if (MODE == 0) {
      *reinterpret_cast<float4*>(out + o) = v;
    } else if (MODE == 1) {
      *reinterpret_cast<float4*>(out + o) = log4(v);
    } else if (MODE == 3) {
      float4* oc = reinterpret_cast<float4*>(out + 2 * o);
      oc[0] = make_float4(v.x, 0.f, v.y, 0.f);
      oc[1] = make_float4(v.z, 0.f, v.w, 0.f);
    } else if (MODE == 4) {
      *reinterpret_cast<float4*>(out + o) = make_float4(logf(hypotf(v.x, v.y)), atan2f(v.y, v.x), logf(hypotf(v.z, v.w)), atan2f(v.w, v.z));
    } else {
      const float pi = 3.14159265358979323846f;
      float4 l = make_float4(logf(fabsf(v.x)), logf(fabsf(v.y)), logf(fabsf(v.z)), logf(fabsf(v.w)));
      float4* oc = reinterpret_cast<float4*>(out + 2 * o);
      oc[0] = make_float4(l.x, v.x < 0.f ? pi : 0.f, l.y, v.y < 0.f ? pi : 0.f);
      oc[1] = make_float4(l.z, v.z < 0.f ? pi : 0.f, l.w, v.w < 0.f ? pi : 0.f);
    }
  }
}

// scalar path: any K.
template <int MODE>
__global__ void __launch_bounds__(256)
    gather_rows_scalar(const float* __restrict__ table, const int32_t* __restrict__ xt,
                       const int64_t* __restrict__ scope, float* __restrict__ out, int B, int K,
                       int C) {
  const int f = blockIdx.y;
  const int64_t var = scope[f];
  const int64_t n = static_cast<int64_t>(B) * K;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / K), k = static_cast<int>(i - static_cast<int64_t>(b) * K);
    int c = xt[var * B + b];
    c = c < 0 ? C : min(c, C - 1);
    const float v = table[(static_cast<int64_t>(f) * (C + 1) + c) * K + k];
    const int64_t o = static_cast<int64_t>(f) * n + i;
    if (MODE == 0) {
      out[o] = v;
    } else if (MODE == 1) {
      out[o] = __logf(v);
    } else if (MODE == 3) {
      out[2 * o] = v;
      out[2 * o + 1] = 0.f;
    } else if (MODE == 4) {  // (k even: the real part of a unit; its imaginary part is the next float)
      if ((k & 1) == 0) {
        const float im = table[(static_cast<int64_t>(f) * (C + 1) + c) * K + k + 1];
        out[o] = logf(hypotf(v, im));
        out[o + 1] = atan2f(im, v);
      }
    } else {
      out[2 * o] = logf(fabsf(v));
      out[2 * o + 1] = v < 0.f ? 3.14159265358979323846f : 0.f;
    }
  }
}

template <int MODE>
int gather_impl(const float* table, const int32_t* xt, const int64_t* scope, float* out, int F,
                int B, int K, int C, int D, void* stream, const char* who) {
  CK_REQUIRE(table && xt && scope && out, "%s: null pointer", who);
  CK_REQUIRE(F > 0 && B > 0 && K > 0 && C > 0 && D > 0, "%s: non-positive size F=%d B=%d K=%d C=%d D=%d",
             who, F, B, K, C, D);
  if (F > ck::kMaxFoldsPerLaunch)
    return ck::chunk_folds(F, [&](int f0, int n) {
      return gather_impl<MODE>(table + static_cast<int64_t>(f0) * (C + 1) * K, xt, scope + f0,
                               out + static_cast<int64_t>(f0) * B * K * (MODE == 2 || MODE == 3 ? 2 : 1), n, B, K, C, D, stream, who);
    });
  const bool vec = (K % 4 == 0) && (K / 4 <= 256) && ck::aligned16(table) && ck::aligned16(out);
  if (vec) {
    const int rows_per_block = 256;
    dim3 grid((B + rows_per_block - 1) / rows_per_block, F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL((gather_rows_vec<MODE>), grid, block, 0, s, table, xt, scope, out, B, K,
                             C, rows_per_block);
          return hipGetLastError();
        },
        stream);
  }
  const int64_t n = static_cast<int64_t>(B) * K;
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 4096)), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL((gather_rows_scalar<MODE>), grid, block, 0, s, table, xt, scope, out, B, K, C);
        return hipGetLastError();
      },
      stream);
}

// ---- Gaussian ----------------------------------------------------------------------------------
// Normal(mean, stddev).log_prob(x) exactly as torch.distributions.Normal computes it:
//   -((x - mean)**2) / (2 * stddev**2) - log(stddev) - log(sqrt(2*pi))
__global__ void __launch_bounds__(256)
    gaussian_kernel(const float* __restrict__ mean, const float* __restrict__ stddev,
                    const float* __restrict__ logz, const float* __restrict__ xt,
                    const int64_t* __restrict__ scope, float* __restrict__ out, int B, int K,
                    int rows_per_block) {
  const int f = blockIdx.y;
  const int kk = K <= 256 ? K : 256;  // lanes along the unit axis
  const int lanes_rows = 256 / kk;    // rows per pass
  const int r_in = threadIdx.x / kk, k0 = threadIdx.x - r_in * kk;
  if (r_in >= lanes_rows) return;
  const float* xrow = xt + scope[f] * B;
  const int b_begin = blockIdx.x * rows_per_block;
  const int b_end = min(B, b_begin + rows_per_block);
  const float kHalfLog2Pi = 0.91893853320467274178f;
  for (int k = k0; k < K; k += kk) {
    const float mu = mean[static_cast<int64_t>(f) * K + k];
    const float sd = stddev[static_cast<int64_t>(f) * K + k];
    // 1 / (2 sd^2) once per unit: a division per (row, unit) made these kernels VALU-bound (one more rounding than
    // dividing, ~1e-7 relative on that term)
    const float inv_two_var = 1.f / (2.f * (sd * sd));
    const float tail = __logf(sd);
    const float lz = logz != nullptr ? logz[static_cast<int64_t>(f) * K + k] : 0.f;
    for (int b = b_begin + r_in; b < b_end; b += lanes_rows) {
      const float xv = xrow[b];
      const float d = xv - mu;
      float lp = -(d * d) * inv_two_var - tail - kHalfLog2Pi;
      if (logz != nullptr) lp += lz;
      // NaN = "marginalised": the layer's integral, log_partition or 0 (input.py:672-679)
      if (xv != xv) lp = logz != nullptr ? lz : 0.f;
      out[(static_cast<int64_t>(f) * B + b) * K + k] = lp;
    }
  }
}

// Fully factorised multivariate input: the Hadamard product of H Gaussian folds (what
// RegionGraph.build_circuit emits for an input region over several variables with
// factorize_multivariate=True, templates/region_graph/graph.py:531-540) without writing the
// H (B, K) blocks of the Gaussian layer:  out[f,b,k] = sum_j logN(x[b, var(g_j)]; mean[g_j,k], stddev[g_j,k]),
// g_j = gfold[f, j].  Thread = one unit k and RPT batch rows; the parameters of (g_j, k) are loaded
// once per thread and x[b, var] is a broadcast read.
template <int RPT>
__global__ void __launch_bounds__(256)
    gaussian_prod_kernel(const float* __restrict__ mean, const float* __restrict__ stddev,
                         const float* __restrict__ logz, const float* __restrict__ xt,
                         const int64_t* __restrict__ scope, const int32_t* __restrict__ gfold,
                         float* __restrict__ out, int H, int B, int K) {
  const int f = blockIdx.y;
  const int kk = K <= 256 ? K : 256;
  const int lanes_rows = 256 / kk;
  const int r_in = threadIdx.x / kk, k0 = threadIdx.x - r_in * kk;
  if (r_in >= lanes_rows) return;
  const int b0 = blockIdx.x * (lanes_rows * RPT) + r_in;
  const float kHalfLog2Pi = 0.91893853320467274178f;
  const int32_t* gf = gfold + static_cast<int64_t>(f) * H;
  for (int k = k0; k < K; k += kk) {
    float acc[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) acc[i] = 0.f;
    for (int j = 0; j < H; ++j) {
      const int64_t g = gf[j];
      const float mu = mean[g * K + k];
      const float sd = stddev[g * K + k];
      const float inv_two_var = 1.f / (2.f * (sd * sd));
      const float tail = __logf(sd);
      const float lz = logz != nullptr ? logz[g * K + k] : 0.f;
      const float* xrow = xt + scope[g] * B;
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int b = min(b0 + i * lanes_rows, B - 1);
        const float xv = xrow[b];
        const float d = xv - mu;
        float lp = -(d * d) * inv_two_var - tail - kHalfLog2Pi;
        if (logz != nullptr) lp += lz;
        if (xv != xv) lp = logz != nullptr ? lz : 0.f;  // NaN = marginalised (input.py:672-679)
        acc[i] += lp;
      }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int b = b0 + i * lanes_rows;
      if (b < B) out[(static_cast<int64_t>(f) * B + b) * K + k] = acc[i];
    }
  }
}

// The same with 16 CONSECUTIVE batch rows per thread (B a multiple of 4): the rows' values of a variable arrive as four
// 16-byte loads (broadcast over the K lanes of the row group) where the kernel above issues one load per row and factor, and
// the per-(factor, unit) constants -- 1 / (2 sigma^2), log sigma -- are amortised over 16 rows.  BASELINE config 4's first
// launch (49 folds x 16 factors, 4096 rows, 64 units): 81 -> ~25 us.
__global__ void __launch_bounds__(256)
    gaussian_prod_rows16_kernel(const float* __restrict__ mean, const float* __restrict__ stddev, const float* __restrict__ logz,
                                const float* __restrict__ xt, const int64_t* __restrict__ scope, const int32_t* __restrict__ gfold,
                                float* __restrict__ out, int H, int B, int K) {
  const int f = blockIdx.y;
  const int groups = 256 / K;  // row groups per block (K = 32, 64, 128, 256)
  const int r_in = threadIdx.x / K, k = threadIdx.x - r_in * K;
  const int b0 = (blockIdx.x * groups + r_in) * 16;
  if (b0 >= B) return;
  const float kHalfLog2Pi = 0.91893853320467274178f;
  const int32_t* gf = gfold + static_cast<int64_t>(f) * H;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int j = 0; j < H; ++j) {
    const int64_t g = gf[j];
    const float mu = mean[g * K + k];
    const float sd = stddev[g * K + k];
    const float inv_two_var = 1.f / (2.f * (sd * sd));
    const float lz = logz != nullptr ? logz[g * K + k] : 0.f;
    const float c0 = lz - __logf(sd) - kHalfLog2Pi;
    const float* xrow = xt + scope[g] * B + b0;
    float xv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = b0 + 4 * q < B ? *reinterpret_cast<const float4*>(xrow + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      xv[4 * q] = t.x, xv[4 * q + 1] = t.y, xv[4 * q + 2] = t.z, xv[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float d = xv[i] - mu;
      float lp = fmaf(-(d * d), inv_two_var, c0);
      if (xv[i] != xv[i]) lp = lz;  // NaN = marginalised (input.py:672-679)
      acc[i] += lp;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (b0 + i < B) out[(static_cast<int64_t>(f) * B + b0 + i) * K + k] = acc[i];
}

// ---- lse-sum -> complex-lse-sum ---------------------------------------------------------------------
// ComplexLSESumSemiring.map_from(x, LSESumSemiring) = x.to(complex) (semiring.py:512-514): (x, 0)
__global__ void __launch_bounds__(256) lse_to_clse_kernel(const float* __restrict__ in, float2* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    out[i] = make_float2(in[i], 0.f);
}

// ---- ConstantValue -------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    constant_kernel(const float* __restrict__ value, float* __restrict__ out, int B, int K,
                    int log_space, int value_is_complex, int complex_out) {
  const int f = blockIdx.y;
  const int64_t n = static_cast<int64_t>(B) * K;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int k = static_cast<int>(i % K);
    const int64_t vi = static_cast<int64_t>(f) * K + k;
    ck::c32 v;
    if (value_is_complex) {
      v = {value[2 * vi], value[2 * vi + 1]};
    } else {
      v = {value[vi], 0.f};
    }
    const int64_t o = static_cast<int64_t>(f) * n + i;
    if (complex_out) {
      ck::c32 r = log_space ? v : ck::c_log_shift(v, 0.f);
      out[2 * o] = r.re;
      out[2 * o + 1] = r.im;
    } else {
      out[o] = log_space ? v.re : __logf(v.re);
    }
  }
}

}  // namespace

extern "C" {

int ck_transpose_i64_to_i32(const int64_t* x, int32_t* xt, int B, int D, void* stream) {
  return transpose_impl<int64_t, int32_t>(x, xt, B, D, stream, "ck_transpose_i64_to_i32");
}

int ck_stage_categories(const int64_t* x, int32_t* xt, int B, int D, const int32_t* num_states, int32_t* flag, int clamp,
                        int64_t* x_copy, void* stream) {
  CK_REQUIRE(x && xt && num_states && flag, "ck_stage_categories: null pointer");
  CK_REQUIRE(B > 0 && D > 0, "ck_stage_categories: B=%d D=%d must be positive", B, D);
  dim3 grid((D + kTile - 1) / kTile, (B + kTile - 1) / kTile), block(kTile, 8);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(stage_categories_kernel, grid, block, 0, s, x, xt, B, D, num_states, flag, clamp, x_copy);
        return hipGetLastError();
      },
      stream);
}

int ck_poison_outputs(float* out, int64_t n, const int32_t* flag, void* stream) {
  CK_REQUIRE(out && flag && n > 0, "ck_poison_outputs: null pointer or empty output");
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 256))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(poison_kernel, grid, block, 0, s, out, n, flag);
        return hipGetLastError();
      },
      stream);
}

int ck_zero_if_flag(float* p, int64_t n, const int32_t* flag, void* stream) {
  CK_REQUIRE(p && flag && n > 0, "ck_zero_if_flag: null pointer or empty buffer");
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 1024))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(zero_if_flag_kernel, grid, block, 0, s, p, n, flag);
        return hipGetLastError();
      },
      stream);
}

int ck_transpose_f32(const float* x, float* xt, int B, int D, void* stream) {
  return transpose_impl<float, float>(x, xt, B, D, stream, "ck_transpose_f32");
}

int ck_categorical_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out,
                       int F, int B, int K, int C, int D, void* stream) {
  return gather_impl<0>(table, xt, scope, out, F, B, K, C, D, stream, "ck_categorical_fwd");
}

int ck_categorical_clog_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out_c,
                            int F, int B, int K, int C, int D, void* stream) {
  return gather_impl<3>(table, xt, scope, out_c, F, B, K, C, D, stream, "ck_categorical_clog_fwd");
}

int ck_lse_to_clse(const float* in, float* out_c, int64_t n, void* stream) {
  CK_REQUIRE(in && out_c && n > 0, "ck_lse_to_clse: null pointer or empty input");
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 8192))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(lse_to_clse_kernel, grid, block, 0, s, in, reinterpret_cast<float2*>(out_c), n);
        return hipGetLastError();
      },
      stream);
}

int ck_embedding_log_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out,
                         int F, int B, int K, int C, int D, void* stream) {
  return gather_impl<1>(table, xt, scope, out, F, B, K, C, D, stream, "ck_embedding_log_fwd");
}

int ck_embedding_clog_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out_c,
                          int F, int B, int K, int C, int D, void* stream) {
  return gather_impl<2>(table, xt, scope, out_c, F, B, K, C, D, stream, "ck_embedding_clog_fwd");
}

int ck_embedding_clog_c_fwd(const float* table_c, const int32_t* xt, const int64_t* scope, float* out_c,
                            int F, int B, int K, int C, int D, void* stream) {
  CK_REQUIRE(K > 0 && K <= (1 << 29), "ck_embedding_clog_c_fwd: K=%d", K);
  return gather_impl<4>(table_c, xt, scope, out_c, F, B, 2 * K, C, D, stream, "ck_embedding_clog_c_fwd");
}

int ck_gaussian_fwd(const float* mean, const float* stddev, const float* log_partition,
                    const float* xt, const int64_t* scope, float* out, int F, int B, int K, int D,
                    void* stream) {
  CK_REQUIRE(mean && stddev && xt && scope && out, "ck_gaussian_fwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && K > 0 && D > 0, "ck_gaussian_fwd: non-positive size");
  if (F > ck::kMaxFoldsPerLaunch)
    return ck::chunk_folds(F, [&](int f0, int n) {
      const int64_t o = static_cast<int64_t>(f0) * K;
      return ck_gaussian_fwd(mean + o, stddev + o, log_partition == nullptr ? nullptr : log_partition + o, xt, scope + f0,
                             out + static_cast<int64_t>(f0) * B * K, n, B, K, D, stream);
    });
  const int rows_per_block = 256;
  dim3 grid((B + rows_per_block - 1) / rows_per_block, F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gaussian_kernel, grid, block, 0, s, mean, stddev, log_partition, xt, scope,
                           out, B, K, rows_per_block);
        return hipGetLastError();
      },
      stream);
}

int ck_gaussian_prod_fwd(const float* mean, const float* stddev, const float* log_partition, const float* xt,
                         const int64_t* scope, const int32_t* gfold, float* out, int F, int H, int B, int K,
                         void* stream) {
  CK_REQUIRE(mean && stddev && xt && scope && gfold && out, "ck_gaussian_prod_fwd: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && K > 0, "ck_gaussian_prod_fwd: non-positive size");
  CK_REQUIRE(F <= 65535, "ck_gaussian_prod_fwd: F=%d exceeds grid.y", F);
  if ((K == 32 || K == 64 || K == 128 || K == 256) && B % 4 == 0 && (reinterpret_cast<uintptr_t>(xt) & 15u) == 0) {
    const int rows_per_block = (256 / K) * 16;
    dim3 grid16((B + rows_per_block - 1) / rows_per_block, F), block16(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(gaussian_prod_rows16_kernel, grid16, block16, 0, s, mean, stddev, log_partition, xt, scope, gfold, out, H, B, K);
          return hipGetLastError();
        },
        stream);
  }
  constexpr int RPT = 8;
  const int lanes_rows = 256 / (K <= 256 ? K : 256);
  CK_REQUIRE(lanes_rows >= 1, "ck_gaussian_prod_fwd: unsupported K=%d", K);
  const int rows_per_block = lanes_rows * RPT;
  dim3 grid((B + rows_per_block - 1) / rows_per_block, F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gaussian_prod_kernel<RPT>, grid, block, 0, s, mean, stddev, log_partition, xt, scope,
                           gfold, out, H, B, K);
        return hipGetLastError();
      },
      stream);
}

int ck_constant_fwd(const float* value, float* out, int F, int B, int K, int log_space,
                    int value_is_complex, int complex_out, void* stream) {
  CK_REQUIRE(value && out, "ck_constant_fwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && K > 0, "ck_constant_fwd: non-positive size");
  CK_REQUIRE(complex_out || !value_is_complex, "ck_constant_fwd: complex value needs complex output");
  if (F > ck::kMaxFoldsPerLaunch)
    return ck::chunk_folds(F, [&](int f0, int n) {
      return ck_constant_fwd(value + static_cast<int64_t>(f0) * K * (value_is_complex ? 2 : 1),
                             out + static_cast<int64_t>(f0) * B * K * (complex_out ? 2 : 1), n, B, K, log_space, value_is_complex,
                             complex_out, stream);
    });
  const int64_t n = static_cast<int64_t>(B) * K;
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 1024)), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(constant_kernel, grid, block, 0, s, value, out, B, K, log_space,
                           value_is_complex, complex_out);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
