// Parameter-graph kernels (softmax / sigmoid / mixing weights / matmul / ...) on fold-stacked
// blocks, and the log-likelihood reduction that feeds the data-parallel all-reduce.
//
// The reference re-evaluates these on every forward with one ATen launch per node
// (cirkit/backend/torch/parameters/parameter.py:180-188); total volume is the parameter size
// (32 MB at the north-star config), i.e. ~1 % of the activation traffic.
#include <algorithm>

#include "ck_internal.h"
#include "ck_softmax.h"
#include "ck_tile.h"

namespace {

// softmax over `len` with stride `inner`; one wave per (outer, inner) line when inner == 1,
// otherwise one thread per line (lines are then adjacent in memory -> coalesced across threads).
__global__ void __launch_bounds__(256)
    softmax_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t rows, int len,
                        int log_space) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * static_cast<int64_t>(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = in + row * len;
  float* dst = out + row * len;
  float mx = -INFINITY;
  for (int i = lane; i < len; i += 64) mx = fmaxf(mx, src[i]);
  mx = ck::wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < len; i += 64) sum += expf(src[i] - mx);
  sum = ck::wave_sum(sum);
  if (log_space) {
    const float ls = logf(sum);
    for (int i = lane; i < len; i += 64) dst[i] = (src[i] - mx) - ls;
  } else {
    for (int i = lane; i < len; i += 64) dst[i] = expf(src[i] - mx) / sum;
  }
}

// TorchReduceProductParameter / TorchReduceLSEParameter (nodes.py:754-761) along one axis of a tensor viewed (outer, len, inner):
// a thread per (outer, inner) pair.  BWD: dx = dy y / x (product; an entry that is 0 takes the product of the others) and
// dx = dy exp(x - y) (log-sum-exp).  TorchOuterSumParameter (nodes.py:615-653): out[o, i1 * n2 + i2, r] = a[o, i1, r] + b[o, i2, r].
template <bool BWD>
__global__ void __launch_bounds__(256)
    reduce_axis_kernel(int op, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ out,
                       int64_t outer, int len, int64_t inner) {
  const int64_t t = blockIdx.x * static_cast<int64_t>(256) + threadIdx.x;
  if (t >= outer * inner) return;
  const int64_t o = t / inner, r = t - o * inner;
  const float* src = x + o * len * inner + r;
  if constexpr (!BWD) {
    if (op == 0) {
      float p = 1.f;
      for (int j = 0; j < len; ++j) p *= src[j * inner];
      out[t] = p;
    } else {
      float m = -INFINITY;
      for (int j = 0; j < len; ++j) m = fmaxf(m, src[j * inner]);
      if (!(fabsf(m) < INFINITY)) {  // (all -inf: -inf; an +inf entry: +inf -- torch.logsumexp)
        out[t] = m;
        return;
      }
      float sum = 0.f;
      for (int j = 0; j < len; ++j) sum += expf(src[j * inner] - m);
      out[t] = m + logf(sum);
    }
  } else {
    float* dst = out + o * len * inner + r;
    const float g = dy[t], yy = y[t];
    for (int j = 0; j < len; ++j) {
      float d;
      if (op == 0) {
        const float xv = src[j * inner];
        if (xv != 0.f) {
          d = yy / xv;
        } else {  // (the product of the other entries)
          d = 1.f;
          for (int k = 0; k < len; ++k)
            if (k != j) d *= src[k * inner];
        }
      } else {
        d = fabsf(yy) < INFINITY ? expf(src[j * inner] - yy) : 0.f;
      }
      dst[j * inner] = g == 0.f ? 0.f : g * d;
    }
  }
}
__global__ void __launch_bounds__(256)
    outer_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t outer, int n1, int n2, int64_t inner) {
  const int64_t t = blockIdx.x * static_cast<int64_t>(256) + threadIdx.x;
  if (t >= outer * n1 * n2 * inner) return;
  const int64_t r = t % inner, q = t / inner;
  const int i2 = static_cast<int>(q % n2);
  const int64_t q2 = q / n2;
  const int i1 = static_cast<int>(q2 % n1);
  const int64_t o = q2 / n1;
  out[t] = a[(o * n1 + i1) * inner + r] + b[(o * n2 + i2) * inner + r];
}
// its backward: da[o, i1, r] = sum over i2 of dout[o, i1, i2, r] (which = 0) / db[o, i2, r] = sum over i1 (which = 1)
__global__ void __launch_bounds__(256)
    outer_sum_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int64_t outer, int n1, int n2, int64_t inner, int which) {
  const int n_keep = which == 0 ? n1 : n2, n_red = which == 0 ? n2 : n1;
  const int64_t t = blockIdx.x * static_cast<int64_t>(256) + threadIdx.x;
  if (t >= outer * n_keep * inner) return;
  const int64_t r = t % inner, q = t / inner;
  const int ik = static_cast<int>(q % n_keep);
  const int64_t o = q / n_keep;
  float acc = 0.f;
  for (int j = 0; j < n_red; ++j) {
    const int i1 = which == 0 ? ik : j, i2 = which == 0 ? j : ik;
    acc += dout[((o * n1 + i1) * n2 + i2) * inner + r];
  }
  dx[t] = acc;
}

__global__ void __launch_bounds__(256)
    softmax_strided_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t outer,
                           int len, int64_t inner, int log_space) {
  const int64_t line = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (line >= outer * inner) return;
  const int64_t o = line / inner, i = line - o * inner;
  const float* src = in + o * len * inner + i;
  float* dst = out + o * len * inner + i;
  float mx = -INFINITY;
  for (int j = 0; j < len; ++j) mx = fmaxf(mx, src[j * inner]);
  float sum = 0.f;
  for (int j = 0; j < len; ++j) sum += expf(src[j * inner] - mx);
  const float ls = logf(sum);
  for (int j = 0; j < len; ++j)
    dst[j * inner] = log_space ? (src[j * inner] - mx) - ls : expf(src[j * inner] - mx) / sum;
}

__global__ void __launch_bounds__(256)
    unary_kernel(int op, const float* __restrict__ in, float* __restrict__ out, int64_t n, float a,
                 float b) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float x = in[i];
    float y;
    switch (op) {
      case CK_UNARY_SIGMOID:
        y = 1.f / (1.f + expf(-x));
        break;
      case CK_UNARY_SCALED_SIGMOID:
        y = (1.f / (1.f + expf(-x))) * (b - a) + a;
        break;
      case CK_UNARY_EXP:
        y = expf(x);
        break;
      case CK_UNARY_LOG:
        y = logf(x);
        break;
      case CK_UNARY_CLAMP:  // torch.clamp(x, min=a, max=b): NaN stays NaN
        y = x < a ? a : (x > b ? b : x);
        break;
      case CK_UNARY_SOFTPLUS:  // torch.nn.functional.softplus (beta 1, threshold 20)
        y = x > 20.f ? x : log1pf(expf(x));
        break;
      default:
        y = x * x;
        break;
    }
    out[i] = y;
  }
}

__global__ void __launch_bounds__(256)
    gather_folds_kernel(const float* __restrict__ in, const int64_t* __restrict__ idx,
                        float* __restrict__ out, int64_t per_fold) {
  const int64_t f = blockIdx.y;
  const float* src = in + idx[f] * per_fold;
  float* dst = out + f * per_fold;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < per_fold;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    dst[i] = src[i];
}

__global__ void __launch_bounds__(256)
    conj_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    out[2 * i] = in[2 * i];
    out[2 * i + 1] = -in[2 * i + 1];
  }
}

// (F,K,H) -> (F,K,H*K): out[f,k,h*K+k'] = in[f,k,h] * (k == k')
__global__ void __launch_bounds__(256)
    mixing_weight_kernel(const float* __restrict__ in, float* __restrict__ out, int K, int H) {
  const int64_t f = blockIdx.y;
  const int64_t n = static_cast<int64_t>(K) * H * K;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int k = static_cast<int>(i / (static_cast<int64_t>(H) * K));
    const int r = static_cast<int>(i - static_cast<int64_t>(k) * H * K);
    const int h = r / K, k2 = r - h * K;
    out[f * n + i] = (k == k2) ? in[(f * K + k) * H + h] : 0.f;
  }
}

// out[f,m,n] = sum_k A(m,k) B(k,n); 16x16 LDS tiles.
constexpr int kMM = 16;
__global__ void __launch_bounds__(256)
    bmm_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
               int M, int N, int Kd, int trans_a, int trans_b, int accumulate) {
  __shared__ float as[kMM][kMM + 1], bs[kMM][kMM + 1];
  const int64_t f = blockIdx.z;
  const float* af = a + f * static_cast<int64_t>(M) * Kd;
  const float* bf = b + f * static_cast<int64_t>(Kd) * N;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * kMM + ty, n = blockIdx.x * kMM + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < Kd; k0 += kMM) {
    const int ka = k0 + tx, kb = k0 + ty;
    as[ty][tx] = (m < M && ka < Kd) ? (trans_a ? af[static_cast<int64_t>(ka) * M + m] : af[static_cast<int64_t>(m) * Kd + ka]) : 0.f;
    bs[ty][tx] = (kb < Kd && n < N) ? (trans_b ? bf[static_cast<int64_t>(n) * Kd + kb] : bf[static_cast<int64_t>(kb) * N + n]) : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMM; ++k) acc = fmaf(as[ty][k], bs[k][tx], acc);
    __syncthreads();
  }
  if (m < M && n < N) {
    float* o = out + f * static_cast<int64_t>(M) * N + static_cast<int64_t>(m) * N + n;
    *o = accumulate ? *o + acc : acc;
  }
}

// The same product on (32, 64) output tiles, 2 x 4 outputs per thread, 32 contracted entries per barrier pair, every access to
// memory a coalesced run along the operand's fastest axis (the 16 x 16 form above reads and writes a line per 16 threads and
// spends two barriers per 16 multiply-adds: 40 - 48 us for the 784 Gram matrices W W^T / their backward (G + G^T) W of a squared
// circuit -- 25.7 MB operands, i.e. ~5 us of traffic).  The sum over k runs in the same order: bit-identical.
constexpr int kBT_M = 32, kBT_N = 64, kBT_K = 32;
template <bool TA, bool TB, bool VEC>
__global__ void __launch_bounds__(256)
    bmm_tile_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int M, int N, int Kd, int accumulate) {
  __shared__ __attribute__((aligned(16))) float as[kBT_K][kBT_M + 4];   // [k][m]
  __shared__ __attribute__((aligned(16))) float bs[kBT_K][kBT_N + 4];   // [k][n]
  const int64_t f = blockIdx.z;
  const float* af = a + f * static_cast<int64_t>(M) * Kd;
  const float* bf = b + f * static_cast<int64_t>(Kd) * N;
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int m0 = blockIdx.y * kBT_M, n0 = blockIdx.x * kBT_N;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // the operands of chunk k0 + 32 travel while chunk k0 is multiplied (the Gram matrices contract 256 entries: eight chunks of a
  // load -> LDS -> barrier chain were 32 us for 784 resident workgroups).  VEC (every extent a multiple of 4, 16-byte aligned
  // operands): 16-byte loads along each operand's fastest axis -- 3 load instructions per thread and chunk instead of 12.
  constexpr int NA = VEC ? 1 : 4, NB = VEC ? 2 : 8;
  float4 ar[NA], br[NB];  // (scalar form: .x only)
  auto fetch = [&](int k0) {
    if constexpr (VEC) {
      {  // A: 32 x 32 = 256 float4
        const int row = t >> 3, c4 = (t & 7) * 4;
        const int m = m0 + (TA ? c4 : row), k = k0 + (TA ? row : c4);
        ar[0] = (m < M && k < Kd) ? *reinterpret_cast<const float4*>(TA ? af + static_cast<int64_t>(k) * M + m : af + static_cast<int64_t>(m) * Kd + k)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {  // B: 32 x 64 = 512 float4
        const int i = t + 256 * r;
        const int nn = TB ? (i >> 3) : (i & 15) * 4, kk = TB ? (i & 7) * 4 : (i >> 4);
        const int n = n0 + nn, k = k0 + kk;
        br[r] = (n < N && k < Kd) ? *reinterpret_cast<const float4*>(TB ? bf + static_cast<int64_t>(n) * Kd + k : bf + static_cast<int64_t>(k) * N + n)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // A: 32 x 32
        const int i = t + 256 * r;
        const int mm = TA ? (i & 31) : (i >> 5), kk = TA ? (i >> 5) : (i & 31);
        const int m = m0 + mm, k = k0 + kk;
        ar[r].x = (m < M && k < Kd) ? (TA ? af[static_cast<int64_t>(k) * M + m] : af[static_cast<int64_t>(m) * Kd + k]) : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // B: 32 x 64
        const int i = t + 256 * r;
        const int nn = TB ? (i >> 5) : (i & 63), kk = TB ? (i & 31) : (i >> 6);
        const int n = n0 + nn, k = k0 + kk;
        br[r].x = (n < N && k < Kd) ? (TB ? bf[static_cast<int64_t>(n) * Kd + k] : bf[static_cast<int64_t>(k) * N + n]) : 0.f;
      }
    }
  };
  auto stage = [&]() {
    if constexpr (VEC) {
      const int row = t >> 3, c4 = (t & 7) * 4;
      if (TA) {
        *reinterpret_cast<float4*>(&as[row][c4]) = ar[0];
      } else {
        as[c4 + 0][row] = ar[0].x;
        as[c4 + 1][row] = ar[0].y;
        as[c4 + 2][row] = ar[0].z;
        as[c4 + 3][row] = ar[0].w;
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = t + 256 * r;
        if (TB) {
          const int nn = i >> 3, k4 = (i & 7) * 4;
          bs[k4 + 0][nn] = br[r].x;
          bs[k4 + 1][nn] = br[r].y;
          bs[k4 + 2][nn] = br[r].z;
          bs[k4 + 3][nn] = br[r].w;
        } else {
          *reinterpret_cast<float4*>(&bs[i >> 4][(i & 15) * 4]) = br[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = t + 256 * r;
        as[TA ? (i >> 5) : (i & 31)][TA ? (i & 31) : (i >> 5)] = ar[r].x;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int i = t + 256 * r;
        bs[TB ? (i & 31) : (i >> 6)][TB ? (i >> 5) : (i & 63)] = br[r].x;
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < Kd; k0 += kBT_K) {
    stage();
    __syncthreads();
    if (k0 + kBT_K < Kd) fetch(k0 + kBT_K);
#pragma unroll 8
    for (int k = 0; k < kBT_K; ++k) {
      const float2 av = *reinterpret_cast<const float2*>(&as[k][2 * ty]);
      const float4 bv = *reinterpret_cast<const float4*>(&bs[k][4 * tx]);
      acc[0][0] = fmaf(av.x, bv.x, acc[0][0]);
      acc[0][1] = fmaf(av.x, bv.y, acc[0][1]);
      acc[0][2] = fmaf(av.x, bv.z, acc[0][2]);
      acc[0][3] = fmaf(av.x, bv.w, acc[0][3]);
      acc[1][0] = fmaf(av.y, bv.x, acc[1][0]);
      acc[1][1] = fmaf(av.y, bv.y, acc[1][1]);
      acc[1][2] = fmaf(av.y, bv.z, acc[1][2]);
      acc[1][3] = fmaf(av.y, bv.w, acc[1][3]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int m = m0 + 2 * ty + u;
    if (m >= M) continue;
    float* o = out + f * static_cast<int64_t>(M) * N + static_cast<int64_t>(m) * N + n0 + 4 * tx;
    if (n0 + 4 * tx + 3 < N && (N & 3) == 0) {
      float4 r = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
      if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(o);
        r = make_float4(old.x + r.x, old.y + r.y, old.z + r.z, old.w + r.w);
      }
      *reinterpret_cast<float4*>(o) = r;
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (n0 + 4 * tx + v < N) o[v] = accumulate ? o[v] + acc[u][v] : acc[u][v];
    }
  }
}

// The same product on the matrix cores, one wavefront per (32, 32) output tile: `v_mfma_f32_32x32x2_f32` with A = op(a) rows
// and B = op(b) columns as operands, 16 steps per 32 contracted entries.  Lane (i, kh) of a step's operand holds entry
// k = 16 kh + s of its row / column (ANY pairing of the 32 entries into 16 steps is the same sum as long as both operands use
// it): an operand whose fastest axis is k is four 16-byte loads per lane and chunk, the other form sixteen 4-byte loads that
// coalesce across the lanes.  The multiply-add tiles above are bound by issue where the output is small (the Gram matrices
// W W^T of a squared circuit: (32, 32) outputs over 256 entries, 27 us for 784 folds -- 128 MFMAs per fold here).
// Extents: multiples of 32 (the launcher falls back otherwise).  Order of the sum: chunks of 32 in order, inside a chunk the
// pairs (s, 16 + s) -- not the tile kernels' order (fp32 rounding differs in the last bits).
// TA = 2: the A operand SYMMETRISED, A[m][k] = a[m][k] + a[k][m] (M = Kd): both operand gradients of a Gram product y = x x^T
// in one launch, d x = (d y + d y^T) x.
template <int TA, bool TB>
__global__ void __launch_bounds__(64)
    bmm_mfma_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int M, int N, int Kd, int accumulate) {
  const int64_t f = blockIdx.z;
  const float* af = a + f * static_cast<int64_t>(M) * Kd;
  const float* bf = b + f * static_cast<int64_t>(Kd) * N;
  const int lane = threadIdx.x, i32 = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  float av[16], bv[16], an[16], bn[16];
  auto fetch = [&](int k0, float (&x)[16], float (&y)[16]) {
    const int kb = k0 + 16 * kh;
    if (TA == 2) {  // a[m][k] + a[k][m]
      const float4* p = reinterpret_cast<const float4*>(af + static_cast<int64_t>(m0 + i32) * Kd + kb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = p[q];
        x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) x[s2] += af[static_cast<int64_t>(kb + s2) * M + m0 + i32];
    } else if (TA == 1) {  // a[k][m]
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) x[s2] = af[static_cast<int64_t>(kb + s2) * M + m0 + i32];
    } else {   // a[m][k]
      const float4* p = reinterpret_cast<const float4*>(af + static_cast<int64_t>(m0 + i32) * Kd + kb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = p[q];
        x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
      }
    }
    if (TB) {  // b[n][k]
      const float4* p = reinterpret_cast<const float4*>(bf + static_cast<int64_t>(n0 + i32) * Kd + kb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = p[q];
        y[4 * q] = v.x, y[4 * q + 1] = v.y, y[4 * q + 2] = v.z, y[4 * q + 3] = v.w;
      }
    } else {   // b[k][n]
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) y[s2] = bf[static_cast<int64_t>(kb + s2) * N + n0 + i32];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  fetch(0, av, bv);
  for (int k0 = 0; k0 < Kd; k0 += 32) {
    if (k0 + 32 < Kd) fetch(k0 + 32, an, bn);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) av[s2] = an[s2], bv[s2] = bn[s2];
  }
  // D[m][n] sits in lane (n, hi) register r with m = 8 (r >> 2) + 4 hi + (r & 3)
  float* o = out + f * static_cast<int64_t>(M) * N + n0 + i32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float* dst = o + static_cast<int64_t>(m0 + 8 * (r >> 2) + 4 * kh + (r & 3)) * N;
    *dst = accumulate ? *dst + acc[r] : acc[r];
  }
}

// (R, A, Bd) -> (R, Bd, A) through a 32x32 LDS tile, optional log.
template <class T>
__global__ void __launch_bounds__(256)
    transpose_last2_kernel(const T* __restrict__ in, T* __restrict__ out, int A, int Bd,
                           int take_log, int out_rows) {
  __shared__ T tile[32][33];
  const int64_t r = blockIdx.z;
  const T* src = in + r * static_cast<int64_t>(A) * Bd;
  T* dst = out + r * static_cast<int64_t>(A) * out_rows;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // (32, 8)
  const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int ai = a0 + ty + j, bi = b0 + tx;
    if (ai < A && bi < Bd) {
      const T v = src[static_cast<int64_t>(ai) * Bd + bi];
      if constexpr (sizeof(T) == sizeof(float)) tile[ty + j][tx] = take_log ? logf(v) : v;
      else tile[ty + j][tx] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int bi = b0 + ty + j, ai = a0 + tx;
    if (ai < A && bi < Bd) dst[static_cast<int64_t>(bi) * A + ai] = tile[tx][ty + j];
  }
}

// integral row of a gather table (F, C+1, K): mode 0 zeros (normalised probs), 1 logsumexp over the
// C rows (unnormalised logits, TorchCategoricalLayer.log_partition_function input.py:414-421),
// 2 ones (embedding tables: never selected, kept finite), 3 complex ones (K floats = K / 2 pairs (1, 0))
__global__ void __launch_bounds__(256)
    table_integral_row_kernel(float* __restrict__ table, int C, int K, int mode) {
  float* t = table + static_cast<int64_t>(blockIdx.x) * (C + 1) * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float v = (mode == 2 || (mode == 3 && (k & 1) == 0)) ? 1.f : 0.f;
    if (mode == 1) {
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, t[c * K + k]);
      float sacc = 0.f;
      for (int c = 0; c < C; ++c) sacc += expf(t[c * K + k] - mx);
      v = mx + logf(sacc);
    }
    t[static_cast<int64_t>(C) * K + k] = v;
  }
}

// ---- batched softmax prologue -------------------------------------------------------------------
// All `tensor -> softmax(last axis)` parameters of a circuit in ONE launch (the reference spends one
// ATen launch per node per layer, parameters/parameter.py:180-188).  kind 0: linear-space rows
// (sum-layer weights).  kind 1: Categorical probs (F, K, C) -> log-probability table (F, C, K),
// i.e. softmax + log (input.py:405-408) + the transpose the gather kernels want, through LDS.
constexpr int kMaxJobs = 48;
struct JobTable {
  ck_softmax_job job[kMaxJobs];
  int n;
};

// wavefronts per workgroup of the batched prologue: a table job is one workgroup per fold whose phases are
// separated by barriers, so its latency -- which IS the kernel time when all folds run at once -- scales
// is not simply shorter with more waves (measured on config 2: 2 waves 39 us, 4 waves 29 us, 8 waves 38 us, 16 waves 60 us)
constexpr int kPW = 4;
// the 64-unit table jobs (WIDE launch): 83 KB of LDS allow one workgroup per CU, so the workgroup itself has to
// supply the parallelism -- 9 category tiles, 64 weight rows and 64 statistics rows over 8 waves
constexpr int kPW64 = 8;

// Rows of 33..256 entries (kind 0): NE entries per lane (lane, lane + 64, ...), 16 / NE rows of a wave in flight
// at once.  Same arithmetic, in the same order, as the row-at-a-time loop at the end of softmax_job_rows
// (per-lane partial results in entry order, then the wave reduction) -- only the loads are issued up front.
template <int NE>
__device__ __forceinline__ void softmax_rows_medium(const ck_softmax_job& j, int blk) {
  constexpr int RB = 16 / NE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int len = j.len;
  for (int it0 = 0; it0 < 16; it0 += RB) {
    float x[RB][NE];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int64_t row = static_cast<int64_t>(blk) * (16 * kPW) + (it0 + r) * kPW + wave;
#pragma unroll
      for (int k = 0; k < NE; ++k)
        x[r][k] = (row < j.rows && lane + 64 * k < len) ? j.in[row * len + lane + 64 * k] : -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int64_t row = static_cast<int64_t>(blk) * (16 * kPW) + (it0 + r) * kPW + wave;
      if (row >= j.rows) continue;
      float mx = x[r][0];
#pragma unroll
      for (int k = 1; k < NE; ++k) mx = fmaxf(mx, x[r][k]);
      mx = ck::wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NE; ++k)
        if (lane + 64 * k < len) sum += __expf(x[r][k] - mx);
      sum = ck::wave_sum(sum);
#pragma unroll
      for (int k = 0; k < NE; ++k)
        if (lane + 64 * k < len) j.out[row * len + lane + 64 * k] = __expf(x[r][k] - mx) / sum;
    }
  }
}

__device__ __forceinline__ void softmax_job_rows(const ck_softmax_job& j, int blk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int len = j.len;
  if (j.kind == 0 && len > 32 && len <= 256) {
    if (len <= 64)
      softmax_rows_medium<1>(j, blk);
    else if (len <= 128)
      softmax_rows_medium<2>(j, blk);
    else
      softmax_rows_medium<4>(j, blk);
    return;
  }
  if (len <= 32) {
    // two rows per wave pass (one per 32-lane half); 16 rows per wave and block; all loads issued up front
    const int half = lane >> 5, l = lane & 31;
    float x[8];
    bool ok[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t row = static_cast<int64_t>(blk) * (16 * kPW) + it * (2 * kPW) + wave * 2 + half;
      ok[it] = row < j.rows && l < len;
      x[it] = ok[it] ? j.in[row * len + l] : -INFINITY;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t row = static_cast<int64_t>(blk) * (16 * kPW) + it * (2 * kPW) + wave * 2 + half;
      const float mx = half_reduce_dpp<true>(x[it]);
      const float e = ok[it] ? __expf(x[it] - mx) : 0.f;
      const float sum = half_reduce_dpp<false>(e);
      if (ok[it]) {
        const float p = e / sum;
        if (j.kind == 0) {
          j.out[row * len + l] = p;
        } else {
          // MFMA-tiled layouts of ck_tile.h: row = fold*32 + o, l = input unit
          const int64_t fold = row >> 5;
          const int o = static_cast<int>(row & 31);
          const int g = l >> 3, kh2 = (l >> 2) & 1, t = l & 3;
          const int ln = o + 32 * kh2;
          float* base = j.out + fold * 1024;
          base[g * 256 + ln * 4 + t] = p;  // (kind 2)
        }
      }
    }
  } else {
    for (int it = 0; it < 16; ++it) {
      const int64_t row = static_cast<int64_t>(blk) * (16 * kPW) + it * kPW + wave;
      if (row >= j.rows) break;
      const float* src = j.in + row * len;
      float mx = -INFINITY;
      for (int i = lane; i < len; i += 64) mx = fmaxf(mx, src[i]);
      mx = ck::wave_max(mx);
      float sum = 0.f;
      for (int i = lane; i < len; i += 64) sum += __expf(src[i] - mx);
      sum = ck::wave_sum(sum);
      for (int i = lane; i < len; i += 64) j.out[row * len + i] = __expf(src[i] - mx) / sum;
    }
  }
}

__device__ __forceinline__ void softmax_job_table(const ck_softmax_job& j, int f, float* tile) {
  // tile[k][c], row stride C+1 (conflict-free both for the row reductions and the transposed read)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = j.len, K = j.k, ld = C + 1;
  const float* src = j.in + static_cast<int64_t>(f) * K * C;
  float* stat = tile + K * ld;  // [K] max, [K] log-sum
  // phase 1: the whole (K, C) block of logits, coalesced 16-byte loads, all in flight at once
  if ((C & 3) == 0) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    for (int i = threadIdx.x; i < (K * C) >> 2; i += blockDim.x) {
      const float4 v = src4[i];
      const int e = i << 2, k = e / C, c = e - k * C;
      float* d = tile + k * ld + c;
      d[0] = v.x;
      d[1] = v.y;
      d[2] = v.z;
      d[3] = v.w;
    }
  } else {
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
      const int k = i / C, c = i - k * C;
      tile[k * ld + c] = src[i];
    }
  }
  __syncthreads();
  // phase 2: per-row max and log-sum-exp from LDS; 8 rows per wave pass so that the 2 x 6 shuffle
  // steps of the reductions of different rows overlap
  for (int k0 = wave; k0 < K; k0 += 8 * kPW) {
    float mx[8], sum[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = min(k0 + kPW * r, K - 1);
      const float* row = tile + k * ld;
      float m = -INFINITY;
      for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
      mx[r] = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 8; ++r) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = min(k0 + kPW * r, K - 1);
      const float* row = tile + k * ld;
      float sacc = 0.f;
      for (int c = lane; c < C; c += 64) sacc += __expf(row[c] - mx[r]);
      sum[r] = sacc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 8; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = k0 + kPW * r;
        if (k < K) {
          stat[k] = mx[r];
          stat[K + k] = __logf(sum[r]);
        }
      }
    }
  }
  __syncthreads();
  // phase 3: out[c][k] = log(exp(d)/sum), d = theta - max; written coalesced (k fastest), 16 B per lane
  // exp(d) underflows to 0 below ~-103.97 -> the reference yields log(0) = -inf
  float* dst = j.out + static_cast<int64_t>(f) * (C + 1) * K;  // (C + 1, K): row C = integral row
  for (int k = threadIdx.x; k < K; k += blockDim.x) dst[static_cast<int64_t>(C) * K + k] = 0.f;  // log sum_c p = 0
  if ((K & 3) == 0) {
    const int k4n = K >> 2;
    for (int i = threadIdx.x; i < C * k4n; i += blockDim.x) {
      const int c = i / k4n, k = (i - c * k4n) << 2;
      float4 o;
      float d;
      d = tile[(k + 0) * ld + c] - stat[k + 0];
      o.x = d < -103.9f ? -INFINITY : d - stat[K + k + 0];
      d = tile[(k + 1) * ld + c] - stat[k + 1];
      o.y = d < -103.9f ? -INFINITY : d - stat[K + k + 1];
      d = tile[(k + 2) * ld + c] - stat[k + 2];
      o.z = d < -103.9f ? -INFINITY : d - stat[K + k + 2];
      d = tile[(k + 3) * ld + c] - stat[k + 3];
      o.w = d < -103.9f ? -INFINITY : d - stat[K + k + 3];
      reinterpret_cast<float4*>(dst)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < C * K; i += blockDim.x) {
      const int c = i / K, k = i - c * K;
      const float d = tile[k * ld + c] - stat[k];
      dst[i] = d < -103.9f ? -INFINITY : d - stat[K + k];
    }
  }
}

// kind 4: the Categorical log-table of fold idx[d] pushed through dense layer fold d in the same block:
//   T'[d, c, :] = log(W_d . exp(T[c, :] - m_c)) + m_c,  T = log softmax_C(theta_cat[idx[d]]) transposed,
//   W_d = softmax(theta_dense[d]) (rows of 32), c = 0..C (row C: the integral row, T = 0).
// Exactly the arithmetic of the kind-1 job followed by ck_sum_lse_fwd on the table (the register-tile
// step of ck_tile.h on 32-row tiles of categories) and of the 32-wide rows job for W -- bit-identical
// to running them one after the other -- without the (F, C+1, K) table ever reaching memory.
__device__ __forceinline__ void softmax_job_table_dense(const ck_softmax_job& j, int d, float* tile) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = j.len, K = j.k, ld = C + 1;  // K == 32
  const int64_t f = j.idx != nullptr ? j.idx[d] : d;
  const float* src = j.in + f * K * C;
  float* stat = tile + K * ld;   // [K] max, [K] log-sum
  float* w_s = stat + 2 * K;     // [32][32] row-major linear weights of dense fold d
  if ((C & 3) == 0) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    for (int i = threadIdx.x; i < (K * C) >> 2; i += blockDim.x) {
      const float4 v = src4[i];
      const int e = i << 2, k = e / C, c = e - k * C;
      float* dd = tile + k * ld + c;
      dd[0] = v.x;
      dd[1] = v.y;
      dd[2] = v.z;
      dd[3] = v.w;
    }
  } else {
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
      const int k = i / C, c = i - k * C;
      tile[k * ld + c] = src[i];
    }
  }
  {  // W_d: 32 rows of 32, two rows per wave pass (same reduction tree as softmax_job_rows)
    const int half = lane >> 5, l = lane & 31;
    const float* th = j.in2 + static_cast<int64_t>(d) * 1024;
#pragma unroll
    for (int it = 0; it < 16 / kPW; ++it) {
      const int row = it * (2 * kPW) + wave * 2 + half;
      const float x = th[row * 32 + l];
      const float mx = half_reduce_dpp<true>(x);
      const float e = __expf(x - mx);
      const float sum = half_reduce_dpp<false>(e);
      w_s[row * 32 + l] = e / sum;
    }
  }
  __syncthreads();
  for (int k0 = wave; k0 < K; k0 += 8 * kPW) {  // per-unit max and log-sum-exp over the categories (as kind 1)
    float mx[8], sum[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = min(k0 + kPW * r, K - 1);
      const float* row = tile + k * ld;
      float m = -INFINITY;
      for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
      mx[r] = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 8; ++r) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = min(k0 + kPW * r, K - 1);
      const float* row = tile + k * ld;
      float sacc = 0.f;
      for (int c = lane; c < C; c += 64) sacc += __expf(row[c] - mx[r]);
      sum[r] = sacc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 8; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = k0 + kPW * r;
        if (k < K) {
          stat[k] = mx[r];
          stat[K + k] = __logf(sum[r]);
        }
      }
    }
  }
  __syncthreads();
  WRegs wr;
  load_w<CK_W_ROWMAJOR>(w_s, lane, wr);
  const int b_in = lane & 31, kh = lane >> 5;
  float* dst = j.out + static_cast<int64_t>(d) * (C + 1) * K;
  for (int t = wave; t * 32 <= C; t += kPW) {  // 32 categories per register tile, rows 0..C
    const int c = t * 32 + b_in;
    const int cl = min(c, C - 1);
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int k = 8 * g + 4 * kh + tt;
        const float dlt = tile[k * ld + cl] - stat[k];
        const float val = dlt < -103.9f ? -INFINITY : dlt - stat[K + k];
        v[4 * g + tt] = c >= C ? 0.f : val;  // row C: log sum_c p = 0
      }
    if (j.kind == 4) {
      sum_step<CK_W_ROWMAJOR>(wr, v);
    } else {  // kind 5: the row stays in linear space, y = W . exp(v - m), with its log scale m stored aside
      const float m = row_max16(v);
      const float nml = exp_offset(m, 0.f);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_exp2f(fmaf(v[r], kL2E, nml));
      contract_linear<CK_W_ROWMAJOR>(wr, v);
      if (c <= C && kh == 0) j.out2[static_cast<int64_t>(d) * (C + 1) + c] = m;
    }
    if (c <= C) tile_store(dst + static_cast<int64_t>(c) * K + 4 * kh, v);
  }
}

// kind 5 for C <= 256, C % 4 == 0 (the usual Categorical sizes): a wave holds a whole (unit, all categories) row of
// logits in registers -- one float4 per lane -- so the per-unit maximum and log-sum-exp are wave reductions in registers
// (the job above makes two passes over the logits in LDS for them: 8 of its 22 us at config 2), and the tile in LDS
// already holds the normalised log-probabilities the dense layer is applied to.
//   T[c, k] = (theta[k, c] - max_c theta[k, .]) - log sum_c exp(theta[k, c] - max)      (-inf below -103.9, as above)
//   out[d, c, :] = W_d . exp(T[c, :] - m_c),  out2[d, c] = m_c = max_k T[c, k];  row C: T = 0 (the integral row)
// Same arithmetic as the job above up to the order of the two reductions over the categories.
__device__ __forceinline__ void softmax_job_table_dense_rows(const ck_softmax_job& j, int d, float* tile) {
  // (the job itself lives in ck_softmax.h: the persistent leaf launch runs it too)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = j.len;
  constexpr int K = 32;
  const int64_t f = j.idx != nullptr ? j.idx[d] : d;
  const float* theta = j.in + f * K * C;
  const float* theta_w = j.in2 + static_cast<int64_t>(d) * 1024;
  float* dst = j.out + static_cast<int64_t>(d) * (C + 1) * K;
  const int kh = lane >> 5;
  auto sync = [] { __syncthreads(); };
  if (j.kind == 4) {
    table_dense_rows<kPW, false>(theta, theta_w, C, tile, wave, lane, sync, [&](int c, const float (&v)[16], float) {
      if (c <= C) tile_store(dst + static_cast<int64_t>(c) * K + 4 * kh, v);
    });
  } else {
    table_dense_rows<kPW, true>(theta, theta_w, C, tile, wave, lane, sync, [&](int c, const float (&v)[16], float m) {
      if (c <= C && kh == 0) j.out2[static_cast<int64_t>(d) * (C + 1) + c] = m;
      if (c <= C) tile_store(dst + static_cast<int64_t>(c) * K + 4 * kh, v);
    });
  }
}

// kind 1 for C <= 256, C % 4 == 0: the rows-in-registers form of the table job (as softmax_job_table_dense_rows: the same
// reductions, so the tables of the three jobs agree bit for bit) -- one pass through LDS, transposed on the way in:
// tile_T[c][k], row stride K + 1 (the float4 columns of the output are then read without bank conflicts).
__device__ __forceinline__ void softmax_job_table_rows(const ck_softmax_job& j, int f, float* tile) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = j.len, K = j.k, n4 = C >> 2, ldT = K + 1;
  const float4* src = reinterpret_cast<const float4*>(j.in + static_cast<int64_t>(f) * K * C);
  const bool on = lane < n4;
  for (int k0 = wave; k0 < K; k0 += 8 * kPW) {  // 8 rows (units) per wave and pass, all loads in flight
    float4 x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = k0 + kPW * r;
      x[r] = on && k < K ? src[k * n4 + lane] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    log_softmax_rows<8>(x);  // (ck_softmax.h; rows k >= K hold -inf throughout and are not stored)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = k0 + kPW * r;
      if (k >= K) break;  // (uniform over the wave)
      if (on) {
        float* col = tile + (4 * lane) * ldT + k;
        col[0] = x[r].x;
        col[ldT] = x[r].y;
        col[2 * ldT] = x[r].z;
        col[3 * ldT] = x[r].w;
      }
    }
  }
  __syncthreads();
  float* dst = j.out + static_cast<int64_t>(f) * (C + 1) * K;  // (C + 1, K): row C = integral row
  for (int k = threadIdx.x; k < K; k += blockDim.x) dst[static_cast<int64_t>(C) * K + k] = 0.f;  // log sum_c p = 0
  if ((K & 3) == 0) {
    const int k4n = K >> 2;
    for (int i = threadIdx.x; i < C * k4n; i += blockDim.x) {
      const int c = i / k4n, k = (i - c * k4n) << 2;
      const float* row = tile + c * ldT + k;
      reinterpret_cast<float4*>(dst)[i] = make_float4(row[0], row[1], row[2], row[3]);
    }
  } else {
    for (int i = threadIdx.x; i < C * K; i += blockDim.x) {
      const int c = i / K, k = i - c * K;
      dst[i] = tile[c * ldT + k];
    }
  }
}

// kind 4 with 64 units: the same job on the two-block register tile (weights in LDS in MFMA operand layout,
// as in ck_cp.hip); W rows of 64 follow the arithmetic of the long-row branch of softmax_job_rows.
__device__ __forceinline__ void softmax_job_table_dense64(const ck_softmax_job& j, int d, float* tile) {
  constexpr int K = 64, NK = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = j.len;
  // rows form (C <= 256, C % 4 == 0): a (unit, all categories) row is one float4 per lane -- maximum and log-sum-exp are
  // wave reductions in registers and the tile in LDS holds the NORMALISED log-probabilities (one pass through LDS, the
  // same reductions as softmax_job_table_rows / softmax_job_table_dense_rows); otherwise two passes over the logits in LDS
  const bool rows_form = C <= 256 && (C & 3) == 0;
  const int ld = rows_form ? C + 4 : C + 1;
  const int64_t f = j.idx != nullptr ? j.idx[d] : d;
  const float* src = j.in + f * K * C;
  float* stat = tile + K * (C + 4);  // [K] max, [K] log-sum
  float* w_s = stat + 2 * K;         // [p][q][g][lane][4] linear weights of dense fold d
  // every global load of the job is issued before the first use: the (K, C) logits as 16-byte loads and the
  // K / kPW64 weight rows of this wave (the job's latency is the kernel time, see kPW)
  const float* th = j.in2 + static_cast<int64_t>(d) * (K * K);
  float wx[K / kPW64];
#pragma unroll
  for (int r = 0; r < K / kPW64; ++r) wx[r] = th[(wave + kPW64 * r) * K + lane];
  float4 x[K / kPW64];
  if (rows_form) {
    const int n4r = C >> 2;
#pragma unroll
    for (int r = 0; r < K / kPW64; ++r)
      x[r] = lane < n4r ? reinterpret_cast<const float4*>(src)[(wave + kPW64 * r) * n4r + lane] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  } else if ((C & 3) == 0) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    const int n4 = (K * C) >> 2;
    for (int base = threadIdx.x; base < n4; base += 8 * kPW64 * 64) {  // 8 loads in flight per thread
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * (kPW64 * 64);
        if (i < n4) v[u] = src4[i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * (kPW64 * 64);
        if (i < n4) {
          const int e = i << 2, k = e / C, c = e - k * C;
          float* dd = tile + k * ld + c;
          dd[0] = v[u].x;
          dd[1] = v[u].y;
          dd[2] = v[u].z;
          dd[3] = v[u].w;
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
      const int k = i / C, c = i - k * C;
      tile[k * ld + c] = src[i];
    }
  }
  {  // the rows of this wave are reduced together, so that the shuffle steps of different rows overlap
    constexpr int R = K / kPW64;
    float mx[R], e[R], sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mx[r] = wx[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < R; ++r) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] = e[r] = __expf(wx[r] - mx[r]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < R; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int o = wave + kPW64 * r;
      const int k = lane, p = o >> 5, q = k >> 5, g = (k >> 3) & 3, ln = (o & 31) + 32 * ((k >> 2) & 1);
      w_s[((((p * NK + q) * 4 + g) * 64) + ln) * 4 + (k & 3)] = e[r] / sum[r];
    }
  }
  if (rows_form) {
    const bool on = lane < (C >> 2);
    log_softmax_rows<K / kPW64>(x);  // (ck_softmax.h)
    if (on) {
#pragma unroll
      for (int r = 0; r < K / kPW64; ++r) *reinterpret_cast<float4*>(tile + (wave + kPW64 * r) * ld + 4 * lane) = x[r];
    }
  }
  __syncthreads();
  if (!rows_form) {  // per-unit max and log-sum-exp over the categories, K / kPW64 units per wave, reduced together
    constexpr int R = K / kPW64;
    float mx[R], sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* row = tile + (wave + kPW64 * r) * ld;
      float m = -INFINITY;
      for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
      mx[r] = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < R; ++r) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* row = tile + (wave + kPW64 * r) * ld;
      float sacc = 0.f;
      for (int c = lane; c < C; c += 64) sacc += __expf(row[c] - mx[r]);
      sum[r] = sacc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < R; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        stat[wave + kPW64 * r] = mx[r];
        stat[K + wave + kPW64 * r] = __logf(sum[r]);
      }
    }
  }
  __syncthreads();
  const int b_in = lane & 31, kh = lane >> 5;
  float* dst = j.out + static_cast<int64_t>(d) * (C + 1) * K;
  for (int t = wave; t * 32 <= C; t += kPW64) {
    const int c = t * 32 + b_in;
    const int cl = min(c, C - 1);
    float v[NK][16];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int k = 32 * q + 8 * g + 4 * kh + tt;
          float val = tile[k * ld + cl];
          if (!rows_form) {
            const float dlt = val - stat[k];
            val = dlt < -103.9f ? -INFINITY : dlt - stat[K + k];
          }
          v[q][4 * g + tt] = c >= C ? 0.f : val;
          m = fmaxf(m, v[q][4 * g + tt]);
        }
    m = ck::clamp_finite(ck::xhalf_max(m));
    const float nml = exp_offset(m, 0.f);
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[q][r] = __builtin_amdgcn_exp2f(fmaf(v[q][r], kL2E, nml));
#pragma unroll
    for (int p = 0; p < NK; ++p) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(w_s + ((((p * NK + q) * 4 + g) * 64) + lane) * 4);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, v[q][4 * g + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, v[q][4 * g + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, v[q][4 * g + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, v[q][4 * g + 3], acc, 0, 0, 0);
        }
      if (c <= C) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 o4;
          o4.x = fmaf(__builtin_amdgcn_logf(acc[4 * g + 0]), kLN2, m);
          o4.y = fmaf(__builtin_amdgcn_logf(acc[4 * g + 1]), kLN2, m);
          o4.z = fmaf(__builtin_amdgcn_logf(acc[4 * g + 2]), kLN2, m);
          o4.w = fmaf(__builtin_amdgcn_logf(acc[4 * g + 3]), kLN2, m);
          *reinterpret_cast<float4*>(dst + static_cast<int64_t>(c) * K + 32 * p + 8 * g + 4 * kh) = o4;
        }
      }
    }
  }
}

// WIDE: the launch that carries the 64-unit table jobs -- kept apart so that their two-block tile does not set
// the register budget (and with it the occupancy) of every other job
template <bool WIDE>
__global__ void __launch_bounds__((WIDE ? kPW64 : kPW) * 64) softmax_batch_kernel(const JobTable t) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int bid = blockIdx.x;
  int ji = 0;
  while (ji + 1 < t.n && bid >= t.job[ji + 1].block_begin) ++ji;
  const ck_softmax_job& j = t.job[ji];
  const int blk = bid - j.block_begin;
  if constexpr (WIDE) {
    softmax_job_table_dense64(j, blk, tile);
    return;
  }
  const bool rows_form = j.len <= 256 && (j.len & 3) == 0;  // a (unit, all categories) row is one float4 per lane
  if (j.kind == 1 && rows_form)
    softmax_job_table_rows(j, blk, tile);
  else if (j.kind == 1)
    softmax_job_table(j, blk, tile);
  else if ((j.kind == 4 || j.kind == 5) && j.k == 32 && rows_form)
    softmax_job_table_dense_rows(j, blk, tile);
  else if (j.kind == 4 || j.kind == 5)
    softmax_job_table_dense(j, blk, tile);
  else
    softmax_job_rows(j, blk);
}

// Binomial log-pmf table (TorchBinomialLayer.log_unnormalized_likelihood, input.py:530-541, through
// torch.distributions.Binomial.log_prob): table[f, c, k] = c l - lgamma(c + 1) - lgamma(n - c + 1)
//   - (n max(l, 0) + n log1p(exp(-|l|)) - lgamma(n + 1)),  l = logits[f, k] or log(p) - log1p(-p) of the clamped
// probability (probs_to_logits); row n + 1 is the integral row (a normalised distribution: 0).
__global__ void __launch_bounds__(256) binomial_table_kernel(const float* __restrict__ p, int is_logits, float* __restrict__ table,
                                                             int64_t F, int K, int n) {
  const int64_t rows = n + 2;
  const int64_t total = F * rows * K;
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int k = static_cast<int>(e % K);
    const int c = static_cast<int>((e / K) % rows);
    const int64_t f = e / (K * rows);
    if (c == n + 1) {
      table[e] = 0.f;
      continue;
    }
    float l = p[f * K + k];
    if (!is_logits) {
      const float eps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps: clamp_probs
      const float q = fminf(fmaxf(l, eps), 1.f - eps);
      l = logf(q) - log1pf(-q);
    }
    const float nf = static_cast<float>(n), cf = static_cast<float>(c);
    const float norm = nf * fmaxf(l, 0.f) + nf * log1pf(expf(-fabsf(l))) - lgammaf(nf + 1.f);
    table[e] = cf * l - lgammaf(cf + 1.f) - lgammaf(nf - cf + 1.f) - norm;
  }
}

// log of the integral of the product of two Gaussian densities, for every pair of units (nodes.py:975-988):
//   out[f, i * K2 + j] = -0.5 (log 2 pi + log(s1_i^2 + s2_j^2) + (m1_i - m2_j)^2 / (s1_i^2 + s2_j^2))
__global__ void __launch_bounds__(256) gaussian_product_logz_kernel(const float* __restrict__ m1, const float* __restrict__ s1,
                                                                    const float* __restrict__ m2, const float* __restrict__ s2,
                                                                    float* __restrict__ out, int64_t F, int K1, int K2) {
  const int64_t n = F * K1 * K2;
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < n;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t f = e / (static_cast<int64_t>(K1) * K2);
    const int r = static_cast<int>(e - f * K1 * K2), i = r / K2, j = r - i * K2;
    const float a = s1[f * K1 + i], b = s2[f * K2 + j];
    const float var = a * a + b * b;
    const float d = m1[f * K1 + i] - m2[f * K2 + j];
    out[e] = -0.5f * (1.8378770664093453f + logf(var) + d * d * (1.f / var));
  }
}

// TorchGaussianProductMean / TorchGaussianProductStddev (nodes.py:865-938): the mean and the standard deviation of the product of
// two Gaussian densities for every unit pair (i, j): with v1 = s1_i^2, v2 = s2_j^2,
//   mean = (m1_i v2 + m2_j v1) / (v1 + v2)          stddev = sqrt(v1 v2 / (v1 + v2))
// op 0: mean (m1, m2 read), op 1: stddev.
__global__ void __launch_bounds__(256) gaussian_product_ms_kernel(int op, const float* __restrict__ m1, const float* __restrict__ s1,
                                                                  const float* __restrict__ m2, const float* __restrict__ s2,
                                                                  float* __restrict__ out, int64_t F, int K1, int K2) {
  const int64_t n = F * K1 * K2;
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < n;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t f = e / (static_cast<int64_t>(K1) * K2);
    const int r = static_cast<int>(e - f * K1 * K2), i = r / K2, j = r - i * K2;
    const float a = s1[f * K1 + i], b = s2[f * K2 + j];
    const float v1 = a * a, v2 = b * b;
    if (op == 0)
      out[e] = (m1[f * K1 + i] * v2 + m2[f * K2 + j] * v1) * (1.f / (v1 + v2));
    else
      out[e] = sqrtf(1.f / (1.f / v1 + 1.f / v2));
  }
}
// Their backward, one thread per (fold, unit) of either operand walking the other operand's units (no atomics).  With D = v1 + v2:
//   mean:    d/dm1 = v2 / D,  d/dm2 = v1 / D,  d/ds1 = 2 s1 (m2 - mean) / D,  d/ds2 = 2 s2 (m1 - mean) / D
//   stddev:  d/ds1 = (s1 / out) v2^2 / D^2,  d/ds2 = (s2 / out) v1^2 / D^2
__global__ void __launch_bounds__(256) gaussian_product_ms_bwd_kernel(int op, const float* __restrict__ m1, const float* __restrict__ s1,
                                                                      const float* __restrict__ m2, const float* __restrict__ s2,
                                                                      const float* __restrict__ dout, float* __restrict__ dm1,
                                                                      float* __restrict__ ds1, float* __restrict__ dm2,
                                                                      float* __restrict__ ds2, int64_t F, int K1, int K2) {
  const int64_t n = F * (K1 + K2);
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < n;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t f = e / (K1 + K2);
    const int u = static_cast<int>(e - f * (K1 + K2));
    const bool first = u < K1;
    const int i = first ? u : u - K1;
    const float sd = first ? s1[f * K1 + i] : s2[f * K2 + i];
    const float mu = op == 0 ? (first ? m1[f * K1 + i] : m2[f * K2 + i]) : 0.f;
    const int Ko = first ? K2 : K1;
    const float vt = sd * sd;
    float gm = 0.f, gs = 0.f;
    for (int j = 0; j < Ko; ++j) {
      const float os = first ? s2[f * K2 + j] : s1[f * K1 + j];
      const float g = first ? dout[(f * K1 + i) * K2 + j] : dout[(f * K1 + j) * K2 + i];
      const float vo = os * os, rd = 1.f / (vt + vo);
      if (op == 0) {
        const float om = first ? m2[f * K2 + j] : m1[f * K1 + j];
        const float mean = (mu * vo + om * vt) * rd;
        gm += g * vo * rd;
        gs += g * 2.f * sd * (om - mean) * rd;
      } else {
        const float out = sqrtf(vt * vo * rd);
        gs += g * (sd / out) * vo * vo * rd * rd;
      }
    }
    if (first) {
      if (op == 0) dm1[f * K1 + i] = gm;
      ds1[f * K1 + i] = gs;
    } else {
      if (op == 0) dm2[f * K2 + i] = gm;
      ds2[f * K2 + i] = gs;
    }
  }
}
// Backward of gaussian_product_logz_kernel: with v = s1_i^2 + s2_j^2, d = m1_i - m2_j and g = dout[f, i K2 + j],
//   dm1_i = sum_j -g d / v      ds1_i = sum_j g s1_i (d^2 / v - 1) / v      (and the mirror image for operand 2; four distinct buffers).
// One thread per (fold, unit) of either operand walks the other operand's units: no atomics, deterministic.
__global__ void __launch_bounds__(256) gaussian_product_logz_bwd_kernel(const float* __restrict__ m1, const float* __restrict__ s1,
                                                                        const float* __restrict__ m2, const float* __restrict__ s2,
                                                                        const float* __restrict__ dout, float* __restrict__ dm1,
                                                                        float* __restrict__ ds1, float* __restrict__ dm2,
                                                                        float* __restrict__ ds2, int64_t F, int K1, int K2) {
  const int64_t n = F * (K1 + K2);
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < n;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t f = e / (K1 + K2);
    const int u = static_cast<int>(e - f * (K1 + K2));
    const bool first = u < K1;
    const int i = first ? u : u - K1;
    const float mu = first ? m1[f * K1 + i] : m2[f * K2 + i];
    const float sd = first ? s1[f * K1 + i] : s2[f * K2 + i];
    const int Ko = first ? K2 : K1;
    float gm = 0.f, gs = 0.f;
    for (int j = 0; j < Ko; ++j) {
      const float om = first ? m2[f * K2 + j] : m1[f * K1 + j];
      const float os = first ? s2[f * K2 + j] : s1[f * K1 + j];
      const float g = first ? dout[(f * K1 + i) * K2 + j] : dout[(f * K1 + j) * K2 + i];
      const float v = sd * sd + os * os, d = mu - om;  // (d: this unit minus the other -- the sign of dm flips with the operand, d^2 does not)
      const float rv = 1.f / v;
      gm -= g * d * rv;
      gs += g * sd * (d * d * rv - 1.f) * rv;
    }
    if (first) {
      dm1[f * K1 + i] = gm;
      ds1[f * K1 + i] = gs;
    } else {
      dm2[f * K2 + i] = gm;
      ds2[f * K2 + i] = gs;
    }
  }
}

// ---- log-likelihood sum ------------------------------------------------------------------------
// Single workgroup: B is a batch (<= a few 10^5 rows), and a one-block tree gives a
// run-to-run deterministic fp64 sum (no atomics).
__global__ void __launch_bounds__(1024)
    ll_sum_kernel(const float* __restrict__ ll, int64_t B, int64_t stride, double* __restrict__ out) {
  __shared__ double part[16];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) acc += static_cast<double>(ll[i * stride]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < static_cast<int>(blockDim.x >> 6); ++i) t += part[i];
    out[0] = t;
    out[1] = static_cast<double>(B);
  }
}

unsigned grid1d(int64_t n, int cap = 2048) {
  return static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, cap));
}

// Softmax over LONG rows (Tucker weights: Ki^2 = 1024 / 4096 entries per row): one wavefront per row, the row
// held in registers between the single read and the single write (N4 float4 per lane, all loads in flight
// at once), so the kernel moves 2 x 4 bytes per entry -- the row-at-a-time loop of softmax_job_rows
// reads every row three times with one dependent load per pass.
template <int N4>
__global__ void __launch_bounds__(256) softmax_long_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int64_t rows, int len) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = len >> 2;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave; row < rows; row += static_cast<int64_t>(gridDim.x) * 4) {
    const float4* src = reinterpret_cast<const float4*>(in + row * len);
    float4 x[N4];
#pragma unroll
    for (int k = 0; k < N4; ++k) {
      const int i = lane + 64 * k;
      x[k] = i < n4 ? src[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < N4; ++k) mx = fmaxf(fmaxf(mx, fmaxf(x[k].x, x[k].y)), fmaxf(x[k].z, x[k].w));
    mx = ck::wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < N4; ++k) {
      x[k].x = __expf(x[k].x - mx);
      x[k].y = __expf(x[k].y - mx);
      x[k].z = __expf(x[k].z - mx);
      x[k].w = __expf(x[k].w - mx);
      sum += (x[k].x + x[k].y) + (x[k].z + x[k].w);
    }
    sum = ck::wave_sum(sum);
    float4* dst = reinterpret_cast<float4*>(out + row * len);
#pragma unroll
    for (int k = 0; k < N4; ++k) {
      const int i = lane + 64 * k;
      if (i < n4) dst[i] = make_float4(x[k].x / sum, x[k].y / sum, x[k].z / sum, x[k].w / sum);
    }
  }
}

bool long_row_job(const ck_softmax_job& j) {
  return j.kind == 0 && j.len >= 512 && j.len <= 4096 && j.len % 4 == 0 && ck::aligned16(j.in) && ck::aligned16(j.out);
}

int launch_long_rows(const ck_softmax_job& j, void* stream) {
  const int n4 = (static_cast<int>(j.len) / 4 + 63) / 64;
  const int64_t want = (j.rows + 3) / 4;
  dim3 grid(static_cast<unsigned>(std::min<int64_t>(want, 256 * 16))), block(256);
  const float* in = j.in;
  float* out = j.out;
  const int64_t rows = j.rows;
  const int len = static_cast<int>(j.len);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (n4 <= 2)
          hipLaunchKernelGGL(softmax_long_rows_kernel<2>, grid, block, 0, s, in, out, rows, len);
        else if (n4 <= 4)
          hipLaunchKernelGGL(softmax_long_rows_kernel<4>, grid, block, 0, s, in, out, rows, len);
        else if (n4 <= 8)
          hipLaunchKernelGGL(softmax_long_rows_kernel<8>, grid, block, 0, s, in, out, rows, len);
        else
          hipLaunchKernelGGL(softmax_long_rows_kernel<16>, grid, block, 0, s, in, out, rows, len);
        return hipGetLastError();
      },
      stream);
}

}  // namespace

extern "C" {

int ck_param_reduce(int op, const float* x, float* y, int64_t outer, int len, int64_t inner, void* stream) {
  CK_REQUIRE(x && y && outer > 0 && len > 0 && inner > 0, "ck_param_reduce: bad arguments");
  CK_REQUIRE(op == 0 || op == 1, "ck_param_reduce: op %d (0 product, 1 log-sum-exp)", op);
  dim3 grid(static_cast<unsigned>((outer * inner + 255) / 256)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(reduce_axis_kernel<false>, grid, block, 0, s, op, x, static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), y,
                           outer, len, inner);
        return hipGetLastError();
      },
      stream);
}

int ck_param_reduce_bwd(int op, const float* x, const float* y, const float* dy, float* dx, int64_t outer, int len, int64_t inner, void* stream) {
  CK_REQUIRE(x && y && dy && dx && outer > 0 && len > 0 && inner > 0, "ck_param_reduce_bwd: bad arguments");
  CK_REQUIRE(op == 0 || op == 1, "ck_param_reduce_bwd: op %d (0 product, 1 log-sum-exp)", op);
  dim3 grid(static_cast<unsigned>((outer * inner + 255) / 256)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(reduce_axis_kernel<true>, grid, block, 0, s, op, x, y, dy, dx, outer, len, inner);
        return hipGetLastError();
      },
      stream);
}

int ck_param_outer_sum(const float* a, const float* b, float* out, int64_t outer, int n1, int n2, int64_t inner, void* stream) {
  CK_REQUIRE(a && b && out && outer > 0 && n1 > 0 && n2 > 0 && inner > 0, "ck_param_outer_sum: bad arguments");
  const int64_t n = outer * n1 * n2 * inner;
  dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(outer_sum_kernel, grid, block, 0, s, a, b, out, outer, n1, n2, inner);
        return hipGetLastError();
      },
      stream);
}

int ck_param_outer_sum_bwd(const float* dout, float* dx, int64_t outer, int n1, int n2, int64_t inner, int which, void* stream) {
  CK_REQUIRE(dout && dx && outer > 0 && n1 > 0 && n2 > 0 && inner > 0 && (which == 0 || which == 1), "ck_param_outer_sum_bwd: bad arguments");
  const int64_t n = outer * (which == 0 ? n1 : n2) * inner;
  dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(outer_sum_bwd_kernel, grid, block, 0, s, dout, dx, outer, n1, n2, inner, which);
        return hipGetLastError();
      },
      stream);
}

int ck_param_softmax(const float* in, float* out, int64_t outer, int len, int64_t inner,
                     int log_space, void* stream) {
  CK_REQUIRE(in && out, "ck_param_softmax: null pointer");
  CK_REQUIRE(outer > 0 && len > 0 && inner > 0, "ck_param_softmax: non-positive size");
  if (inner == 1) {
    const int64_t rows = outer;
    dim3 grid(static_cast<unsigned>((rows + 3) / 4)), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(softmax_rows_kernel, grid, block, 0, s, in, out, rows, len, log_space);
          return hipGetLastError();
        },
        stream);
  }
  dim3 grid(static_cast<unsigned>((outer * inner + 255) / 256)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(softmax_strided_kernel, grid, block, 0, s, in, out, outer, len, inner, log_space);
        return hipGetLastError();
      },
      stream);
}

int ck_param_unary(int op, const float* in, float* out, int64_t n, float a, float b, void* stream) {
  CK_REQUIRE(in && out, "ck_param_unary: null pointer");
  CK_REQUIRE(n > 0, "ck_param_unary: n must be positive");
  CK_REQUIRE(op >= CK_UNARY_SIGMOID && op <= CK_UNARY_SOFTPLUS, "ck_param_unary: unknown op %d", op);
  dim3 grid(grid1d(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(unary_kernel, grid, block, 0, s, op, in, out, n, a, b);
        return hipGetLastError();
      },
      stream);
}

int ck_param_gather_folds(const float* in, const int64_t* idx, float* out, int64_t F_out,
                          int64_t per_fold, void* stream) {
  CK_REQUIRE(in && idx && out, "ck_param_gather_folds: null pointer");
  CK_REQUIRE(F_out > 0 && per_fold > 0, "ck_param_gather_folds: non-positive size");
  CK_REQUIRE(F_out <= 65535, "ck_param_gather_folds: F_out exceeds grid.y");
  dim3 grid(grid1d(per_fold, 64), static_cast<unsigned>(F_out)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gather_folds_kernel, grid, block, 0, s, in, idx, out, per_fold);
        return hipGetLastError();
      },
      stream);
}

int ck_param_conj(const float* in_c, float* out_c, int64_t n, void* stream) {
  CK_REQUIRE(in_c && out_c, "ck_param_conj: null pointer");
  CK_REQUIRE(n > 0, "ck_param_conj: n must be positive");
  dim3 grid(grid1d(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(conj_kernel, grid, block, 0, s, in_c, out_c, n);
        return hipGetLastError();
      },
      stream);
}

int ck_param_mixing_weight(const float* in, float* out, int F, int K, int H, void* stream) {
  CK_REQUIRE(in && out, "ck_param_mixing_weight: null pointer");
  CK_REQUIRE(F > 0 && K > 0 && H > 0, "ck_param_mixing_weight: non-positive size");
  CK_REQUIRE(F <= 65535, "ck_param_mixing_weight: F exceeds grid.y");
  dim3 grid(grid1d(static_cast<int64_t>(K) * H * K, 256), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(mixing_weight_kernel, grid, block, 0, s, in, out, K, H);
        return hipGetLastError();
      },
      stream);
}

int ck_param_bmm(const float* a, const float* b, float* out, int F, int M, int N, int Kd, int trans_a, int trans_b, int accumulate,
                     void* stream) {
  CK_REQUIRE(a && b && out, "ck_param_bmm: null pointer");
  CK_REQUIRE(F > 0 && M > 0 && N > 0 && Kd > 0, "ck_param_bmm: non-positive size");
  CK_REQUIRE(F <= 65535, "ck_param_bmm: F exceeds grid.z");
  if (trans_a == 2 && !(M == Kd && (M & 31) == 0 && (N & 31) == 0 && ck::aligned16(a) && ck::aligned16(b)))
    return ck::fail(CK_ERR_UNSUPPORTED, "ck_param_bmm: the symmetrised form (trans_a = 2) needs M = Kd and extents that are multiples of 32");
  if ((M & 31) == 0 && (N & 31) == 0 && (Kd & 31) == 0 && ck::aligned16(a) && ck::aligned16(b) && N <= 65535 * 32 && M <= 65535 * 32) {
    dim3 grid(N / 32, M / 32, F), block(64);
    return ck::dispatch(
        [=](hipStream_t s) {
          auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, 0, s, a, b, out, M, N, Kd, accumulate); };
          if (trans_a == 2) trans_b ? go(bmm_mfma_kernel<2, true>) : go(bmm_mfma_kernel<2, false>);
          else if (trans_a && trans_b) go(bmm_mfma_kernel<1, true>);
          else if (trans_a) go(bmm_mfma_kernel<1, false>);
          else if (trans_b) go(bmm_mfma_kernel<0, true>);
          else go(bmm_mfma_kernel<0, false>);
          return hipGetLastError();
        },
        stream);
  }
  if (M * static_cast<int64_t>(N) >= 512 && ck::aligned16(out)) {  // (small products, e.g. a row of ones times a block: the 16 x 16 form)
    dim3 grid((N + kBT_N - 1) / kBT_N, (M + kBT_M - 1) / kBT_M, F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          const bool vec = (M & 3) == 0 && (N & 3) == 0 && (Kd & 3) == 0 && ck::aligned16(a) && ck::aligned16(b);
          auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, 0, s, a, b, out, M, N, Kd, accumulate); };
          if (vec) {
            if (trans_a && trans_b) go(bmm_tile_kernel<true, true, true>);
            else if (trans_a) go(bmm_tile_kernel<true, false, true>);
            else if (trans_b) go(bmm_tile_kernel<false, true, true>);
            else go(bmm_tile_kernel<false, false, true>);
          } else {
            if (trans_a && trans_b) go(bmm_tile_kernel<true, true, false>);
            else if (trans_a) go(bmm_tile_kernel<true, false, false>);
            else if (trans_b) go(bmm_tile_kernel<false, true, false>);
            else go(bmm_tile_kernel<false, false, false>);
          }
          return hipGetLastError();
        },
        stream);
  }
  dim3 grid((N + kMM - 1) / kMM, (M + kMM - 1) / kMM, F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(bmm_kernel, grid, block, 0, s, a, b, out, M, N, Kd, trans_a, trans_b, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_param_transpose_last2(const float* in, float* out, int64_t R, int A, int Bd, int take_log,
                             int out_rows, void* stream) {
  CK_REQUIRE(in && out, "ck_param_transpose_last2: null pointer");
  CK_REQUIRE(R > 0 && A > 0 && Bd > 0, "ck_param_transpose_last2: non-positive size");
  CK_REQUIRE(out_rows >= Bd, "ck_param_transpose_last2: out_rows=%d < Bd=%d", out_rows, Bd);
  CK_REQUIRE(R <= 65535, "ck_param_transpose_last2: R exceeds grid.z");
  dim3 grid((Bd + 31) / 32, (A + 31) / 32, static_cast<unsigned>(R)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(transpose_last2_kernel<float>, grid, block, 0, s, in, out, A, Bd, take_log, out_rows);
        return hipGetLastError();
      },
      stream);
}

int ck_param_transpose_last2_c(const float* in_c, float* out_c, int64_t R, int A, int Bd, int out_rows, void* stream) {
  CK_REQUIRE(in_c && out_c, "ck_param_transpose_last2_c: null pointer");
  CK_REQUIRE(R > 0 && A > 0 && Bd > 0, "ck_param_transpose_last2_c: non-positive size");
  CK_REQUIRE(out_rows >= Bd, "ck_param_transpose_last2_c: out_rows=%d < Bd=%d", out_rows, Bd);
  CK_REQUIRE(R <= 65535, "ck_param_transpose_last2_c: R exceeds grid.z");
  dim3 grid((Bd + 31) / 32, (A + 31) / 32, static_cast<unsigned>(R)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(transpose_last2_kernel<float2>, grid, block, 0, s, reinterpret_cast<const float2*>(in_c),
                           reinterpret_cast<float2*>(out_c), A, Bd, 0, out_rows);
        return hipGetLastError();
      },
      stream);
}

int ck_param_table_integral_row(float* table, int F, int C, int K, int mode, void* stream) {
  CK_REQUIRE(table != nullptr, "ck_param_table_integral_row: null pointer");
  CK_REQUIRE(F > 0 && C > 0 && K > 0 && mode >= 0 && mode <= 3, "ck_param_table_integral_row: bad arguments");
  dim3 grid(F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(table_integral_row_kernel, grid, block, 0, s, table, C, K, mode);
        return hipGetLastError();
      },
      stream);
}

int ck_param_softmax_batch(const ck_softmax_job* jobs, int njobs, void* stream) {
  CK_REQUIRE(jobs != nullptr && njobs > 0, "ck_param_softmax_batch: no jobs");
  for (int i = 0; i < njobs; ++i)
    if (jobs[i].in && jobs[i].out && jobs[i].rows > 0 && long_row_job(jobs[i]))
      if (int st = launch_long_rows(jobs[i], stream)) return st;
  for (int wide = 0; wide < 2; ++wide) {
    int start = 0;
    while (start < njobs) {
      JobTable t{};
      int blocks = 0;
      size_t lds = 0;
      while (start < njobs && t.n < kMaxJobs) {
        ck_softmax_job j = jobs[start];
        const int idx = start++;
        CK_REQUIRE(j.in && j.out && j.rows > 0 && j.len > 0, "ck_param_softmax_batch: bad job %d", idx);
        CK_REQUIRE(j.kind >= 0 && j.kind <= 5 && j.kind != 3, "ck_param_softmax_batch: job %d has unknown kind %d", idx, j.kind);
        CK_REQUIRE(j.kind < 2 || j.kind >= 4 || (j.len == 32 && j.rows % 32 == 0),
                   "ck_param_softmax_batch: tiled job %d needs len = 32 and rows %% 32 = 0", idx);
        CK_REQUIRE(j.kind < 4 || ((j.k == 32 || (j.k == 64 && j.kind == 4)) && j.in2 != nullptr),
                   "ck_param_softmax_batch: job %d (kind 4/5) needs k = 32 (kind 4: or 64) and in2", idx);
        CK_REQUIRE(j.kind != 5 || j.out2 != nullptr, "ck_param_softmax_batch: job %d (kind 5) needs out2", idx);
        if ((j.kind == 4 && j.k == 64) != (wide == 1)) continue;  // the other pass takes it
        if (long_row_job(j)) continue;                             // done by softmax_long_rows_kernel
        j.block_begin = blocks;
        if (j.kind != 1 && j.kind < 4) {
          blocks += static_cast<int>((j.rows + 16 * kPW - 1) / (16 * kPW));
        } else {
          CK_REQUIRE(j.k > 0, "ck_param_softmax_batch: job %d needs k > 0", idx);
          size_t need = (static_cast<size_t>(j.k) * (j.len + 4) + 2 * j.k + (j.kind >= 4 ? j.k * j.k : 0)) * sizeof(float);  // (row stride up to len + 4)
          if (j.kind == 1) need = std::max(need, static_cast<size_t>(j.len) * (j.k + 1) * sizeof(float));  // (transposed: [len][k + 1])
          if (need > 160 * 1024)
            return ck::fail(CK_ERR_UNSUPPORTED, "ck_param_softmax_batch: C*K=%d too large for the table job", j.len * j.k);
          lds = std::max(lds, need);
          blocks += static_cast<int>(j.rows);
        }
        t.job[t.n++] = j;
      }
      if (t.n == 0) continue;
      const dim3 grid(blocks), block((wide ? kPW64 : kPW) * 64);
      int st = ck::dispatch(
          [=](hipStream_t s) {
            auto go = [&](auto kern) {
              if (lds > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
                if (e != hipSuccess) return e;
              }
              hipLaunchKernelGGL(kern, grid, block, lds, s, t);
              return hipGetLastError();
            };
            return wide ? go(softmax_batch_kernel<true>) : go(softmax_batch_kernel<false>);
          },
          stream);
      if (st != CK_OK) return st;
    }
  }
  return CK_OK;
}

int ck_param_binomial_table(const float* p, int is_logits, float* table, int64_t F, int K, int total_count, void* stream) {
  CK_REQUIRE(p && table, "ck_param_binomial_table: null pointer");
  CK_REQUIRE(F > 0 && K > 0 && total_count >= 0, "ck_param_binomial_table: bad sizes");
  dim3 grid(grid1d(F * (total_count + 2) * K)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(binomial_table_kernel, grid, block, 0, s, p, is_logits, table, F, K, total_count);
        return hipGetLastError();
      },
      stream);
}

int ck_param_gaussian_product_logz(const float* mean1, const float* stddev1, const float* mean2, const float* stddev2,
                                   float* out, int64_t F, int K1, int K2, void* stream) {
  CK_REQUIRE(mean1 && stddev1 && mean2 && stddev2 && out, "ck_param_gaussian_product_logz: null pointer");
  CK_REQUIRE(F > 0 && K1 > 0 && K2 > 0, "ck_param_gaussian_product_logz: non-positive size");
  dim3 grid(grid1d(F * K1 * K2)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gaussian_product_logz_kernel, grid, block, 0, s, mean1, stddev1, mean2, stddev2, out, F, K1, K2);
        return hipGetLastError();
      },
      stream);
}

int ck_param_gaussian_product_ms(int op, const float* mean1, const float* stddev1, const float* mean2, const float* stddev2, float* out,
                                 int F, int K1, int K2, void* stream) {
  CK_REQUIRE(op == 0 || op == 1, "ck_param_gaussian_product_ms: op %d (0 mean, 1 stddev)", op);
  CK_REQUIRE(stddev1 && stddev2 && out && (op == 1 || (mean1 && mean2)), "ck_param_gaussian_product_ms: null pointer");
  CK_REQUIRE(F > 0 && K1 > 0 && K2 > 0, "ck_param_gaussian_product_ms: non-positive size");
  const int64_t n = static_cast<int64_t>(F) * K1 * K2;
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 4096))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gaussian_product_ms_kernel, grid, block, 0, s, op, mean1, stddev1, mean2, stddev2, out, static_cast<int64_t>(F), K1, K2);
        return hipGetLastError();
      },
      stream);
}

int ck_param_gaussian_product_ms_bwd(int op, const float* mean1, const float* stddev1, const float* mean2, const float* stddev2,
                                     const float* dout, float* dmean1, float* dstddev1, float* dmean2, float* dstddev2, int F, int K1, int K2,
                                     void* stream) {
  CK_REQUIRE(op == 0 || op == 1, "ck_param_gaussian_product_ms_bwd: op %d (0 mean, 1 stddev)", op);
  CK_REQUIRE(stddev1 && stddev2 && dout && dstddev1 && dstddev2 && (op == 1 || (mean1 && mean2 && dmean1 && dmean2)),
             "ck_param_gaussian_product_ms_bwd: null pointer");
  CK_REQUIRE(F > 0 && K1 > 0 && K2 > 0, "ck_param_gaussian_product_ms_bwd: non-positive size");
  const int64_t n = static_cast<int64_t>(F) * (K1 + K2);
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 4096))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gaussian_product_ms_bwd_kernel, grid, block, 0, s, op, mean1, stddev1, mean2, stddev2, dout, dmean1, dstddev1, dmean2,
                           dstddev2, static_cast<int64_t>(F), K1, K2);
        return hipGetLastError();
      },
      stream);
}

int ck_param_gaussian_product_logz_bwd(const float* mean1, const float* stddev1, const float* mean2, const float* stddev2,
                                       const float* dout, float* dmean1, float* dstddev1, float* dmean2, float* dstddev2, int64_t F,
                                       int K1, int K2, void* stream) {
  CK_REQUIRE(mean1 && stddev1 && mean2 && stddev2 && dout && dmean1 && dstddev1 && dmean2 && dstddev2,
             "ck_param_gaussian_product_logz_bwd: null pointer");
  CK_REQUIRE(F > 0 && K1 > 0 && K2 > 0, "ck_param_gaussian_product_logz_bwd: non-positive size");
  dim3 grid(grid1d(F * (K1 + K2))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gaussian_product_logz_bwd_kernel, grid, block, 0, s, mean1, stddev1, mean2, stddev2, dout, dmean1, dstddev1,
                           dmean2, dstddev2, F, K1, K2);
        return hipGetLastError();
      },
      stream);
}

int ck_ll_sum(const float* ll, int64_t B, int64_t stride, double* out_dev, void* stream) {
  CK_REQUIRE(ll && out_dev, "ck_ll_sum: null pointer");
  CK_REQUIRE(B > 0 && stride > 0, "ck_ll_sum: non-positive size");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(ll_sum_kernel, dim3(1), dim3(1024), 0, s, ll, B, stride, out_dev);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
