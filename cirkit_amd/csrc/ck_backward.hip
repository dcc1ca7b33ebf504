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
#include "ck_opt.h"
#include "ck_tile.h"

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
                        const int64_t* __restrict__ row_off, const int64_t* __restrict__ grow_off,
                        const float* __restrict__ w,
                        const float* __restrict__ /*out: not needed, y is recomputed*/,
                        const float* __restrict__ gout, float* __restrict__ dw, int H, int B, int Ki,
                        int Ko, int mode, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int N = mode == CK_SUM_PROD ? Ki : H * Ki;
  if (mode == CK_SUM_KRON) {  // TorchTuckerLayer (optimized.py:89-103): the weight contracts the Kronecker product of the children
    N = 1;
    for (int h = 0; h < H; ++h) N *= Ki;
  }
  float* e_s = reinterpret_cast<float*>(smem);   // [TB][N]
  float* gv_s = e_s + static_cast<size_t>(TB) * N;  // [TB][N]
  float* gy_s = gv_s + static_cast<size_t>(TB) * N;  // [TB][Ko]
  float* w_s = gy_s + static_cast<size_t>(TB) * Ko;  // [64][kBwdNC+1]
  float* m_s = w_s + 64 * (kBwdNC + 1);              // [TB]
  float* eh_s = m_s + TB;                            // CK_SUM_KRON: [TB][H * Ki] per-child exp(v_h - max_h)
  const int f = blockIdx.y;
  const int b0 = blockIdx.x * TB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t* gro = grow_off + static_cast<int64_t>(f) * H;  // where the children's gradients go
  const float* wf = w + static_cast<int64_t>(f) * Ko * N;
  float* dwf = dw + static_cast<int64_t>(f) * Ko * N;

  if (mode == CK_SUM_KRON) {
    // phase A (Tucker): one maximum PER CHILD (semiring.py:383-408 applied to the einsum's operands), e = e_0 (x) e_1 (x) ...
    // with child 0 most significant (n = i_0 Ki^(H-1) + ... + i_(H-1), optimized.py:89-103)
    for (int rh = wave; rh < TB * H; rh += 4) {
      const int r = rh / H, h = rh - r * H, b = b0 + r;
      float mx = -INFINITY;
      for (int i = lane; i < Ki; i += 64) {
        const float v = b < B ? arena[ro[h] + static_cast<int64_t>(b) * Ki + i] : -INFINITY;
        eh_s[(r * H + h) * Ki + i] = v;
        mx = fmaxf(mx, v);
      }
      mx = ck::clamp_finite(ck::wave_max(mx));
      for (int i = lane; i < Ki; i += 64) eh_s[(r * H + h) * Ki + i] = expf(eh_s[(r * H + h) * Ki + i] - mx);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TB * N; i += 256) {
      const int r = i / N;
      int n = i - r * N;
      float p = 1.f;
      for (int h = H - 1; h >= 0; --h) {
        p *= eh_s[(r * H + h) * Ki + n % Ki];
        n /= Ki;
      }
      e_s[i] = p;
    }
    __syncthreads();
  }
  // phase A: v, m, e (rows beyond B are zeroed so they contribute nothing)
  for (int r = wave; r < TB && mode != CK_SUM_KRON; r += 4) {
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
  if (mode == CK_SUM_KRON) {
    // phase C (Tucker): d loss / d v_h[i] = sum over the n whose digit h is i of e[n] (W^T gy)[n]  (the factor e_h[i] is
    // part of e[n]); one thread per (row, child, unit), the other digits in a fixed order: deterministic
    for (int i = threadIdx.x; i < TB * N; i += 256) gv_s[i] *= e_s[i];
    __syncthreads();
    const int rest = N / Ki;
    for (int t = threadIdx.x; t < TB * H * Ki; t += 256) {
      const int r = t / (H * Ki), h = (t / Ki) % H, i = t % Ki, b = b0 + r;
      if (b >= B) continue;
      int sh = 1;  // stride of digit h
      for (int k = h + 1; k < H; ++k) sh *= Ki;
      float acc = 0.f;
      for (int m = 0; m < rest; ++m) acc += gv_s[static_cast<size_t>(r) * N + (m / sh) * (sh * Ki) + i * sh + (m % sh)];
      grad_store(garena + gro[h] + static_cast<int64_t>(b) * Ki + i, acc, accumulate);
    }
    return;
  }
  // phase C: gv *= e, scatter to the children
  for (int i = threadIdx.x; i < TB * N; i += 256) {
    const int r = i / N, n = i - r * N, b = b0 + r;
    if (b >= B) continue;
    const float g = gv_s[i] * e_s[i];
    if (mode == CK_SUM_PROD) {
      for (int h = 0; h < H; ++h) grad_store(garena + gro[h] + static_cast<int64_t>(b) * Ki + n, g, accumulate);
    } else {
      const int h = n / Ki, k = n - h * Ki;
      grad_store(garena + gro[h] + static_cast<int64_t>(b) * Ki + k, g, accumulate);
    }
  }
}

// One output unit over 32 inputs (the scalar root of a circuit, CK_SUM_PROD / H = 1): a half-wave per batch row, lane = input
// unit, 32 half-waves per block and rows_per_block rows per block, so few blocks share the 32 words of dW (atomics on ONE
// cache line retire ~8 ns apart: 16 k of them, one per lane and 8-row block, made the launch 133 us).  dW: lane registers ->
// LDS over the block's half-waves -> one atomic per input unit and block.
__global__ void __launch_bounds__(1024)
    sum_lse_bwd_scalar32(const float* __restrict__ arena, float* __restrict__ garena, const int64_t* __restrict__ row_off,
                         const int64_t* __restrict__ grow_off, const float* __restrict__ w, const float* __restrict__ gout,
                         float* __restrict__ dw, int H, int B, int rows_per_block, int accumulate) {
  __shared__ float part[32][33];
  const int f = blockIdx.y, i = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t* gro = grow_off + static_cast<int64_t>(f) * H;
  const float wi = w[static_cast<int64_t>(f) * kK + i];
  float dwi = 0.f;
  const int b0 = blockIdx.x * rows_per_block;
  for (int b = b0 + hw; b < min(b0 + rows_per_block, B); b += 32) {
    float v = 0.f;
    for (int h = 0; h < H; ++h) v += arena[ro[h] + static_cast<int64_t>(b) * kK + i];
    float m = v;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 32));
    m = ck::clamp_finite(m);
    const float e = expf(v - m);
    float y = wi * e;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) y += __shfl_xor(y, s, 32);
    const float go = gout[static_cast<int64_t>(f) * B + b];
    const float gy = (y > 0.f && go != 0.f) ? go / y : 0.f;
    dwi = fmaf(gy, e, dwi);
    const float gv = wi * e * gy;
    for (int h = 0; h < H; ++h) grad_store(garena + gro[h] + static_cast<int64_t>(b) * kK + i, gv, accumulate);
  }
  part[hw][i] = dwi;
  __syncthreads();
  if (hw == 0) {
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) sacc += part[k][i];
    if (sacc != 0.f) atomicAdd(dw + static_cast<int64_t>(f) * kK + i, sacc);
  }
}

// K = 32 product-type sum layers (dense / CP-T) on the register tile of ck_tile.h: three exact fp32
// MFMA contractions per 32-row tile --  y = W e,  gv = e * (W^T gy),  dW += gy^T e.  The last one
// contracts over the batch rows, so gy and e are transposed through LDS (8 KB per wave) into the
// MFMA A / B operand layouts; dW accumulates in registers across the wave's tiles, is reduced over
// the workgroup's waves in LDS and leaves with one atomic per element and workgroup.
__global__ void __launch_bounds__(256)
    sum_lse_bwd_tile32(const float* __restrict__ arena, float* __restrict__ garena,
                       const int64_t* __restrict__ row_off, const int64_t* __restrict__ grow_off,
                       const float* __restrict__ w, const float* __restrict__ gout, float* __restrict__ dw, int H, int B,
                       int tiles_per_wave, int accumulate) {
  __shared__ __attribute__((aligned(16))) float lds[4][2][32 * 32];  // per wave: gy[b][o], e[b][n]
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const float* wf = w + static_cast<int64_t>(f) * kK * kK;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t* gro = grow_off + static_cast<int64_t>(f) * H;  // where the children's gradients go
  WRegs wr;  // A operand of y = W e: lane (o, kh) holds W[o][u(s, kh)]
  load_w<CK_W_ROWMAJOR>(wf, lane, wr);
  float wt[16];  // A operand of gv = W^T gy: lane (n, kh) holds W[u(s, kh)][n]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t) wt[4 * g + t] = wf[(8 * g + 4 * kh + t) * kK + b_in];
  f32x16 dwacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dwacc[r] = 0.f;
  float* gy_s = lds[wave][0];
  float* e_s = lds[wave][1];
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
    const float m = row_max16(v);
    float e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) e[j] = live ? __builtin_amdgcn_exp2f((v[j] - m) * kL2E) : 0.f;  // (v_exp_f32, as the forward: ~5e-7 relative)
    // y = W e
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr.q[g].x, e[4 * g + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr.q[g].y, e[4 * g + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr.q[g].z, e[4 * g + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr.q[g].w, e[4 * g + 3], acc, 0, 0, 0);
    }
    // gy = gout / y
    float gy[16];
    {
      float go[16];
      tile_load(gout + (static_cast<int64_t>(f) * B + bl) * kK + 4 * kh, go);
#pragma unroll
      for (int r = 0; r < 16; ++r) gy[r] = (live && acc[r] > 0.f && go[r] != 0.f) ? go[r] * __builtin_amdgcn_rcpf(acc[r]) : 0.f;
    }
    // gv = e * (W^T gy)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[s2], gy[s2], acc, 0, 0, 0);
    if (live) {
      for (int h = 0; h < H; ++h) {
        float* dst = garena + gro[h] + static_cast<int64_t>(b) * kK + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 gv = make_float4(acc[4 * g] * e[4 * g], acc[4 * g + 1] * e[4 * g + 1], acc[4 * g + 2] * e[4 * g + 2],
                                        acc[4 * g + 3] * e[4 * g + 3]);
          float4* d4 = reinterpret_cast<float4*>(dst + 8 * g);
          if (accumulate == 0) {
            *d4 = gv;
          } else if (accumulate == 1) {
            const float4 o = *d4;
            *d4 = make_float4(o.x + gv.x, o.y + gv.y, o.z + gv.z, o.w + gv.w);
          } else {
            atomicAdd(dst + 8 * g + 0, gv.x);
            atomicAdd(dst + 8 * g + 1, gv.y);
            atomicAdd(dst + 8 * g + 2, gv.z);
            atomicAdd(dst + 8 * g + 3, gv.w);
          }
        }
      }
    }
    // dW += gy^T e : transpose both tiles through LDS (row b, unit u at [b * 32 + u])
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<float4*>(gy_s + b_in * 32 + 8 * g + 4 * kh) = make_float4(gy[4 * g], gy[4 * g + 1], gy[4 * g + 2], gy[4 * g + 3]);
      *reinterpret_cast<float4*>(e_s + b_in * 32 + 8 * g + 4 * kh) = make_float4(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int bb = 16 * kh + s2;  // batch row contracted by lanes (., kh) at step s2
      dwacc = __builtin_amdgcn_mfma_f32_32x32x2f32(gy_s[bb * 32 + b_in], e_s[bb * 32 + b_in], dwacc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // reduce dW over the 4 waves, then one atomic per element: D[o][n] in lane (n, hi) reg r, o = u(r, hi)
  __syncthreads();
  float* red = &lds[0][0][0];  // 4 x 1024 floats
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = 8 * (r >> 2) + 4 * kh + (r & 3);
    red[wave * 1024 + o * 32 + b_in] = dwacc[r];
  }
  __syncthreads();
  float* dwf = dw + static_cast<int64_t>(f) * kK * kK;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const float sacc = red[i] + red[1024 + i] + red[2048 + i] + red[3072 + i];
    if (sacc != 0.f) atomicAdd(&dwf[i], sacc);
  }
}

// The same three contractions for K = 64 (two 32-unit blocks per row): both operand layouts of W -- for y = W e and
// for gv = W^T gy -- are staged in LDS once per workgroup and shared by its four waves (they do not fit in
// registers next to the 64 x 64 dW accumulator); gy and e go through per-wave LDS tiles (row stride 68: the
// transposed 16-byte writes of 32 rows spread over all banks) for the batch-contracted dW += gy^T e.
constexpr int kT64 = 68;  // row stride of the per-wave transpose tiles
__global__ void __launch_bounds__(256)
    sum_lse_bwd_tile64(const float* __restrict__ arena, float* __restrict__ garena, const int64_t* __restrict__ row_off,
                       const int64_t* __restrict__ grow_off, const float* __restrict__ w, const float* __restrict__ gout, float* __restrict__ dw, int H, int B,
                       int tiles_per_wave, int accumulate) {
  constexpr int K = 64;
  extern __shared__ __attribute__((aligned(16))) float sm64[];
  float* w_a = sm64;          // [p][q][g][lane][4]: W[32p + (lane & 31)][32q + 8g + 4(lane >> 5) + t]
  float* w_t = sm64 + 4096;   // [q][p][g][lane][4]: W[32p + 8g + 4(lane >> 5) + t][32q + (lane & 31)]
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  float* gy_s = sm64 + 8192 + wave * (2 * 32 * kT64);
  float* e_s = gy_s + 32 * kT64;
  const float* wf = w + static_cast<int64_t>(f) * K * K;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t* gro = grow_off + static_cast<int64_t>(f) * H;  // where the children's gradients go
  {  // all loads of the staging first (a load -> store loop would pay the memory latency once per iteration)
    float4 a4[4];
    float tv[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = threadIdx.x + 256 * u;
      const int ln = i & 63, g = (i >> 6) & 3, q = (i >> 8) & 1, p = i >> 9;
      a4[u] = *reinterpret_cast<const float4*>(wf + (32 * p + (ln & 31)) * K + 32 * q + 8 * g + 4 * (ln >> 5));
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = threadIdx.x + 256 * u;
      const int t = i & 3, ln = (i >> 2) & 63, g = (i >> 8) & 3, p = (i >> 10) & 1, q = i >> 11;
      tv[u] = wf[(32 * p + 8 * g + 4 * (ln >> 5) + t) * K + 32 * q + (ln & 31)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(w_a + 4 * (threadIdx.x + 256 * u)) = a4[u];
#pragma unroll
    for (int u = 0; u < 16; ++u) w_t[threadIdx.x + 256 * u] = tv[u];
  }
  __syncthreads();
  f32x16 dwacc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) dwacc[p][q][r] = 0.f;
  const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const int b0 = (tile0 + tt) * 32;
    if (b0 >= B) break;
    const int b = b0 + b_in;
    const bool live = b < B;
    const int bl = live ? b : B - 1;
    float e[2][16];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int j = 0; j < 16; ++j) e[q][j] = 0.f;
      for (int h = 0; h < H; ++h) tile_load_add(arena + ro[h] + static_cast<int64_t>(bl) * K + 32 * q + 4 * kh, e[q]);
    }
    float m = e[0][0];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) m = fmaxf(m, e[q][j]);
    m = ck::clamp_finite(ck::xhalf_max(m));
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) e[q][j] = live ? __builtin_amdgcn_exp2f((e[q][j] - m) * kL2E) : 0.f;  // (v_exp_f32, as the forward)
    // y = W e, gy = gout / y
    float gy[2][16];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(w_a + ((((p * 2 + q) * 4 + g) * 64) + lane) * 4);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, e[q][4 * g + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, e[q][4 * g + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, e[q][4 * g + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, e[q][4 * g + 3], acc, 0, 0, 0);
        }
      float go[16];
      tile_load(gout + (static_cast<int64_t>(f) * B + bl) * K + 32 * p + 4 * kh, go);
#pragma unroll
      for (int r = 0; r < 16; ++r) gy[p][r] = (live && acc[r] > 0.f && go[r] != 0.f) ? go[r] * __builtin_amdgcn_rcpf(acc[r]) : 0.f;
    }
    // gv = e * (W^T gy)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(w_t + ((((q * 2 + p) * 4 + g) * 64) + lane) * 4);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, gy[p][4 * g + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, gy[p][4 * g + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, gy[p][4 * g + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, gy[p][4 * g + 3], acc, 0, 0, 0);
        }
      if (live) {
        for (int h = 0; h < H; ++h) {
          float* dst = garena + gro[h] + static_cast<int64_t>(b) * K + 32 * q + 4 * kh;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 gv = make_float4(acc[4 * g] * e[q][4 * g], acc[4 * g + 1] * e[q][4 * g + 1],
                                          acc[4 * g + 2] * e[q][4 * g + 2], acc[4 * g + 3] * e[q][4 * g + 3]);
            float4* d4 = reinterpret_cast<float4*>(dst + 8 * g);
            if (accumulate == 0) {
              *d4 = gv;
            } else if (accumulate == 1) {
              const float4 o = *d4;
              *d4 = make_float4(o.x + gv.x, o.y + gv.y, o.z + gv.z, o.w + gv.w);
            } else {
              atomicAdd(dst + 8 * g + 0, gv.x);
              atomicAdd(dst + 8 * g + 1, gv.y);
              atomicAdd(dst + 8 * g + 2, gv.z);
              atomicAdd(dst + 8 * g + 3, gv.w);
            }
          }
        }
      }
    }
    // dW += gy^T e: both tiles row-major through LDS (row b, unit u at [b * kT64 + u])
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(gy_s + b_in * kT64 + 32 * q + 8 * g + 4 * kh) =
            make_float4(gy[q][4 * g], gy[q][4 * g + 1], gy[q][4 * g + 2], gy[q][4 * g + 3]);
        *reinterpret_cast<float4*>(e_s + b_in * kT64 + 32 * q + 8 * g + 4 * kh) =
            make_float4(e[q][4 * g], e[q][4 * g + 1], e[q][4 * g + 2], e[q][4 * g + 3]);
      }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int bb = 16 * kh + s2;  // batch row contracted by lanes (., kh) at step s2
      const float a0 = gy_s[bb * kT64 + b_in], a1 = gy_s[bb * kT64 + 32 + b_in];
      const float c0 = e_s[bb * kT64 + b_in], c1 = e_s[bb * kT64 + 32 + b_in];
      dwacc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, dwacc[0][0], 0, 0, 0);
      dwacc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c1, dwacc[0][1], 0, 0, 0);
      dwacc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c0, dwacc[1][0], 0, 0, 0);
      dwacc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, dwacc[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // reduce dW over the 4 waves, then one atomic per element: D[o][n] in lane (n, hi) reg r, o = u(r, hi)
  __syncthreads();
  float* red = sm64 + 8192;  // 4 x 4096 floats (the transpose tiles are no longer needed)
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 32 * p + 8 * (r >> 2) + 4 * kh + (r & 3);
        red[wave * 4096 + o * K + 32 * q + b_in] = dwacc[p][q][r];
      }
  __syncthreads();
  float* dwf = dw + static_cast<int64_t>(f) * K * K;
  for (int i = threadIdx.x; i < 4096; i += 256) {
    const float sacc = (red[i] + red[4096 + i]) + (red[8192 + i] + red[12288 + i]);
    if (sacc != 0.f) atomicAdd(&dwf[i], sacc);
  }
}

// Gradient of children that several folds of ONE layer share (a region used by several partitionings): the layer's
// backward writes every (fold, child slot) contribution to its own block of a temporary -- plain stores -- and this
// kernel adds the blocks of each distinct child in list order: no float atomics (69 M of them per launch at
// BASELINE config 4, the bulk of that backward) and a summation order that does not depend on scheduling.
__global__ void __launch_bounds__(256)
    segment_add_kernel(const float* __restrict__ tmp, const int32_t* __restrict__ cptr, const int32_t* __restrict__ clist,
                       const int64_t* __restrict__ coff, float* __restrict__ garena, int64_t block_elems) {
  const int c = blockIdx.y;
  const int s0 = cptr[c], s1 = cptr[c + 1];
  float* dst = garena + coff[c];
  if ((block_elems & 3) == 0) {
    const int64_t n4 = block_elems >> 2;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n4;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
      float4 a = reinterpret_cast<float4*>(dst)[i];
      for (int j = s0; j < s1; ++j) {
        const float4 v = reinterpret_cast<const float4*>(tmp + static_cast<int64_t>(clist[j]) * block_elems)[i];
        a.x += v.x;
        a.y += v.y;
        a.z += v.z;
        a.w += v.w;
      }
      reinterpret_cast<float4*>(dst)[i] = a;
    }
  } else {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < block_elems;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
      float a = dst[i];
      for (int j = s0; j < s1; ++j) a += tmp[static_cast<int64_t>(clist[j]) * block_elems + i];
      dst[i] = a;
    }
  }
}

// TorchKroneckerLayer backward (inner.py:178-187): out[b, n] = sum_h v_h[b, digit_h(n)] in log space, child 0 most
// significant, so d loss / d v_h[b, i] = sum of gout[b, n] over the n whose digit h is i.  One thread per (row, child, unit),
// the other digits in a fixed order.
__global__ void __launch_bounds__(256)
    kronecker_bwd_kernel(float* __restrict__ garena, const int64_t* __restrict__ row_off, const float* __restrict__ gout, int H,
                         int B, int K, int N, int accumulate) {
  const int f = blockIdx.y;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* g = gout + static_cast<int64_t>(f) * B * N;
  const int rest = N / K;
  for (int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; t < static_cast<int64_t>(B) * H * K;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(t / (H * K)), h = static_cast<int>((t / K) % H), i = static_cast<int>(t % K);
    int sh = 1;
    for (int k = h + 1; k < H; ++k) sh *= K;
    float acc = 0.f;
    for (int m = 0; m < rest; ++m) acc += g[static_cast<int64_t>(b) * N + (m / sh) * (sh * K) + i * sh + (m % sh)];
    grad_store(garena + ro[h] + static_cast<int64_t>(b) * K + i, acc, accumulate);
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
    categorical_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ gfold, const int32_t* __restrict__ xt,
                           const int64_t* __restrict__ scope, float* __restrict__ dtable, int B, int K,
                           int C, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float hist[];  // [C+1][K] (row C: marginalised rows)
  const int f = blockIdx.x;
  for (int i = threadIdx.x; i < (C + 1) * K; i += blockDim.x) hist[i] = 0.f;
  __syncthreads();
  const int32_t* xrow = xt + scope[f] * static_cast<int64_t>(B);
  const float* g = gout + static_cast<int64_t>(gfold != nullptr ? gfold[f] : f) * B * K;
  if ((K & 3) == 0 && K <= 1024) {
    // float4 per lane, 4 independent rows per thread in flight (the loop is load-latency bound)
    const int kq = K >> 2;                    // float4 per row
    const int rows_pass = blockDim.x / kq;    // rows covered by the block per pass
    const int r_in = threadIdx.x / kq, q = threadIdx.x - r_in * kq;
    if (r_in < rows_pass) {
      for (int b0 = r_in; b0 < B; b0 += 4 * rows_pass) {
        float4 v[4];
        int c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = b0 + u * rows_pass;
          c[u] = -2;
          if (b < B) {
            const int cc = xrow[b];
            c[u] = cc < 0 ? C : min(cc, C - 1);
            v[u] = reinterpret_cast<const float4*>(g + static_cast<int64_t>(b) * K)[q];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (c[u] == -2) continue;
          float* h = hist + c[u] * K + 4 * q;
          if (v[u].x != 0.f) atomicAdd(h + 0, v[u].x);
          if (v[u].y != 0.f) atomicAdd(h + 1, v[u].y);
          if (v[u].z != 0.f) atomicAdd(h + 2, v[u].z);
          if (v[u].w != 0.f) atomicAdd(h + 3, v[u].w);
        }
      }
    }
  } else {
    for (int64_t i = threadIdx.x; i < static_cast<int64_t>(B) * K; i += blockDim.x) {
      const int b = static_cast<int>(i / K), k = static_cast<int>(i - static_cast<int64_t>(b) * K);
      int c = xrow[b];
      c = c < 0 ? C : min(c, C - 1);
      const float v = g[i];
      if (v != 0.f) atomicAdd(&hist[c * K + k], v);
    }
  }
  __syncthreads();
  float* dst = dtable + static_cast<int64_t>(f) * (C + 1) * K;
  for (int i = threadIdx.x; i < (C + 1) * K; i += blockDim.x) dst[i] = accumulate ? dst[i] + hist[i] : hist[i];
}

// The same reduction without floating-point atomics in the inner loop (LDS float atomics retire ~4 cycles per lane on
// gfx950: 0.68 ms at config 2).  The rows of a fold are counting-sorted by category in LDS (integer atomics on 257
// counters), then every category's rows are summed by ONE wave with plain 128-byte row loads -- lanes = 32 units x
// 2 rows, 4 independent loads in flight per lane -- and added to the category's row of the table.  A category is
// only ever touched by the wave that owns it, so no atomics and no races; 16 waves per workgroup keep enough rows in
// flight to stream gout at HBM rate.  K must be a multiple of 32.
constexpr int kCatChunk = 4096;  // rows sorted at a time
template <int GS, bool EMB>
__global__ void __launch_bounds__(1024)
    categorical_bwd_sorted_kernel(const float* __restrict__ gout, const int32_t* __restrict__ gfold, const int32_t* __restrict__ xt,
                                  const int64_t* __restrict__ scope, float* __restrict__ dtable, int B, int K, int C, int accumulate,
                                  const int32_t* __restrict__ fold_order, const float* __restrict__ wtable) {
  // GS: floats between consecutive gout entries (2: the real parts of complex gradients).  EMB (the Embedding layer under a
  // log, ck_embedding_bwd): the result is d w (F, K, C) = hist / wtable (an entry nobody selected: 0), transposed through LDS
  // rows of K + 1 floats.  (Template parameters: the Categorical instantiation <1, false> is the kernel it was.)
  constexpr int gs = GS;
  const int KS = EMB ? K + 1 : K;
  extern __shared__ __attribute__((aligned(16))) float hist[];  // [C+1][KS], then the int arrays below
  int* start = reinterpret_cast<int*>(hist + (C + 1) * KS);     // [C+2] exclusive prefix of the counts
  int* cur = start + (C + 2);                                   // [C+1] scatter cursors
  int* order = cur + (C + 1);                                   // [kCatChunk] row numbers grouped by category
  const int f = fold_order != nullptr ? fold_order[blockIdx.x] : blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int k_in = lane & 31, slot = lane >> 5;
  for (int i = threadIdx.x; i < (C + 1) * KS; i += blockDim.x) hist[i] = 0.f;
  const int32_t* xrow = xt + scope[f] * static_cast<int64_t>(B);
  const float* g = gout + static_cast<int64_t>(gfold != nullptr ? gfold[f] : f) * B * K * gs;
  const int rs = K * gs;  // floats between rows
  for (int b0 = 0; b0 < B; b0 += kCatChunk) {
    const int nb = min(kCatChunk, B - b0);
    for (int i = threadIdx.x; i < C + 2; i += blockDim.x) start[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
      const int cc = xrow[b0 + i];
      atomicAdd(&start[(cc < 0 ? C : min(cc, C - 1)) + 1], 1);
    }
    __syncthreads();
    if (wave == 0) {  // inclusive scan of start[1 .. C+1]: each lane a run of bins, then a wave scan of the run totals
      const int per = (C + 1 + 63) / 64;
      int tot = 0;
      for (int j = 0; j < per; ++j) {
        const int idx = 1 + lane * per + j;
        if (idx <= C + 1) tot += start[idx];
      }
      int incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      int run = incl - tot;
      for (int j = 0; j < per; ++j) {
        const int idx = 1 + lane * per + j;
        if (idx <= C + 1) {
          run += start[idx];
          start[idx] = run;
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= C; i += blockDim.x) cur[i] = start[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
      const int cc = xrow[b0 + i];
      order[atomicAdd(&cur[cc < 0 ? C : min(cc, C - 1)], 1)] = b0 + i;
    }
    __syncthreads();
    for (int c = wave; c <= C; c += nw) {
      const int s0 = start[c], s1 = start[c + 1];
      if (s0 == s1) continue;
      for (int k = k_in; k < K; k += 32) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = s0 + slot;
        for (; i + 6 < s1; i += 8) {
          const float v0 = g[static_cast<int64_t>(order[i]) * rs + k * gs];
          const float v1 = g[static_cast<int64_t>(order[i + 2]) * rs + k * gs];
          const float v2 = g[static_cast<int64_t>(order[i + 4]) * rs + k * gs];
          const float v3 = g[static_cast<int64_t>(order[i + 6]) * rs + k * gs];
          a0 += v0;
          a1 += v1;
          a2 += v2;
          a3 += v3;
        }
        for (; i < s1; i += 2) a0 += g[static_cast<int64_t>(order[i]) * rs + k * gs];
        float acc = (a0 + a1) + (a2 + a3);
        acc += __shfl_xor(acc, 32, 64);
        if (slot == 0) hist[c * KS + k] += acc;
      }
    }
    __syncthreads();
  }
  if constexpr (EMB) {
    const float* t = wtable + static_cast<int64_t>(f) * (C + 1) * K;
    for (int i = threadIdx.x; i < C * K; i += blockDim.x) {
      const int c = i / K, k = i - c * K;
      const float h = hist[c * KS + k];
      hist[c * KS + k] = h == 0.f ? 0.f : h / t[i];
    }
    __syncthreads();
    float* dst = dtable + static_cast<int64_t>(f) * K * C;
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
      const int k = i / C, c = i - k * C;
      dst[i] = accumulate ? dst[i] + hist[c * KS + k] : hist[c * KS + k];
    }
    return;
  }
  float* dst = dtable + static_cast<int64_t>(f) * (C + 1) * K;
  for (int i = threadIdx.x; i < (C + 1) * K; i += blockDim.x) dst[i] = accumulate ? dst[i] + hist[i] : hist[i];
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

// The same for a list of parameter tensors in ONE launch (a training step has one per sum layer): block b belongs to the job
// with first_block <= b < first_block of the next one, 4 rows per block.
struct SoftmaxBwdJob {
  const float* w;
  const float* dw;
  float* dtheta;
  int64_t rows;
  int32_t len;
  int32_t first_block;
  int64_t part_stride;  // n_part > 1: dW = the sum of n_part slots, part_stride floats apart (ck_tail_bwd.hip)
  int32_t n_part;
  int32_t reserved;
  float* theta;  // the optimizer in the epilogue (len == 32): logits, moments, next step's softmax
  float* m1;
  float* m2;
  float* w_out;
};
// the optimizer's step on one entry of a 32-wide row held by the lanes 0 .. 31 (i = lane) and the row's next softmax
// (the entry's logit and moments are requested by `softmax_row_fetch` before the row's gradient is formed: one round trip, not two)
struct RowOpt {
  float th, m1, m2;
  bool on;
};
__device__ __forceinline__ RowOpt softmax_row_fetch(const SoftmaxBwdJob& j, const ck_opt_state* opt, int64_t at, int lane) {
  RowOpt r{0.f, 0.f, 0.f, opt != nullptr && j.theta != nullptr && opt->skip_now == 0};
  if (r.on && lane < 32) {
    r.th = ck::as_global(j.theta)[at];
    if (opt->kind != 0) {
      r.m1 = ck::as_global(j.m1)[at];
      r.m2 = ck::as_global(j.m2)[at];
    }
  }
  return r;
}
__device__ __forceinline__ void softmax_row_step(const SoftmaxBwdJob& j, const ck_opt_state* opt, const RowOpt& r, int64_t at, float g, int lane) {
  if (!r.on) return;
  const OptK ok = opt_k(*opt);
  float th = 0.f;
  if (lane < 32) {
    float m1 = r.m1, m2 = r.m2;
    th = opt_update(ok, r.th, g, m1, m2);
    ck::as_global(j.theta)[at] = th;
    if (ok.kind != 0) {
      ck::as_global(j.m1)[at] = m1;
      ck::as_global(j.m2)[at] = m2;
    }
  }
  float mx = th;
#pragma unroll
  for (int s = 1; s < 32; s <<= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
  const float e = expf(th - mx);
  float sum = e;
#pragma unroll
  for (int s = 1; s < 32; s <<= 1) sum += __shfl_xor(sum, s, 64);
  if (lane < 32) ck::as_global(j.w_out)[at] = e / sum;
}
__global__ void __launch_bounds__(256) softmax_bwd_batch_kernel(const SoftmaxBwdJob* __restrict__ jobs, int n_jobs,
                                                                const ck_opt_state* __restrict__ opt) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {  // last job whose first block is <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= static_cast<int>(blockIdx.x)) lo = mid;
    else hi = mid - 1;
  }
  const SoftmaxBwdJob j = jobs[lo];
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x - j.first_block) * 4 + (threadIdx.x >> 6);
  if (row >= j.rows) return;
  const auto* wr = ck::as_global(j.w + row * j.len);  // (pointers out of the job table: device memory, not FLAT)
  const auto* dr = ck::as_global(j.dw + row * j.len);
  auto* dth = ck::as_global(j.dtheta);
  if (j.n_part > 1 && j.len == 32) {
    // the row's gradient is spread over n_part slots: the two half-waves take every other slot, eight loads in flight each
    const int i = lane & 31;
    const RowOpt ro = softmax_row_fetch(j, opt, row * 32 + i, lane);
    float acc = 0.f;
    int p = lane >> 5;
    for (; p + 14 < j.n_part; p += 16) {
      float t8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t8[q] = dr[(p + 2 * q) * j.part_stride + i];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += t8[q];
    }
    for (; p < j.n_part; p += 2) acc += dr[p * j.part_stride + i];
    acc += __shfl_xor(acc, 32, 64);
    const float w = wr[i];
    float dot = w * acc;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) dot += __shfl_xor(dot, s, 64);
    const float g = w * (acc - dot);
    if (lane < 32) dth[row * j.len + i] = g;
    softmax_row_step(j, opt, ro, row * 32 + i, g, lane);
    return;
  }
  if (j.len == 32) {  // (one entry per lane of the lower half: the form the optimizer epilogue takes)
    const int i = lane & 31;
    const RowOpt ro = softmax_row_fetch(j, opt, row * 32 + i, lane);
    const float w = wr[i], dv = dr[i];
    float dot = lane < 32 ? w * dv : 0.f;
    dot = ck::wave_sum(dot);
    const float g = w * (dv - dot);
    if (lane < 32) dth[row * 32 + i] = g;
    softmax_row_step(j, opt, ro, row * 32 + i, g, lane);
    return;
  }
  float dot = 0.f;
  for (int i = lane; i < j.len; i += 64) dot = fmaf(wr[i], dr[i], dot);
  dot = ck::wave_sum(dot);
  for (int i = lane; i < j.len; i += 64) dth[row * j.len + i] = wr[i] * (dr[i] - dot);
}

// Categorical parameter backward: table (F, C+1, K) = log softmax_C(theta (F, K, C)) transposed.
// dtheta[f,k,c] = dT[f,c,k] - exp(T[f,c,k]) * sum_c' dT[f,c',k].   One block per fold; both (C, K)
// tiles go through LDS (row stride K+1) so that global reads (k fastest) and writes (c fastest) are
// both coalesced.
__global__ void __launch_bounds__(256)
    log_table_bwd_kernel(const float* __restrict__ table, const float* __restrict__ dtable,
                         float* __restrict__ dtheta, int K, int C, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = K + 1;
  float* d_s = sm;                 // [C][K+1] dT
  float* p_s = d_s + C * ld;       // [C][K+1] exp(T)
  float* colsum = p_s + C * ld;    // [K]
  const int f = blockIdx.x;
  const float* T = table + static_cast<int64_t>(f) * (C + 1) * K;  // rows 0..C-1; row C (integral, = 0) has no gradient
  const float* dT = dtable + static_cast<int64_t>(f) * (C + 1) * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x) colsum[k] = 0.f;
  for (int i = threadIdx.x; i < C * K; i += blockDim.x) {
    const int c = i / K, k = i - c * K;
    d_s[c * ld + k] = dT[i];
    p_s[c * ld + k] = expf(T[i]);
  }
  __syncthreads();
  {
    const int kk = threadIdx.x % K, c0 = threadIdx.x / K, cstep = blockDim.x / K;
    if (cstep > 0) {
      if (c0 < cstep) {
        float sacc = 0.f;
        for (int c = c0; c < C; c += cstep) sacc += d_s[c * ld + kk];
        atomicAdd(&colsum[kk], sacc);
      }
    } else {
      for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float sacc = 0.f;
        for (int c = 0; c < C; ++c) sacc += d_s[c * ld + k];
        colsum[k] = sacc;
      }
    }
  }
  __syncthreads();
  float* dst = dtheta + static_cast<int64_t>(f) * K * C;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    const int k = i / C, c = i - k * C;
    const float g = d_s[c * ld + k] - p_s[c * ld + k] * colsum[k];
    if (accumulate)
      dst[i] += g;
    else
      dst[i] = g;
  }
}

// ---- Gaussian input layer ------------------------------------------------------------------------
// lp = -(x - mu)^2 / (2 sd^2) - log sd - log sqrt(2 pi)   (input.py:661-670)
//   d lp / d mu = (x - mu) / sd^2,   d lp / d sd = (x - mu)^2 / sd^3 - 1 / sd
// One workgroup per fold: thread = unit k x row group; partial sums over the batch in registers,
// then an LDS reduction over the row groups.  A NaN input (marginalised variable) contributes nothing.
__global__ void __launch_bounds__(256)
    gaussian_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ xt, const int64_t* __restrict__ scope,
                        const float* __restrict__ mean, const float* __restrict__ stddev, float* __restrict__ dmean,
                        float* __restrict__ dstd, int B, int K) {
  extern __shared__ float red[];  // [2][groups][kk]
  const int f = blockIdx.x;
  const int kk = K <= 256 ? K : 256;
  const int groups = 256 / kk;
  const int r_in = threadIdx.x / kk, k0 = threadIdx.x - r_in * kk;
  const float* xrow = xt + scope[f] * B;
  const float* g = gout + static_cast<int64_t>(f) * B * K;
  for (int k = k0; k < K; k += kk) {
    float am = 0.f, as = 0.f;
    if (r_in < groups) {
      const float mu = mean[static_cast<int64_t>(f) * K + k];
      const float sd = stddev[static_cast<int64_t>(f) * K + k];
      const float inv_var = 1.f / (sd * sd), inv_sd = 1.f / sd;
      for (int b = r_in; b < B; b += groups) {
        const float xv = xrow[b];
        if (xv != xv) continue;
        const float d = xv - mu, gg = g[static_cast<int64_t>(b) * K + k];
        am = fmaf(gg, d * inv_var, am);
        as = fmaf(gg, d * d * inv_var * inv_sd - inv_sd, as);
      }
      red[r_in * kk + k0] = am;
      red[(groups + r_in) * kk + k0] = as;
    }
    __syncthreads();
    if (r_in == 0) {
      float tm = 0.f, ts = 0.f;
      for (int r = 0; r < groups; ++r) {
        tm += red[r * kk + k0];
        ts += red[(groups + r) * kk + k0];
      }
      dmean[static_cast<int64_t>(f) * K + k] = tm;
      dstd[static_cast<int64_t>(f) * K + k] = ts;
    }
    __syncthreads();
  }
}

// ---- Mixing layer ----------------------------------------------------------------------------------
// out[k] = log(sum_h w[k,h] e_h[k]) + m, e_h[k] = exp(x_h[k] - m), m = max over (h, k) of the row
// (ck_mixing_lse_fwd).  With y[k] = exp(out[k] - m) recomputed as sum_h w e:
//   g x_h[k] = gout[k] w[k,h] e_h[k] / y[k],     d w[k,h] += sum_b gout[k] e_h[k] / y[k].
// One wave per batch row (lanes over k), dW partials in LDS per workgroup, one atomic per (k, h).
__global__ void __launch_bounds__(256)
    mixing_bwd_kernel(const float* __restrict__ arena, float* __restrict__ garena, const int64_t* __restrict__ row_off,
                      const int64_t* __restrict__ grow_off,
                      const float* __restrict__ mw, const float* __restrict__ gout, float* __restrict__ dmw, int H,
                      int B, int K, int rows_per_block, int accumulate) {
  extern __shared__ float dw_s[];  // [K][H]
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t* gro = grow_off + static_cast<int64_t>(f) * H;  // where the children's gradients go
  const float* mwf = mw + static_cast<int64_t>(f) * K * H;
  for (int i = threadIdx.x; i < K * H; i += blockDim.x) dw_s[i] = 0.f;
  __syncthreads();
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
      float y = 0.f;
      for (int h = 0; h < H; ++h)
        y = fmaf(mwf[static_cast<int64_t>(k) * H + h], expf(arena[ro[h] + static_cast<int64_t>(b) * K + k] - mx), y);
      const float gy = gout[(static_cast<int64_t>(f) * B + b) * K + k] / y;
      for (int h = 0; h < H; ++h) {
        const float e = expf(arena[ro[h] + static_cast<int64_t>(b) * K + k] - mx);
        const float ge = gy * e;
        float* dst = garena + gro[h] + static_cast<int64_t>(b) * K + k;
        const float gv = ge * mwf[static_cast<int64_t>(k) * H + h];
        if (accumulate == 0)
          *dst = gv;
        else if (accumulate == 1)
          *dst += gv;
        else
          atomicAdd(dst, gv);
        atomicAdd(&dw_s[k * H + h], ge);
      }
    }
  }
  __syncthreads();
  float* dwf = dmw + static_cast<int64_t>(f) * K * H;
  for (int i = threadIdx.x; i < K * H; i += blockDim.x) atomicAdd(&dwf[i], dw_s[i]);
}

// The same for K = 32 / 64 and H <= HMAX with the coefficient gradients in REGISTERS: a lane owns one unit k (64 / K
// rows per wave pass), keeps e_h[k] of its row and accumulates d w[k, h] over all rows the wave visits; the four waves
// are added through LDS with plain stores and leave with one atomic per (k, h) and workgroup.  (The LDS float
// atomics of the general kernel retire ~4 cycles per lane and were 3/4 of its time.)
template <int HMAX>
__global__ void __launch_bounds__(256)
    mixing_bwd_reg_kernel(const float* __restrict__ arena, float* __restrict__ garena, const int64_t* __restrict__ row_off,
                          const int64_t* __restrict__ grow_off, const float* __restrict__ mw,
                          const float* __restrict__ gout, float* __restrict__ dmw, int H, int B, int K, int rows_per_block,
                          int accumulate) {
  __shared__ float red[4][64][HMAX + 1];
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpw = 64 / K;               // rows per wave pass
  const int k = lane % K, sub = lane / K;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t* gro = grow_off + static_cast<int64_t>(f) * H;
  float wk[HMAX], dacc[HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; ++h) {
    wk[h] = h < H ? mw[(static_cast<int64_t>(f) * K + k) * H + h] : 0.f;
    dacc[h] = 0.f;
  }
  const int b_begin = blockIdx.x * rows_per_block;
  const int b_end = min(B, b_begin + rows_per_block);
  for (int b0 = b_begin + wave * rpw; b0 < b_end; b0 += 4 * rpw) {
    const int b = b0 + sub;
    const bool live = b < b_end;
    const int bl = live ? b : b_end - 1;
    float x[HMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
      if (h < H) {
        x[h] = arena[ro[h] + static_cast<int64_t>(bl) * K + k];
        mx = fmaxf(mx, x[h]);
      }
    for (int o = K >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));  // over the K lanes of the row
    mx = ck::clamp_finite(mx);
    float y = 0.f;
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
      if (h < H) {
        x[h] = expf(x[h] - mx);
        y = fmaf(wk[h], x[h], y);
      }
    const float gy = live ? gout[(static_cast<int64_t>(f) * B + bl) * K + k] / y : 0.f;
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
      if (h < H) {
        const float ge = gy * x[h];
        dacc[h] += ge;
        if (live) grad_store(garena + gro[h] + static_cast<int64_t>(b) * K + k, ge * wk[h], accumulate);
      }
  }
#pragma unroll
  for (int h = 0; h < HMAX; ++h) red[wave][lane][h] = dacc[h];
  __syncthreads();
  for (int i = threadIdx.x; i < K * H; i += 256) {
    const int kk = i / H, h = i - kk * H;
    float t = 0.f;
    for (int wv = 0; wv < 4; ++wv)
      for (int sb = 0; sb < rpw; ++sb) t += red[wv][sb * K + kk][h];
    if (t != 0.f) atomicAdd(&dmw[(static_cast<int64_t>(f) * K + kk) * H + h], t);
  }
}

// ---- parameter-graph backward pieces -------------------------------------------------------------
// scaled sigmoid y = s (vmax - vmin) + vmin:  dx = dy (y - vmin)(vmax - y) / (vmax - vmin)
__global__ void __launch_bounds__(256)
    scaled_sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                              int64_t n, float vmin, float vmax, int accumulate) {
  const float inv = 1.f / (vmax - vmin);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float g = dy[i] * (y[i] - vmin) * (vmax - y[i]) * inv;
    dx[i] = accumulate ? dx[i] + g : g;
  }
}
// softmax / log-softmax backward along ANY axis of a tensor viewed as (outer, len, inner) (TorchSoftmaxParameter /
// TorchLogSoftmaxParameter, nodes.py:764-783: dim is any axis of the unfolded shape): a thread per (outer, inner) pair walks
// the axis twice; neighbouring threads read neighbouring `inner` positions.  y is the node's OUTPUT.
//   softmax:      dx = y (dy - sum_l y dy)          log-softmax:  dx = dy - exp(y) sum_l dy
__global__ void __launch_bounds__(256)
    softmax_bwd_strided_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int64_t outer,
                               int len, int64_t inner, int log_space, int accumulate) {
  for (int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; t < outer * inner;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t base = (t / inner) * len * inner + t % inner;
    float acc = 0.f;
    for (int l = 0; l < len; ++l) acc += log_space ? dy[base + l * inner] : y[base + l * inner] * dy[base + l * inner];
    for (int l = 0; l < len; ++l) {
      const int64_t i = base + l * inner;
      const float g = log_space ? dy[i] - expf(y[i]) * acc : y[i] * (dy[i] - acc);
      dx[i] = accumulate ? dx[i] + g : g;
    }
  }
}
// entrywise parameter nodes (nodes.py:656-699) from input x and output y: sigmoid y (1 - y), exp y, log 1 / x, square 2 x
__global__ void __launch_bounds__(256)
    unary_bwd_kernel(int op, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                     float* __restrict__ dx, int64_t n, int accumulate) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    float d;
    if (op == CK_UNARY_SIGMOID) d = y[i] * (1.f - y[i]);
    else if (op == CK_UNARY_EXP) d = y[i];
    else if (op == CK_UNARY_LOG) d = 1.f / x[i];
    else if (op == CK_UNARY_CLAMP) d = y[i] == x[i] ? 1.f : 0.f;  // (inside [vmin, vmax], bounds included: torch.clamp's backward)
    else if (op == CK_UNARY_SOFTPLUS) d = x[i] > 20.f ? 1.f : 1.f / (1.f + expf(-x[i]));
    else d = 2.f * x[i];
    // (an entry nobody selected has dy == 0: its gradient is 0 whatever the derivative -- 0 * inf at log(0) would be NaN)
    const float g = dy[i] == 0.f ? 0.f : dy[i] * d;
    dx[i] = accumulate ? dx[i] + g : g;
  }
}
// mixing weight (F, K, H) -> (F, K, H*K) block diagonal: dx[f,k,h] = dy[f,k,h*K + k]
__global__ void __launch_bounds__(256)
    mixing_weight_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t n, int K, int H,
                             int accumulate) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int h = static_cast<int>(i % H);
    const int64_t fk = i / H;
    const int k = static_cast<int>(fk % K);
    const float g = dy[fk * (static_cast<int64_t>(H) * K) + static_cast<int64_t>(h) * K + k];
    dx[i] = accumulate ? dx[i] + g : g;
  }
}
// backward of a fold gather (parameter address book, parameter.py:41-47): ddst[idx[i]] += dsrc[i]
__global__ void __launch_bounds__(256)
    scatter_add_folds_kernel(const float* __restrict__ dsrc, const int64_t* __restrict__ idx, float* __restrict__ ddst,
                             int64_t per_fold) {
  const int64_t i = blockIdx.y;
  const float* src = dsrc + i * per_fold;
  float* dst = ddst + idx[i] * per_fold;
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < per_fold;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x)
    atomicAdd(dst + e, src[e]);
}
__global__ void __launch_bounds__(256)
    axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, int64_t n) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    y[i] = fmaf(a, x[i], y[i]);
}

// latch (nullable triple): *step = *src, *sticky |= *src, *src = 0 -- the validation flag a forward raised becomes this
// step's flag (what the optimizer launch reads) and the sticky one (what check_inputs() reports) in the launch that zeroes
// the gradient buffers anyway
// opt (nullable, with the latch): the optimizer's clock of this step as well (ck_opt_tick: a flagged step is dropped -- skip_now,
// counted in `skipped` --, otherwise the step count and Adam's bias corrections advance), for the launches of the step whose
// epilogues update parameters
__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ p, int64_t n, float v, int32_t* __restrict__ src,
                                                    int32_t* __restrict__ step, int32_t* __restrict__ sticky, ck_opt_state* __restrict__ opt) {
  if (src != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    const int32_t f = *src;
    *step = f;
    if (f != 0) {
      *sticky |= f;
      *src = 0;
    }
    if (opt != nullptr) {
      if (f != 0) {
        opt->skip_now = 1;
        opt->skipped += 1;
      } else {
        opt->skip_now = 0;
        opt->step += 1;
        const double t = static_cast<double>(opt->step);  // (in double from the double betas, as ck_opt_tick)
        opt->bc1 = static_cast<float>(-expm1(t * log(opt->b1d)));
        opt->bc2 = static_cast<float>(-expm1(t * log(opt->b2d)));
      }
    }
  }
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

// Adam step (torch.optim.Adam defaults semantics, no weight decay / amsgrad), fused over one tensor.
// skip_flag (nullable): nonzero = this step saw an invalid batch: parameters, moments and the step count stay as they are
// (`skipped` counts such launches; the bias corrections use step - skipped).
__global__ void __launch_bounds__(256)
    adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1,
                float* __restrict__ m2, int64_t n, float lr, float b1, float b2, float eps, int step,
                float gscale, const int32_t* __restrict__ skip_flag, int32_t* __restrict__ skipped) {
  if (skip_flag != nullptr && *skip_flag != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && skipped != nullptr) *skipped += 1;  // (no launch reads it while this one runs)
    return;
  }
  const float t = static_cast<float>(step - (skipped != nullptr ? *skipped : 0));
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
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
    sgd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr, float gscale,
               const int32_t* __restrict__ skip_flag) {
  if (skip_flag != nullptr && *skip_flag != 0) return;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] -= lr * gscale * g[i];
}

__global__ void latch_flag_kernel(int32_t* __restrict__ src, int32_t* __restrict__ dst) {
  if (*src != 0) {
    *dst |= *src;
    *src = 0;
  }
}

// ---- squared circuits: the pieces of `loss = -mean(2 Re c(x) - Re Z)` that are neither a layer nor a parameter node -------
// dst[i * dst_stride] = src[i * src_stride] (the real parts of complex values; one column of a table)
__global__ void __launch_bounds__(256)
    copy_strided_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int64_t src_stride, int64_t dst_stride) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) dst[i * dst_stride] = src[i * src_stride];
}
__global__ void __launch_bounds__(256) fill_strided_kernel(float* __restrict__ p, int64_t n, int64_t stride, float v) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) p[i * stride] = v;
}
// d w[f, k, c] = d log|w| / w from the gather table (F, C + 1, K) and the scattered gradient in the same layout (an entry nobody
// selected has gradient 0 whatever w is): TorchEmbeddingLayer under the log of csafelog (input.py:258-266, utils.py:32-50)
__global__ void __launch_bounds__(256)
    embedding_weight_bwd_kernel(const float* __restrict__ table, const float* __restrict__ dtable, float* __restrict__ dw, int C, int K) {
  const int f = blockIdx.y;
  const float* t = table + static_cast<int64_t>(f) * (C + 1) * K;
  const float* d = dtable + static_cast<int64_t>(f) * (C + 1) * K;
  float* o = dw + static_cast<int64_t>(f) * K * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < K * C; i += gridDim.x * 256) {
    const int k = i / C, c = i - k * C;
    const float g = d[c * K + k];
    o[i] = g == 0.f ? 0.f : g / t[c * K + k];
  }
}
// out = [2 sum_b yc[b * stride] - B z[0], B] in fp64 (one workgroup: a deterministic tree, as ck_ll_sum)
__global__ void __launch_bounds__(1024)
    squared_ll_kernel(const float* __restrict__ yc, int64_t B, int64_t stride, const float* __restrict__ z, double* __restrict__ out) {
  __shared__ double part[16];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) acc += static_cast<double>(yc[i * stride]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < static_cast<int>(blockDim.x >> 6); ++i) t += part[i];
    out[0] = 2.0 * t - static_cast<double>(B) * static_cast<double>(z[0]);
    out[1] = static_cast<double>(B);
  }
}

bool g_bwd_force_generic = false;

unsigned grid1(int64_t n, int cap = 2048) { return static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, cap)); }

}  // namespace

extern "C" {

// Test hook: route ck_sum_lse_bwd through the shape-generic kernel even for K = 32.
int ck_debug_force_generic_bwd(int on) {
  g_bwd_force_generic = on != 0;
  return CK_OK;
}

int ck_fill_f32(float* p, int64_t n, float value, void* stream) {
  CK_REQUIRE(p != nullptr && n > 0, "ck_fill_f32: bad arguments");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(fill_kernel, grid, block, 0, s, p, n, value, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                           static_cast<int32_t*>(nullptr), static_cast<ck_opt_state*>(nullptr));
        return hipGetLastError();
      },
      stream);
}

int ck_fill_latch(float* p, int64_t n, float value, int32_t* src, int32_t* step_flag, int32_t* sticky, ck_opt_state* opt, void* stream) {
  CK_REQUIRE(p != nullptr && n > 0 && src && step_flag && sticky, "ck_fill_latch: bad arguments");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(fill_kernel, grid, block, 0, s, p, n, value, src, step_flag, sticky, opt);
        return hipGetLastError();
      },
      stream);
}

int ck_gaussian_bwd(const float* gout, const float* xt, const int64_t* scope, const float* mean, const float* stddev,
                    float* dmean, float* dstddev, int F, int B, int K, void* stream) {
  CK_REQUIRE(gout && xt && scope && mean && stddev && dmean && dstddev, "ck_gaussian_bwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && K > 0, "ck_gaussian_bwd: non-positive size");
  const int kk = K <= 256 ? K : 256;
  const size_t lds = static_cast<size_t>(2) * (256 / kk) * kk * sizeof(float);
  dim3 grid(F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(gaussian_bwd_kernel, grid, block, lds, s, gout, xt, scope, mean, stddev, dmean, dstddev, B, K);
        return hipGetLastError();
      },
      stream);
}

int ck_mixing_lse_bwd(const float* arena, float* garena, const int64_t* row_off, const int64_t* grad_row_off,
                      const float* mw, const float* gout,
                      float* dmw, int F, int H, int B, int K, int accumulate, void* stream) {
  CK_REQUIRE(arena && garena && row_off && mw && gout && dmw, "ck_mixing_lse_bwd: null pointer");
  const int64_t* grow = grad_row_off ? grad_row_off : row_off;
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && K > 0, "ck_mixing_lse_bwd: non-positive size");
  CK_REQUIRE(accumulate >= 0 && accumulate <= 2, "ck_mixing_lse_bwd: accumulate must be 0, 1 or 2");
  CK_REQUIRE(F <= 65535, "ck_mixing_lse_bwd: F=%d exceeds grid.y", F);
  const size_t lds = static_cast<size_t>(K) * H * sizeof(float);
  CK_REQUIRE(lds <= 64 * 1024, "ck_mixing_lse_bwd: K*H=%d coefficients do not fit in LDS", K * H);
  if ((K == 32 || K == 64) && H <= 16) {
    int rpb = 16;  // rows per workgroup: more amortise the final atomics, fewer fill the chip when there are few folds
    while (rpb < 512 && static_cast<int64_t>(F) * ((B + 2 * rpb - 1) / (2 * rpb)) >= 2048) rpb *= 2;
    dim3 grid((B + rpb - 1) / rpb, F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          if (H <= 4)
            hipLaunchKernelGGL(mixing_bwd_reg_kernel<4>, grid, block, 0, s, arena, garena, row_off, grow, mw, gout, dmw, H, B, K, rpb, accumulate);
          else if (H <= 8)
            hipLaunchKernelGGL(mixing_bwd_reg_kernel<8>, grid, block, 0, s, arena, garena, row_off, grow, mw, gout, dmw, H, B, K, rpb, accumulate);
          else
            hipLaunchKernelGGL(mixing_bwd_reg_kernel<16>, grid, block, 0, s, arena, garena, row_off, grow, mw, gout, dmw, H, B, K, rpb, accumulate);
          return hipGetLastError();
        },
        stream);
  }
  const int rows_per_block = 64;
  dim3 grid((B + rows_per_block - 1) / rows_per_block, F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(mixing_bwd_kernel, grid, block, lds, s, arena, garena, row_off, grow, mw, gout, dmw, H, B, K,
                           rows_per_block, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_param_scaled_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, float vmin, float vmax,
                                int accumulate, void* stream) {
  CK_REQUIRE(y && dy && dx && n > 0, "ck_param_scaled_sigmoid_bwd: bad arguments");
  CK_REQUIRE(vmax > vmin, "ck_param_scaled_sigmoid_bwd: vmax must exceed vmin");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(scaled_sigmoid_bwd_kernel, grid, block, 0, s, y, dy, dx, n, vmin, vmax, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_param_softmax_bwd_strided(const float* y, const float* dy, float* dx, int64_t outer, int len, int64_t inner, int log_space,
                                 int accumulate, void* stream) {
  CK_REQUIRE(y && dy && dx && outer > 0 && len > 0 && inner > 0, "ck_param_softmax_bwd_strided: bad arguments");
  dim3 grid(grid1(outer * inner)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(softmax_bwd_strided_kernel, grid, block, 0, s, y, dy, dx, outer, len, inner, log_space, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_param_unary_bwd(int op, const float* x, const float* y, const float* dy, float* dx, int64_t n, int accumulate, void* stream) {
  CK_REQUIRE(x && y && dy && dx && n > 0, "ck_param_unary_bwd: bad arguments");
  CK_REQUIRE(op == CK_UNARY_SIGMOID || op == CK_UNARY_EXP || op == CK_UNARY_LOG || op == CK_UNARY_SQUARE || op == CK_UNARY_CLAMP ||
                 op == CK_UNARY_SOFTPLUS,
             "ck_param_unary_bwd: op %d (scaled sigmoid: ck_param_scaled_sigmoid_bwd)", op);
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(unary_bwd_kernel, grid, block, 0, s, op, x, y, dy, dx, n, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_param_mixing_weight_bwd(const float* dy, float* dx, int F, int K, int H, int accumulate, void* stream) {
  CK_REQUIRE(dy && dx && F > 0 && K > 0 && H > 0, "ck_param_mixing_weight_bwd: bad arguments");
  const int64_t n = static_cast<int64_t>(F) * K * H;
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(mixing_weight_bwd_kernel, grid, block, 0, s, dy, dx, n, K, H, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_param_scatter_add_folds(const float* dsrc, const int64_t* idx, float* ddst, int64_t n, int64_t per_fold,
                               void* stream) {
  CK_REQUIRE(dsrc && idx && ddst && n > 0 && per_fold > 0, "ck_param_scatter_add_folds: bad arguments");
  CK_REQUIRE(n <= 65535, "ck_param_scatter_add_folds: n exceeds grid.y");
  dim3 grid(grid1(per_fold, 64), static_cast<unsigned>(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(scatter_add_folds_kernel, grid, block, 0, s, dsrc, idx, ddst, per_fold);
        return hipGetLastError();
      },
      stream);
}

int ck_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream) {
  CK_REQUIRE(y && x && n > 0, "ck_axpy_f32: bad arguments");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(axpy_kernel, grid, block, 0, s, y, x, a, n);
        return hipGetLastError();
      },
      stream);
}

int ck_segment_add_rows(const float* tmp, const int32_t* cptr, const int32_t* clist, const int64_t* coff, float* garena,
                        int n_child, int64_t block_elems, void* stream) {
  CK_REQUIRE(tmp && cptr && clist && coff && garena, "ck_segment_add_rows: null pointer");
  CK_REQUIRE(n_child > 0 && block_elems > 0, "ck_segment_add_rows: non-positive size");
  CK_REQUIRE(n_child <= 65535, "ck_segment_add_rows: %d children exceed grid.y", n_child);
  CK_REQUIRE((block_elems & 3) != 0 || (ck::aligned16(tmp) && ck::aligned16(garena)), "ck_segment_add_rows: unaligned buffers");
  const int64_t work = (block_elems & 3) == 0 ? block_elems >> 2 : block_elems;
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((work + 255) / 256, 4096)), n_child), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(segment_add_kernel, grid, block, 0, s, tmp, cptr, clist, coff, garena, block_elems);
        return hipGetLastError();
      },
      stream);
}

int ck_sum_lse_bwd(const float* arena, float* garena, const int64_t* row_off, const int64_t* grad_row_off, const float* w,
                   const float* out, const float* gout, float* dw, int F, int H, int B, int Ki, int Ko,
                   int mode, int accumulate, void* stream) {
  CK_REQUIRE(arena && garena && row_off && w && out && gout && dw, "ck_sum_lse_bwd: null pointer");
  const int64_t* grow = grad_row_off ? grad_row_off : row_off;
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && Ki > 0 && Ko > 0, "ck_sum_lse_bwd: non-positive size");
  CK_REQUIRE(mode == CK_SUM_CAT || mode == CK_SUM_PROD || mode == CK_SUM_KRON, "ck_sum_lse_bwd: unsupported mode %d", mode);
  CK_REQUIRE(accumulate >= 0 && accumulate <= 2, "ck_sum_lse_bwd: accumulate must be 0, 1 or 2");
  CK_REQUIRE(F <= 65535, "ck_sum_lse_bwd: F=%d exceeds grid.y", F);
  if ((mode == CK_SUM_PROD || H == 1) && Ki == kK && Ko == 1 && !g_bwd_force_generic) {
    const int rpb = 256;  // rows per block of 32 half-waves (8 rows each)
    dim3 grid((B + rpb - 1) / rpb, F), block(1024);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(sum_lse_bwd_scalar32, grid, block, 0, s, arena, garena, row_off, grow, w, gout, dw, H, B, rpb, accumulate);
          return hipGetLastError();
        },
        stream);
  }
  if ((mode == CK_SUM_PROD || H == 1) && Ki == kK && Ko == kK && !g_bwd_force_generic && ck::aligned16(arena) &&
      ck::aligned16(garena) && ck::aligned16(w) && ck::aligned16(gout)) {
    const int tiles = (B + 31) / 32;
    // tiles per wave: more amortise the weight loads and the dW atomics (one per element and workgroup), fewer give
    // a launch with few folds enough workgroups to fill the chip -- keep >= ~2048 workgroups where the batch allows
    int tpw = 1;
    while (tpw < 8 && 4 * tpw * 2 <= tiles && static_cast<int64_t>(F) * ((tiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) >= 2048)
      tpw *= 2;
    dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(sum_lse_bwd_tile32, grid, block, 0, s, arena, garena, row_off, grow, w, gout, dw, H, B, tpw, accumulate);
          return hipGetLastError();
        },
        stream);
  }
  if ((mode == CK_SUM_PROD || H == 1) && Ki == 64 && Ko == 64 && !g_bwd_force_generic && ck::aligned16(arena) &&
      ck::aligned16(garena) && ck::aligned16(w) && ck::aligned16(gout)) {
    const int tiles = (B + 31) / 32;
    int tpw = 1;  // (never more than the four waves of a workgroup can share: a small batch keeps all of them busy)
    while (tpw < 8 && 4 * tpw * 2 <= tiles && static_cast<int64_t>(F) * ((tiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) >= 1024)
      tpw *= 2;
    dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), F), block(256);
    const size_t lds = (8192 + 4 * 2 * 32 * kT64) * sizeof(float);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sum_lse_bwd_tile64),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
          if (e != hipSuccess) return e;
          hipLaunchKernelGGL(sum_lse_bwd_tile64, grid, block, lds, s, arena, garena, row_off, grow, w, gout, dw, H, B, tpw, accumulate);
          return hipGetLastError();
        },
        stream);
  }
  int64_t N64 = mode == CK_SUM_PROD ? Ki : static_cast<int64_t>(H) * Ki;
  if (mode == CK_SUM_KRON) {
    N64 = 1;
    for (int h = 0; h < H && N64 <= (1 << 20); ++h) N64 *= Ki;
  }
  if (N64 > (1 << 16)) return ck::fail(CK_ERR_UNSUPPORTED, "ck_sum_lse_bwd: %lld contracted inputs do not fit in LDS", static_cast<long long>(N64));
  const int N = static_cast<int>(N64);
  auto lds_bytes = [&](int tb) {
    return (static_cast<size_t>(2) * tb * N + static_cast<size_t>(tb) * Ko + 64 * (kBwdNC + 1) + tb +
            (mode == CK_SUM_KRON ? static_cast<size_t>(tb) * H * Ki : 0)) * sizeof(float);
  };
  int tb = 16;
  while (tb > 4 && lds_bytes(tb) > 64 * 1024) tb >>= 1;
  if (mode == CK_SUM_KRON && lds_bytes(tb) > 160 * 1024) tb = 1;  // (one row per workgroup: K = 64 Tucker layers, 4096 products a row)
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
          hipLaunchKernelGGL(kern, grid, block, lds, s, arena, garena, row_off, grow, w, out, gout, dw, H, B, Ki, Ko, mode,
                             accumulate);
          return hipGetLastError();
        };
        if (tb == 16) return go(sum_lse_bwd_generic<16>);
        if (tb == 8) return go(sum_lse_bwd_generic<8>);
        if (tb == 1) return go(sum_lse_bwd_generic<1>);
        return go(sum_lse_bwd_generic<4>);
      },
      stream);
}

int ck_kronecker_bwd(float* garena, const int64_t* row_off, const float* gout, int F, int H, int B, int K,
                     int accumulate, void* stream) {
  CK_REQUIRE(garena && row_off && gout, "ck_kronecker_bwd: null pointer");
  CK_REQUIRE(F > 0 && H >= 2 && B > 0 && K > 0, "ck_kronecker_bwd: non-positive size or arity below 2");
  CK_REQUIRE(accumulate >= 0 && accumulate <= 2, "ck_kronecker_bwd: accumulate must be 0, 1 or 2");
  CK_REQUIRE(F <= 65535, "ck_kronecker_bwd: F=%d exceeds grid.y", F);
  int64_t N = 1;
  for (int h = 0; h < H; ++h) {
    N *= K;
    CK_REQUIRE(N <= (int64_t{1} << 24), "ck_kronecker_bwd: K^H too large");
  }
  const int64_t words = static_cast<int64_t>(B) * H * K;
  dim3 grid(grid1(words), F), block(256);
  const int n = static_cast<int>(N);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(kronecker_bwd_kernel, grid, block, 0, s, garena, row_off, gout, H, B, K, n, accumulate);
        return hipGetLastError();
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

int ck_categorical_bwd(const float* gout, const int32_t* gfold, const int32_t* xt, const int64_t* scope, float* dtable, int F,
                       int B, int K, int C, int accumulate, const int32_t* fold_order, void* stream) {
  CK_REQUIRE(gout && xt && scope && dtable, "ck_categorical_bwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && K > 0 && C > 0, "ck_categorical_bwd: non-positive size");
  const size_t lds = static_cast<size_t>(C + 1) * K * sizeof(float);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_categorical_bwd: C*K=%d does not fit in LDS", C * K);
  const size_t lds_sorted = lds + (static_cast<size_t>(2) * C + 3 + kCatChunk) * sizeof(int);
  if (K % 32 == 0 && lds_sorted <= 160 * 1024 && B >= 256) {
    dim3 grid(F), block(1024);
    return ck::dispatch(
        [=](hipStream_t s) {
          if (lds_sorted > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(categorical_bwd_sorted_kernel<1, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_sorted));
            if (e != hipSuccess) return e;
          }
          hipLaunchKernelGGL((categorical_bwd_sorted_kernel<1, false>), grid, block, lds_sorted, s, gout, gfold, xt, scope, dtable, B, K, C, accumulate,
                             fold_order, static_cast<const float*>(nullptr));
          return hipGetLastError();
        },
        stream);
  }
  CK_REQUIRE(fold_order == nullptr, "ck_categorical_bwd: fold_order is only read by the sorted launch (K %% 32 == 0, B >= 256)");
  dim3 grid(F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (lds > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(categorical_bwd_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
          if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(categorical_bwd_kernel, grid, block, lds, s, gout, gfold, xt, scope, dtable, B, K, C, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_embedding_bwd(const float* gout, int gout_stride, const int32_t* gfold, const int32_t* fold_order, const int32_t* xt,
                     const int64_t* scope, const float* table, float* dw, int F, int B, int K, int C, void* stream) {
  CK_REQUIRE(gout && xt && scope && table && dw, "ck_embedding_bwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && K > 0 && C > 0 && (gout_stride == 1 || gout_stride == 2), "ck_embedding_bwd: bad arguments");
  const size_t lds = (static_cast<size_t>(C + 1) * (K + 1) + static_cast<size_t>(2) * C + 3 + kCatChunk) * sizeof(float);
  if (K % 32 != 0 || lds > 160 * 1024)
    return ck::fail(CK_ERR_UNSUPPORTED, "ck_embedding_bwd: needs K %% 32 == 0 and (C + 1) (K + 1) + 2 C + %d words of LDS (K=%d, C=%d)",
                    kCatChunk + 3, K, C);
  dim3 grid(F), block(1024);
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern) {
          if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(160 * 1024));
            if (e != hipSuccess) return e;
          }
          hipLaunchKernelGGL(kern, grid, block, lds, s, gout, gfold, xt, scope, dw, B, K, C, 0, fold_order, table);
          return hipGetLastError();
        };
        return gout_stride == 2 ? go(categorical_bwd_sorted_kernel<2, true>) : go(categorical_bwd_sorted_kernel<1, true>);
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

int ck_param_softmax_bwd_batch(const ck_softmax_bwd_job* jobs, int n_jobs, int n_blocks, const ck_opt_state* opt, void* stream) {
  CK_REQUIRE(jobs != nullptr && n_jobs > 0 && n_blocks > 0, "ck_param_softmax_bwd_batch: bad arguments");
  static_assert(sizeof(SoftmaxBwdJob) == sizeof(ck_softmax_bwd_job), "job layout");
  dim3 grid(static_cast<unsigned>(n_blocks)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(softmax_bwd_batch_kernel, grid, block, 0, s, reinterpret_cast<const SoftmaxBwdJob*>(jobs), n_jobs, opt);
        return hipGetLastError();
      },
      stream);
}

int ck_param_log_table_bwd(const float* table, const float* dtable, float* dtheta, int F, int K, int C,
                           int accumulate, void* stream) {
  CK_REQUIRE(table && dtable && dtheta, "ck_param_log_table_bwd: null pointer");
  CK_REQUIRE(F > 0 && K > 0 && C > 0, "ck_param_log_table_bwd: non-positive size");
  dim3 grid(F), block(256);
  const size_t lds = (static_cast<size_t>(2) * C * (K + 1) + K) * sizeof(float);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_param_log_table_bwd: C*K=%d does not fit in LDS", C * K);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (lds > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(log_table_bwd_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
          if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(log_table_bwd_kernel, grid, block, lds, s, table, dtable, dtheta, K, C, accumulate);
        return hipGetLastError();
      },
      stream);
}

int ck_adam_step(float* p, const float* g, float* m1, float* m2, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, const int32_t* skip_flag, int32_t* skipped,
                 void* stream) {
  CK_REQUIRE(p && g && m1 && m2, "ck_adam_step: null pointer");
  CK_REQUIRE(n > 0 && step > 0, "ck_adam_step: n and step must be positive");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(adam_kernel, grid, block, 0, s, p, g, m1, m2, n, lr, beta1, beta2, eps, step, grad_scale, skip_flag, skipped);
        return hipGetLastError();
      },
      stream);
}

int ck_sgd_step(float* p, const float* g, int64_t n, float lr, float grad_scale, const int32_t* skip_flag, void* stream) {
  CK_REQUIRE(p && g, "ck_sgd_step: null pointer");
  CK_REQUIRE(n > 0, "ck_sgd_step: n must be positive");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(sgd_kernel, grid, block, 0, s, p, g, n, lr, grad_scale, skip_flag);
        return hipGetLastError();
      },
      stream);
}

int ck_copy_strided_f32(const float* src, float* dst, int64_t n, int64_t src_stride, int64_t dst_stride, void* stream) {
  CK_REQUIRE(src && dst && n > 0 && src_stride > 0 && dst_stride > 0, "ck_copy_strided_f32: bad arguments");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(copy_strided_kernel, grid, block, 0, s, src, dst, n, src_stride, dst_stride);
        return hipGetLastError();
      },
      stream);
}

int ck_fill_strided_f32(float* p, int64_t n, int64_t stride, float value, void* stream) {
  CK_REQUIRE(p && n > 0 && stride > 0, "ck_fill_strided_f32: bad arguments");
  dim3 grid(grid1(n)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(fill_strided_kernel, grid, block, 0, s, p, n, stride, value);
        return hipGetLastError();
      },
      stream);
}

int ck_embedding_weight_bwd(const float* table, const float* dtable, float* dw, int F, int C, int K, void* stream) {
  CK_REQUIRE(table && dtable && dw && F > 0 && F <= 65535 && C > 0 && K > 0, "ck_embedding_weight_bwd: bad arguments");
  dim3 grid(static_cast<unsigned>(std::min((K * C + 255) / 256, 64)), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(embedding_weight_bwd_kernel, grid, block, 0, s, table, dtable, dw, C, K);
        return hipGetLastError();
      },
      stream);
}

int ck_squared_ll(const float* yc, int64_t B, int64_t stride, const float* z, double* out, void* stream) {
  CK_REQUIRE(yc && z && out && B > 0 && stride > 0, "ck_squared_ll: bad arguments");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(squared_ll_kernel, dim3(1), dim3(1024), 0, s, yc, B, stride, z, out);
        return hipGetLastError();
      },
      stream);
}

int ck_latch_flag(int32_t* src, int32_t* dst, void* stream) {
  CK_REQUIRE(src && dst, "ck_latch_flag: null pointer");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(latch_flag_kernel, dim3(1), dim3(1), 0, s, src, dst);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
