// Job-list launches of the training step for circuits of 64-unit CP layers (the circuit of the
// reference's learning notebook -- QuadGraph, CP, K = 64, notebooks/learning-a-circuit.ipynb cells 4 / 16 / 18 -- and BASELINE
// config 4, Poon-Domingos with Gaussian leaves).
//
// The reference differentiates its layer-by-layer forward with autograd (layers/inner.py:126-127, 266-273 through
// semiring.py:383-408 and utils.py:10-30).  Here a training step is a short list of LEVEL launches over JOBS:
//
// * a SUM job is one fold of a dense / CP-T layer with 64 inputs and 64 outputs: v = the sum of its input blocks (a Hadamard
//   product in log space: the product layer itself is never evaluated), e = exp(v - max v), y = W e, out = log y + max v.
//   Backward (ck_backward.hip, head comment): gy = G / y with G the SUM of the job's gradient blocks, gx = e * (W^T gy) --
//   ONE block, the gradient of every input of the product --, dW = gy^T e accumulated over the job's rows in registers,
//   reduced in LDS and pushed through the softmax parameterisation W = softmax(theta) by the same workgroup
//   (nodes.py:764-772): d theta = W (dW - <W, dW>) is written once, no atomics, no zero fill, no dW in memory -- and, with the
//   fused optimizer, Adam's update of theta and the softmax of the NEXT step's weights follow in the same epilogue;
// * a MIX job is one fold of a mixing layer (nodes.py:847-862): elementwise over units, H slots that are sums of blocks;
// * an NSUM job adds blocks (a product layer that is kept, or a gradient with many readers);
// * the ROOT launch evaluates the scalar sum folds and the final mixing layer at the top of the circuit, sums the
//   log-likelihood, and runs their backward in the same launch (the seed of the mean log-likelihood is a constant).
//
// Every gradient block has ONE writer (plain stores); a reader adds the blocks of its list.  Blocks are (rows, 64) fp32,
// row-major -- the reference's layout.  All tables (jobs, pointer pool) live in device memory and are built once per batch
// size by cirkit_amd/train_jobs.py.
#include "ck_internal.h"
#include "ck_opt.h"
#include "ck_tile.h"

namespace {

constexpr int kU = 64;    // units per block row
constexpr int kWS = 65;   // row stride of a weight matrix in LDS (odd: column AND row walks are conflict-free)
constexpr int kTS = 68;   // row stride of the transpose tiles (16-byte aligned rows)

typedef ck_sum_job SumJob;
typedef ck_mix_job MixJob;

__device__ __forceinline__ float row16_reduce_max(float v) {
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x140, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true));
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));
  return v;
}

// The (64, 64) row-major weights of a job into LDS, row stride kWS: all loads first, then the stores.
template <int THREADS>
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, float* __restrict__ w_s, int tid) {
  constexpr int N = 1024 / THREADS;
  float4 v[N];
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] = ck::gload4(w + 4 * (tid + THREADS * u));
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int i = tid + THREADS * u;
    float* d = w_s + (i >> 4) * kWS + (i & 15) * 4;
    d[0] = v[u].x;
    d[1] = v[u].y;
    d[2] = v[u].z;
    d[3] = v[u].w;
  }
}

// The row of the input blocks a batch row reads: itself, or -- a job whose input is a Categorical fold (`xrow`) -- the row of the
// fold's (C + 1, 64) log-probability table its category selects (layers/input.py:399-412; negative = marginalised = row C).
__device__ __forceinline__ int64_t input_row(const int32_t* __restrict__ xrow, int C, int64_t bl) {
  if (xrow == nullptr) return bl;
  const int cc = xrow[bl];
  return cc < 0 ? C : min(cc, C - 1);
}

// v[q][.] = sum over the job's input blocks of row bl, units 32 q + 8 g + 4 kh + t (register 4 g + t)
__device__ __forceinline__ void load_inputs(const float* const* __restrict__ pool, int off, int n, int64_t bl, int kh,
                                            float (&e)[2][16]) {
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) e[q][j] = 0.f;
  for (int s = 0; s < n; ++s) {
    const float* src = pool[off + s] + bl * kU + 4 * kh;
    tile_load_add(src, e[0]);
    tile_load_add(src + 32, e[1]);
  }
}

__device__ __forceinline__ float exp_tile(float (&e)[2][16], bool live) {
  float m = e[0][0];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, e[q][j]);
  m = ck::clamp_finite(ck::xhalf_max(m));
  const float nml = exp_offset(m, 0.f);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) e[q][j] = live ? __builtin_amdgcn_exp2f(fmaf(e[q][j], kL2E, nml)) : 0.f;
  return m;
}

// y[p] = W e for the 32 outputs 32 p + ...: A operand lane (o, kh) = W[32 p + o][32 q + 8 g + 4 kh + t]
__device__ __forceinline__ f32x16 contract_rows(const float* __restrict__ w_s, int p, int b_in, int kh, const float (&e)[2][16]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* wr = w_s + (32 * p + b_in) * kWS + 4 * kh;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[32 * q + 8 * g + t], e[q][4 * g + t], acc, 0, 0, 0);
  return acc;
}

// ---- forward ------------------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    jobs_sum64_fwd_kernel(const SumJob* __restrict__ jobs, const float* const* __restrict__ pool) {
  __shared__ float w_s[kU * kWS];
  const SumJob& J = jobs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int in_off = J.in_off, n_in = J.n_in, row1 = J.row1;
  float e[2][16];
  int b0 = J.row0 + 32 * wave;
  const int32_t* xrow = reinterpret_cast<const int32_t*>(J.xrow);
  const int Cn = J.C;
  if (b0 < row1) load_inputs(pool, in_off, n_in, input_row(xrow, Cn, b0 + b_in < row1 ? b0 + b_in : row1 - 1), kh, e);  // (in flight while the weights are staged)
  stage_weights<WAVES * 64>(J.w, w_s, threadIdx.x);
  __syncthreads();
  float* __restrict__ out = J.out;
  for (bool first = true; b0 < row1; b0 += 32 * WAVES, first = false) {
    const int b = b0 + b_in;
    const bool live = b < row1;
    const int64_t bl = live ? b : row1 - 1;
    if (!first) load_inputs(pool, in_off, n_in, input_row(xrow, Cn, bl), kh, e);
    const float m = exp_tile(e, true);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x16 acc = contract_rows(w_s, p, b_in, kh, e);
      if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ck::gstore4(out + bl * kU + 32 * p + 8 * g + 4 * kh,
                      make_float4(fmaf(__builtin_amdgcn_logf(acc[4 * g]), kLN2, m), fmaf(__builtin_amdgcn_logf(acc[4 * g + 1]), kLN2, m),
                                  fmaf(__builtin_amdgcn_logf(acc[4 * g + 2]), kLN2, m), fmaf(__builtin_amdgcn_logf(acc[4 * g + 3]), kLN2, m)));
      }
    }
  }
}

// ---- backward -----------------------------------------------------------------------------------------------------------
#ifndef CK_JOBS_BWD_OCC
#define CK_JOBS_BWD_OCC 2  // waves per SIMD the backward kernel is compiled for (lab builds: 3 spills 34 registers)
#endif
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(CK_JOBS_BWD_OCC, CK_JOBS_BWD_OCC)))
    jobs_sum64_bwd_kernel(const SumJob* __restrict__ jobs, const float* const* __restrict__ pool,
                          const ck_opt_state* __restrict__ opt) {
  static_assert(WAVES == 4 || WAVES == 8, "the epilogue assigns WAVES threads to a weight row");
  // [weights 64 x 65][per wave: gy 16 x 68, e 16 x 68 -- later WAVES / 2 sum buffers of 64 x 64]
  constexpr int kBufs = WAVES / 2;
  constexpr int kArea = WAVES * 2 * 16 * kTS > kBufs * 4096 ? WAVES * 2 * 16 * kTS : kBufs * 4096;
  __shared__ __attribute__((aligned(16))) float lds[kU * kWS + kArea];
  __shared__ unsigned int s_ticket;
  __shared__ float lw_s[kU];  // a job under a mixing fold: log of its slot's coefficient per unit
  float* w_s = lds;
  float* area = lds + kU * kWS;  // (64 x 65 floats: a multiple of 16 bytes)
  const SumJob& J = jobs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int in_off = J.in_off, n_in = J.n_in, g_off = J.g_off, n_g = J.n_g, row1 = J.row1;
  float e[2][16], gy[2][16];
  // the operands of a tile: v (into e) and G = the sum of the job's gradient blocks (into gy: the layout of the outputs)
  const int32_t* xrow = reinterpret_cast<const int32_t*>(J.xrow);
  const int Cn = J.C;
  auto load_tile = [&](int64_t bl) {
    load_inputs(pool, in_off, n_in, input_row(xrow, Cn, bl), kh, e);
    load_inputs(pool, g_off, n_g, bl, kh, gy);
  };
  int b0 = J.row0 + 32 * wave;
  if (b0 < row1) load_tile(b0 + b_in < row1 ? b0 + b_in : row1 - 1);  // (the first tile's loads fly while the weights are staged)
  const float* __restrict__ mix_out = J.mix_out;
  const int partner_off = J.partner_off, n_partner = J.n_partner;
  if (mix_out != nullptr && threadIdx.x < kU) lw_s[threadIdx.x] = logf(J.mix_w[threadIdx.x * J.mix_H + J.mix_h]);
  stage_weights<WAVES * 64>(J.w, w_s, threadIdx.x);
  __syncthreads();
  float* gy_s = area + wave * (2 * 16 * kTS);
  float* e_s = gy_s + 16 * kTS;
  f32x16 dwacc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) dwacc[p][q][r] = 0.f;
  float* __restrict__ gx = J.gx;
  for (bool first = true; b0 < row1; b0 += 32 * WAVES, first = false) {
    const int b = b0 + b_in;
    const bool live = b < row1;
    const int64_t bl = live ? b : row1 - 1;
    if (!first) load_tile(bl);
    const float m_in = exp_tile(e, live);
    // y = W e, gy = G / y
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x16 acc = contract_rows(w_s, p, b_in, kh, e);
      if (mix_out != nullptr) {
        // the gradient of this job's output through the mixing fold above it: (the fold's G) * w_h * exp(x_h - out_mix), x_h = the
        // job's own output -- recomputed exactly as the forward stored it -- plus the other factors of its slot
        float xh[16], om[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xh[r] = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m_in);
        for (int s2 = 0; s2 < n_partner; ++s2) tile_load_add(pool[partner_off + s2] + bl * kU + 32 * p + 4 * kh, xh);
        tile_load(mix_out + bl * kU + 32 * p + 4 * kh, om);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t = xh[r] + lw_s[32 * p + 8 * (r >> 2) + 4 * kh + (r & 3)] - om[r];
          gy[p][r] = (acc[r] > 0.f && gy[p][r] != 0.f) ? gy[p][r] * __builtin_amdgcn_exp2f(t * kL2E) : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) gy[p][r] = (live && acc[r] > 0.f && gy[p][r] != 0.f) ? gy[p][r] * __builtin_amdgcn_rcpf(acc[r]) : 0.f;
    }
    // gx = e * (W^T gy): A operand lane (n, kh) = W[32 p + 8 g + 4 kh + t][32 q + n]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w_s[(32 * p + 8 * g + 4 * kh + t) * kWS + 32 * q + b_in], gy[p][4 * g + t], acc, 0, 0, 0);
      if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ck::gstore4(gx + bl * kU + 32 * q + 8 * g + 4 * kh,
                      make_float4(acc[4 * g] * e[q][4 * g], acc[4 * g + 1] * e[q][4 * g + 1], acc[4 * g + 2] * e[q][4 * g + 2],
                                  acc[4 * g + 3] * e[q][4 * g + 3]));
      }
    }
    // dW += gy^T e: contracts over the rows, so both tiles go through LDS, 16 rows at a time (row r, unit u at [r * kTS + u])
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((b_in >> 4) == half) {
        const int r = b_in & 15;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(gy_s + r * kTS + 32 * q + 8 * g + 4 * kh) =
                make_float4(gy[q][4 * g], gy[q][4 * g + 1], gy[q][4 * g + 2], gy[q][4 * g + 3]);
            *reinterpret_cast<float4*>(e_s + r * kTS + 32 * q + 8 * g + 4 * kh) =
                make_float4(e[q][4 * g], e[q][4 * g + 1], e[q][4 * g + 2], e[q][4 * g + 3]);
          }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        const int bb = 8 * kh + s2;  // row (of this half) contracted by lanes (., kh) at step s2
        const float a0 = gy_s[bb * kTS + b_in], a1 = gy_s[bb * kTS + 32 + b_in];
        const float c0 = e_s[bb * kTS + b_in], c1 = e_s[bb * kTS + 32 + b_in];
        dwacc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, dwacc[0][0], 0, 0, 0);
        dwacc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c1, dwacc[0][1], 0, 0, 0);
        dwacc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c0, dwacc[1][0], 0, 0, 0);
        dwacc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, dwacc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);  // (the reads are done before the next half overwrites the tiles)
      __builtin_amdgcn_wave_barrier();
    }
  }
  // mode 2: the optimizer's operands of this thread's 16 entries (row o, quarter c0), requested now -- they arrive while the
  // accumulators are reduced below
  constexpr int kTPR = WAVES;       // threads per weight row
  constexpr int kEPT = kU / kTPR;   // entries per thread: 16 | 8
  const int o = threadIdx.x / kTPR, c0 = (threadIdx.x % kTPR) * kEPT;
  float4 pth[kEPT / 4], pm1[kEPT / 4], pm2[kEPT / 4];
  const bool fused_opt = J.mode == 2;
  const bool adam = fused_opt && opt->kind != 0;
  if (fused_opt) {
#pragma unroll
    for (int k = 0; k < kEPT / 4; ++k) {
      pth[k] = ck::gload4(J.theta + o * kU + c0 + 4 * k);
      if (adam) {
        pm1[k] = ck::gload4(J.m1 + o * kU + c0 + 4 * k);
        pm2[k] = ck::gload4(J.m2 + o * kU + c0 + 4 * k);
      }
    }
  }
  // dW of the job's rows: the waves' accumulators added in LDS (WAVES / 2 buffers of (64, 64): the first half of the waves
  // stores, the second half adds)
  __syncthreads();
  float* buf = area + (wave % kBufs) * 4096;
  if (wave < kBufs) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[(32 * p + 8 * (r >> 2) + 4 * kh + (r & 3)) * kU + 32 * q + b_in] = dwacc[p][q][r];
  }
  __syncthreads();
  if (wave >= kBufs) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[(32 * p + 8 * (r >> 2) + 4 * kh + (r & 3)) * kU + 32 * q + b_in] += dwacc[p][q][r];
  }
  __syncthreads();
  float* dw_s = area;  // (64, 64) after the sum below
  constexpr int kPer = 4096 / (WAVES * 64);  // entries of the sum per thread
  auto sum_bufs = [&](int i) {
    float t = area[i];
#pragma unroll
    for (int k = 1; k < kBufs; ++k) t += area[k * 4096 + i];
    return t;
  };
  if (J.n_split > 1) {
    // the job's rows are cut over several workgroups: partial sums go to slots (write-through), the last arrival adds them
    float* slot = J.part + static_cast<int64_t>(J.split) * 4096;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int i = threadIdx.x + WAVES * 64 * u;
      __hip_atomic_store(slot + i, sum_bufs(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(J.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != static_cast<unsigned int>(J.n_split - 1)) return;
    if (threadIdx.x == 0) __hip_atomic_store(J.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next step
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int i = threadIdx.x + WAVES * 64 * u;
      float t = 0.f;
      for (int sp = 0; sp < J.n_split; ++sp)
        t += __hip_atomic_load(J.part + static_cast<int64_t>(sp) * 4096 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dw_s[i] = t;
    }
  } else {
    float t[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) t[u] = sum_bufs(threadIdx.x + WAVES * 64 * u);
    if (kBufs > 1) __syncthreads();  // (every thread has read its entries of all buffers before buffer 0 is overwritten)
#pragma unroll
    for (int u = 0; u < kPer; ++u) dw_s[threadIdx.x + WAVES * 64 * u] = t[u];
  }
  __syncthreads();
  // epilogue: thread (o, part) owns kEPT consecutive entries of weight row o
  auto row_sum = [&](float v) {
    v = quad_sum(v);
    if (kTPR == 8) v += __shfl_xor(v, 4, 64);
    return v;
  };
  auto row_max = [&](float v) {
    v = quad_max(v);
    if (kTPR == 8) v = fmaxf(v, __shfl_xor(v, 4, 64));
    return v;
  };
  float wv[kEPT], dv[kEPT];
#pragma unroll
  for (int k = 0; k < kEPT; ++k) {
    wv[k] = w_s[o * kWS + c0 + k];
    dv[k] = dw_s[o * kU + c0 + k];
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kEPT; ++k) s = fmaf(wv[k], dv[k], s);
  s = row_sum(s);  // = sum_b G[b, o]: sum_n W[o, n] dW[o, n] = sum_b gy[b, o] y[b, o]
#ifdef CK_JOBS_LAB_DEFECT  // lab build only (scripts/defect_injection.sh): a wrong <W, dW> that the gradient tests must see
  s *= 1.f + CK_JOBS_LAB_DEFECT;
#endif
  if (J.mix_dw != nullptr && threadIdx.x % kTPR == 0) {
    // d w[o, h] = s / w.  A coefficient that underflowed to 0 (a logit gap beyond ~87) made every G of this slot 0, so s = 0
    // and the quotient would be 0 / 0: its logit's gradient w (dw - s') is 0 whatever dw is, as in the reference's autograd
    const float wmix = J.mix_w[o * J.mix_H + J.mix_h];
    J.mix_dw[o * J.mix_H + J.mix_h] = wmix > 0.f ? s / wmix : 0.f;
  }
  if (J.mode == 0) {  // the gradient of the linear weights, for a parameter graph this epilogue does not know
#pragma unroll
    for (int k = 0; k < kEPT; k += 4) ck::gstore4(J.dtheta + o * kU + c0 + k, make_float4(dv[k], dv[k + 1], dv[k + 2], dv[k + 3]));
    return;
  }
#pragma unroll
  for (int k = 0; k < kEPT; ++k) dv[k] = wv[k] * (dv[k] - s);  // d theta (nodes.py:764-772 under autograd)
  if (J.mode == 1) {
#pragma unroll
    for (int k = 0; k < kEPT; k += 4) ck::gstore4(J.dtheta + o * kU + c0 + k, make_float4(dv[k], dv[k + 1], dv[k + 2], dv[k + 3]));
    return;
  }
  // mode 2: the optimizer's update of theta and the softmax of the next step's weights, here
  const ck_opt_state os = *opt;
  const OptK ok = opt_k(os);
  if (os.skip_now) return;
  float th[kEPT];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < kEPT; k += 4) {
    const float4 t4 = pth[k >> 2];
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
    if (os.kind != 0) {
      a4 = pm1[k >> 2];
      b4 = pm2[k >> 2];
    }
    th[k] = opt_update(ok, t4.x, dv[k], a4.x, b4.x);
    th[k + 1] = opt_update(ok, t4.y, dv[k + 1], a4.y, b4.y);
    th[k + 2] = opt_update(ok, t4.z, dv[k + 2], a4.z, b4.z);
    th[k + 3] = opt_update(ok, t4.w, dv[k + 3], a4.w, b4.w);
    ck::gstore4(J.theta + o * kU + c0 + k, make_float4(th[k], th[k + 1], th[k + 2], th[k + 3]));
    if (os.kind != 0) {
      ck::gstore4(J.m1 + o * kU + c0 + k, a4);
      ck::gstore4(J.m2 + o * kU + c0 + k, b4);
    }
    mx = fmaxf(fmaxf(mx, fmaxf(th[k], th[k + 1])), fmaxf(th[k + 2], th[k + 3]));
  }
  mx = row_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < kEPT; ++k) {
    th[k] = expf(th[k] - mx);
    sum += th[k];
  }
  sum = row_sum(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < kEPT; k += 4)
    ck::gstore4(J.w_out + o * kU + c0 + k, make_float4(th[k] * inv, th[k + 1] * inv, th[k + 2] * inv, th[k + 3] * inv));
}

// ---- mixing jobs ----------------------------------------------------------------------------------------------------------
// out[k] = log(sum_h w[k, h] e_h[k]) + m[k],  e_h[k] = exp(x_h[k] - m[k]),  m[k] = max_h x_h[k]  (a TorchSumLayer whose weight is a
// TorchMixingWeightParameter, nodes.py:847-862, under semiring.py:383-408: the reference shifts by the maximum of the whole row,
// the value is the same for any shift that bounds the unit's own inputs);  x_h = the sum of slot h's blocks.
// 256 threads: 16 rows per pass, thread (r, c) owns units 4 c .. 4 c + 3 of its row.
//
// One slot into a unit's running sum: y <- w exp(x - m') + y exp(m - m'), m' = clamp(max(m, x)), m <- m'.  The larger of (m, x)
// IS m' unless it is infinite, so its exponential is 1 (inf / 0 at +-inf): ONE exponential per slot and unit, of the smaller
// one -- the launch was bound by issuing two library exponentials per slot and unit (config 4: 0.56 ms of a 4.6 ms step).
#ifndef CK_MIX_LIBM
#define CK_MIX_LIBM 0  // 1: expf / logf of the library instead of v_exp_f32 / v_log_f32 (as the sum jobs: ~1e-7 relative)
#endif
__device__ __forceinline__ float mix_exp(float d) {  // d <= 0 (or NaN)
#if CK_MIX_LIBM
  return expf(d);
#else
  const float t = d * kL2E;
  const float lo = fmaf(d, kL2E, -t);  // the rounding error of the product, put back: exp2(t + lo) = exp2(t) (1 + lo ln 2)
  const float e = __builtin_amdgcn_exp2f(t);
  return fabsf(t) < 1e30f ? fmaf(e, lo * kLN2, e) : e;
#endif
}
__device__ __forceinline__ float mix_log(float y) {
#if CK_MIX_LIBM
  return logf(y);
#else
  return __builtin_amdgcn_logf(y) * kLN2;
#endif
}
__device__ __forceinline__ void mix_step(float w, float x, float& m, float& y) {
  const bool x_big = x >= m;
  const float big = fmaxf(m, x);  // (a NaN x is the "smaller" one: its exponential is NaN, as before)
  const float mn = ck::clamp_finite(big);
  const float t = mix_exp((x_big ? m : x) - mn);
  const float db = big - mn;
  const float eb = db == 0.f ? 1.f : (db > 0.f ? INFINITY : (db < 0.f ? 0.f : db));
  y = fmaf(w, x_big ? eb : t, y * (x_big ? t : eb));
  m = mn;
}
__global__ void __launch_bounds__(256)
    jobs_mix_fwd_kernel(const MixJob* __restrict__ jobs, const float* const* __restrict__ pool) {
  __shared__ float w_s[kU * 17];
  const MixJob& J = jobs[blockIdx.x];
  const int H = J.H, S = J.S, in_off = J.in_off, row1 = J.row1;
  for (int i = threadIdx.x; i < kU * H; i += 256) w_s[(i / H) * 17 + (i % H)] = J.w[i];
  __syncthreads();
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  for (int b0 = J.row0; b0 < row1; b0 += 16) {
    const int b = b0 + r;
    const bool live = b < row1;
    const int64_t bl = live ? b : row1 - 1;
    // ONE pass over the slots: a running maximum PER UNIT (the sum of a unit only involves that unit's inputs, so any
    // shift that bounds them gives the reference's value; the reference shifts by the maximum of the whole row)
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), y = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h0 = 0; h0 < H; h0 += 4) {
      // four slots' blocks in flight (slot after slot, block after block was a chain of H S round trips per 16 rows); a slot's
      // blocks are added in list order, the slots enter the running sum in order
      float4 xs[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) xs[k] = ck::gload4(pool[in_off + min(h0 + k, H - 1) * S] + bl * kU + 4 * c);
      for (int s = 1; s < S; ++s) {
        float4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = ck::gload4(pool[in_off + min(h0 + k, H - 1) * S + s] + bl * kU + 4 * c);
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[k].x += t[k].x, xs[k].y += t[k].y, xs[k].z += t[k].z, xs[k].w += t[k].w;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int h = h0 + k;
        if (h >= H) break;
        const float4 x = xs[k];
        mix_step(w_s[(4 * c + 0) * 17 + h], x.x, m.x, y.x);
        mix_step(w_s[(4 * c + 1) * 17 + h], x.y, m.y, y.y);
        mix_step(w_s[(4 * c + 2) * 17 + h], x.z, m.z, y.z);
        mix_step(w_s[(4 * c + 3) * 17 + h], x.w, m.w, y.w);
      }
    }
    if (live) ck::gstore4(J.out + bl * kU + 4 * c, make_float4(mix_log(y.x) + m.x, mix_log(y.y) + m.y, mix_log(y.z) + m.z, mix_log(y.w) + m.w));
  }
}

// Backward: gx_h[k] = G[k] w[k, h] e_h[k] / y[k] (one block per slot), d w[k, h] = sum_b G[k] e_h[k] / y[k] in registers over the
// job's rows, reduced over the workgroup, then the softmax over h behind the mixing coefficients (nodes.py:764-772, 847-862).
template <int HMAX>
__global__ void __launch_bounds__(256)
    jobs_mix_bwd_kernel(const MixJob* __restrict__ jobs, const float* const* __restrict__ pool, const ck_opt_state* __restrict__ opt,
                        int64_t gx_stride) {
  __shared__ float w_s[kU * 17];
  __shared__ float red[4][kU * HMAX];
  __shared__ unsigned int s_ticket;
  const MixJob& J = jobs[blockIdx.x];
  const int H = J.H, S = J.S, in_off = J.in_off, g_off = J.g_off, n_g = J.n_g, row1 = J.row1;
  for (int i = threadIdx.x; i < kU * H; i += 256) w_s[(i / H) * 17 + (i % H)] = J.w[i];
  __syncthreads();
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  float4 dacc[HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; ++h) dacc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = J.row0; b0 < row1; b0 += 16) {
    const int b = b0 + r;
    const bool live = b < row1;
    const int64_t bl = live ? b : row1 - 1;
    float4 x[HMAX];
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
      if (h < H) {
        x[h] = ck::gload4(pool[in_off + h * S] + bl * kU + 4 * c);
        for (int s = 1; s < S; ++s) {
          const float4 t = ck::gload4(pool[in_off + h * S + s] + bl * kU + 4 * c);
          x[h].x += t.x, x[h].y += t.y, x[h].z += t.z, x[h].w += t.w;
        }
        m = make_float4(fmaxf(m.x, x[h].x), fmaxf(m.y, x[h].y), fmaxf(m.z, x[h].z), fmaxf(m.w, x[h].w));  // (per unit, as the forward)
      }
    m = make_float4(ck::clamp_finite(m.x), ck::clamp_finite(m.y), ck::clamp_finite(m.z), ck::clamp_finite(m.w));
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
      if (h < H) {
        x[h] = make_float4(expf(x[h].x - m.x), expf(x[h].y - m.y), expf(x[h].z - m.z), expf(x[h].w - m.w));
        y.x = fmaf(w_s[(4 * c + 0) * 17 + h], x[h].x, y.x);
        y.y = fmaf(w_s[(4 * c + 1) * 17 + h], x[h].y, y.y);
        y.z = fmaf(w_s[(4 * c + 2) * 17 + h], x[h].z, y.z);
        y.w = fmaf(w_s[(4 * c + 3) * 17 + h], x[h].w, y.w);
      }
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < n_g; ++s) {
      const float4 t = ck::gload4(pool[g_off + s] + bl * kU + 4 * c);
      g.x += t.x, g.y += t.y, g.z += t.z, g.w += t.w;
    }
    g.x = (live && y.x > 0.f && g.x != 0.f) ? g.x / y.x : 0.f;
    g.y = (live && y.y > 0.f && g.y != 0.f) ? g.y / y.y : 0.f;
    g.z = (live && y.z > 0.f && g.z != 0.f) ? g.z / y.z : 0.f;
    g.w = (live && y.w > 0.f && g.w != 0.f) ? g.w / y.w : 0.f;
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
      if (h < H) {
        const float4 ge = make_float4(g.x * x[h].x, g.y * x[h].y, g.z * x[h].z, g.w * x[h].w);
        dacc[h].x += ge.x, dacc[h].y += ge.y, dacc[h].z += ge.z, dacc[h].w += ge.w;
        if (live)
          ck::gstore4(J.gx + h * gx_stride + bl * kU + 4 * c,
                      make_float4(ge.x * w_s[(4 * c + 0) * 17 + h], ge.y * w_s[(4 * c + 1) * 17 + h], ge.z * w_s[(4 * c + 2) * 17 + h],
                                  ge.w * w_s[(4 * c + 3) * 17 + h]));
      }
  }
  // the four row groups of a wave (lanes 16 apart), then the four waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int h = 0; h < HMAX; ++h) {
    float4 v = dacc[h];
    v.x += __shfl_xor(v.x, 16, 64), v.y += __shfl_xor(v.y, 16, 64), v.z += __shfl_xor(v.z, 16, 64), v.w += __shfl_xor(v.w, 16, 64);
    v.x += __shfl_xor(v.x, 32, 64), v.y += __shfl_xor(v.y, 32, 64), v.z += __shfl_xor(v.z, 32, 64), v.w += __shfl_xor(v.w, 32, 64);
    if (lane < 16) {
      red[wave][(4 * c + 0) * HMAX + h] = v.x;
      red[wave][(4 * c + 1) * HMAX + h] = v.y;
      red[wave][(4 * c + 2) * HMAX + h] = v.z;
      red[wave][(4 * c + 3) * HMAX + h] = v.w;
    }
  }
  __syncthreads();
  float* dw_s = red[0];
  for (int i = threadIdx.x; i < kU * HMAX; i += 256) dw_s[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  __syncthreads();
  if (J.n_split > 1) {
    float* slot = J.part + static_cast<int64_t>(J.split) * (kU * HMAX);
    for (int i = threadIdx.x; i < kU * HMAX; i += 256) __hip_atomic_store(slot + i, dw_s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(J.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != static_cast<unsigned int>(J.n_split - 1)) return;
    if (threadIdx.x == 0) __hip_atomic_store(J.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = threadIdx.x; i < kU * HMAX; i += 256) {
      float t = 0.f;
      for (int sp = 0; sp < J.n_split; ++sp)
        t += __hip_atomic_load(J.part + static_cast<int64_t>(sp) * (kU * HMAX) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dw_s[i] = t;
    }
    __syncthreads();
  }
  if (threadIdx.x >= kU) return;
  const int k = threadIdx.x;
  if (J.mode == 0) {
    for (int h = 0; h < H; ++h) J.dtheta[k * H + h] = dw_s[k * HMAX + h];
    return;
  }
  float s = 0.f;
  for (int h = 0; h < H; ++h) s = fmaf(w_s[k * 17 + h], dw_s[k * HMAX + h], s);
  if (J.mode == 1) {
    for (int h = 0; h < H; ++h) J.dtheta[k * H + h] = w_s[k * 17 + h] * (dw_s[k * HMAX + h] - s);
    return;
  }
  const ck_opt_state os = *opt;
  const OptK ok = opt_k(os);
  if (os.skip_now) return;
  float mx = -INFINITY;
  for (int h = 0; h < H; ++h) {
    const float g = w_s[k * 17 + h] * (dw_s[k * HMAX + h] - s);
    float a = os.kind ? J.m1[k * H + h] : 0.f, v = os.kind ? J.m2[k * H + h] : 0.f;
    const float th = opt_update(ok, J.theta[k * H + h], g, a, v);
    J.theta[k * H + h] = th;
    if (os.kind) {
      J.m1[k * H + h] = a;
      J.m2[k * H + h] = v;
    }
    dw_s[k * HMAX + h] = th;
    mx = fmaxf(mx, th);
  }
  float sum = 0.f;
  for (int h = 0; h < H; ++h) {
    const float ex = expf(dw_s[k * HMAX + h] - mx);
    dw_s[k * HMAX + h] = ex;
    sum += ex;
  }
  for (int h = 0; h < H; ++h) J.w_out[k * H + h] = dw_s[k * HMAX + h] / sum;
}

// The parameter step of mixing folds whose backward ran inside their factors' sum jobs: d w (64, H) is in `part`.
__global__ void __launch_bounds__(64) jobs_mix_params_kernel(const MixJob* __restrict__ jobs, const ck_opt_state* __restrict__ opt) {
  const MixJob& J = jobs[blockIdx.x];
  const int H = J.H, k = threadIdx.x;
  const float* dw = J.part + k * H;
  const float* w = J.w + k * H;
  if (J.mode == 0) {
    for (int h = 0; h < H; ++h) J.dtheta[k * H + h] = dw[h];
    return;
  }
  float s = 0.f;
  for (int h = 0; h < H; ++h) s = fmaf(w[h], dw[h], s);
  if (J.mode == 1) {
    for (int h = 0; h < H; ++h) J.dtheta[k * H + h] = w[h] * (dw[h] - s);
    return;
  }
  const ck_opt_state os = *opt;
  const OptK ok = opt_k(os);
  if (os.skip_now) return;
  float th[16];
  float mx = -INFINITY;
  for (int h = 0; h < H; ++h) {
    const float g = w[h] * (dw[h] - s);
    float a = os.kind ? J.m1[k * H + h] : 0.f, v = os.kind ? J.m2[k * H + h] : 0.f;
    th[h] = opt_update(ok, J.theta[k * H + h], g, a, v);
    J.theta[k * H + h] = th[h];
    if (os.kind) {
      J.m1[k * H + h] = a;
      J.m2[k * H + h] = v;
    }
    mx = fmaxf(mx, th[h]);
  }
  float sum = 0.f;
  for (int h = 0; h < H; ++h) {
    th[h] = expf(th[h] - mx);
    sum += th[h];
  }
  for (int h = 0; h < H; ++h) J.w_out[k * H + h] = th[h] / sum;
}

// ---- block sums ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    jobs_nsum_kernel(const ck_nsum_job* __restrict__ jobs, const float* const* __restrict__ pool, int64_t elems) {
  const ck_nsum_job& J = jobs[blockIdx.y];
  const int n_in = J.n_in;
  const float* const* __restrict__ src = pool + J.in_off;
  for (int64_t i = (blockIdx.x * 256ll + threadIdx.x) * 4; i < elems; i += gridDim.x * 1024ll) {
    // four blocks' loads in flight (a thread has ONE iteration at 1024 rows: block after block is a chain of round trips),
    // added in list order
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < n_in; s0 += 4) {
      float4 t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = ck::gload4(src[min(s0 + k, n_in - 1)] + i);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (s0 + k < n_in) {
          if (s0 + k == 0) a = t[k];
          else a.x += t[k].x, a.y += t[k].y, a.z += t[k].z, a.w += t[k].w;
        }
    }
    ck::gstore4(J.out + i, a);
  }
}

// ---- the top of the circuit -----------------------------------------------------------------------------------------------------
// R scalar sum folds (64 inputs each: o_r = log(sum_n w_r[n] exp(v_r[n] - m_r)) + m_r, v_r the sum of the fold's blocks) under a
// final mixing layer out = log(sum_r c_r exp(o_r - M)) + M (or out = o_0 when there is none); the log-likelihood sum; and, with
// `gx`, their backward for the loss  sum_b seed_b out_b:  gx_r[n] = seed p_r w_r[n] e_r[n] / y_r  with  p_r = c_r exp(o_r - M) / Y,
// d w_r[n] = sum_b seed p_r e_r[n] / y_r,  d c_r = sum_b seed exp(o_r - M) / Y -- per-workgroup partial sums, the last arrival adds
// them in workgroup order and applies the softmax parameterisations.  One wave per row, lane = input unit.
constexpr int kRootMax = 16;
__global__ void __launch_bounds__(256) jobs_root_kernel(const ck_root_launch a) {
  __shared__ float red[4][kRootMax * kU + kRootMax];
  __shared__ double red_ll[4];
  __shared__ unsigned int s_ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = a.R, B = a.B;
  const bool bad = a.bad_flag != nullptr && *a.bad_flag != 0;
  // the block pointers of every (fold, list position) once, in LDS: a row's loads are then issued together (one round trip)
  __shared__ const float* ptr_s[kRootMax * 4];
  const int S = a.S;
  for (int i = threadIdx.x; i < R * S; i += 256) ptr_s[(i / S) * 4 + (i % S)] = a.pool[a.in_off[i / S] + (i % S)];
  __syncthreads();
  float wr[kRootMax], dw[kRootMax], dc[kRootMax], cr[kRootMax];
#pragma unroll
  for (int r = 0; r < kRootMax; ++r) {
    wr[r] = r < R ? a.w[r][lane] : 0.f;
    cr[r] = (r < R && a.c != nullptr) ? a.c[r] : 1.f;
    dw[r] = 0.f;
    dc[r] = 0.f;
  }
  double ll = 0.0;
  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    float e[kRootMax], o[kRootMax], yr[kRootMax];
    float M = -INFINITY;
#pragma unroll
    for (int r = 0; r < kRootMax; ++r) {
      e[r] = 0.f;
      if (r < R) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
          if (s2 < S) e[r] += *ck::as_global(ptr_s[r * 4 + s2] + static_cast<int64_t>(b) * kU + lane);
      }
    }
#pragma unroll
    for (int r = 0; r < kRootMax; ++r)
      if (r < R) {
        const float v = e[r];
        const float m = ck::clamp_finite(ck::wave_max(v));
        e[r] = expf(v - m);
        yr[r] = ck::wave_sum(wr[r] * e[r]);
        o[r] = logf(yr[r]) + m;
        M = fmaxf(M, o[r]);
      }
    float out;
    float Y = 0.f;
    if (a.c != nullptr) {
      M = ck::clamp_finite(M);
#pragma unroll
      for (int r = 0; r < kRootMax; ++r)
        if (r < R) {
          o[r] = expf(o[r] - M);  // from here on: exp(o_r - M)
          Y = fmaf(cr[r], o[r], Y);
        }
      out = logf(Y) + M;
    } else {
      out = o[0];
      o[0] = 1.f;
      Y = 1.f;
    }
    if (bad) out = __builtin_nanf("");
    if (lane == 0) a.out[b] = out;
    ll += static_cast<double>(out);
    if (a.gx != nullptr) {
      const float seed = a.seed != nullptr ? a.seed[b] : a.seed_const;
#pragma unroll
      for (int r = 0; r < kRootMax; ++r)
        if (r < R) {
          const float q = (Y > 0.f) ? seed * o[r] / Y : 0.f;  // seed * exp(o_r - M) / Y
          const float t = (yr[r] > 0.f) ? q * cr[r] * e[r] / yr[r] : 0.f;
          dc[r] += q;
          dw[r] += t;
          a.gx[(static_cast<int64_t>(r) * B + b) * kU + lane] = t * wr[r];
        }
    }
  }
  // partial sums of this workgroup
#pragma unroll
  for (int r = 0; r < kRootMax; ++r)
    if (r < R) {
      red[wave][r * kU + lane] = dw[r];
      if (lane == 0) red[wave][kRootMax * kU + r] = dc[r];
    }
  if (lane == 0) red_ll[wave] = ll;
  __syncthreads();
  const int n_part = R * kU + R;
  float* slot = a.part + static_cast<int64_t>(blockIdx.x) * (kRootMax * kU + kRootMax + 2);
  for (int i = threadIdx.x; i < n_part; i += 256) {
    const int j = i < R * kU ? i : kRootMax * kU + (i - R * kU);
    __hip_atomic_store(slot + i, (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    const double t = (red_ll[0] + red_ll[1]) + (red_ll[2] + red_ll[3]);
    __hip_atomic_store(reinterpret_cast<double*>(slot + kRootMax * kU + kRootMax), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;
  if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* tot = &red[0][0];
  for (int i = threadIdx.x; i < n_part; i += 256) {
    float t = 0.f;
    for (unsigned int g0 = 0; g0 < gridDim.x; g0 += 8) {  // eight slots' loads in flight, added in workgroup order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        v[k] = g0 + k < gridDim.x ? __hip_atomic_load(a.part + static_cast<int64_t>(g0 + k) * (kRootMax * kU + kRootMax + 2) + i, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)
                                  : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += v[k];
    }
    tot[i] = t;
  }
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (unsigned int g = 0; g < gridDim.x; ++g)
      t += __hip_atomic_load(reinterpret_cast<const double*>(a.part + static_cast<int64_t>(g) * (kRootMax * kU + kRootMax + 2) + kRootMax * kU + kRootMax),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.ll[0] = t;
    a.ll[1] = static_cast<double>(B);
  }
  __syncthreads();
  if (a.gx == nullptr || a.mode == 0) {
    if (a.gx != nullptr) {  // raw gradients of the linear weights
      for (int i = threadIdx.x; i < R * kU; i += 256) a.dtheta_w[i / kU][i % kU] = tot[i];
      if (a.c != nullptr && threadIdx.x < R) a.dtheta_c[threadIdx.x] = tot[R * kU + threadIdx.x];
    }
    return;
  }
  const ck_opt_state os = a.mode == 2 ? *a.opt : ck_opt_state{};
  const OptK ok = opt_k(os);
  if (a.mode == 2 && os.skip_now) return;
  // softmax behind every weight row (wave w takes folds w, w + 4, ...) ...
  for (int r = wave; r < R; r += 4) {
    const float w = a.w[r][lane], d = tot[r * kU + lane];
    const float s = ck::wave_sum(w * d);
    const float g = w * (d - s);
    if (a.mode == 1) {
      a.dtheta_w[r][lane] = g;
    } else {
      float m1 = os.kind ? a.m1_w[r][lane] : 0.f, m2 = os.kind ? a.m2_w[r][lane] : 0.f;
      const float th = opt_update(ok, a.theta_w[r][lane], g, m1, m2);
      a.theta_w[r][lane] = th;
      if (os.kind) {
        a.m1_w[r][lane] = m1;
        a.m2_w[r][lane] = m2;
      }
      const float mx = ck::wave_max(th);
      const float ex = expf(th - mx);
      const float sum = ck::wave_sum(ex);
      a.w_out[r][lane] = ex / sum;
    }
  }
  // ... and behind the mixing coefficients (wave 0 after its folds: lanes r < R)
  if (a.c != nullptr && wave == 0) {
    const bool on = lane < R;
    const float c = on ? a.c[lane] : 0.f, d = on ? tot[R * kU + lane] : 0.f;
    const float s = ck::wave_sum(c * d);
    const float g = c * (d - s);
    if (a.mode == 1) {
      if (on) a.dtheta_c[lane] = g;
    } else {
      float m1 = (on && os.kind) ? a.m1_c[lane] : 0.f, m2 = (on && os.kind) ? a.m2_c[lane] : 0.f;
      const float th = on ? opt_update(ok, a.theta_c[lane], g, m1, m2) : -INFINITY;
      const float mx = ck::wave_max(th);
      const float ex = on ? expf(th - mx) : 0.f;
      const float sum = ck::wave_sum(ex);
      if (on) {
        a.theta_c[lane] = th;
        if (os.kind) {
          a.m1_c[lane] = m1;
          a.m2_c[lane] = m2;
        }
        a.c_out[lane] = ex / sum;
      }
    }
  }
}

// ---- Categorical input layers ---------------------------------------------------------------------------------------------------
// One fold of a TorchCategoricalLayer with 64 units (layers/input.py:399-412: out[b, :] = log softmax(theta)[:, x_b]), backward in
// ONE workgroup: G = the sum of the fold's gradient blocks; dT[c, :] = sum over the rows with x_b = c of G[b, :] (the scatter-add
// autograd performs for the advanced indexing) -- rows counting-sorted by category in LDS (integer atomics), every category
// owned by one wave, plain row loads eight at a time, no float atomics --; then through the log-softmax:
// d theta[k, c] = dT[c, k] - p[k, c] sum_c' dT[c', k]  (nodes.py:764-783), with p = exp(theta - lse) and lse read off the forward's
// table.  mode 2: the optimizer's update of theta and the next step's (C + 1, 64) table, transposed through LDS.
constexpr int kCatRows = 512;   // rows sorted at a time (LDS: two workgroups per CU at 256 categories)
constexpr int kHS = 65;         // row stride of the (C + 1, 64) histogram in LDS
__global__ void __launch_bounds__(1024)
    jobs_cat_bwd_kernel(const ck_cat_job* __restrict__ jobs, const float* const* __restrict__ pool, const ck_opt_state* __restrict__ opt,
                        int B, int C) {
  extern __shared__ __attribute__((aligned(16))) float cat_smem[];
  const int per = (C + 1 + 15) / 16;  // categories per wave; key(c) = (c % 16) * per + c / 16: a wave's categories are contiguous keys
  const int nkeys = 16 * per;
  float* hist = cat_smem;                                   // [C + 1][kHS]
  float* red = hist + (C + 1) * kHS;                        // [16][64]
  float* tk = red + 16 * kU;                                // [64] column sums, then the new log-normalisers
  float* lse = tk + kU;                                     // [64]
  int* start = reinterpret_cast<int*>(lse + kU);            // [nkeys + 1]
  int* cur = start + nkeys + 1;                             // [nkeys]
  int* order = cur + nkeys;                                 // [kCatRows]
  int* keys = order + kCatRows;                             // [kCatRows]
  const ck_cat_job& J = jobs[blockIdx.x];
  const int cshift = (C & (C - 1)) == 0 ? 31 - __clz(C) : -1;  // entry e = k C + c of theta (64, C): shifts where C is a power of two
  auto unit_of = [&](int e) { return cshift >= 0 ? e >> cshift : e / C; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < (C + 1) * kHS; i += 1024) hist[i] = 0.f;
  const int g_off = J.g_off, n_g = J.n_g;
  for (int b0 = 0; b0 < B; b0 += kCatRows) {
    const int nb = min(kCatRows, B - b0);
    for (int i = threadIdx.x; i <= nkeys; i += 1024) start[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 1024) {
      const int cc = J.x[b0 + i];
      const int c = cc < 0 ? C : min(cc, C - 1);
      atomicAdd(&start[(c & 15) * per + (c >> 4) + 1], 1);
    }
    __syncthreads();
    if (wave == 0) {  // inclusive scan of start[1 .. nkeys]: a run of bins per lane, then a wave scan of the run totals
      const int run_len = (nkeys + 63) / 64;
      int tot = 0;
      for (int j = 0; j < run_len; ++j) {
        const int idx = 1 + lane * run_len + j;
        if (idx <= nkeys) tot += start[idx];
      }
      int incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      int run = incl - tot;
      for (int j = 0; j < run_len; ++j) {
        const int idx = 1 + lane * run_len + j;
        if (idx <= nkeys) {
          run += start[idx];
          start[idx] = run;
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nkeys; i += 1024) cur[i] = start[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 1024) {
      const int cc = J.x[b0 + i];
      const int c = cc < 0 ? C : min(cc, C - 1);
      const int key = (c & 15) * per + (c >> 4);
      const int pos = atomicAdd(&cur[key], 1);
      order[pos] = b0 + i;
      keys[pos] = key;
    }
    __syncthreads();
#ifndef CK_CAT_LAB
#define CK_CAT_LAB 0
#endif
    if (!(CK_CAT_LAB & 1)) {  // this wave's rows: positions [start[wave per], start[(wave + 1) per]), lane = unit
      const int s0 = start[wave * per], s1 = start[(wave + 1) * per];
      int cur_key = -1;
      float acc = 0.f;
      for (int i = s0; i < s1; i += 8) {
        float v[8];
        int64_t row[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = 0.f;
          row[u] = i + u < s1 ? order[i + u] : -1;
        }
        // (four blocks of the list at a time: 32 independent loads in flight -- a loop over the list inside the loop over the rows
        //  is a load, a wait and an add per (row, block): 75 us of this launch at 256 rows and lists of 2 .. 4)
        for (int s2 = 0; s2 < n_g; s2 += 4) {
          const float* gp[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) gp[q] = pool[g_off + min(s2 + q, n_g - 1)];
          float x[8][4];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              x[u][q] = (row[u] >= 0 && s2 + q < n_g) ? *ck::as_global(gp[q] + row[u] * kU + lane) : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = (((v[u] + x[u][0]) + x[u][1]) + x[u][2]) + x[u][3];  // (list order, as before)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (i + u < s1) {
            const int key = keys[i + u];
            if (key != cur_key) {
              if (cur_key >= 0) hist[((cur_key % per) * 16 + cur_key / per) * kHS + lane] += acc;
              acc = 0.f;
              cur_key = key;
            }
            acc += v[u];
          }
        }
      }
      if (cur_key >= 0) hist[((cur_key % per) * 16 + cur_key / per) * kHS + lane] += acc;
    }
    __syncthreads();
  }
  if (CK_CAT_LAB & 2) return;
  // column sums over the C categories (the integral row C carries no parameter) and the forward's log-normalisers
  {
    float t = 0.f;
    for (int c = wave; c < C; c += 16) t += hist[c * kHS + lane];
    red[wave * kU + lane] = t;
  }
  __syncthreads();
  if (threadIdx.x < kU) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w * kU + threadIdx.x];
    tk[threadIdx.x] = t;
    lse[threadIdx.x] = J.theta[static_cast<int64_t>(threadIdx.x) * C] - J.table[threadIdx.x];  // theta[k, 0] - log p[k, 0]
  }
  __syncthreads();
  const int n = kU * C;
  // (four entries per thread and pass, every load of a pass issued before its arithmetic: a pass is one memory round trip)
  if (J.mode != 2) {
    for (int e0 = threadIdx.x; e0 < n; e0 += 4096) {
      float th[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) th[u] = e0 + 1024 * u < n ? *ck::as_global(J.theta + e0 + 1024 * u) : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 1024 * u;
        if (e < n) {
          const int k = unit_of(e), c = e - k * C;
          *ck::as_global(J.dtheta + e) = hist[c * kHS + k] - __expf(th[u] - lse[k]) * tk[k];
        }
      }
    }
    return;
  }
  const ck_opt_state os = *opt;
  const OptK ok = opt_k(os);
  if (os.skip_now) return;
  for (int e0 = threadIdx.x; e0 < n; e0 += 4096) {
    float th[4], a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool on = e0 + 1024 * u < n;
      th[u] = on ? *ck::as_global(J.theta_out + e0 + 1024 * u) : 0.f;
      a[u] = (on && os.kind) ? *ck::as_global(J.m1 + e0 + 1024 * u) : 0.f;
      b[u] = (on && os.kind) ? *ck::as_global(J.m2 + e0 + 1024 * u) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + 1024 * u;
      if (e < n) {
        const int k = unit_of(e), c = e - k * C;
        const float g = hist[c * kHS + k] - __expf(th[u] - lse[k]) * tk[k];
        const float t = opt_update(ok, th[u], g, a[u], b[u]);
        *ck::as_global(J.theta_out + e) = t;
        if (os.kind) {
          *ck::as_global(J.m1 + e) = a[u];
          *ck::as_global(J.m2 + e) = b[u];
        }
        hist[c * kHS + k] = t;  // (this thread's own slot: the updated logits, transposed, for the next table)
      }
    }
  }
  __syncthreads();
  if (CK_CAT_LAB & 4) return;
  {  // log-sum-exp over the categories of every unit: wave w takes c = w, w + 16, ...
    float mx = -INFINITY;
    for (int c = wave; c < C; c += 16) mx = fmaxf(mx, hist[c * kHS + lane]);
    red[wave * kU + lane] = mx;
    __syncthreads();
    if (threadIdx.x < kU) {
      float m2 = red[threadIdx.x];
#pragma unroll
      for (int w = 1; w < 16; ++w) m2 = fmaxf(m2, red[w * kU + threadIdx.x]);
      tk[threadIdx.x] = m2;
    }
    __syncthreads();
    float sm = 0.f;
    for (int c = wave; c < C; c += 16) sm += __expf(hist[c * kHS + lane] - tk[lane]);
    __syncthreads();
    red[wave * kU + lane] = sm;
    __syncthreads();
    if (threadIdx.x < kU) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += red[w * kU + threadIdx.x];
      lse[threadIdx.x] = tk[threadIdx.x] + logf(t);
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < n; e += 1024) {  // table[c, k] = log softmax(theta')[k, c]
    const int c = e >> 6, k = e & 63;
    J.table_out[e] = hist[c * kHS + k] - lse[k];
  }
}

// ---- Gaussian input layers -------------------------------------------------------------------------------------------------------
// One fold of a TorchGaussianLayer (layers/input.py:661-670): log N(x; mu_k, sigma_k) per unit k.  With G the sum of the fold's
// gradient blocks:  d mu_k = sum_b G[b, k] (x_b - mu_k) / sigma_k^2,   d sigma_k = sum_b G[b, k] ((x_b - mu_k)^2 / sigma_k^2 - 1) / sigma_k
// (a NaN x_b -- a marginalised variable -- contributes nothing), then through the scaled sigmoid behind sigma
// (nodes.py:698-699: sigma = vmin + (vmax - vmin) sigmoid(theta), d theta = d sigma (sigma - vmin)(vmax - sigma) / (vmax - vmin)).
__global__ void __launch_bounds__(256)
    jobs_gauss_bwd_kernel(const ck_gauss_job* __restrict__ jobs, const float* const* __restrict__ pool, const ck_opt_state* __restrict__ opt,
                          int B) {
  __shared__ float red[4][2][kU];
  const ck_gauss_job& J = jobs[blockIdx.x];
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const float4 mu = ck::gload4(J.mean + 4 * c), sd = ck::gload4(J.stddev + 4 * c);
  const float4 is = make_float4(1.f / sd.x, 1.f / sd.y, 1.f / sd.z, 1.f / sd.w);
  float4 dm = make_float4(0.f, 0.f, 0.f, 0.f), ds = dm;
  const int g_off = J.g_off, n_g = J.n_g;
  for (int b = r; b < B; b += 16) {
    const float x = J.x[b];
    float4 g = ck::gload4(pool[g_off] + static_cast<int64_t>(b) * kU + 4 * c);
    for (int s2 = 1; s2 < n_g; ++s2) {
      const float4 t = ck::gload4(pool[g_off + s2] + static_cast<int64_t>(b) * kU + 4 * c);
      g.x += t.x, g.y += t.y, g.z += t.z, g.w += t.w;
    }
    if (x == x) {
      const float4 z = make_float4((x - mu.x) * is.x, (x - mu.y) * is.y, (x - mu.z) * is.z, (x - mu.w) * is.w);
      dm.x = fmaf(g.x, z.x * is.x, dm.x), dm.y = fmaf(g.y, z.y * is.y, dm.y), dm.z = fmaf(g.z, z.z * is.z, dm.z), dm.w = fmaf(g.w, z.w * is.w, dm.w);
      ds.x = fmaf(g.x, (z.x * z.x - 1.f) * is.x, ds.x), ds.y = fmaf(g.y, (z.y * z.y - 1.f) * is.y, ds.y);
      ds.z = fmaf(g.z, (z.z * z.z - 1.f) * is.z, ds.z), ds.w = fmaf(g.w, (z.w * z.w - 1.f) * is.w, ds.w);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[8] = {dm.x, dm.y, dm.z, dm.w, ds.x, ds.y, ds.z, ds.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] += __shfl_xor(v[k], 16, 64);
    v[k] += __shfl_xor(v[k], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[wave][0][4 * c + k] = v[k];
      red[wave][1][4 * c + k] = v[4 + k];
    }
  }
  __syncthreads();
  if (threadIdx.x >= 2 * kU) return;
  const int which = threadIdx.x >> 6, k = threadIdx.x & 63;  // wave 0: the means, wave 1: the standard deviations
  float g = (red[0][which][k] + red[1][which][k]) + (red[2][which][k] + red[3][which][k]);
  const float sdk = J.stddev[k];
  if (which == 1 && J.has_ss) g *= (sdk - J.vmin) * (J.vmax - sdk) / (J.vmax - J.vmin);
  float* dst = which == 0 ? J.dmean : J.dsd;
  if (J.mode != 2) {
    dst[k] = g;
    return;
  }
  const ck_opt_state os = *opt;
  const OptK ok = opt_k(os);
  if (os.skip_now) return;
  float* th = which == 0 ? J.th_mean : J.th_sd;
  float* m1 = which == 0 ? J.m1_mean : J.m1_sd;
  float* m2 = which == 0 ? J.m2_mean : J.m2_sd;
  float a = os.kind ? m1[k] : 0.f, b2 = os.kind ? m2[k] : 0.f;
  const float t = opt_update(ok, th[k], g, a, b2);
  th[k] = t;
  if (os.kind) {
    m1[k] = a;
    m2[k] = b2;
  }
  if (which == 0) {
    if (J.mean_out != nullptr) J.mean_out[k] = t;
  } else {
    J.sd_out[k] = J.has_ss ? J.vmin + (J.vmax - J.vmin) / (1.f + expf(-t)) : t;
  }
}

// The optimizer step on one flat range with the constants and clock of a DEVICE ck_opt_state (recordable: nothing of the step
// count is baked into the launch) -- the tensors no job epilogue updates.
__global__ void __launch_bounds__(256)
    opt_range_kernel(float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ g2, float* __restrict__ m1,
                     float* __restrict__ m2, int64_t n, const ck_opt_state* __restrict__ opt) {
  const ck_opt_state os = *opt;
  const OptK ok = opt_k(os);
  if (os.skip_now) return;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    float a = os.kind ? m1[i] : 0.f, b = os.kind ? m2[i] : 0.f;
    p[i] = opt_update(ok, p[i], g2 != nullptr ? g[i] + g2[i] : g[i], a, b);
    if (os.kind) {
      m1[i] = a;
      m2[i] = b;
    }
  }
}

// The optimizer's clock, once per step on the device: a batch with an illegal category (the circuit's flag) makes this step's
// updates no-ops -- parameters, moments and the bias corrections' step count stay -- and the flag is latched for `check_inputs`.
__global__ void opt_tick_kernel(ck_opt_state* __restrict__ o, int32_t* __restrict__ flag, int32_t* __restrict__ sticky) {
  const bool bad = flag != nullptr && *flag != 0;
  if (bad) {
    if (sticky != nullptr) *sticky |= *flag;
    *flag = 0;
    o->skip_now = 1;
    o->skipped += 1;
    return;
  }
  o->skip_now = 0;
  o->step += 1;
  // in double from the caller's double betas, as torch.optim.Adam and ck_adam_step's host side do (1 - 0.999f^1 in fp32 is
  // off by 1.3e-5 relative: the fused and the separate optimizer would disagree in the first steps)
  const double t = static_cast<double>(o->step);
  o->bc1 = static_cast<float>(-expm1(t * log(o->b1d)));
  o->bc2 = static_cast<float>(-expm1(t * log(o->b2d)));
}

}  // namespace

extern "C" {

int ck_jobs_sum64_fwd(const ck_sum_job* jobs, int n_units, const float* const* pool, void* stream) {
  CK_REQUIRE(jobs && pool && n_units > 0, "ck_jobs_sum64_fwd: bad arguments");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(jobs_sum64_fwd_kernel<4>, dim3(n_units), dim3(256), 0, s, jobs, pool);
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_sum64_bwd(const ck_sum_job* jobs, int n_units, const float* const* pool, const ck_opt_state* opt, int waves, void* stream) {
  CK_REQUIRE(jobs && pool && n_units > 0, "ck_jobs_sum64_bwd: bad arguments");
  CK_REQUIRE(waves == 4 || waves == 8, "ck_jobs_sum64_bwd: 4 or 8 waves per workgroup");
  return ck::dispatch(
      [=](hipStream_t s) {
        if (waves == 8) {
          hipLaunchKernelGGL(jobs_sum64_bwd_kernel<8>, dim3(n_units), dim3(512), 0, s, jobs, pool, opt);
        } else {
          hipLaunchKernelGGL(jobs_sum64_bwd_kernel<4>, dim3(n_units), dim3(256), 0, s, jobs, pool, opt);
        }
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_mix_fwd(const ck_mix_job* jobs, int n_units, const float* const* pool, int h_max, void* stream) {
  CK_REQUIRE(jobs && pool && n_units > 0, "ck_jobs_mix_fwd: bad arguments");
  CK_REQUIRE(h_max >= 1 && h_max <= 16, "ck_jobs_mix_fwd: at most 16 slots per mixing job");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(jobs_mix_fwd_kernel, dim3(n_units), dim3(256), 0, s, jobs, pool);
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_mix_params(const ck_mix_job* jobs, int n_jobs, const ck_opt_state* opt, void* stream) {
  CK_REQUIRE(jobs && n_jobs > 0, "ck_jobs_mix_params: bad arguments");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(jobs_mix_params_kernel, dim3(n_jobs), dim3(64), 0, s, jobs, opt);
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_mix_bwd(const ck_mix_job* jobs, int n_units, const float* const* pool, int h_max, int64_t gx_stride,
                    const ck_opt_state* opt, void* stream) {
  CK_REQUIRE(jobs && pool && n_units > 0 && gx_stride > 0, "ck_jobs_mix_bwd: bad arguments");
  CK_REQUIRE(h_max >= 1 && h_max <= 16, "ck_jobs_mix_bwd: at most 16 slots per mixing job");
  return ck::dispatch(
      [=](hipStream_t s) {
        if (h_max <= 2)
          hipLaunchKernelGGL(jobs_mix_bwd_kernel<2>, dim3(n_units), dim3(256), 0, s, jobs, pool, opt, gx_stride);
        else if (h_max <= 4)
          hipLaunchKernelGGL(jobs_mix_bwd_kernel<4>, dim3(n_units), dim3(256), 0, s, jobs, pool, opt, gx_stride);
        else if (h_max <= 8)
          hipLaunchKernelGGL(jobs_mix_bwd_kernel<8>, dim3(n_units), dim3(256), 0, s, jobs, pool, opt, gx_stride);
        else
          hipLaunchKernelGGL(jobs_mix_bwd_kernel<16>, dim3(n_units), dim3(256), 0, s, jobs, pool, opt, gx_stride);
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_nsum(const ck_nsum_job* jobs, int n_jobs, const float* const* pool, int64_t elems, void* stream) {
  CK_REQUIRE(jobs && pool && n_jobs > 0 && n_jobs <= 65535 && elems > 0 && elems % 4 == 0, "ck_jobs_nsum: bad arguments");
  const unsigned gx = static_cast<unsigned>(std::min<int64_t>((elems / 4 + 255) / 256, 64));
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(jobs_nsum_kernel, dim3(gx, n_jobs), dim3(256), 0, s, jobs, pool, elems);
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_root(const ck_root_launch* a, void* stream) {
  CK_REQUIRE(a && a->pool && a->in_off && a->n_in && a->w && a->out && a->ll && a->part && a->ticket, "ck_jobs_root: null pointer");
  CK_REQUIRE(a->R >= 1 && a->R <= kRootMax && a->B > 0, "ck_jobs_root: 1..16 scalar folds");
  CK_REQUIRE(a->c != nullptr || a->R == 1, "ck_jobs_root: several scalar folds need the final mixing coefficients");
  CK_REQUIRE(a->n_wg >= 1 && a->n_wg <= 64, "ck_jobs_root: 1..64 workgroups");
  CK_REQUIRE(a->S >= 1 && a->S <= 4, "ck_jobs_root: 1..4 blocks per scalar fold");
  CK_REQUIRE(a->gx == nullptr || a->mode == 0 || a->mode == 1 || (a->mode == 2 && a->opt), "ck_jobs_root: bad mode");
  const ck_root_launch v = *a;
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(jobs_root_kernel, dim3(v.n_wg), dim3(256), 0, s, v);
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_cat_bwd(const ck_cat_job* jobs, int n_jobs, const float* const* pool, int B, int C, const ck_opt_state* opt, void* stream) {
  CK_REQUIRE(jobs && pool && n_jobs > 0 && B > 0 && C > 0, "ck_jobs_cat_bwd: bad arguments");
  const int per = (C + 1 + 15) / 16, nkeys = 16 * per;
  const size_t lds = (static_cast<size_t>(C + 1) * kHS + 16 * kU + 2 * kU) * sizeof(float) + (2 * nkeys + 1 + 2 * kCatRows) * sizeof(int);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_jobs_cat_bwd: %d categories do not fit one workgroup's LDS", C);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(jobs_cat_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(jobs_cat_bwd_kernel, dim3(n_jobs), dim3(1024), lds, s, jobs, pool, opt, B, C);
        return hipGetLastError();
      },
      stream);
}

int ck_jobs_gauss_bwd(const ck_gauss_job* jobs, int n_jobs, const float* const* pool, int B, const ck_opt_state* opt, void* stream) {
  CK_REQUIRE(jobs && pool && n_jobs > 0 && B > 0, "ck_jobs_gauss_bwd: bad arguments");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(jobs_gauss_bwd_kernel, dim3(n_jobs), dim3(256), 0, s, jobs, pool, opt, B);
        return hipGetLastError();
      },
      stream);
}

int ck_opt_step_range(float* p, const float* g, const float* g2, float* m1, float* m2, int64_t n, const ck_opt_state* opt, void* stream) {
  CK_REQUIRE(p && g && opt && n > 0, "ck_opt_step_range: bad arguments");
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 2048));
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(opt_range_kernel, dim3(grid), dim3(256), 0, s, p, g, g2, m1, m2, n, opt);
        return hipGetLastError();
      },
      stream);
}

int ck_opt_tick(ck_opt_state* state, int32_t* flag, int32_t* sticky, void* stream) {
  CK_REQUIRE(state, "ck_opt_tick: null state");
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(1), 0, s, state, flag, sticky);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
