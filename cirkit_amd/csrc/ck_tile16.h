// The 16-row x 32-unit register tile (v_mfma_f32_16x16x4_f32): half the rows of ck_tile.h's tile per wavefront, so twice
// the wavefronts for the same batch -- what latency-bound launches want (the tail of a circuit: a handful of folds per
// level; the fused leaf region at batch 4096: 24.5 tiles of 32 rows per CU).
//
// Lane l of a wave: b = l & 15 (batch row of the 16-row tile), kq = l >> 4.  Lane (b, kq) holds units
// 16 beta + 4 kq + r (beta in 0..1, r in 0..3) of row b in register j = 4 beta + r.  With the WEIGHTS as the MFMA A
// operand and the activations as B, contraction step s (0..7) of output block beta' uses
//     A: lane (o', kq) = W[16 beta' + o'][16 (s >> 2) + 4 kq + (s & 3)]        B: register s of lane (b, kq)
// and D[o][b] arrives in lane (b, kq') register r as o = 16 beta' + 4 kq' + r: the OUTPUT layout of a step is the
// INPUT layout of the next, as for the 32-row tile.  The two output blocks are independent accumulators, issued
// alternately: 16 MFMAs x 32 cycles with the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 covered.
// Exact fp32 (an fmaf chain over k); the order of the 32 products differs from the 32-row tile's
// (units 0,4,8,12, 1,5,9,13, ... instead of 0,4,1,5, ...), so the two agree to fp32 rounding, not bit for bit.
//
// Weights of one fold in registers: 16 floats per lane = float4 (beta', g) = W[16 beta' + o'][16 g + 4 kq .. + 3], g = s >> 2.
// In LDS ("TILE16" order): float4 index (beta' * 2 + g) * 64 + lane -- one contiguous KiB per wave read.
#pragma once

#include "ck_internal.h"
#include "ck_tile.h"

namespace {

struct WRegs16 {
  float4 q[4];  // q[beta' * 2 + g]
};

// offset (in floats) of the float4 (beta', g) of lane (o', kq) inside one fold's 1024-float weight block
template <int LAYOUT>
__device__ __forceinline__ int w16_offset(int lane, int bg) {
  const int op = lane & 15, kq = lane >> 4, beta = bg >> 1, g = bg & 1;
  if constexpr (LAYOUT == CK_W_ROWMAJOR) {
    return (16 * beta + op) * kK + 16 * g + 4 * kq;
  } else {  // CK_W_TILED_F32: dword (q, lane', t) = W[lane' & 31][8 q + 4 (lane' >> 5) + t]
    return (2 * g + (kq >> 1)) * 256 + (16 * beta + op + 32 * (kq & 1)) * 4;
  }
}
template <int LAYOUT>
__device__ __forceinline__ void load_w16(const float* __restrict__ wf, int lane, WRegs16& w) {
#pragma unroll
  for (int bg = 0; bg < 4; ++bg) w.q[bg] = ck::gload4(wf + w16_offset<LAYOUT>(lane, bg));  // (device memory)
}
__device__ __forceinline__ float w16_elem(const WRegs16& w, int beta, int s) {
  const float4& q = w.q[beta * 2 + (s >> 2)];
  return (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
}

__device__ __forceinline__ void tile16_load(const float* __restrict__ row_kq, float (&v)[8]);

// maximum of a value over the four lanes (b, 0..3) that hold one row (v_permlane16_swap / v_permlane32_swap: no LDS)
__device__ __forceinline__ float xquad_max(float m) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  return ck::xhalf_max(m);
}
__device__ __forceinline__ float xquad_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float row_max8(const float (&v)[8]) {
  const float m0 = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(v[3], v[4]), v[5]);
  return xquad_max(__builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(v[6], v[7])));
}

// e <- W . e in linear space on the 16-row tile
__device__ __forceinline__ void contract16(const WRegs16& w, float (&e)[8]) {
  f32x4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w16_elem(w, 0, s), e[s], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w16_elem(w, 1, s), e[s], a1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    e[r] = a0[r];
    e[4 + r] = a1[r];
  }
}

// One log-einsum-exp step  v <- log(W . exp(v - max v)) + max v  (semiring.py:383-408) on the 16-row tile
__device__ __forceinline__ void sum_step16(const WRegs16& w, float (&v)[8]) {
  const float m = ck::clamp_finite(row_max8(v));  // torch.clamp(amax, finfo.min, finfo.max), semiring.py:392-399
  const float nml = exp_offset(m, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
  contract16(w, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = fmaf(__builtin_amdgcn_logf(v[j]), kLN2, m);
}

// The same for SIGNED values (log|v|, sign bit j of sg): exp with the sign, contraction, log|y| and the sign of y
__device__ __forceinline__ void sum_step16_signed(const WRegs16& w, float (&v)[8], uint32_t& sg) {
  const float m = ck::clamp_finite(row_max8(v));
  const float nml = exp_offset(m, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float e = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
    v[j] = (sg >> j) & 1u ? -e : e;
  }
  contract16(w, v);
  sg = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sg |= (v[j] < 0.f ? 1u : 0u) << j;
    v[j] = fmaf(__builtin_amdgcn_logf(__builtin_fabsf(v[j])), kLN2, m);
  }
}

// the tile as complex logarithms (v[j], pi if bit j of sg else 0) into a (B, 32) complex64 block; row_kq points at the
// lane's first complex element (float offset 2 (32 b + 4 kq))
__device__ __forceinline__ void tile16_store_clog(float* __restrict__ row_kq, const float (&v)[8], uint32_t sg) {
  constexpr float kPi = 3.14159265358979323846f;
#pragma unroll
  for (int beta = 0; beta < 2; ++beta) {
    const float p0 = (sg >> (4 * beta)) & 1u ? kPi : 0.f, p1 = (sg >> (4 * beta + 1)) & 1u ? kPi : 0.f;
    const float p2 = (sg >> (4 * beta + 2)) & 1u ? kPi : 0.f, p3 = (sg >> (4 * beta + 3)) & 1u ? kPi : 0.f;
    ck::gstore4(row_kq + 32 * beta, make_float4(v[4 * beta + 0], p0, v[4 * beta + 1], p1));
    ck::gstore4(row_kq + 32 * beta + 4, make_float4(v[4 * beta + 2], p2, v[4 * beta + 3], p3));
  }
}

// (B, 32) blocks <-> register tile: lane (b, kq) moves 2 x 16 bytes of row b (units 16 beta + 4 kq ..)
__device__ __forceinline__ void tile16_load(const float* __restrict__ row_kq, float (&v)[8]) {
#pragma unroll
  for (int beta = 0; beta < 2; ++beta) {
    const float4 t4 = ck::gload4(row_kq + 16 * beta);
    v[4 * beta + 0] = t4.x;
    v[4 * beta + 1] = t4.y;
    v[4 * beta + 2] = t4.z;
    v[4 * beta + 3] = t4.w;
  }
}
__device__ __forceinline__ void tile16_load_add(const float* __restrict__ row_kq, float (&v)[8]) {
#pragma unroll
  for (int beta = 0; beta < 2; ++beta) {
    const float4 t4 = ck::gload4(row_kq + 16 * beta);
    v[4 * beta + 0] += t4.x;
    v[4 * beta + 1] += t4.y;
    v[4 * beta + 2] += t4.z;
    v[4 * beta + 3] += t4.w;
  }
}
__device__ __forceinline__ void tile16_store(float* __restrict__ row_kq, const float (&v)[8]) {
#pragma unroll
  for (int beta = 0; beta < 2; ++beta)
    ck::gstore4(row_kq + 16 * beta, make_float4(v[4 * beta + 0], v[4 * beta + 1], v[4 * beta + 2], v[4 * beta + 3]));
}

}  // namespace
