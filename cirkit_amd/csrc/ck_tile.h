// The 32-row x 32-unit register tile shared by sum_lse_tile32 (ck_sum.hip), cp_lse_kernel (ck_cp.hip), the fused leaf kernel
// (ck_fused.hip) and the fused tail (ck_tail.hip).
//
// Lane l of a wave: b = l & 31 (batch row of the 32-row tile), kh = l >> 5.  Lane (b, kh) holds
// units 8g + 4kh + t (g, t in 0..3) of row b in register j = 4g + t.  v_mfma_f32_32x32x2_f32 returns D[o][b] in lane
// (b, hi) register r with o = 8(r>>2) + 4hi + (r&3): the OUTPUT layout of a step is the INPUT layout of the next.
//
// Weight layouts of one fold (32 outputs x 32 inputs = 1024 dwords):
//   CK_W_ROWMAJOR    W[o][i] fp32 -- the reference's layout; lane (o, kh) reads 4 x 16 B at stride 32 B
//   CK_W_TILED_F32   dword (q, lane, t) = W[o = lane&31][8q + 4(lane>>5) + t]: every wave load
//                    instruction reads one contiguous KiB (8 lines instead of 32)
// The tiled layout is produced by the softmax prologue (ck_param.hip, kind 2).
#pragma once

#include <type_traits>
#include <utility>

#include "ck_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kK = 32;

struct WRegs {
  float4 q[4];
};

template <int LAYOUT>
__device__ __forceinline__ void load_w(const float* __restrict__ wf, int lane, WRegs& w) {
  if constexpr (LAYOUT == CK_W_ROWMAJOR) {
    const float* wrow = wf + (lane & 31) * kK + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) w.q[g] = *reinterpret_cast<const float4*>(wrow + 8 * g);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) w.q[q] = *reinterpret_cast<const float4*>(wf + q * 256 + lane * 4);
  }
}

__device__ __forceinline__ float row_max16(const float (&v)[16]) {
  float m = v[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) m = fmaxf(m, v[j]);
  m = ck::xhalf_max(m);
  return ck::clamp_finite(m);  // torch.clamp(amax, finfo.min, finfo.max), semiring.py:392-399
}

constexpr float kL2E = 1.44269504088896340736f;
constexpr float kLN2 = 0.69314718055994530942f;

// exp(v - m) is evaluated as exp2(fma(v, log2 e, c)) with c = -m log2 e (+ a scaling exponent).
// When the row maximum was clamped from -inf / +inf to -+FLT_MAX, -m log2 e overflows and the FMA
// would produce NaN where the reference gets exp(-inf - m) = 0 (an all -inf row must come out as
// -inf, semiring.py:392-408); c = `shift` alone keeps those limits.
__device__ __forceinline__ float exp_offset(float m, float shift) {
  return fabsf(m) > 1e38f ? shift : fmaf(-m, kL2E, shift);
}

// One log-einsum-exp step  v <- log(W . exp(v - max v)) + max v  on a register tile.
//
// LAYOUT ROWMAJOR / TILED_F32: exact fp32 contraction on v_mfma_f32_32x32x2_f32 (an fmaf chain).
//   Measured on MI355X: this MFMA shares the fp32 ALUs with the VALU, the two never
//   co-execute, so its 16 x 64 cycles add to the exp/log work.
// (A split-fp16 variant -- three v_mfma_f32_32x32x16_f16 products per step, ~22-bit significand -- lived here until round 3:
//  0.154 ms per step against 0.105 for this exact path, and its low parts underflowed on small weights.  Removed.)
template <int LAYOUT>
__device__ __forceinline__ void sum_step(const WRegs& w, float (&v)[16]) {
  const float m = row_max16(v);
  {
    const float nml = exp_offset(m, 0.f);
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].x, v[4 * g + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].y, v[4 * g + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].z, v[4 * g + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].w, v[4 * g + 3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m);
  }
}

// e <- W . e in LINEAR space (e in [0, 1]): the contraction of `sum_step` without the exp before and the
// log after it.  Used by kernels that carry a value between fused levels as (linear tile, per-row log
// scale) instead of going through log and exp again (ck_fused.hip).
template <int LAYOUT>
__device__ __forceinline__ void contract_linear(const WRegs& w, float (&e)[16]) {
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].x, e[4 * g + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].y, e[4 * g + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].z, e[4 * g + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].w, e[4 * g + 3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = acc[r];
  }
}

// e <- W . e with the fp32 operands SPLIT into bf16 pieces and contracted on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16,
// fp32 accumulation) -- a labelled VARIANT of contract_linear, never the default: the exact path is an fp32 fmaf chain on the
// fp32 lanes (16 x 64 cycles, shared with every other vector instruction), the bf16 pipe is separate hardware at 16x the rate.
// bf16 keeps fp32's exponent, so the pieces of a small weight do not underflow (what killed the split-fp16 variant of
// rounds 1-2).  A value is cut by TRUNCATION into h + m (+ l): 8 significant bits each, residuals exact.
//   PIECES 2 ("bf16x3"): W_h e_h + W_h e_m + W_m e_h          dropped terms <= 2^-15 of a product
//   PIECES 3 ("bf16x6"): + W_h e_l + W_m e_m + W_l e_h         dropped terms <= 2^-23: fp32-like
// wnode: the node's weight pieces in LDS -- dword ((p * 2 + m) * 64 + lane) * 4 + d holds, as two bf16, piece p of
// W[lane & 31][u(8m + 2d, lane >> 5)] and of W[lane & 31][u(8m + 2d + 1, lane >> 5)], u(j, kh) = 8 (j >> 2) + 4 kh + (j & 3): the A
// operand of half m (units of registers 8m .. 8m + 7), laid out so that one ds_read_b128 per (piece, half) fetches it.
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
template <int PIECES>
__device__ __forceinline__ void contract_bf16(const float* wnode, int lane, float (&e)[16]) {
  u32x4v ep[PIECES][2];
  float r[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) r[j] = e[j];
#pragma unroll
  for (int p = 0; p < PIECES; ++p) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int d = 0; d < 4; ++d)  // the high halves of two residuals, the first in the low 16 bits
        ep[p][m][d] = __builtin_amdgcn_perm(__float_as_uint(r[8 * m + 2 * d + 1]), __float_as_uint(r[8 * m + 2 * d]), 0x07060302u);
    if (p + 1 < PIECES) {
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] -= __uint_as_float(__float_as_uint(r[j]) & 0xffff0000u);  // exact
    }
  }
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const u32x4v* wp = reinterpret_cast<const u32x4v*>(wnode);
  auto mm = [&](u32x4v a, u32x4v b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, a), __builtin_bit_cast(bf16x8v, b), acc, 0, 0, 0);
  };
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    u32x4v w[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) w[p] = wp[(p * 2 + m) * 64 + lane];
    if constexpr (PIECES == 3) {  // (smallest terms first)
      mm(w[2], ep[0][m]);
      mm(w[1], ep[1][m]);
      mm(w[0], ep[2][m]);
    }
    mm(w[1], ep[0][m]);
    mm(w[0], ep[1][m]);
    mm(w[0], ep[0][m]);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) e[k] = acc[k];
}
// Where piece p of W[o][i] goes in that layout (halfword index inside the node's block).
__device__ __forceinline__ int bf16_piece_index(int p, int o, int i) {
  const int g = i >> 3, kh = (i >> 2) & 1, t = i & 3;
  return (((p * 2 + (g >> 1)) * 64 + o + 32 * kh) * 4 + (2 * (g & 1) + (t >> 1))) * 2 + (t & 1);
}

// ---- linear-domain chaining of fused CP-T levels (ck_fused.hip, ck_leaf.hip) -------------------------------------
// A node is carried as (y: linear tile, s: per-row log scale), value = log y + s.  A level forms e = y_l * y_r
// (linear_product) and contracts y = W e (contract_linear).  What this costs matters: fp32-input MFMA and the VALU
// share the fp32 lanes of a SIMD, so every VALU instruction of a step ADDS ~4 cycles to its 1024 cycles of MFMA
// (measured, scripts/ubench/leaf_step.hip: the former max + division + log step ran at 1.45x the MFMA time whatever
// the number of waves).  Hence:
//   * RESCALE = false (the first fused level, 8 of the 15 nodes of a depth-4 subtree): the bare product, 16 multiplies;
//   * RESCALE = true: the product is renormalised by a POWER OF TWO, 2^-k with k the exponent of the row maximum --
//     exact (no rounding at all, unlike a division by the maximum), s += k ln 2; v_frexp_exp + v_ldexp instead of an
//     IEEE division and a log;
//   * no branch per step: `bad` collects rows whose maximum fell below 2^-80 ~ 8e-25 (children with disjoint large
//     units; also NaN); the caller then redoes the whole tile in log space (tile_walk_logspace), exactly the
//     reference's arithmetic (semiring.py:383-408).  An unscaled level cannot hide an underflow from the next
//     rescaled level or from the final check of the root tile: a row maximum above 2^-80 there means the terms lost
//     below it (< 2^-126) were smaller than 2^-46 of it.
constexpr float kLinearFloor = 8.2718061e-25f;  // 2^-80

__device__ __forceinline__ float tile_row_max(const float (&p)[16]) {
  const float m0 = __builtin_fmaxf(__builtin_fmaxf(p[0], p[1]), p[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(p[3], p[4]), p[5]);
  const float m2 = __builtin_fmaxf(__builtin_fmaxf(p[6], p[7]), p[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(p[9], p[10]), p[11]);
  const float m4 = __builtin_fmaxf(__builtin_fmaxf(p[12], p[13]), p[14]);
  const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), p[15]));
  return ck::xhalf_max(m);
}

// the same of |p| (SIGNED tiles: a real-valued circuit under complex-lse-sum carries signed linear values; the sign is
// the phase 0 / pi of the reference's complex logarithm, semiring.py:441-476)
__device__ __forceinline__ float tile_row_max_abs(const float (&p)[16]) {
  float m = __builtin_fabsf(p[0]);
#pragma unroll
  for (int j = 1; j < 16; j += 3) m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(p[j])), __builtin_fmaxf(__builtin_fabsf(p[j + 1]), __builtin_fabsf(p[j + 2])));
  return ck::xhalf_max(m);
}

// cur <- cur * sib as eight packed multiplies (v_pk_mul_f32: two products per VALU instruction -- every VALU
// instruction of a step adds to its MFMA time, see above)
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tile_mul(float (&cur)[16], const float (&sib)[16]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x2v a = {cur[2 * j], cur[2 * j + 1]};
    const f32x2v b = {sib[2 * j], sib[2 * j + 1]};
    a = a * b;
    cur[2 * j] = a.x;
    cur[2 * j + 1] = a.y;
  }
}
__device__ __forceinline__ void tile_scale(float (&cur)[16], float sc) {
  // (a multiply by a splat is scalarised by the compiler, so the second half is laundered: it cannot tell; spelling
  // the instruction out in inline asm is WRONG here -- the hazard recogniser does not see inside an asm block and omits
  // the wait states between a VALU write and the MFMA that reads it)
  float sc2 = sc;
  asm volatile("" : "+v"(sc2));
  const f32x2v b = {sc, sc2};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x2v a = {cur[2 * j], cur[2 * j + 1]};
    a = a * b;
    cur[2 * j] = a.x;
    cur[2 * j + 1] = a.y;
  }
}

// cur <- cur * sib (renormalised), s <- s_cur + s_sib (+ k ln 2)
template <bool RESCALE, bool SIGNED = false>
__device__ __forceinline__ void linear_product(float (&cur)[16], const float (&sib)[16], float& s, float s_sib, bool& bad) {
  tile_mul(cur, sib);
  s += s_sib;
  if constexpr (RESCALE) {
    const float mx = SIGNED ? tile_row_max_abs(cur) : tile_row_max(cur);
    const int k = __builtin_amdgcn_frexp_expf(mx);  // mx = f 2^k, f in [0.5, 1)
    const float sc = __builtin_amdgcn_ldexpf(1.f, -k);
    bad |= !(mx > kLinearFloor);
    tile_scale(cur, sc);
    s = fmaf(static_cast<float>(k), kLN2, s);
  }
}

// LDS reads of buffers that global_load_lds_dwordx4 also writes, spelled out: for a plain C++ load the compiler puts
// s_waitcnt vmcnt(0) in front -- it cannot tell the read from the DMA in flight -- i.e. it waits for every request the wave
// has just issued (ck_cp.hip, ck_gemm.hip, ck_leaf.hip).
__device__ __forceinline__ void lds_read4(f32x4v& a, f32x4v& b, f32x4v& c, f32x4v& d, uint32_t a0, uint32_t a1,
                                          uint32_t a2, uint32_t a3) {
  asm volatile(
      "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
      : "memory");
}
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void lds_read4_off(f32x4v& a, f32x4v& b, f32x4v& c, f32x4v& d, uint32_t base) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t"
      "ds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(base), "n"(O0), "n"(O1), "n"(O2), "n"(O3)
      : "memory");
}

// compile-time loop: the walks below are fully unrolled (the sibling stack and the step order are static)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// CP-T steps that follow leaf i in the depth-first walk (= trailing one bits of i) and the steps before leaf i
__host__ __device__ constexpr int steps_after(int i) {
  int n = 0;
  while (i & 1) {
    ++n;
    i >>= 1;
  }
  return n;
}
__host__ __device__ constexpr int steps_before(int i) {
  int n = 0;
  for (int j = 0; j < i; ++j) n += steps_after(j);
  return n;
}

// The depth-first walk of one subtree tile entirely in LOG space (the fallback of the linear chain): leaves are
// log(table row) + scale, a level adds the siblings and applies sum_step -- the reference's arithmetic.
//   leaf(ic, v):               fills v with the LINEAR table row of leaf ic.value, returns its log scale
//   weights(sc, ic, lc, w):    loads the weights of contraction number sc.value = level lc.value + 1 after leaf ic.value
template <int D, int LAYOUT, class LeafFn, class WFn>
__device__ __forceinline__ void tile_walk_logspace(LeafFn&& leaf, WFn&& weights, float (&cur)[16]) {
  float stack[D][16];
  static_for<0, (1 << D)>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const float s = leaf(ic, cur);
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] = logf(cur[j]) + s;
    static_for<0, steps_after(i)>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
#pragma unroll
      for (int j = 0; j < 16; ++j) cur[j] += stack[l][j];
      WRegs w;
      weights(std::integral_constant<int, steps_before(i) + l>{}, ic, lc, w);
      sum_step<LAYOUT>(w, cur);
    });
    if constexpr (steps_after(i) < D) {
#pragma unroll
      for (int j = 0; j < 16; ++j) stack[steps_after(i)][j] = cur[j];
    }
  });
}

// The same walk for SIGNED values: a node is (log|v|, sign) -- the reference's complex logarithm of a real number,
// (log|v|, 0 or pi) -- `sg` holds the 16 sign bits of a lane's registers.  A product adds the logarithms and xors the
// signs; a sum step exponentiates with the sign, contracts, and takes log|y| and the sign of y.
template <int LAYOUT>
__device__ __forceinline__ void sum_step_signed(const WRegs& w, float (&v)[16], uint32_t& sg) {
  const float m = row_max16(v);
  const float nml = exp_offset(m, 0.f);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float e = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
    v[j] = (sg >> j) & 1u ? -e : e;
  }
  contract_linear<LAYOUT>(w, v);
  sg = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sg |= (v[j] < 0.f ? 1u : 0u) << j;
    v[j] = fmaf(__builtin_amdgcn_logf(__builtin_fabsf(v[j])), kLN2, m);
  }
}
template <int D, int LAYOUT, class LeafFn, class WFn>
__device__ __forceinline__ void tile_walk_logspace_signed(LeafFn&& leaf, WFn&& weights, float (&cur)[16], uint32_t& sg) {
  float stack[D][16];
  uint32_t sstack[D];
  static_for<0, (1 << D)>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const float s = leaf(ic, cur);
    sg = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      sg |= (cur[j] < 0.f ? 1u : 0u) << j;
      cur[j] = logf(__builtin_fabsf(cur[j])) + s;
    }
    static_for<0, steps_after(i)>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
#pragma unroll
      for (int j = 0; j < 16; ++j) cur[j] += stack[l][j];
      sg ^= sstack[l];
      WRegs w;
      weights(std::integral_constant<int, steps_before(i) + l>{}, ic, lc, w);
      sum_step_signed<LAYOUT>(w, cur, sg);
    });
    if constexpr (steps_after(i) < D) {
#pragma unroll
      for (int j = 0; j < 16; ++j) stack[steps_after(i)][j] = cur[j];
      sstack[steps_after(i)] = sg;
    }
  });
}

__device__ __forceinline__ void tile_load(const float* __restrict__ src_row, float (&v)[16]);

// Where the log-space fallback of a fused leaf launch finds its operands (plain pointers and ints: the fallback is an
// out-of-line function, so that its registers -- sibling stack, weights -- are not part of the hot path's allocation).
struct SubtreeSource {
  const float* table;     // (F0, C+1, 32) LINEAR table rows
  const float* scale;     // (F0, C+1) their log scales
  const int32_t* xt;      // (Dvars, B) staged batch, or nullptr:
  const int64_t* x64;     // ... the raw (B, D) int64 batch (ck_leaf_walk_fwd with x_rows: low dwords, row min_u32(x, C))
  int D;
  const int64_t* scope;   // variable of each input-layer fold
  const int32_t* leaf_ids;  // (2^D) input-layer fold of each leaf of this root
  const int32_t* fold0;     // (2^D) table fold of each leaf
  const float* w_steps;     // the root's weights in step order (a flat pointer to their LDS copy), or nullptr:
  const float* w[4];        // ... then level l + 1 weights are w[l] + fold * 1024,
  const int32_t* nodes;     //     fold = nodes[node_off[l + 1] + t * (2^D >> (l + 1)) + (i >> (l + 1))]
  int node_off[5];
  int t, B, C, bl;
};

template <int D, int LAYOUT, bool SIGNED = false>
__device__ __noinline__ void subtree_tile_logspace(const SubtreeSource src, int lane, float (&out)[16], uint32_t* sign_out = nullptr) {
  const int kh = lane >> 5;
  auto leaf = [&](auto ic, float (&v)[16]) {
    constexpr int i = decltype(ic)::value;
    const int64_t var = src.scope[src.leaf_ids[i]];
    int64_t r = static_cast<int64_t>(src.fold0[i]) * (src.C + 1);
    if (src.xt != nullptr) {
      const int x = src.xt[var * static_cast<int64_t>(src.B) + src.bl];
      r += x < 0 ? src.C : min(x, src.C - 1);
    } else {  // the same mapping as the walk that noted this tile
      r += min(static_cast<uint32_t>(src.x64[static_cast<int64_t>(src.bl) * src.D + var]), static_cast<uint32_t>(src.C));
    }
    tile_load(src.table + r * kK + 4 * kh, v);
    return src.scale[r];
  };
  auto weights = [&](auto sc, auto ic, auto lc, WRegs& w) {
    constexpr int i = decltype(ic)::value, l = decltype(lc)::value;
    if (src.w_steps != nullptr) {
      load_w<CK_W_TILED_F32>(src.w_steps + decltype(sc)::value * 1024, lane, w);  // (tiled layouts: one KiB per q)
    } else {
      const int fold = src.nodes[src.node_off[l + 1] + src.t * ((1 << D) >> (l + 1)) + (i >> (l + 1))];
      load_w<LAYOUT>(src.w[l] + static_cast<int64_t>(fold) * (kK * kK), lane, w);
    }
  };
  if constexpr (SIGNED) {
    uint32_t sg = 0;
    tile_walk_logspace_signed<D, LAYOUT>(leaf, weights, out, sg);
    *sign_out = sg;
  } else {
    tile_walk_logspace<D, LAYOUT>(leaf, weights, out);
  }
}

// Read one (32 rows x 32 units) tile of a (B, 32) block in register layout, adding it to v.
__device__ __forceinline__ void tile_load_add(const float* __restrict__ src_row, float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t4 = *reinterpret_cast<const float4*>(src_row + 8 * g);
    v[4 * g + 0] += t4.x;
    v[4 * g + 1] += t4.y;
    v[4 * g + 2] += t4.z;
    v[4 * g + 3] += t4.w;
  }
}

__device__ __forceinline__ void tile_load(const float* __restrict__ src_row, float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t4 = *reinterpret_cast<const float4*>(src_row + 8 * g);
    v[4 * g + 0] = t4.x;
    v[4 * g + 1] = t4.y;
    v[4 * g + 2] = t4.z;
    v[4 * g + 3] = t4.w;
  }
}

// The tile as complex logarithms of real numbers, (v[j], pi if bit j of sg else 0), into a (B, 32) complex64 block:
// dst_row points at the lane's first complex element (float offset 2 (32 b + 4 kh) of the block).
__device__ __forceinline__ void tile_store_clog(float* __restrict__ dst_row, const float (&v)[16], uint32_t sg) {
  constexpr float kPi = 3.14159265358979323846f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float p0 = (sg >> (4 * g)) & 1u ? kPi : 0.f, p1 = (sg >> (4 * g + 1)) & 1u ? kPi : 0.f;
    const float p2 = (sg >> (4 * g + 2)) & 1u ? kPi : 0.f, p3 = (sg >> (4 * g + 3)) & 1u ? kPi : 0.f;
    *reinterpret_cast<float4*>(dst_row + 16 * g) = make_float4(v[4 * g + 0], p0, v[4 * g + 1], p1);
    *reinterpret_cast<float4*>(dst_row + 16 * g + 4) = make_float4(v[4 * g + 2], p2, v[4 * g + 3], p3);
  }
}

// The same tile in TILE-NATIVE order: a 4 KB block per (fold, 32-row tile) whose dword (g, lane, t) is unit 8g + 4 (lane >> 5) + t
// of row lane & 31 -- the register layout itself, so that a wave instruction moves one contiguous KiB (8 cache lines) where the
// row-major forms above touch 32 lines, 32 bytes of each.  Used for tiles that only tile kernels read (the training forward's
// kept tiles, gradient tiles between the backward launches): the texture addresser's time per line is what bounds those
// launches (scripts/exp_leaf_bwd.sh).  `block`: the tile's 1024 floats.
__device__ __forceinline__ void tile_load_native(const float* __restrict__ block, int lane, float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t4 = *reinterpret_cast<const float4*>(block + g * 256 + lane * 4);
    v[4 * g + 0] = t4.x;
    v[4 * g + 1] = t4.y;
    v[4 * g + 2] = t4.z;
    v[4 * g + 3] = t4.w;
  }
}
__device__ __forceinline__ void tile_store_native(float* __restrict__ block, int lane, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(block + g * 256 + lane * 4) = make_float4(v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

__device__ __forceinline__ void tile_store(float* __restrict__ dst_row, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(dst_row + 8 * g) =
        make_float4(v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

}  // namespace
