// Device pieces of the parameter prologue that TWO translation units run: ck_param.hip (the batched prologue launch,
// ck_param_softmax_batch) and ck_leaf.hip (the persistent leaf launch evaluates the parameters of its own roots, see
// leaf_persistent_kernel<.., PARAMS>).  One source, one arithmetic: the two paths produce the same bits
// (tests/test_gpu_parity.py::test_leaf_launch_evaluates_its_parameters).
//
// Reference: the parameter graphs tensor -> softmax re-evaluated on every forward (parameters/parameter.py:180-188,
// nodes.py softmax over the last axis), TorchCategoricalLayer.log_unnormalized_likelihood's log-probabilities
// (layers/input.py:399-412) and the dense TorchSumLayer applied to them (layers/inner.py:266-273, semiring.py:383-408).
#pragma once

#include "ck_internal.h"
#include "ck_tile.h"

namespace {

// maximum / sum over each 32-lane half of a wave (rows of 32 weights, two rows per wave), without LDS: DPP inside the
// 16-lane rows, v_permlane16_swap across them (a __shfl_xor is an LDS round trip per step, ten of them per row in a chain)
// Maximum reductions by hand.  fmaxf() costs three instructions per DPP step (v_mov_dpp, a canonicalising v_max x x that
// quietens a signalling NaN the move could carry, v_max) plus the wait states of a dependent chain, and there is no way to tell
// this compiler that the values are not NaNs (#pragma float_control is not supported on the target).  A row of logits does
// not need fmaxf's NaN rule: a NaN anywhere in it makes the whole softmax row NaN either way (exp(NaN - m) enters the sum).
// FOUR independent values go through the tree together, their steps interleaved: one v_max_f32_dpp per value and step, and
// every DPP / permlane operand was written four instructions earlier (two wait states needed; the s_nop covers the first).
// The maximum is exact: same value as fmaxf for every non-NaN input.
__device__ __forceinline__ float max_plain(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#define CK_DPP4(ctrl)                                                                     \
  "v_max_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
  "v_max_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
  "v_max_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
  "v_max_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CK_SWAP4(op)                                                                      \
  "v_mov_b32 %4, %0\n\tv_mov_b32 %5, %1\n\tv_mov_b32 %6, %2\n\tv_mov_b32 %7, %3\n\t"       \
  op " %0, %4\n\t" op " %1, %5\n\t" op " %2, %6\n\t" op " %3, %7\n\t"                     \
  "v_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %5\n\tv_max_f32 %2, %2, %6\n\tv_max_f32 %3, %3, %7\n\t"
// maximum over each 32-lane half (WAVE = false) or over the wave of four values at once
template <bool WAVE>
__device__ __forceinline__ void max4_dpp(float& a, float& b, float& c, float& d) {
  float t0, t1, t2, t3;
  if constexpr (WAVE)
    asm("s_nop 1\n\t" CK_DPP4("quad_perm:[1,0,3,2]") CK_DPP4("quad_perm:[2,3,0,1]") CK_DPP4("row_half_mirror") CK_DPP4("row_mirror")
            CK_SWAP4("v_permlane16_swap_b32") CK_SWAP4("v_permlane32_swap_b32")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
  else
    asm("s_nop 1\n\t" CK_DPP4("quad_perm:[1,0,3,2]") CK_DPP4("quad_perm:[2,3,0,1]") CK_DPP4("row_half_mirror") CK_DPP4("row_mirror")
            CK_SWAP4("v_permlane16_swap_b32")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
}
#undef CK_DPP4
#undef CK_SWAP4
// (one value: the same steps with their wait states)
#define CK_DPP_MAX(name, ctrl)                                                                                     \
  __device__ __forceinline__ float name(float v) {                                                                 \
    float r;                                                                                                       \
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v)); \
    return r;                                                                                                      \
  }
CK_DPP_MAX(dpp_max_q1, "quad_perm:[1,0,3,2]")
CK_DPP_MAX(dpp_max_q2, "quad_perm:[2,3,0,1]")
CK_DPP_MAX(dpp_max_hm, "row_half_mirror")
CK_DPP_MAX(dpp_max_rm, "row_mirror")
#undef CK_DPP_MAX
__device__ __forceinline__ float row16_max(float v) { return dpp_max_rm(dpp_max_hm(dpp_max_q2(dpp_max_q1(v)))); }
template <bool MAX>
__device__ __forceinline__ float half_reduce_dpp(float v);
template <bool MAX>
__device__ __forceinline__ float wave_reduce_dpp(float v);
// the maxima of N independent values, four at a time
template <bool WAVE, int N>
__device__ __forceinline__ void max_n_dpp(float (&v)[N]) {
  int i = 0;
#pragma unroll
  for (; i + 4 <= N; i += 4) max4_dpp<WAVE>(v[i], v[i + 1], v[i + 2], v[i + 3]);
#pragma unroll
  for (; i < N; ++i) v[i] = WAVE ? wave_reduce_dpp<true>(v[i]) : half_reduce_dpp<true>(v[i]);
}
template <bool MAX>
__device__ __forceinline__ float half_reduce_dpp(float v) {
  if constexpr (MAX) {
    v = row16_max(v);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);  // rows 0<->1, 2<->3
    return max_plain(__uint_as_float(r[0]), __uint_as_float(r[1]));
  } else {
    auto op = [](float a, float b) { return a + b; };
    v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
    v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
    v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
    v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x140, 0xF, 0xF, true)));  // row_mirror
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);  // rows 0<->1, 2<->3
    return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
}
template <bool MAX>
__device__ __forceinline__ float wave_reduce_dpp(float v) {
  if constexpr (MAX) {
    v = row16_max(v);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = max_plain(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max_plain(__uint_as_float(q[0]), __uint_as_float(q[1]));
  } else {
    return ck::wave_reduce<false>(v);  // (ck_internal.h: v_add_f32_dpp steps already)
  }
}

// Where a softmaxed 32 x 32 weight matrix goes: dword index of W[o][l] inside the fold's 1024-dword block
//   row-major: o * 32 + l;   CK_W_TILED_F32 (ck_tile.h): (l >> 3) * 256 + (o + 32 * ((l >> 2) & 1)) * 4 + (l & 3)
__device__ __forceinline__ int w32_index(int o, int l, bool tiled) {
  return tiled ? (l >> 3) * 256 + (o + 32 * ((l >> 2) & 1)) * 4 + (l & 3) : o * 32 + l;
}

// softmax of the rows of a (rows, 32) block of logits, two rows per wave pass (one per 32-lane half), `PASSES` passes of
// this wave in flight at once: pass p handles rows 2 * (first_pair + p * pair_stride) + half.  The reduction tree and
// the order of operations are those of every 32-wide softmax of the prologue (softmax_job_rows, len <= 32).
//   put(row, l, p): stores the probability of entry l of `row`.
// (in two halves, so that a caller with several blocks to do can request all of them before it evaluates the first)
template <int PASSES>
__device__ __forceinline__ void softmax_rows32_load(const float* __restrict__ in, int rows, int first_pair, int pair_stride, int lane,
                                                    float (&x)[PASSES]) {
  const int half = lane >> 5, l = lane & 31;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int row = 2 * (first_pair + p * pair_stride) + half;
    x[p] = row < rows ? ck::as_global(in)[row * 32 + l] : -INFINITY;  // (`in` is device memory in every caller)
  }
}
template <int PASSES, class Put>
__device__ __forceinline__ void softmax_rows32_apply(const float (&x)[PASSES], int rows, int first_pair, int pair_stride, int lane, Put&& put) {
  const int half = lane >> 5, l = lane & 31;
  float mx[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) mx[p] = x[p];
  max_n_dpp<false, PASSES>(mx);
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int row = 2 * (first_pair + p * pair_stride) + half;
    const bool ok = row < rows;
    const float e = ok ? __expf(x[p] - mx[p]) : 0.f;
    const float sum = half_reduce_dpp<false>(e);
    if (ok) put(row, l, e / sum);
  }
}
template <int PASSES, class Put>
__device__ __forceinline__ void softmax_rows32(const float* __restrict__ in, int rows, int first_pair, int pair_stride, int lane,
                                               Put&& put) {
  float x[PASSES];
  softmax_rows32_load<PASSES>(in, rows, first_pair, pair_stride, lane, x);
  softmax_rows32_apply<PASSES>(x, rows, first_pair, pair_stride, lane, put);
}

// Log-softmax of R rows of logits held by a wave, four per lane (lanes past a row hold -inf), in place:
//   x <- (x - max) - log sum exp(x - max),   -inf where x - max < -103.9 (where the reference's softmax underflows to 0 and
//   its logarithm gives -inf: layers/input.py:399-412).
// Stage by stage over the rows, so that their independent chains interleave (one row after the other is a chain of ~60
// dependent instructions with wait states in between), and branch-free: lanes past the row hold -inf, exp2 makes them 0.
// The straightforward form -- fmaxf reductions, __logf, a select per step, row by row -- was 105 instructions per row, and a
// table job is bound by instruction issue.  Every table job of the prologue uses this function: their tables
// agree bit for bit.
template <int R>
__device__ __forceinline__ void log_softmax_rows(float4 (&x)[R]) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  float mx[R], ls[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mx[r] = max_plain(max_plain(x[r].x, x[r].y), max_plain(x[r].z, x[r].w));
  max_n_dpp<true, R>(mx);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    x[r] = make_float4(x[r].x - mx[r], x[r].y - mx[r], x[r].z - mx[r], x[r].w - mx[r]);
    const f32x2v a01 = f32x2v{x[r].x, x[r].y} * 1.44269504088896340736f, a23 = f32x2v{x[r].z, x[r].w} * 1.44269504088896340736f;  // (= __expf's multiply)
    ls[r] = (__builtin_amdgcn_exp2f(a01.x) + __builtin_amdgcn_exp2f(a01.y)) + (__builtin_amdgcn_exp2f(a23.x) + __builtin_amdgcn_exp2f(a23.y));
  }
#pragma unroll
  for (int r = 0; r < R; ++r) ls[r] = __builtin_amdgcn_logf(wave_reduce_dpp<false>(ls[r]));  // log2 of the sum, in [0, log2 C]: no denormal to rescale
#pragma unroll
  for (int r = 0; r < R; ++r) {
    // x - ln 2 * log2(sum) as ONE explicit FMA: a multiply and a subtraction would be contracted by the compiler in one caller
    // and not in another, and the tables of the different jobs have to agree bit for bit
    x[r].x = x[r].x < -103.9f ? -INFINITY : __builtin_fmaf(-kLN2, ls[r], x[r].x);
    x[r].y = x[r].y < -103.9f ? -INFINITY : __builtin_fmaf(-kLN2, ls[r], x[r].y);
    x[r].z = x[r].z < -103.9f ? -INFINITY : __builtin_fmaf(-kLN2, ls[r], x[r].z);
    x[r].w = x[r].w < -103.9f ? -INFINITY : __builtin_fmaf(-kLN2, ls[r], x[r].w);
  }
}

// The Categorical log-table of one fold pushed through one dense fold, by `NW` waves (this wave is number `w` of them):
//   T[c, k]   = (theta[k, c] - max_c theta[k, .]) - log sum_c exp(theta[k, c] - max)        (-inf below -103.9)
//   KIND5:  out[c, :] = W . exp(T[c, :] - m_c),  out2[c] = m_c = max_k T[c, k]      (rows stay linear, scale aside)
//   else:   out[c, :] = log(W . exp(T[c, :] - m_c)) + m_c                              (the dense layer's own output)
//   W = softmax(theta_dense) (rows of 32), c = 0 .. C (row C: the integral row, T = 0); C <= 256, C % 4 == 0, K = 32.
// A wave holds whole (unit, all categories) rows of logits in registers -- one float4 per lane -- so the per-unit
// maximum and log-sum-exp are wave reductions; the tile in LDS (`tile`: 32 x (C + 4) floats, then 1024 floats of W)
// holds the normalised log-probabilities the dense layer is applied to on 32-category register tiles.
// `sync()` is the barrier of the waves that share `tile` (all of them call this function together);
// `store(c, v, m)` receives the finished row tile: category c = tile * 32 + (lane & 31) of this lane, its 16 values, and m.
// theta: the fold's (32, C) logits; theta_w: the dense fold's (32, 32) logits (both may be nullptr: the wave then only
// takes part in the barriers -- a group with nothing to do in this round).
// (table_dense_rows_x: the same with the wave's logits already in registers -- x[r] = categories 4 lane .. 4 lane + 3 of unit
//  w + NW r, -inf in the lanes past C / 4 -- and theta_w anywhere (an LDS copy): the epilogue of ck_table_dense_bwd, which has just
//  updated both, builds the next step's table from them)
template <int NW, bool KIND5, bool WLDS = false, class Sync, class Store>
__device__ __forceinline__ void table_dense_rows_x(float4 (&x)[32 / NW], bool have, const float* theta_w, int C, float* tile, int w, int lane,
                                                   Sync&& sync, Store&& store);

template <int NW, bool KIND5, class Sync, class Store>
__device__ __forceinline__ void table_dense_rows(const float* __restrict__ theta, const float* __restrict__ theta_w, int C,
                                                 float* tile, int w, int lane, Sync&& sync, Store&& store) {
  constexpr int RPW = 32 / NW;  // units (rows of logits) per wave
  const int n4 = C >> 2;
  float4 x[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) x[r] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  if (lane < n4 && theta != nullptr) {  // unit k = w + NW r: one row of C logits per wave and r, all loads in flight
#pragma unroll
    for (int r = 0; r < RPW; ++r) x[r] = reinterpret_cast<const float4*>(theta)[(w + NW * r) * n4 + lane];
  }
  table_dense_rows_x<NW, KIND5, false>(x, theta != nullptr, theta_w, C, tile, w, lane, sync, store);
}

template <int NW, bool KIND5, bool WLDS, class Sync, class Store>
__device__ __forceinline__ void table_dense_rows_x(float4 (&x)[32 / NW], bool have, const float* theta_w, int C, float* tile, int w, int lane,
                                                   Sync&& sync, Store&& store) {
  constexpr int K = 32;
  static_assert(K % NW == 0 && 16 % NW == 0, "waves per job");
  constexpr int RPW = K / NW;  // units (rows of logits) per wave
  __builtin_assume(w >= 0 && w < NW);  // (every row of W this wave takes exists: no bounds branch per pass)
  const int n4 = C >> 2, ld = C + 4;  // row stride of tile[k][c]: 16-byte aligned rows, conflict-free both ways
  float* w_s = tile + K * ld;          // [32][32] row-major linear weights of the dense fold
  const bool on = lane < n4 && have;
  if (theta_w != nullptr) {  // W: 32 rows of 32, two rows per wave pass
    auto put = [&](int row, int l, float p) { w_s[row * 32 + l] = p; };
    if constexpr (WLDS) {  // (the logits sit in LDS: plain reads, the same reductions)
      float xw[16 / NW];
#pragma unroll
      for (int p = 0; p < 16 / NW; ++p) xw[p] = theta_w[(2 * (w + p * NW) + (lane >> 5)) * 32 + (lane & 31)];
      softmax_rows32_apply<16 / NW>(xw, 32, w, NW, lane, put);
    } else {
      softmax_rows32<16 / NW>(theta_w, 32, w, NW, lane, put);
    }
  }
  log_softmax_rows<RPW>(x);
  if (on) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) *reinterpret_cast<float4*>(tile + (w + NW * r) * ld + 4 * lane) = x[r];
  }
  sync();
  if (!have) return;
  WRegs wr;
  load_w<CK_W_ROWMAJOR>(w_s, lane, wr);
  const int b_in = lane & 31, kh = lane >> 5;
  for (int t = w; t * 32 <= C; t += NW) {  // 32 categories per register tile, rows 0..C
    const int c = t * 32 + b_in;
    const int cl = min(c, C - 1);
    float v[16];
    const float* col = tile + 4 * kh * ld + cl;  // (read unconditionally -- cl is a valid column --, then selected: a load under a
    const bool past = c >= C;                    //  condition is a branch per element)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const float t = col[(8 * g + tt) * ld];
        v[4 * g + tt] = past ? 0.f : t;  // row C: log sum_c p = 0
      }
    float m = 0.f;
    if constexpr (!KIND5) {
      sum_step<CK_W_ROWMAJOR>(wr, v);
    } else {
      m = row_max16(v);
      const float nml = exp_offset(m, 0.f);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_exp2f(fmaf(v[r], kL2E, nml));
      contract_linear<CK_W_ROWMAJOR>(wr, v);
    }
    store(c, v, m);
  }
}

}  // namespace
