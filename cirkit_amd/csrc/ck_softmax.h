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
template <bool MAX>
__device__ __forceinline__ float half_reduce_dpp(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
  v = op(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x140, 0xF, 0xF, true)));  // row_mirror
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);  // rows 0<->1, 2<->3
  return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <bool MAX>
__device__ __forceinline__ float wave_reduce_dpp(float v) {
  return ck::wave_reduce<MAX>(v);  // (ck_internal.h)
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
template <int PASSES, class Put>
__device__ __forceinline__ void softmax_rows32(const float* __restrict__ in, int rows, int first_pair, int pair_stride, int lane,
                                               Put&& put) {
  const int half = lane >> 5, l = lane & 31;
  float x[PASSES];
  bool ok[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int row = 2 * (first_pair + p * pair_stride) + half;
    ok[p] = row < rows;
    x[p] = ok[p] ? in[row * 32 + l] : -INFINITY;
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int row = 2 * (first_pair + p * pair_stride) + half;
    const float mx = half_reduce_dpp<true>(x[p]);
    const float e = ok[p] ? __expf(x[p] - mx) : 0.f;
    const float sum = half_reduce_dpp<false>(e);
    if (ok[p]) put(row, l, e / sum);
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
template <int NW, bool KIND5, class Sync, class Store>
__device__ __forceinline__ void table_dense_rows(const float* __restrict__ theta, const float* __restrict__ theta_w, int C,
                                                 float* tile, int w, int lane, Sync&& sync, Store&& store) {
  constexpr int K = 32;
  static_assert(K % NW == 0 && 16 % NW == 0, "waves per job");
  constexpr int RPW = K / NW;  // units (rows of logits) per wave
  const int n4 = C >> 2, ld = C + 4;  // row stride of tile[k][c]: 16-byte aligned rows, conflict-free both ways
  float* w_s = tile + K * ld;          // [32][32] row-major linear weights of the dense fold
  const bool on = lane < n4 && theta != nullptr;
  float4 x[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)  // unit k = w + NW r: one row of C logits per wave and r, all loads in flight
    x[r] = on ? reinterpret_cast<const float4*>(theta)[(w + NW * r) * n4 + lane] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  if (theta_w != nullptr)  // W: 32 rows of 32, two rows per wave pass
    softmax_rows32<16 / NW>(theta_w, 32, w, NW, lane, [&](int row, int l, float p) { w_s[row * 32 + l] = p; });
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int k = w + NW * r;
    const float mx = wave_reduce_dpp<true>(fmaxf(fmaxf(x[r].x, x[r].y), fmaxf(x[r].z, x[r].w)));
    const float4 dl = make_float4(x[r].x - mx, x[r].y - mx, x[r].z - mx, x[r].w - mx);
    const float part = on ? (__expf(dl.x) + __expf(dl.y)) + (__expf(dl.z) + __expf(dl.w)) : 0.f;
    const float ls = __logf(wave_reduce_dpp<false>(part));
    if (on) {
      float4 o;
      o.x = dl.x < -103.9f ? -INFINITY : dl.x - ls;
      o.y = dl.y < -103.9f ? -INFINITY : dl.y - ls;
      o.z = dl.z < -103.9f ? -INFINITY : dl.z - ls;
      o.w = dl.w < -103.9f ? -INFINITY : dl.w - ls;
      *reinterpret_cast<float4*>(tile + k * ld + 4 * lane) = o;
    }
  }
  sync();
  if (theta == nullptr) return;
  WRegs wr;
  load_w<CK_W_ROWMAJOR>(w_s, lane, wr);
  const int b_in = lane & 31, kh = lane >> 5;
  for (int t = w; t * 32 <= C; t += NW) {  // 32 categories per register tile, rows 0..C
    const int c = t * 32 + b_in;
    const int cl = min(c, C - 1);
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) v[4 * g + tt] = c >= C ? 0.f : tile[(8 * g + 4 * kh + tt) * ld + cl];  // row C: log sum_c p = 0
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
