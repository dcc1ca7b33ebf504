// Register-tile helpers of the backward kernels (ck_leaf_bwd.hip, ck_tail_bwd.hip): the three contractions of a sum node with
// e = exp(v - m), y = W e, out = log y + m on the 32 x 32 tiles of ck_tile.h --
//     gy = g_out / y          dW += gy^T e          g_child = e * (W^T gy)
// (autograd through LSESumSemiring.apply_reduce, semiring.py:383-408).  dW contracts over the batch rows, which the register
// layout keeps on the lanes: both operands go through a swizzled 4 KB LDS tile per wave to get rows onto the MFMA's k index.
#pragma once

#include "ck_internal.h"
#include "ck_tile.h"

namespace {

// Row r, unit u of a 32 x 32 tile in a wave's LDS scratch: 16-byte chunks XOR-swizzled by the row so that the b128 writes of
// the register layout (lane = row) and the b32 reads of the transposed layout (lane = unit) are both conflict-free.
__device__ __forceinline__ int tsw(int r, int u) { return r * 32 + 4 * ((u >> 2) ^ (r & 7)) + (u & 3); }

__device__ __forceinline__ void tile_to_lds(float* s, int b_in, int kh, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(s + b_in * 32 + 4 * ((2 * g + kh) ^ (b_in & 7))) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// acc += gy^T e over the 32 rows of the tile: A[m = o][k = row] = gy[row][o], B[k = row][n = i] = e[row][i]; lanes (., kb)
// contract rows 16 kb + j at step j.  Result D[o][i] in lane (i, hi) register r, o = 8 (r >> 2) + 4 hi + (r & 3).
__device__ __forceinline__ void dw_accumulate(f32x16& acc, float* s_gy, float* s_e, int b_in, int kh, const float (&gy)[16], const float (&e)[16]) {
  tile_to_lds(s_gy, b_in, kh, gy);
  tile_to_lds(s_e, b_in, kh, e);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
  float a[16], b[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int row = 16 * kh + j;
    a[j] = s_gy[tsw(row, b_in)];
    b[j] = s_e[tsw(row, b_in)];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
}

// The same through ONE 4 KB tile (the two operands take turns): for kernels that are short of LDS, not of time.
__device__ __forceinline__ void dw_accumulate_seq(f32x16& acc, float* s_t, int b_in, int kh, const float (&gy)[16], const float (&e)[16]) {
  float a[16], b[16];
  tile_to_lds(s_t, b_in, kh, gy);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = s_t[tsw(16 * kh + j, b_in)];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // (the reads have returned before the tile is overwritten)
  tile_to_lds(s_t, b_in, kh, e);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
  for (int j = 0; j < 16; ++j) b[j] = s_t[tsw(16 * kh + j, b_in)];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
}

// v <- e * (W^T gy): wt = the node's weights in "transposed tiled" order (dword (q, lane, t) = W[8q + 4 (lane >> 5) + t][lane & 31])
__device__ __forceinline__ void child_gradient(const float* wt_lds, int lane, const float (&gy)[16], const float (&e)[16], float (&out)[16]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 w = *reinterpret_cast<const float4*>(wt_lds + q * 256 + lane * 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, gy[4 * q + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, gy[4 * q + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, gy[4 * q + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, gy[4 * q + 3], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) out[r] = acc[r] * e[r];
}

// gy = g / y (0 where the row is padding, the gradient is 0 or y is 0)
__device__ __forceinline__ void grad_over_y(const float (&g)[16], const float (&y)[16], bool live, float (&gy)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float q = g[r] * __builtin_amdgcn_rcpf(y[r]);
    gy[r] = (live && y[r] > 0.f && g[r] != 0.f) ? q : 0.f;
  }
}

}  // namespace
