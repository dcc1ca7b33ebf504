// Backward of the fused leaf region (ck_leaf.hip's training forward): the gradient walk of Categorical table -> CP-T levels
// without materialised layer gradients.
//
// The reference obtains these gradients from autograd through TorchCPTLayer.forward (optimized.py:171-178) and
// LSESumSemiring.apply_reduce (semiring.py:383-408), keeping every layer output alive.  Per CP-T node n with children a, b
// (log-space values v), e = exp(v_a + v_b - m), y = W e, out = log y + m:
//     gy = g_out / y          dW += gy^T e          g_a = g_b = e * (W^T gy)
// -- the SAME log-space gradient tile goes to both children of a product, so one tile per node walks down the tree.  Nothing
// depends on the row scale m as long as e and y carry the same one: the training forward keeps the LINEAR tile y of every
// node (LeafArgs::keep), this walk forms e from the kept children with the forward's own instructions (bare product at the
// first level, exact power-of-two renormalisation above: ck_tile.h linear_product), so the kept y of the node IS W e bit for
// bit and is not recomputed.
//
// One launch covers TWO levels: a unit is (node P, its children Q0, Q1, their four children c0..c3) for one 32-row batch
// tile; P's gradient tile is read (written by the launch above: the tile its parent left for its two children; at the top,
// the gradient of the root layer's outputs), the tiles of Q0 and Q1 are written for the launch below (LEAF: for the
// Categorical scatter, ck_categorical_bwd with a fold index -- float atomics on gfx950 retire one dword per L2 channel and
// clock: 103 M additions took 0.35 ms, scripts/ubench/atomic_scatter.hip).  Resident workgroups of 8 waves walk
// (P, tile range) segments like the forward; a wave keeps the three 32 x 32 weight-gradient accumulators of its segment
// in registers (dW += gy^T e is an MFMA contraction over the batch rows: both operands go through a swizzled 4 KB LDS tile
// per wave to get rows onto the k index), they are reduced over the waves in LDS and leave with ONE atomic per element
// and segment.  W^T (the A operand of W^T gy) is staged per segment from the row-major weights.
//
// Tiles whose forward walk left the linear range (LeafArgs::redo) are skipped here and taken by leaf_bwd_redo_kernel:
// the whole subtree of that (root, tile) again in log space -- the reference's arithmetic -- forward and backward.
#include <algorithm>

#include "ck_internal.h"
#include "ck_tile.h"

namespace {

struct BwdArgs {
  const int32_t* unit_tab;  // (n_p, 16): gin fold, P fold, Q0 fold, Q1 fold, c0..c3 (fold of the child level / table fold), v0..v3 (variables), root
  const int32_t* work;      // (n_seg, 4): row of unit_tab, first tile, end tile, 0
  int n_seg, B, C, D;
  const float* gin;  // (F, B, 32) log-space gradient tiles, indexed by unit_tab[.., 0]
  const float* y_p;  // kept linear tiles of P's level
  const float* y_q;  // ... of the level below
  const float* y_c;  // !LEAF: ... of the level below that
  const float* table;    // LEAF: (F0, C + 1, 32) linear table rows
  const int64_t* x64;    // LEAF: raw (B, D) batch
  const float* w_p;      // (F_p, 32, 32) row-major linear weights of P's level
  const float* w_q;
  float* dw_p;           // += gy^T e
  float* dw_q;
  float* gout;           // (F_q, B, 32): the gradient tile node Q leaves for its two children
  const int32_t* redo;   // (n_roots, tiles) flags of the forward, or nullptr
};

constexpr int kUnitTab = 16;

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

// e <- a * b as the forward forms it (ck_tile.h): the bare product at the first fused level, renormalised by the power of two
// of the row maximum above it -- the kept y of the node is W e for exactly this e.
template <bool RESCALE>
__device__ __forceinline__ void forward_product(float (&cur)[16], const float (&sib)[16]) {
  float s = 0.f;
  bool bad = false;
  linear_product<RESCALE>(cur, sib, s, 0.f, bad);
}

template <bool LEAF>
__global__ void __launch_bounds__(512) leaf_bwd_kernel(const BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float wt_lds[3 * 1024];      // W^T of P, Q0, Q1 ("transposed tiled")
  __shared__ __attribute__((aligned(16))) float scratch[8 * 2 * 1024];  // per wave: the gy and e tiles of a dW contraction
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b_in = lane & 31, kh = lane >> 5;
  float* const s_gy = scratch + wave * 2048;
  float* const s_e = s_gy + 1024;
  const int n_tiles = (a.B + 31) >> 5;
  for (int seg = blockIdx.x; seg < a.n_seg; seg += gridDim.x) {
    const int p = a.work[4 * seg], tile_begin = a.work[4 * seg + 1], tile_end = a.work[4 * seg + 2];
    const int32_t* ut = a.unit_tab + static_cast<int64_t>(p) * kUnitTab;
    const int gin_fold = ut[0], p_fold = ut[1], q_fold[2] = {ut[2], ut[3]};
    const int c_fold[4] = {ut[4], ut[5], ut[6], ut[7]};
    const int var[4] = {ut[8], ut[9], ut[10], ut[11]};
    const int root = ut[12];
    if (seg != static_cast<int>(blockIdx.x)) __syncthreads();  // every wave has left the previous segment
    // W^T of the three nodes: row-major W[o][i] -> dword (o >> 3) * 256 + (i + 32 ((o >> 2) & 1)) * 4 + (o & 3)
    for (int n = 0; n < 3; ++n) {
      const float* w = n == 0 ? a.w_p + static_cast<int64_t>(p_fold) * 1024 : a.w_q + static_cast<int64_t>(q_fold[n - 1]) * 1024;
      for (int idx = threadIdx.x; idx < 1024; idx += 512) {
        const int o = idx >> 5, i = idx & 31;
        wt_lds[n * 1024 + (o >> 3) * 256 + (i + 32 * ((o >> 2) & 1)) * 4 + (o & 3)] = w[idx];
      }
    }
    __syncthreads();
    f32x16 dw[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[n][r] = 0.f;
    const float* gin = a.gin + static_cast<int64_t>(gin_fold) * a.B * kK;
    const float* yp = a.y_p + static_cast<int64_t>(p_fold) * a.B * kK;
    for (int tile = tile_begin + wave; tile < tile_end; tile += 8) {
      if (a.redo != nullptr && a.redo[static_cast<int64_t>(root) * n_tiles + tile] != 0) continue;  // (leaf_bwd_redo_kernel's)
      const int b = tile * 32 + b_in;
      const bool live = b < a.B;
      const int bl = live ? b : a.B - 1;
      const int64_t off = static_cast<int64_t>(bl) * kK + 4 * kh;
      float g[16], y[16], yq0[16], yq1[16];
      tile_load(gin + off, g);
      tile_load(yp + off, y);
      tile_load(a.y_q + static_cast<int64_t>(q_fold[0]) * a.B * kK + off, yq0);
      tile_load(a.y_q + static_cast<int64_t>(q_fold[1]) * a.B * kK + off, yq1);
      // the four bottom tiles: table rows of the four leaves (LEAF), kept tiles of the level below otherwise
      const float* cptr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (LEAF) {
          const uint32_t lo = static_cast<uint32_t>(a.x64[static_cast<int64_t>(bl) * a.D + var[i]]);
          const uint32_t row = min(lo, static_cast<uint32_t>(a.C));  // negative: the integral row (as the forward)
          cptr[i] = a.table + (static_cast<int64_t>(c_fold[i]) * (a.C + 1) + row) * kK + 4 * kh;
        } else {
          cptr[i] = a.y_c + static_cast<int64_t>(c_fold[i]) * a.B * kK + off;
        }
      }
      float gy[16], e[16], gq[16];
      // ---- node P
#pragma unroll
      for (int r = 0; r < 16; ++r) e[r] = yq1[r];
      forward_product<true>(e, yq0);
      grad_over_y(g, y, live, gy);
      dw_accumulate(dw[0], s_gy, s_e, b_in, kh, gy, e);
      child_gradient(wt_lds, lane, gy, e, gq);
      // ---- nodes Q0, Q1
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float c0[16];
        tile_load(cptr[2 * n], c0);
        tile_load(cptr[2 * n + 1], e);
        forward_product<!LEAF>(e, c0);
        grad_over_y(gq, n == 0 ? yq0 : yq1, live, gy);
        dw_accumulate(dw[1 + n], s_gy, s_e, b_in, kh, gy, e);
        float gc[16];
        child_gradient(wt_lds + (1 + n) * 1024, lane, gy, e, gc);
        if (live) tile_store(a.gout + static_cast<int64_t>(q_fold[n]) * a.B * kK + static_cast<int64_t>(b) * kK + 4 * kh, gc);
      }
    }
    // the segment's weight gradients: summed over the waves in LDS, one atomic per element
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 3; ++n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_gy[(8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + b_in] = dw[n][r];
      __syncthreads();
      float* dst = n == 0 ? a.dw_p + static_cast<int64_t>(p_fold) * 1024 : a.dw_q + static_cast<int64_t>(q_fold[n - 1]) * 1024;
      for (int idx = threadIdx.x; idx < 1024; idx += 512) {
        float sacc = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) sacc += scratch[w8 * 2048 + idx];
        if (sacc != 0.f) atomicAdd(dst + idx, sacc);
      }
      __syncthreads();
    }
  }
}

}  // namespace

extern "C" {

int ck_leaf_walk_bwd(const ck_leaf_bwd_launch* d, void* stream) {
  CK_REQUIRE(d != nullptr, "ck_leaf_walk_bwd: null descriptor");
  CK_REQUIRE(d->unit_tab && d->work && d->gin && d->y_p && d->y_q && d->w_p && d->w_q && d->dw_p && d->dw_q && d->gout,
             "ck_leaf_walk_bwd: null pointer");
  CK_REQUIRE(d->n_seg > 0 && d->n_wg > 0 && d->B > 0, "ck_leaf_walk_bwd: non-positive size");
  CK_REQUIRE(static_cast<int64_t>(d->B) * kK < (int64_t{1} << 31), "ck_leaf_walk_bwd: B=%d too large", d->B);
  if (d->leaf) CK_REQUIRE(d->table && d->x_rows && d->C > 0 && d->D > 0, "ck_leaf_walk_bwd: the leaf launch needs table, x_rows, C and D");
  else CK_REQUIRE(d->y_c != nullptr, "ck_leaf_walk_bwd: y_c is null");
  BwdArgs a{};
  a.unit_tab = d->unit_tab;
  a.work = d->work;
  a.n_seg = d->n_seg;
  a.B = d->B;
  a.C = d->C;
  a.D = d->D;
  a.gin = d->gin;
  a.y_p = d->y_p;
  a.y_q = d->y_q;
  a.y_c = d->y_c;
  a.table = d->table;
  a.x64 = d->x_rows;
  a.w_p = d->w_p;
  a.w_q = d->w_q;
  a.dw_p = d->dw_p;
  a.dw_q = d->dw_q;
  a.gout = d->gout;
  a.redo = d->redo;
  const bool leaf = d->leaf != 0;
  dim3 grid(static_cast<unsigned>(std::min(d->n_wg, d->n_seg)));
  return ck::dispatch(
      [=](hipStream_t s) {
        if (leaf) hipLaunchKernelGGL((leaf_bwd_kernel<true>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((leaf_bwd_kernel<false>), grid, dim3(512), 0, s, a);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
