// Backward of the fused leaf region (ck_leaf.hip's training forward): the gradient walk of Categorical table -> CP-T levels
// without materialised layer gradients.
//
// The reference obtains these gradients from autograd through TorchCPTLayer.forward (optimized.py:171-178) and
// LSESumSemiring.apply_reduce (semiring.py:383-408), keeping every layer output alive.  Per CP-T node n with children a, b
// (log-space values v), e = exp(v_a + v_b - m), y = W e, out = log y + m:
//     gy = g_out / y          dW += gy^T e          g_a = g_b = e * (W^T gy)
// -- the SAME log-space gradient tile goes to both children of a product, so one tile per node walks down the tree.  Nothing
// depends on the row scale m as long as e and y carry the same one: the training forward keeps the LINEAR tile y of the
// nodes of every SECOND level (LeafArgs::keep: the P's of the units below), this walk forms e from the kept (or gathered)
// children with the forward's own instructions (bare product at the first level, exact power-of-two renormalisation above:
// ck_tile.h linear_product), so the kept y of P IS W e bit for bit; the y of Q0 / Q1 -- the level in between -- is the
// forward's contraction again (W_Q e_Q, same operands, same instruction: the same bits).  The launches move bytes, not flops
// (scripts/exp_leaf_bwd.sh: all contractions removed -8 %; 44 KB per unit through L2 at ~5 TB/s IS the launch time), so 2 more
// contractions per unit are cheaper than 8 KB of kept tiles written by the forward and read here.
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
#include <cstdlib>

#include "ck_bwd_tile.h"
#include "ck_internal.h"
#include "ck_opt.h"
#include "ck_softmax.h"
#include "ck_tile.h"

namespace {

struct BwdArgs {
  const int32_t* unit_tab;  // (n_p, 16): gin fold, P fold, Q0 fold, Q1 fold, c0..c3 (fold of the child level / table fold), v0..v3 (variables), root
  const int32_t* work;      // (n_seg, 4): row of unit_tab, first tile, end tile, 0
  int n_seg, B, C, D;
  const float* gin;  // log-space gradient tiles, indexed by unit_tab[.., 0]: (F, tiles, 1024) tile-native, or (gin_rowmajor) (F, B, 32)
  const float* y_p;  // kept linear tiles of P's level, (F_l, tiles, 1024) tile-native (ck_tile.h)
  const float* y_c;  // !LEAF: ... of the level two below (the level in between is recomputed)
  int gin_rowmajor;
  const float* table;    // LEAF: (F0, C + 1, 32) linear table rows
  const int64_t* x64;    // LEAF: raw (B, D) batch
  const float* w_p;      // (F_p, 32, 32) row-major linear weights of P's level
  const float* w_q;
  float* dw_p;           // += gy^T e
  float* dw_q;
  float* gout;           // the gradient tile node Q leaves for its two children: (F_q, tiles, 1024) tile-native; LEAF: (F_q, B, 32) row-major
                         // (what the Categorical scatter reads row by row)
  const int32_t* redo;   // (n_roots, tiles) flags of the forward, or nullptr
#ifdef CK_BWD_STAMPS  // the lab build of scripts/bwd_stamps.py / scripts/exp_leaf_bwd.sh (never the product)
  long long* stamps;     // shader-clock stamps: [wave][unit < 16][8] of workgroup stamp_wg
  int stamp_wg;
  int exp;               // timing experiments (CK_BWD_EXP, wrong results): 1 no dW contraction, 2 no W^T contraction, 4 every tile reads rows 0..31, 8 no stores
#endif
};

#ifdef CK_BWD_STAMPS
#define CK_EXP(args, bit) (((args).exp & (bit)) != 0)
#else
#define CK_EXP(args, bit) false
#endif

constexpr int kUnitTab = 16;

// e <- a * b as the forward forms it (ck_tile.h): the bare product at the first fused level, renormalised by the power of two
// of the row maximum above it -- the kept y of the node is W e for exactly this e.
template <bool RESCALE, bool SIGNED>
__device__ __forceinline__ void forward_product(float (&cur)[16], const float (&sib)[16]) {
  float s = 0.f;
  bool bad = false;
  linear_product<RESCALE, SIGNED>(cur, sib, s, 0.f, bad);
}
// gy = g / y for SIGNED tiles (y of either sign; ck_bwd_tile.h grad_over_y tests y > 0)
template <bool SIGNED>
__device__ __forceinline__ void grad_over(const float (&g)[16], const float (&y)[16], bool live, float (&gy)[16]) {
  if constexpr (SIGNED) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float q = g[r] * __builtin_amdgcn_rcpf(y[r]);
      gy[r] = (live && y[r] != 0.f && g[r] != 0.f) ? q : 0.f;
    }
  } else {
    grad_over_y(g, y, live, gy);
  }
}

// WAVES = 8: two waves per SIMD, a unit's loads issued at its start (the other wave computes meanwhile).  Measured at the
// north-star configuration the launches are LATENCY-bound in that form (scripts/exp_leaf_bwd.sh: without any MFMA 224 of 245 us;
// loads, stores and the rest add up): a unit is ~15 us of dependent memory round trips against 3.4 us of instructions, and
// two units in flight per SIMD do not cover that.
// WAVES = 4 (one wave per SIMD, up to 512 registers): software-pipelined -- the raw tiles of unit k + 1 travel while unit k
// computes.  An iteration first CONSUMES the raw registers of its unit (gy = g / y, the three products: 96 registers), then
// refills the same raw registers with the loads of the next unit, stores the previous unit's results (carried in registers, so
// that no store is younger than the loads the next iteration waits for first: the compiler's wait there is vmcnt(0)), and
// computes.  Marked units are computed with every row dead (no contribution, nothing stored).
// SIGNED: the tiles of a real-valued circuit under complex-lse-sum (ck_leaf.hip SIGNED + KEEP: a squared circuit's c(x)): linear
// values of either sign, renormalised by the row maximum of |.|; the gradients are those w.r.t. log|.| -- the same three
// contractions (ck_signed.hip).
template <bool LEAF, int WAVES, bool SIGNED = false>
__global__ void __launch_bounds__(WAVES * 64) leaf_bwd_kernel(const BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float wt_lds[3 * 1024];          // W^T of P, Q0, Q1 ("transposed tiled")
  __shared__ __attribute__((aligned(16))) float wq_lds[2 * 1024];          // W of Q0, Q1 (CK_W_TILED_F32: the forward's operand)
  __shared__ __attribute__((aligned(16))) float scratch[WAVES * 2 * 1024];  // per wave: the gy and e tiles of a dW contraction
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b_in = lane & 31, kh = lane >> 5;
  float* const s_gy = scratch + wave * 2048;
  float* const s_e = s_gy + 1024;
  const int n_tiles = (a.B + 31) >> 5;
#ifdef CK_BWD_STAMPS
  int stamp_unit = 0;
#define CK_BSTAMP(id)                                                                                              \
  do {                                                                                                             \
    if (a.stamps != nullptr && static_cast<int>(blockIdx.x) == a.stamp_wg && stamp_unit < 16) {                    \
      const long long c_ = clock64();                                                                              \
      if (lane == 0) a.stamps[(wave * 16 + stamp_unit) * 8 + (id)] = c_;                                           \
    }                                                                                                              \
  } while (0)
#else
#define CK_BSTAMP(id)
#endif
  for (int seg = blockIdx.x; seg < a.n_seg; seg += gridDim.x) {
    const int p = a.work[4 * seg], tile_begin = a.work[4 * seg + 1], tile_end = a.work[4 * seg + 2];
    if (tile_begin >= tile_end) continue;  // (an empty second segment of fusion.balanced_segments' flat schedule; uniform)
    const int32_t* ut = a.unit_tab + static_cast<int64_t>(p) * kUnitTab;
    const int gin_fold = ut[0], p_fold = ut[1], q_fold[2] = {ut[2], ut[3]};
    const int c_fold[4] = {ut[4], ut[5], ut[6], ut[7]};
    const int var[4] = {ut[8], ut[9], ut[10], ut[11]};
    const int root = ut[12];
    if (seg != static_cast<int>(blockIdx.x)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has left the previous segment's LDS
    // W^T of the three nodes: row-major W[o][i] -> dword (o >> 3) * 256 + (i + 32 ((o >> 2) & 1)) * 4 + (o & 3)
    for (int n = 0; n < 3; ++n) {
      const float* w = n == 0 ? a.w_p + static_cast<int64_t>(p_fold) * 1024 : a.w_q + static_cast<int64_t>(q_fold[n - 1]) * 1024;
      for (int idx = threadIdx.x; idx < 1024; idx += WAVES * 64) {
        const int o = idx >> 5, i = idx & 31;
        wt_lds[n * 1024 + (o >> 3) * 256 + (i + 32 * ((o >> 2) & 1)) * 4 + (o & 3)] = w[idx];
        if (n > 0) wq_lds[(n - 1) * 1024 + (i >> 3) * 256 + (o + 32 * ((i >> 2) & 1)) * 4 + (i & 3)] = w[idx];
      }
    }
    __syncthreads();
    f32x16 dw[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[n][r] = 0.f;
    const int64_t fold_stride = static_cast<int64_t>(n_tiles) * 1024;  // floats per fold of a tile-native array
    const float* gin = a.gin + (a.gin_rowmajor ? static_cast<int64_t>(gin_fold) * a.B * kK : gin_fold * fold_stride);
    const float* yp = a.y_p + p_fold * fold_stride;
    // marks of the segment's tiles (the units leaf_bwd_redo_kernel takes): one load per lane, then a bit per tile of this wave
    uint64_t marked = 0;  // bit k: tile tile_begin + wave + WAVES k
    if (a.redo != nullptr) {
      const int tl = tile_begin + wave + WAVES * lane;
      marked = __ballot(tl < tile_end && a.redo[static_cast<int64_t>(root) * n_tiles + tl] != 0);
    }
    auto is_marked = [&](int nth, int tile) {
      return nth < 64 ? ((marked >> nth) & 1) != 0 : (a.redo != nullptr && a.redo[static_cast<int64_t>(root) * n_tiles + tile] != 0);
    };
    // the raw tiles of a unit: batch values (LEAF), gradient tile, kept tiles of P and Q0 / Q1, the four bottom tiles
    uint32_t xlo[4];
    float g[16], y[16], c[4][16];
    auto issue_x = [&](int tile, uint32_t (&x)[4]) {  // LEAF: the batch values of the unit's four leaves
      if constexpr (LEAF) {
        if (CK_EXP(a, 4)) tile = 0;
        const int bl = min(tile * 32 + b_in, a.B - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = static_cast<uint32_t>(a.x64[static_cast<int64_t>(bl) * a.D + var[i]]);
      }
    };
    auto issue_a = [&](int tile) mutable {  // every tile whose address is known
      if (CK_EXP(a, 4)) tile = 0;
      const int bl = min(tile * 32 + b_in, a.B - 1);
      const int64_t blk = static_cast<int64_t>(tile) * 1024;
      if (a.gin_rowmajor) tile_load(gin + static_cast<int64_t>(bl) * kK + 4 * kh, g);
      else tile_load_native(gin + blk, lane, g);
      tile_load_native(yp + blk, lane, y);
      if constexpr (!LEAF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tile_load_native(a.y_c + c_fold[i] * fold_stride + blk, lane, c[i]);
      }
    };
    auto issue_b = [&](const uint32_t (&x)[4]) {  // LEAF: the gathered table rows, once the batch values are here
      if constexpr (LEAF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t row = min(x[i], static_cast<uint32_t>(a.C));  // negative: the integral row (as the forward)
          tile_load(a.table + (static_cast<int64_t>(c_fold[i]) * (a.C + 1) + row) * kK + 4 * kh, c[i]);
        }
      }
    };
    auto node = [&](int n, const float (&gy)[16], const float (&e)[16], float (&out)[16]) {
      if (!CK_EXP(a, 1)) dw_accumulate(dw[n], s_gy, s_e, b_in, kh, gy, e);
      if (!CK_EXP(a, 2)) child_gradient(wt_lds + n * 1024, lane, gy, e, out);
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = e[r] * gy[r];
      }
    };
    auto recompute = [&](int n, const float (&e)[16], float (&yq)[16]) {  // y of Q_n = W e, as the forward computed it
      WRegs w;
      load_w<CK_W_TILED_F32>(wq_lds + n * 1024, lane, w);
#pragma unroll
      for (int r = 0; r < 16; ++r) yq[r] = e[r];
      contract_linear<CK_W_TILED_F32>(w, yq);
    };
    auto store = [&](int tile, const float (&r0)[16], const float (&r1)[16]) {
      if (CK_EXP(a, 8)) return;
      if constexpr (LEAF) {
        const int b = tile * 32 + b_in;
        if (b < a.B) {
          tile_store(a.gout + static_cast<int64_t>(q_fold[0]) * a.B * kK + static_cast<int64_t>(b) * kK + 4 * kh, r0);
          tile_store(a.gout + static_cast<int64_t>(q_fold[1]) * a.B * kK + static_cast<int64_t>(b) * kK + 4 * kh, r1);
        }
      } else {
        tile_store_native(a.gout + q_fold[0] * fold_stride + static_cast<int64_t>(tile) * 1024, lane, r0);
        tile_store_native(a.gout + q_fold[1] * fold_stride + static_cast<int64_t>(tile) * 1024, lane, r1);
      }
    };
    if constexpr (WAVES == 8) {
      int nth = 0;
      for (int tile = tile_begin + wave; tile < tile_end; tile += WAVES, ++nth) {
        if (is_marked(nth, tile)) continue;
        const bool live = tile * 32 + b_in < a.B;
        CK_BSTAMP(0);
        issue_x(tile, xlo);
        issue_a(tile);
        issue_b(xlo);
        __builtin_amdgcn_sched_barrier(0);
        CK_BSTAMP(1);
        float gy[16], e[16], gq[16], r0[16], r1[16], eq[2][16], yq[2][16];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
          for (int r = 0; r < 16; ++r) eq[n][r] = c[2 * n + 1][r];
          forward_product<!LEAF, SIGNED>(eq[n], c[2 * n]);
          recompute(n, eq[n], yq[n]);
        }
        grad_over<SIGNED>(g, y, live, gy);
        CK_BSTAMP(2);
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = yq[1][r];
        forward_product<true, SIGNED>(e, yq[0]);
        node(0, gy, e, gq);
        CK_BSTAMP(3);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          grad_over<SIGNED>(gq, yq[n], live, gy);
          node(1 + n, gy, eq[n], n == 0 ? r0 : r1);
          CK_BSTAMP(4 + n);
        }
        store(tile, r0, r1);
        CK_BSTAMP(6);
#ifdef CK_BWD_STAMPS
        ++stamp_unit;
#endif
      }
    } else {
      const int first = tile_begin + wave;
      const int n_mine = first < tile_end ? (tile_end - first + WAVES - 1) / WAVES : 0;
      float r0[16], r1[16];
      int rtile = -1;  // the unit whose results r0 / r1 hold (-1: none to store)
      // batch values travel TWO units ahead, so that a unit's table rows are requested together with its tiles (a whole
      // unit's compute before they are needed) instead of behind a wait in the middle of the previous unit
      uint32_t xnext[4] = {0, 0, 0, 0};
      if (n_mine > 0) {
        issue_x(first, xlo);
        issue_a(first);
        issue_b(xlo);
        issue_x(first + min(1, n_mine - 1) * WAVES, xnext);
      }
      for (int k = 0; k < n_mine; ++k) {
        const int tile = first + k * WAVES;
        const bool mk = is_marked(k, tile);  // (its kept tiles mean nothing: it contributes nothing and stores nothing)
        const bool live = tile * 32 + b_in < a.B && !mk;
        CK_BSTAMP(0);
        // consume the raw tiles of this unit (the first touch waits for all of them; nothing younger is in flight)
        float gyp[16], ep[16], eq0[16], eq1[16], dq0[16], dq1[16];  // (dq: 1 / y of Q0, Q1)
        grad_over<SIGNED>(g, y, live, gyp);  // (dead rows and marked units: zero here, and with it every gradient below)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          eq0[r] = c[1][r];
          eq1[r] = c[3][r];
        }
        forward_product<!LEAF, SIGNED>(eq0, c[0]);
        forward_product<!LEAF, SIGNED>(eq1, c[2]);
        {
          float yq0[16], yq1[16];
          recompute(0, eq0, yq0);
          recompute(1, eq1, yq1);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            ep[r] = yq1[r];
            dq0[r] = (mk || (SIGNED && yq0[r] == 0.f)) ? 0.f : __builtin_amdgcn_rcpf(yq0[r]);  // (a marked unit's tiles mean nothing: no 1 / 0 into the products)
            dq1[r] = (mk || (SIGNED && yq1[r] == 0.f)) ? 0.f : __builtin_amdgcn_rcpf(yq1[r]);
          }
          forward_product<true, SIGNED>(ep, yq0);
        }
        __builtin_amdgcn_sched_barrier(0);
        CK_BSTAMP(1);
        // refill them with the next unit's (the last unit fetches itself again: nobody reads that)
#pragma unroll
        for (int i = 0; i < 4; ++i) xlo[i] = xnext[i];  // (requested an iteration ago; the wait above covered it)
        issue_a(first + min(k + 1, n_mine - 1) * WAVES);
        issue_b(xlo);
        issue_x(first + min(k + 2, n_mine - 1) * WAVES, xnext);
        if (rtile >= 0) store(rtile, r0, r1);
        __builtin_amdgcn_sched_barrier(0);
        CK_BSTAMP(2);
        float gq[16], gy[16];
        node(0, gyp, ep, gq);  // (marked units and dead rows: gyp = 0, so every gradient and weight-gradient term below is 0)
        CK_BSTAMP(3);
#pragma unroll
        for (int r = 0; r < 16; ++r) gy[r] = gq[r] * dq0[r];
        node(1, gy, eq0, r0);
        CK_BSTAMP(4);
#pragma unroll
        for (int r = 0; r < 16; ++r) gy[r] = gq[r] * dq1[r];
        node(2, gy, eq1, r1);
        CK_BSTAMP(5);
#ifdef CK_BWD_STAMPS
        ++stamp_unit;
#endif
        rtile = mk ? -1 : tile;
      }
      if (rtile >= 0) store(rtile, r0, r1);
    }
    // the segment's weight gradients: summed over the waves in LDS, one atomic per element.  The barriers only publish LDS
    // data (__syncthreads would also wait for the atomics of the previous pass to be acknowledged: ~3 us each, per segment)
    auto lds_barrier = [] {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    lds_barrier();  // (every wave has left the walk: the scratch tiles are free)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      float* mine = (n < 2 ? s_gy + n * 1024 : s_gy);  // P and Q0 in the wave's two tiles, Q1 in the first one again
      if (n == 2) lds_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[(8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + b_in] = dw[n][r];
      if (n == 0) continue;
      lds_barrier();
      for (int m = (n == 1 ? 0 : 2); m <= n; ++m) {
        float* dst = m == 0 ? a.dw_p + static_cast<int64_t>(p_fold) * 1024 : a.dw_q + static_cast<int64_t>(q_fold[m - 1]) * 1024;
        const float* src = scratch + (m == 1 ? 1024 : 0);
        for (int idx = threadIdx.x; idx < 1024; idx += WAVES * 64) {
          float sacc = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < WAVES; ++w8) sacc += src[w8 * 2048 + idx];
          if (sacc != 0.f) atomicAdd(dst + idx, sacc);
        }
      }
    }
  }
}

}  // namespace

namespace {

// ---- tiles whose forward walk left the linear range ---------------------------------------------------------------------
// One wave per marked (root, tile) -- one workgroup of four waves per root looks for marks --: the whole subtree again in LOG space, the reference's arithmetic (semiring.py:383-408),
// forward values recomputed on the way down (a node's children are walked again for its own backward: rare tiles, simple
// code).  Writes the gradient tiles of the level-1 nodes where leaf_bwd_kernel<true> would have left them and adds the
// weight gradients of every level with float atomics.
struct RedoArgs {
  const float* table;    // (F0, C + 1, 32) linear rows
  const float* scale;    // (F0, C + 1) their log scales
  const int64_t* x64;
  int B, C, D;
  const int32_t* nodes;  // packed node tables of the region (as ck_leaf_walk_fwd)
  int node_off[5];
  int leaf_off;
  const int64_t* scope;  // variable of each input-layer fold
  const float* w[4];     // row-major (F_l, 32, 32) weights of level l + 1
  float* dw[4];
  const float* gin;      // (F_root, B, 32)
  const int32_t* gin_fold;  // nullptr, or (n_roots): gin is tile-native, (., tiles, 1024), and root t's tiles are block gin_fold[t] of it
  float* gout1;          // (F_1, B, 32)
  int32_t* redo;         // (n_roots, tiles): cleared here
};

template <int D, bool SIGNED>
struct RedoWalk {
  const RedoArgs& a;
  int t, lane, b_in, kh, bl;
  bool live;
  float* lds;  // 2 x 4 KB

  __device__ __forceinline__ int fold(int level, int j) const { return a.nodes[a.node_off[level] + t * ((1 << D) >> level) + j]; }

  // log-space output of node j of `level` (level 0: the leaf's table row); SIGNED: log|v| and the lane's 16 sign bits in sg
  template <int L>
  __device__ __noinline__ void value(int j, float (&v)[16], uint32_t& sg) const {
    sg = 0;
    if constexpr (L == 0) {
      const int64_t var = a.scope[a.nodes[a.leaf_off + t * (1 << D) + j]];
      const uint32_t row = min(static_cast<uint32_t>(a.x64[static_cast<int64_t>(bl) * a.D + var]), static_cast<uint32_t>(a.C));
      const int64_t r = static_cast<int64_t>(fold(0, j)) * (a.C + 1) + row;
      tile_load(a.table + r * kK + 4 * kh, v);
      const float sc = a.scale[r];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if constexpr (SIGNED) sg |= (v[q] < 0.f ? 1u : 0u) << q;
        v[q] = logf(SIGNED ? __builtin_fabsf(v[q]) : v[q]) + sc;
      }
    } else {
      float u[16];
      uint32_t su = 0;
      value<L - 1>(2 * j, v, sg);
      value<L - 1>(2 * j + 1, u, su);
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] += u[q];
      sg ^= su;
      WRegs w;
      load_w<CK_W_ROWMAJOR>(a.w[L - 1] + static_cast<int64_t>(fold(L, j)) * 1024, lane, w);
      if constexpr (SIGNED) sum_step_signed<CK_W_ROWMAJOR>(w, v, sg);
      else sum_step<CK_W_ROWMAJOR>(w, v);
    }
  }

  template <int L>
  __device__ __noinline__ void backward(int j, const float (&g)[16]) const {
    float e[16], u[16];
    uint32_t se = 0, su = 0;
    value<L - 1>(2 * j, e, se);
    value<L - 1>(2 * j + 1, u, su);
#pragma unroll
    for (int q = 0; q < 16; ++q) e[q] += u[q];
    se ^= su;
    const float m = row_max16(e);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float x = live ? expf(e[q] - m) : 0.f;
      e[q] = (SIGNED && ((se >> q) & 1u)) ? -x : x;
    }
    const float* wf = a.w[L - 1] + static_cast<int64_t>(fold(L, j)) * 1024;
    WRegs w;
    load_w<CK_W_ROWMAJOR>(wf, lane, w);
#pragma unroll
    for (int q = 0; q < 16; ++q) u[q] = e[q];
    contract_linear<CK_W_ROWMAJOR>(w, u);  // y = W e
    float gy[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gy[q] = (live && (SIGNED ? u[q] != 0.f : u[q] > 0.f) && g[q] != 0.f) ? g[q] / u[q] : 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    dw_accumulate(acc, lds, lds + 1024, b_in, kh, gy, e);
    float* dwf = a.dw[L - 1] + static_cast<int64_t>(fold(L, j)) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (acc[r] != 0.f) atomicAdd(dwf + (8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + b_in, acc[r]);
    // gc = e * (W^T gy)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[(8 * (q >> 2) + 4 * kh + (q & 3)) * kK + b_in], gy[q], acc, 0, 0, 0);
    float gc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gc[r] = acc[r] * e[r];
    if constexpr (L == 1) {
      if (live) tile_store(a.gout1 + (static_cast<int64_t>(fold(1, j)) * a.B + bl) * kK + 4 * kh, gc);
    } else {
      backward<L - 1>(2 * j, gc);
      backward<L - 1>(2 * j + 1, gc);
    }
  }
};

template <int D, bool SIGNED>
__global__ void __launch_bounds__(256) leaf_bwd_redo_kernel(const RedoArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4][2048];
  const int t = blockIdx.x, n_tiles = (a.B + 31) >> 5;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int32_t* flags = a.redo + static_cast<int64_t>(t) * n_tiles;
  // one workgroup per root: a wave looks at 64 marks at a time (almost always: none set) and walks the marked tiles
  for (int base = wave * 64; base < n_tiles; base += 256) {
    const int tl = base + lane;
    uint64_t marked = __ballot(tl < n_tiles && flags[tl] != 0);
    while (marked != 0) {
      const int tile = base + __builtin_ctzll(marked);
      marked &= marked - 1;
      const int b = tile * 32 + (lane & 31);
      const RedoWalk<D, SIGNED> walk{a, t, lane, lane & 31, lane >> 5, min(b, a.B - 1), b < a.B, lds[wave]};
      float g[16];
      if (a.gin_fold != nullptr)  // (tile-native blocks of n_tiles tiles: the gradient arena of ck_signed.hip's layers)
        tile_load_native(a.gin + (static_cast<int64_t>(a.gin_fold[t]) * n_tiles + tile) * 1024, lane, g);
      else
        tile_load(a.gin + (static_cast<int64_t>(walk.fold(D, 0)) * a.B + walk.bl) * kK + 4 * walk.kh, g);
      walk.template backward<D>(0, g);
      if (lane == 0) flags[tile] = 0;
    }
  }
}


// ---- the dense layer and the Categorical log-softmax, backward, ON THE TABLE ----------------------------------------------
// The forward pushes the dense layer through the (C + 1)-row log-table of its Categorical fold (ck_softmax.h kind 5:
// T'[c] = dense(T[c]), T = log softmax_C(theta), layers/input.py:399-412 + layers/inner.py:266-273), so the gradient the
// scatter leaves is w.r.t. T' (F, C + 1, 32) and both layers' backward is a (C + 1)-row problem per fold -- B / C times
// less than per batch row.  One workgroup per dense fold: T and W = softmax(theta_d) are rebuilt in LDS from the raw
// parameters, the C + 1 rows go through the three contractions of a sum layer's backward on 32-row tiles
// (y = W e, dW += gy^T e, gT = e * (W^T gy)), then
//     dtheta_d[o, i] = W[o, i] (dW[o, i] - <W[o, :], dW[o, :]>)                   (TorchSoftmaxParameter, nodes.py:764-772)
//     dtheta_c[i, c] = gT[c, i] - exp(T[c, i]) sum_c' gT[c', i]      (c < C)       (log-softmax over the categories)
// are written straight into the parameter gradients: nothing table-sized but the scatter's output is read, nothing but the
// two gradients written.  Row C (the integral row, T = 0) contributes to dW only.
struct TableBwdArgs {
  const float* cat_logits;    // (F_cat, 32, C)
  const int64_t* cat_idx;     // (F) Categorical fold of each dense fold, or nullptr: the identity
  const float* dense_logits;  // (F, 32, 32)
  const float* dtp;           // (F, C + 1, 32) gradient w.r.t. the log-space table T'
  float* g_cat;               // (F_cat, 32, C)
  float* g_dense;             // (F, 32, 32)
  int C;
  // the optimizer in the epilogue (ck_table_opt; all NULL: gradients only): the workgroup that holds d theta of a fold updates its
  // logits and moments in place and builds the fold's table of the NEXT forward from them
  const ck_opt_state* opt;
  float* th_cat;              // = cat_logits, written
  float* th_dense;            // = dense_logits, written
  float* m1_cat;
  float* m2_cat;
  float* m1_dense;
  float* m2_dense;
  float* table;               // (F, C + 1, 32) linear rows of T' = dense(log-table) (ck_softmax.h table_dense_rows, KIND5)
  float* table_scale;         // (F, C + 1) their log scales
};

// Eight waves and exactly 80 KB of LDS per workgroup -- two workgroups per compute unit, four waves per SIMD: the kernel is a
// chain of dependent phases (parameters in, table, tiles, reductions, gradients out: measured 97 us with one 148 KB workgroup
// per CU, 161 us with two 4-wave ones), so what matters is how many waves overlap them.  The tile of T a wave has consumed is
// overwritten in place by its tile of gT; exp(T) of the last phase is recomputed from the logits; the dW contraction's two
// operands share one 4 KB tile per wave.
constexpr int kTbWaves = 8;
// exp on v_exp_f32 (one multiply + the native exp2, as every forward kernel: ~5e-7 relative at |x| <= 10): the library expf is
// ~25 instructions, and a lane evaluates 48 of them per fold here
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * kL2E); }
__global__ void __launch_bounds__(kTbWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) table_dense_bwd_kernel(const TableBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tb_lds[];
  const int C = a.C, rows = C + 1, n_t = (rows + 31) >> 5;
  float* t_s = tb_lds;                 // [n_t * 32][32] T, then gT, swizzled (tsw)
  float* w_t = t_s + n_t * 1024;       // W, CK_W_TILED_F32 (A operand of y = W e)
  float* wt_t = w_t + 1024;            // W^T, "transposed tiled" (A operand of W^T gy)
  float* w_rm = wt_t + 1024;           // W row-major
  float* scratch = w_rm + 1024;        // kTbWaves x 1024: per wave the operand tile of the dW contraction
  const int d = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b_in = lane & 31, kh = lane >> 5;
  const int64_t f = a.cat_idx != nullptr ? a.cat_idx[d] : d;
  const float* theta = a.cat_logits + f * 32 * C;
  // W = softmax over the last axis of the fold's (32, 32) logits: 16 lanes per row, two entries each
  for (int o = threadIdx.x >> 4; o < 32; o += kTbWaves * 4) {
    const int j = threadIdx.x & 15;
    const float2 v = *reinterpret_cast<const float2*>(a.dense_logits + static_cast<int64_t>(d) * 1024 + o * 32 + 2 * j);
    float m = fmaxf(v.x, v.y);
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 16));
    const float e0 = expf(v.x - m), e1 = expf(v.y - m);
    float sum = e0 + e1;
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) sum += __shfl_xor(sum, s, 16);
    const float p[2] = {e0 / sum, e1 / sum};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = 2 * j + k;
      w_rm[o * 32 + i] = p[k];
      w_t[(i >> 3) * 256 + (o + 32 * ((i >> 2) & 1)) * 4 + (i & 3)] = p[k];
      wt_t[(o >> 3) * 256 + (i + 32 * ((o >> 2) & 1)) * 4 + (o & 3)] = p[k];
    }
  }
  // T[c][i] = theta[i][c] - logsumexp_c theta[i][:]: 32 / kTbWaves units per wave, lanes over the categories
  // (the logits of the wave's units are read ONCE, all loads in flight together, and stay in registers for the last phase:
  //  three passes over a row and a fourth at the end were sixteen dependent trips to L2 per wave; C <= 256: four per lane and unit)
  constexpr int kUnits = 32 / kTbWaves;
  float lse_mine[kUnits];
  float th[kUnits][4];
  // (lane l holds categories 4 l .. 4 l + 3 of its units: the layout table_dense_rows_x takes for the next table)
#pragma unroll
  for (int ii = 0; ii < kUnits; ++ii) {
    if ((C & 3) == 0) {
      float4 t4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (4 * lane < C) t4 = *reinterpret_cast<const float4*>(theta + (wave + ii * kTbWaves) * C + 4 * lane);
      th[ii][0] = t4.x, th[ii][1] = t4.y, th[ii][2] = t4.z, th[ii][3] = t4.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * lane + q;
        th[ii][q] = c < C ? theta[(wave + ii * kTbWaves) * C + c] : -INFINITY;
      }
    }
  }
#pragma unroll
  for (int ii = 0; ii < kUnits; ++ii) {
    const int i = wave + ii * kTbWaves;
    float m = fmaxf(fmaxf(th[ii][0], th[ii][1]), fmaxf(th[ii][2], th[ii][3]));
    m = ck::wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += fexp(th[ii][q] - m);  // (lanes beyond C: exp(-inf) = 0)
    sum = ck::wave_sum(sum);
    const float lse = m + logf(sum);
    lse_mine[ii] = lse;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * lane + q;
      if (c < n_t * 32) t_s[tsw(c, i)] = c < C ? th[ii][q] - lse : 0.f;  // (row C: the integral row; beyond: padding)
    }
    for (int c = 256 + lane; c < n_t * 32; c += 64) t_s[tsw(c, i)] = 0.f;  // (C <= 256: rows the four per lane do not reach)
  }
  __syncthreads();
  f32x16 dw;
#pragma unroll
  for (int r = 0; r < 16; ++r) dw[r] = 0.f;
  float* s_gy = scratch + wave * 1024;
  const float* dtp = a.dtp + static_cast<int64_t>(d) * rows * kK;
  for (int tile = wave; tile < n_t; tile += kTbWaves) {
    const int c = tile * 32 + b_in;
    const bool live = c < rows;
    float v[16], e[16], gy[16], go[16];
    tile_load(dtp + static_cast<int64_t>(live ? c : rows - 1) * kK + 4 * kh, go);
    if (tile == n_t - 1 && (rows & 31) != 0 && (rows & 31) <= 4) {
      // a tile without any gradient contributes nothing and leaves gT = 0: with C = 256 the ninth tile holds the integral
      // row alone, whose gradient is zero unless the batch had marginalised variables -- and it would be a second round of
      // the whole chain for one of the eight waves.  (Only asked of such a tile: the test waits for the gradient rows,
      // which every other tile needs three contractions later.)
      bool any = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) any = any || (live && go[r] != 0.f);
      if (__ballot(any) == 0) {
        const float zero[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        tile_to_lds(t_s + tile * 1024, b_in, kh, zero);
        continue;
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // register layout out of the swizzled tile
      const float4 t4 = *reinterpret_cast<const float4*>(t_s + c * 32 + 4 * ((2 * g + kh) ^ (c & 7)));
      v[4 * g + 0] = t4.x;
      v[4 * g + 1] = t4.y;
      v[4 * g + 2] = t4.z;
      v[4 * g + 3] = t4.w;
    }
    const float m = row_max16(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = live ? fexp(v[r] - m) : 0.f;
    WRegs w;
#pragma unroll
    for (int q = 0; q < 4; ++q) w.q[q] = *reinterpret_cast<const float4*>(w_t + q * 256 + lane * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = e[r];
    contract_linear<CK_W_TILED_F32>(w, v);  // y = W e
#pragma unroll
    for (int r = 0; r < 16; ++r) gy[r] = (live && v[r] > 0.f && go[r] != 0.f) ? go[r] / v[r] : 0.f;
    dw_accumulate_seq(dw, s_gy, b_in, kh, gy, e);
    float gt[16];
    child_gradient(wt_t, lane, gy, e, gt);
    tile_to_lds(t_s + tile * 1024, b_in, kh, gt);  // (gT over the wave's own tile of T)
  }
  __syncthreads();
  // dW over the waves -> row-major over W^T's tile (free by now), then the softmax backward of the dense weights
#pragma unroll
  for (int r = 0; r < 16; ++r) s_gy[(8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + b_in] = dw[r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 1024; idx += kTbWaves * 64) {
    float sacc = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < kTbWaves; ++w8) sacc += scratch[w8 * 1024 + idx];
    wt_t[idx] = sacc;
  }
  __syncthreads();
  const bool step_on = a.opt != nullptr && a.opt->skip_now == 0;  // (uniform; a dropped step changes nothing: the table stays)
  OptK ok{};
  if (a.opt != nullptr) ok = opt_k(*a.opt);
  float nd[2] = {0.f, 0.f};  // the thread's two new dense logits (row o = tid >> 4 of the first pass; kTbWaves * 4 = 32 rows: ONE pass)
  static_assert(kTbWaves * 4 == 32, "one pass over the dense rows");
  for (int o = threadIdx.x >> 4; o < 32; o += kTbWaves * 4) {
    const float* dwr = wt_t;
    const int j = threadIdx.x & 15;
    const float w0 = w_rm[o * 32 + 2 * j], w1 = w_rm[o * 32 + 2 * j + 1];
    const float d0 = dwr[o * 32 + 2 * j], d1 = dwr[o * 32 + 2 * j + 1];
    float dot = w0 * d0 + w1 * d1;
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) dot += __shfl_xor(dot, s, 16);
    const float g0 = w0 * (d0 - dot), g1 = w1 * (d1 - dot);
    const int64_t at = static_cast<int64_t>(d) * 1024 + o * 32 + 2 * j;
    *reinterpret_cast<float2*>(a.g_dense + at) = make_float2(g0, g1);
    if (step_on) {
      const float2 t2 = *reinterpret_cast<const float2*>(a.dense_logits + at);
      float2 m1 = make_float2(0.f, 0.f), m2 = make_float2(0.f, 0.f);
      if (ok.kind != 0) {
        m1 = *reinterpret_cast<const float2*>(a.m1_dense + at);
        m2 = *reinterpret_cast<const float2*>(a.m2_dense + at);
      }
      nd[0] = opt_update(ok, t2.x, g0, m1.x, m2.x);
      nd[1] = opt_update(ok, t2.y, g1, m1.y, m2.y);
      *reinterpret_cast<float2*>(a.th_dense + at) = make_float2(nd[0], nd[1]);
      if (ok.kind != 0) {
        *reinterpret_cast<float2*>(a.m1_dense + at) = m1;
        *reinterpret_cast<float2*>(a.m2_dense + at) = m2;
      }
    }
  }
  // column sums of gT over the categories (row C has no gradient), then dtheta_c, coalesced along the categories
  // (the moments of unit ii + 1 travel while unit ii is reduced and updated: a unit is otherwise two dependent round trips)
  const bool vec = (C & 3) == 0 && 4 * lane < C;
  float4 pm1 = make_float4(0.f, 0.f, 0.f, 0.f), pm2 = pm1;
  auto fetch_moments = [&](int ii, float4& m1, float4& m2) {
    if (step_on && ok.kind != 0 && vec) {
      const int64_t at = f * 32 * C + static_cast<int64_t>(wave + ii * kTbWaves) * C + 4 * lane;
      m1 = *reinterpret_cast<const float4*>(a.m1_cat + at);
      m2 = *reinterpret_cast<const float4*>(a.m2_cat + at);
    }
  };
  fetch_moments(0, pm1, pm2);
#pragma unroll
  for (int ii = 0; ii < kUnits; ++ii) {
    const int i = wave + ii * kTbWaves;
    float4 m1 = pm1, m2 = pm2;
    if (ii + 1 < kUnits) fetch_moments(ii + 1, pm1, pm2);
    float gt[4];
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * lane + q;
      gt[q] = c < C ? t_s[tsw(c, i)] : 0.f;
      sacc += gt[q];
    }
    sacc = ck::wave_sum(sacc);
    const float lse = lse_mine[ii];
    const int64_t row = f * 32 * C + static_cast<int64_t>(i) * C;
    if ((C & 3) == 0) {  // (uniform; the lane's four categories as one 16-byte access per array)
      if (4 * lane < C) {
        const int64_t at = row + 4 * lane;
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = gt[q] - fexp(th[ii][q] - lse) * sacc;
        *reinterpret_cast<float4*>(a.g_cat + at) = make_float4(g[0], g[1], g[2], g[3]);
        if (step_on) {
          th[ii][0] = opt_update(ok, th[ii][0], g[0], m1.x, m2.x);
          th[ii][1] = opt_update(ok, th[ii][1], g[1], m1.y, m2.y);
          th[ii][2] = opt_update(ok, th[ii][2], g[2], m1.z, m2.z);
          th[ii][3] = opt_update(ok, th[ii][3], g[3], m1.w, m2.w);
          *reinterpret_cast<float4*>(a.th_cat + at) = make_float4(th[ii][0], th[ii][1], th[ii][2], th[ii][3]);
          if (ok.kind != 0) {
            *reinterpret_cast<float4*>(a.m1_cat + at) = m1;
            *reinterpret_cast<float4*>(a.m2_cat + at) = m2;
          }
        }
      }
    } else {  // (gradients only: the launcher refuses the optimizer for such a C)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * lane + q;
        if (c < C) a.g_cat[row + c] = gt[q] - fexp(th[ii][q] - lse) * sacc;
      }
    }
  }
  if (!step_on) return;
  // the fold's table of the next forward from the logits this workgroup has just updated (the kind-5 job of the prologue,
  // ck_softmax.h: the same function on the same layout of the logits, so the table is the prologue's bit for bit)
  __syncthreads();  // (gT and the dense layer's tiles have been read: the LDS is free)
  float* tile = tb_lds;                       // 32 x (C + 4) log-probabilities + 1024 linear weights (table_dense_rows_x)
  float* nd_s = tb_lds + 32 * (C + 4) + 1024;  // the new dense logits, row-major
  {
    const int o = threadIdx.x >> 4, j = threadIdx.x & 15;
    nd_s[o * 32 + 2 * j] = nd[0];
    nd_s[o * 32 + 2 * j + 1] = nd[1];
  }
  __syncthreads();
  float4 x[kUnits];
#pragma unroll
  for (int ii = 0; ii < kUnits; ++ii) x[ii] = make_float4(th[ii][0], th[ii][1], th[ii][2], th[ii][3]);
  float* dst = a.table + static_cast<int64_t>(d) * rows * kK;
  float* dsc = a.table_scale + static_cast<int64_t>(d) * rows;
  table_dense_rows_x<kTbWaves, true, true>(x, true, nd_s, C, tile, wave, lane, [] { __syncthreads(); },
                                           [&](int c, const float (&v)[16], float m) {
                                             if (c <= C) {
                                               tile_store(dst + static_cast<int64_t>(c) * kK + 4 * kh, v);
                                               if (kh == 0) dsc[c] = m;
                                             }
                                           });
}

}  // namespace

extern "C" {

int ck_leaf_walk_bwd(const ck_leaf_bwd_launch* d, void* stream) {
  CK_REQUIRE(d != nullptr, "ck_leaf_walk_bwd: null descriptor");
  CK_REQUIRE(d->unit_tab && d->work && d->gin && d->y_p && d->w_p && d->w_q && d->dw_p && d->dw_q && d->gout,
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
  a.y_c = d->y_c;
  a.table = d->table;
  a.x64 = d->x_rows;
  a.w_p = d->w_p;
  a.w_q = d->w_q;
  a.dw_p = d->dw_p;
  a.dw_q = d->dw_q;
  a.gout = d->gout;
  a.redo = d->redo;
  a.gin_rowmajor = d->gin_rowmajor;
#ifdef CK_BWD_STAMPS
  a.exp = getenv("CK_BWD_EXP") != nullptr ? atoi(getenv("CK_BWD_EXP")) : 0;
  a.stamps = getenv("CK_BWD_STAMP_PTR") != nullptr && (d->leaf != 0) == (getenv("CK_BWD_STAMP_TOP") == nullptr)
                 ? reinterpret_cast<long long*>(strtoull(getenv("CK_BWD_STAMP_PTR"), nullptr, 0)) : nullptr;
  a.stamp_wg = getenv("CK_BWD_STAMP_WG") != nullptr ? atoi(getenv("CK_BWD_STAMP_WG")) : 8;
#endif
  const bool leaf = d->leaf != 0;
  CK_REQUIRE(d->waves == 4 || d->waves == 8, "ck_leaf_walk_bwd: waves must be 4 or 8 (got %d)", d->waves);
  const int waves = d->waves;
  const bool is_signed = d->is_signed != 0;
  dim3 grid(static_cast<unsigned>(std::min(d->n_wg, d->n_seg)));
  return ck::dispatch(
      [=](hipStream_t s) {
        if (is_signed) {
          if (waves != 8) return hipErrorInvalidValue;
          if (leaf) hipLaunchKernelGGL((leaf_bwd_kernel<true, 8, true>), grid, dim3(512), 0, s, a);
          else hipLaunchKernelGGL((leaf_bwd_kernel<false, 8, true>), grid, dim3(512), 0, s, a);
        } else if (waves == 4) {
          if (leaf) hipLaunchKernelGGL((leaf_bwd_kernel<true, 4>), grid, dim3(256), 0, s, a);
          else hipLaunchKernelGGL((leaf_bwd_kernel<false, 4>), grid, dim3(256), 0, s, a);
        } else {
          if (leaf) hipLaunchKernelGGL((leaf_bwd_kernel<true, 8>), grid, dim3(512), 0, s, a);
          else hipLaunchKernelGGL((leaf_bwd_kernel<false, 8>), grid, dim3(512), 0, s, a);
        }
        return hipGetLastError();
      },
      stream);
}

int ck_table_dense_bwd(const float* cat_logits, const int64_t* cat_idx, const float* dense_logits, const float* dtable, float* g_cat,
                       float* g_dense, int F, int C, const ck_table_opt* opt, void* stream) {
  CK_REQUIRE(cat_logits && dense_logits && dtable && g_cat && g_dense, "ck_table_dense_bwd: null pointer");
  if (opt != nullptr) {
    CK_REQUIRE(opt->state && opt->table && opt->table_scale && opt->m1_cat && opt->m2_cat && opt->m1_dense && opt->m2_dense,
               "ck_table_dense_bwd: null pointer in the optimizer descriptor");
    CK_REQUIRE(cat_idx == nullptr, "ck_table_dense_bwd: the optimizer epilogue needs one Categorical fold per dense fold (cat_idx NULL)");
    if ((C & 3) != 0) return ck::fail(CK_ERR_UNSUPPORTED, "ck_table_dense_bwd: the optimizer epilogue needs C %% 4 == 0 (C=%d)", C);
  }
  CK_REQUIRE(F > 0 && C > 0, "ck_table_dense_bwd: non-positive size");
  if (C > 256) return ck::fail(CK_ERR_UNSUPPORTED, "ck_table_dense_bwd: C=%d (at most 256 categories: a lane keeps four logits per unit)", C);
  const int n_t = (C + 1 + 31) / 32;
  const size_t lds = (static_cast<size_t>(n_t + 3) * 1024 + kTbWaves * 1024) * sizeof(float);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_table_dense_bwd: C=%d does not fit in LDS", C);
  TableBwdArgs a{cat_logits, cat_idx, dense_logits, dtable, g_cat, g_dense, C};
  if (opt != nullptr) {
    a.opt = opt->state;
    a.th_cat = const_cast<float*>(cat_logits);
    a.th_dense = const_cast<float*>(dense_logits);
    a.m1_cat = opt->m1_cat;
    a.m2_cat = opt->m2_cat;
    a.m1_dense = opt->m1_dense;
    a.m2_dense = opt->m2_dense;
    a.table = opt->table;
    a.table_scale = opt->table_scale;
  }
  dim3 grid(static_cast<unsigned>(F));
  return ck::dispatch(
      [=](hipStream_t s) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(table_dense_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(table_dense_bwd_kernel, grid, dim3(kTbWaves * 64), lds, s, a);
        return hipGetLastError();
      },
      stream);
}

int ck_leaf_walk_bwd_redo(const float* table, const float* table_scale, const int64_t* x_rows, int B, int C, int D,
                          const int32_t* nodes, const int32_t* node_off, int leaf_off, const int64_t* scope, int depth,
                          const float* const* w_levels, float* const* dw_levels, const float* gin, float* gout1, int32_t* redo,
                          int n_roots, const int32_t* gin_fold, int is_signed, void* stream) {
  CK_REQUIRE(table && table_scale && x_rows && nodes && node_off && scope && w_levels && dw_levels && gin && gout1 && redo,
             "ck_leaf_walk_bwd_redo: null pointer");
  CK_REQUIRE(B > 0 && C > 0 && D > 0 && n_roots > 0 && n_roots <= 65535, "ck_leaf_walk_bwd_redo: bad sizes");
  CK_REQUIRE(depth == 2 || depth == 4, "ck_leaf_walk_bwd_redo: depth %d (2 or 4)", depth);
  RedoArgs a{};
  a.table = table;
  a.scale = table_scale;
  a.x64 = x_rows;
  a.B = B;
  a.C = C;
  a.D = D;
  a.nodes = nodes;
  for (int l = 0; l <= depth; ++l) a.node_off[l] = node_off[l];
  a.leaf_off = leaf_off;
  a.scope = scope;
  for (int l = 0; l < depth; ++l) {
    CK_REQUIRE(w_levels[l] && dw_levels[l], "ck_leaf_walk_bwd_redo: null level %d", l + 1);
    a.w[l] = w_levels[l];
    a.dw[l] = dw_levels[l];
  }
  a.gin = gin;
  a.gin_fold = gin_fold;
  a.gout1 = gout1;
  a.redo = redo;
  dim3 grid(static_cast<unsigned>(n_roots));
  return ck::dispatch(
      [=](hipStream_t s) {
        if (is_signed) {
          if (depth == 2) hipLaunchKernelGGL((leaf_bwd_redo_kernel<2, true>), grid, dim3(256), 0, s, a);
          else hipLaunchKernelGGL((leaf_bwd_redo_kernel<4, true>), grid, dim3(256), 0, s, a);
        } else if (depth == 2) hipLaunchKernelGGL((leaf_bwd_redo_kernel<2, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((leaf_bwd_redo_kernel<4, false>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
