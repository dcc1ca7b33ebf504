// Cross-layer fusion of the leaf region of a circuit:
//
//     input layer (Categorical table gather)  ->  [dense sum layer]  ->  D levels of CP-T layers
//
// evaluated per (root fold, 32-row batch tile) by ONE wavefront, depth-first, entirely in
// registers.  None of the intermediate layers' (F, B, K) outputs ever reaches HBM: at the
// north-star config (784-var QuadTree, K=32, B=4096) that is 1.9 GB of the 2.5 GB the layer-wise
// evaluation moves.
//
// Why this works in registers: the tile layout of ck_tile.h makes the OUTPUT registers of one
// 32x32x32 log-einsum-exp step the INPUT registers of the next, so a subtree is evaluated with a
// binary-counter walk: leaves are visited left to right; after leaf i, every level whose bit of i
// is set combines the saved left sibling with the fresh right sibling and applies that level's sum
// step.  The leaf loop is fully unrolled, so the sibling stack, the carry conditions and the order
// of the steps are static: the weights of step k+1 and the table row of leaf i+1 are prefetched
// while step k / leaf i computes.
//
// Reference semantics per step (unchanged): TorchCategoricalLayer input.py:399-412, TorchSumLayer
// inner.py:266-273, TorchCPTLayer optimized.py:171-178, LSESumSemiring.apply_reduce
// semiring.py:383-408.
#include <algorithm>

#include "ck_internal.h"
#include "ck_tile.h"

namespace {

constexpr int kMaxDepth = 4;

struct SubtreeArgs {
  const float* table;    // (F0, C+1, K) leaf log-prob table (transposed; row C = integral row)
  const int32_t* xt;     // (Dvars, B) staged batch
  const int64_t* scope;  // (F0) variable of each leaf fold
  const float* w_dense;  // (F_dense, 1024 dwords) weights of the dense layer, or nullptr
  const float* w[kMaxDepth];  // w[l-1]: (F_l, 1024 dwords) weights of CP-T level l = 1..D
  const int32_t* nodes;       // packed node tables, see node_off
  int node_off[kMaxDepth + 1];  // nodes + node_off[l] -> (F_root, 2^(D-l)) fold ids at level l
                                // (level 0 = folds of the dense layer, or of the input layer)
  int leaf_off;                 // nodes + leaf_off -> (F_root, 2^D) input-layer fold of each leaf
  float* out;                   // (F_root, B, K)
  const float* scale;           // LINEAR kernels: (F0, C+1) log scale of each table row
  int B, C, F_root, groups_per_root;
};

// ---- static order of the steps of the walk ------------------------------------------------------
// A step is (leaf i, level lvl): lvl = -1 is the dense layer applied to leaf i, lvl = l >= 0 the
// CP-T level l+1 applied after leaf i (exists iff bits 0..l of i are all set).
template <int D, bool HAS_DENSE>
__device__ __forceinline__ constexpr bool first_step_of_leaf(int i, int& lvl) {
  if (HAS_DENSE) {
    lvl = -1;
    return true;
  }
  if (D > 0 && (i & 1)) {
    lvl = 0;
    return true;
  }
  return false;
}
template <int D, bool HAS_DENSE>
__device__ __forceinline__ constexpr bool next_step(int i, int lvl, int& ni, int& nl) {
  const int l2 = lvl + 1;
  if (l2 < D && ((i >> l2) & 1) && ((i & ((1 << l2) - 1)) == (1 << l2) - 1)) {
    ni = i;
    nl = l2;
    return true;
  }
  for (int j = i + 1; j < (1 << D); ++j)
    if (first_step_of_leaf<D, HAS_DENSE>(j, nl)) {
      ni = j;
      return true;
    }
  return false;
}

template <int D, bool HAS_DENSE, int LAYOUT>
__global__ void __launch_bounds__(256) subtree_cat_cpt_kernel(const SubtreeArgs a) {
  // XCD-aware mapping: workgroup id -> (root fold t, tile group).  Consecutive workgroup ids go
  // to consecutive XCDs (id % 8); all tile groups of one root fold are given the same id % 8 so
  // its leaf tables and weights are fetched into ONE XCD's L2.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, i_in = bid >> 3;
  const int chunk = i_in / a.groups_per_root, tg = i_in - chunk * a.groups_per_root;
  const int t = chunk * 8 + xcd;
  if (t >= a.F_root) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b0 = (tg * 4 + wave) * 32;
  if (b0 >= a.B) return;
  const int b = b0 + b_in;
  const bool live = b < a.B;
  const int bl = live ? b : a.B - 1;

  constexpr int kLeaves = 1 << D;
  const int32_t* leaf_ids = a.nodes + a.leaf_off + t * kLeaves;
  const int32_t* dense_ids = a.nodes + a.node_off[0] + t * kLeaves;

  auto w_ptr = [&](int i, int lvl) -> const float* {
    if (lvl < 0) return a.w_dense + static_cast<int64_t>(dense_ids[i]) * (kK * kK);
    const int fold = a.nodes[a.node_off[lvl + 1] + t * (kLeaves >> (lvl + 1)) + (i >> (lvl + 1))];
    return a.w[lvl] + static_cast<int64_t>(fold) * (kK * kK);
  };

  // batch values of all leaves (one coalesced int per lane and leaf)
  int xv[kLeaves];
#pragma unroll
  for (int i = 0; i < kLeaves; ++i) {
    const int v = a.xt[a.scope[leaf_ids[i]] * static_cast<int64_t>(a.B) + bl];
    xv[i] = v < 0 ? a.C : min(v, a.C - 1);  // negative = marginalised -> integral row C; clamp for memory safety
  }
  // table fold: the input-layer fold when the dense layer runs in this kernel, else the level-0
  // fold (= the input fold for cp-t plans, or the dense fold when the host has already pushed the
  // dense layer through the table, see cirkit_amd/circuit.py `dense_on_table`)
  auto row_ptr = [&](int i) {
    const int tf = HAS_DENSE ? leaf_ids[i] : dense_ids[i];
    return a.table + (static_cast<int64_t>(tf) * (a.C + 1) + xv[i]) * kK + 4 * kh;
  };

  WRegs wcur, wnxt;
  {
    int l0 = 0, i0 = 0;
    bool any = first_step_of_leaf<D, HAS_DENSE>(0, l0);
    if (!any) any = next_step<D, HAS_DENSE>(0, D, i0, l0);  // (lvl = D: no carry) -> first leaf with a step
    if (any) load_w<LAYOUT>(w_ptr(i0, l0), lane, wcur);
  }

  // gathered table rows are prefetched kPF leaves ahead (ring of kPF register tiles)
  // (measured: 2 leaves ahead costs a wave of occupancy at depth 4 and is 1.4x slower; 1 is best)
  constexpr int kPF = 1;
  float stack[D > 0 ? D : 1][16];
  float cur[16], nxt[kPF][16];
#pragma unroll
  for (int p = 0; p < kPF; ++p)
    if (p < kLeaves) tile_load(row_ptr(p), nxt[p]);

#define CK_STEP(I, LVL)                                          \
  {                                                              \
    int ni_ = 0, nl_ = 0;                                        \
    const bool more_ = next_step<D, HAS_DENSE>(I, LVL, ni_, nl_); \
    if (more_) load_w<LAYOUT>(w_ptr(ni_, nl_), lane, wnxt);      \
    sum_step<LAYOUT>(wcur, cur);                                 \
    if (more_) wcur = wnxt;                                      \
  }

#pragma unroll
  for (int i = 0; i < kLeaves; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] = nxt[i % kPF][j];
    if (i + kPF < kLeaves) tile_load(row_ptr(i + kPF), nxt[i % kPF]);  // later leaves' gathers in flight
    if (HAS_DENSE) CK_STEP(i, -1)
#pragma unroll
    for (int l = 0; l < D; ++l) {
      if (((i >> l) & 1) == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) stack[l][j] = cur[j];
        break;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) cur[j] += stack[l][j];
      CK_STEP(i, l)
    }
  }
#undef CK_STEP
  if (live) tile_store(a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh, cur);
}

template <int D, int LAYOUT>
hipError_t launch_dl(const SubtreeArgs& a, bool has_dense, dim3 grid, hipStream_t s) {
  if (has_dense)
    hipLaunchKernelGGL((subtree_cat_cpt_kernel<D, true, LAYOUT>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((subtree_cat_cpt_kernel<D, false, LAYOUT>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

template <int LAYOUT>
hipError_t launch_depth(const SubtreeArgs& a, int depth, bool has_dense, dim3 grid, hipStream_t s) {
  switch (depth) {
    case 0:
      return launch_dl<0, LAYOUT>(a, has_dense, grid, s);
    case 1:
      return launch_dl<1, LAYOUT>(a, has_dense, grid, s);
    case 2:
      return launch_dl<2, LAYOUT>(a, has_dense, grid, s);
    case 3:
      return launch_dl<3, LAYOUT>(a, has_dense, grid, s);
    default:
      return launch_dl<4, LAYOUT>(a, has_dense, grid, s);
  }
}

// ---- linear-domain variant -----------------------------------------------------------------------
// Between two fused CP-T levels the reference takes a log and the next level immediately exponentiates
// again: v = log y_l + m_l + log y_r + m_r, e = exp(v - max v).  Here a node is carried as
// (y: linear tile in [0, 1], s: per-row log scale) and the next level forms
//     p = y_l * y_r,   e = p / max_k p,   s = s_l + s_r + log(max_k p),   y = W . e
// -- the same quantities, with ONE log per row instead of 32 exp + 32 log per row and no rounding of
// a log/exp round trip.  The leaf table comes in the same representation (rows in linear space +
// `scale`, written by the prologue job that pushes the table through the dense layer); the root
// level is converted back, out = log y + s.  If every product of a row underflows fp32 (children
// whose large units do not overlap by more than ~1e-38) the step is redone in log space.

// (leaf, level) of the k-th CP-T step of the depth-first walk without an in-kernel dense layer
template <int D>
__host__ __device__ constexpr int next_step_leaf(int k) {
  for (int i = 0; i < (1 << D); ++i) {
    if (k < steps_after(i)) return i;
    k -= steps_after(i);
  }
  return 0;
}
template <int D>
__host__ __device__ constexpr int next_step_level(int k) {
  for (int i = 0; i < (1 << D); ++i) {
    if (k < steps_after(i)) return k;
    k -= steps_after(i);
  }
  return 0;
}

template <int D, int LAYOUT>
__global__ void __launch_bounds__(256) subtree_linear_kernel(const SubtreeArgs a) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, i_in = bid >> 3;
  const int chunk = i_in / a.groups_per_root, tg = i_in - chunk * a.groups_per_root;
  const int t = chunk * 8 + xcd;
  if (t >= a.F_root) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b0 = (tg * 4 + wave) * 32;
  // Tiled weight layouts: the weights of all 2^D - 1 nodes of this root's subtree (4 KB each, the same for the four
  // waves) are staged in LDS once per workgroup, in step order -- a quarter of the L2 traffic of every wave fetching
  // them itself, no weight registers held across steps, and an LDS read right before use instead of an L2 round trip.
  constexpr bool kLdsW = LAYOUT != CK_W_ROWMAJOR;
  constexpr int kNodes = (1 << D) - 1;
  __shared__ __attribute__((aligned(16))) float w_lds[kLdsW ? kNodes * 1024 : 4];
  if constexpr (kLdsW) {
    float4 stg[kNodes];
    {
      int si = 1, sl = 0, n = 0;  // first step: level 1 after leaf 1
      bool more = true;
#pragma unroll
      for (int k = 0; k < kNodes; ++k) {
        if (more) {
          const int fold = a.nodes[a.node_off[sl + 1] + t * ((1 << D) >> (sl + 1)) + (si >> (sl + 1))];
          stg[n++] = *reinterpret_cast<const float4*>(a.w[sl] + static_cast<int64_t>(fold) * (kK * kK) + 4 * threadIdx.x);
          int ni = 0, nl = 0;
          more = next_step<D, false>(si, sl, ni, nl);
          si = ni;
          sl = nl;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kNodes; ++k) *reinterpret_cast<float4*>(w_lds + k * 1024 + 4 * threadIdx.x) = stg[k];
    __syncthreads();
  }
  if (b0 >= a.B) return;
  const int b = b0 + b_in;
  const bool live = b < a.B;
  const int bl = live ? b : a.B - 1;

  constexpr int kLeaves = 1 << D;
  const int32_t* leaf_ids = a.nodes + a.leaf_off + t * kLeaves;
  const int32_t* fold0 = a.nodes + a.node_off[0] + t * kLeaves;  // table fold of each leaf
  auto w_ptr = [&](int i, int lvl) -> const float* {
    const int fold = a.nodes[a.node_off[lvl + 1] + t * (kLeaves >> (lvl + 1)) + (i >> (lvl + 1))];
    return a.w[lvl] + static_cast<int64_t>(fold) * (kK * kK);
  };
  int32_t row[kLeaves];  // table row (fold, category) of each leaf
#pragma unroll
  for (int i = 0; i < kLeaves; ++i) {
    const int v = a.xt[a.scope[leaf_ids[i]] * static_cast<int64_t>(a.B) + bl];
    const int c = v < 0 ? a.C : min(v, a.C - 1);  // negative = marginalised -> integral row C
    row[i] = fold0[i] * (a.C + 1) + c;
  }
  auto row_of = [&](int i) -> int64_t { return static_cast<int64_t>(row[i]); };
  WRegs wcur, wnxt;
  if constexpr (!kLdsW) load_w<LAYOUT>(w_ptr(1, 0), lane, wcur);  // first step: level 1 after leaf 1
  float stack[D][16], sstack[D];
  // two leaf rows in flight: leaves come in pairs with no contraction between them (L L C L L C C ...), so a
  // gather issued one leaf ahead would be waited for right away at every second leaf
  float cur[16], nxt[16], nx2[16], cs, ns, ns2;
  bool bad = false;
  tile_load(a.table + row_of(0) * kK + 4 * kh, nxt);
  ns = a.scale[row_of(0)];
  if (kLeaves > 1) {
    tile_load(a.table + row_of(1) * kK + 4 * kh, nx2);
    ns2 = a.scale[row_of(1)];
  }
  static_for<0, kLeaves>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      cur[j] = nxt[j];
      nxt[j] = nx2[j];
    }
    cs = ns;
    ns = ns2;
    if constexpr (i + 2 < kLeaves) {
      tile_load(a.table + row_of(i + 2) * kK + 4 * kh, nx2);
      ns2 = a.scale[row_of(i + 2)];
    }
    static_for<0, steps_after(i)>([&](auto lc) {
      constexpr int l = decltype(lc)::value, step = steps_before(i) + l;
      if constexpr (kLdsW) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wcur.q[q] = *reinterpret_cast<const float4*>(w_lds + step * 1024 + q * 256 + lane * 4);
      } else if constexpr (step + 1 < kNodes) {
        constexpr int ni = next_step_leaf<D>(step + 1), nl = next_step_level<D>(step + 1);
        load_w<LAYOUT>(w_ptr(ni, nl), lane, wnxt);
      }
      // the first level is the bare product; deeper levels renormalise by a power of two (ck_tile.h: linear_product)
      if constexpr (l == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) cur[j] *= stack[0][j];
        cs += sstack[0];
      } else {
        linear_product<true>(cur, stack[l], cs, sstack[l], bad);
      }
      contract_linear<LAYOUT>(wcur, cur);
      if constexpr (!kLdsW && step + 1 < kNodes) wcur = wnxt;
    });
    if constexpr (steps_after(i) < D) {
      constexpr int l = steps_after(i);
#pragma unroll
      for (int j = 0; j < 16; ++j) stack[l][j] = cur[j];
      sstack[l] = cs;
    }
  });
  if constexpr (D == 1) bad |= !(tile_row_max(cur) > kLinearFloor);  // (deeper roots are renormalised steps)
  if (__builtin_expect(__any(bad), 0)) {
    // rare: a row of products fell out of the fp32 range -> the whole tile again in log space (semiring.py:383-408)
    SubtreeSource src{};
    src.table = a.table;
    src.scale = a.scale;
    src.xt = a.xt;
    src.scope = a.scope;
    src.leaf_ids = leaf_ids;
    src.fold0 = fold0;
    src.w_steps = kLdsW ? w_lds : nullptr;
#pragma unroll
    for (int l = 0; l < D; ++l) src.w[l] = a.w[l];
    src.nodes = a.nodes;
#pragma unroll
    for (int l = 0; l <= D; ++l) src.node_off[l] = a.node_off[l];
    src.t = t;
    src.B = a.B;
    src.C = a.C;
    src.bl = bl;
    float fb[16];  // (its address escapes into the out-of-line call: never `cur`, which must stay in registers)
    subtree_tile_logspace<D, LAYOUT>(src, lane, fb);
    if (live) tile_store(a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh, fb);
    return;
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] = fmaf(__builtin_amdgcn_logf(cur[j]), kLN2, cs);
    tile_store(a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh, cur);
  }
}

template <int LAYOUT>
hipError_t launch_linear(const SubtreeArgs& a, int depth, dim3 grid, hipStream_t s) {
  switch (depth) {
    case 1:
      hipLaunchKernelGGL((subtree_linear_kernel<1, LAYOUT>), grid, dim3(256), 0, s, a);
      break;
    case 2:
      hipLaunchKernelGGL((subtree_linear_kernel<2, LAYOUT>), grid, dim3(256), 0, s, a);
      break;
    case 3:
      hipLaunchKernelGGL((subtree_linear_kernel<3, LAYOUT>), grid, dim3(256), 0, s, a);
      break;
    default:
      hipLaunchKernelGGL((subtree_linear_kernel<4, LAYOUT>), grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace

extern "C" {

int ck_subtree_cat_cpt_fwd(const float* table, const float* table_scale, const int32_t* xt, const int64_t* scope,
                           const float* w_dense, const float* const* w_levels,
                           const int32_t* nodes, const int32_t* node_off, int leaf_off, float* out,
                           int depth, int F_root, int B, int K, int C, int w_layout, void* stream) {
  CK_REQUIRE(table_scale == nullptr || (w_dense == nullptr && depth >= 1),
             "ck_subtree_cat_cpt_fwd: a linear table needs depth >= 1 and no in-kernel dense layer");
  CK_REQUIRE(table && xt && scope && nodes && node_off && out, "ck_subtree_cat_cpt_fwd: null pointer");
  CK_REQUIRE(depth >= 0 && depth <= kMaxDepth, "ck_subtree_cat_cpt_fwd: depth %d outside [0, %d]", depth, kMaxDepth);
  CK_REQUIRE(depth == 0 || w_levels != nullptr, "ck_subtree_cat_cpt_fwd: w_levels is null");
  CK_REQUIRE(depth > 0 || w_dense != nullptr, "ck_subtree_cat_cpt_fwd: nothing to fuse (depth 0, no dense layer)");
  CK_REQUIRE(F_root > 0 && B > 0 && C > 0, "ck_subtree_cat_cpt_fwd: non-positive size");
  CK_REQUIRE(w_layout == CK_W_ROWMAJOR || w_layout == CK_W_TILED_F32, "ck_subtree_cat_cpt_fwd: unknown w_layout %d", w_layout);
  if (K != kK) return ck::fail(CK_ERR_UNSUPPORTED, "ck_subtree_cat_cpt_fwd: K=%d (only K=32 is fused)", K);
  CK_REQUIRE(ck::aligned16(table) && ck::aligned16(out) && (!w_dense || ck::aligned16(w_dense)),
             "ck_subtree_cat_cpt_fwd: buffers must be 16-byte aligned");
  SubtreeArgs a{};
  a.table = table;
  a.xt = xt;
  a.scope = scope;
  a.w_dense = w_dense;
  for (int l = 0; l < depth; ++l) {
    CK_REQUIRE(w_levels[l] != nullptr && ck::aligned16(w_levels[l]), "ck_subtree_cat_cpt_fwd: bad weights of level %d", l + 1);
    a.w[l] = w_levels[l];
  }
  a.nodes = nodes;
  for (int l = 0; l <= depth; ++l) a.node_off[l] = node_off[l];
  a.leaf_off = leaf_off;
  a.out = out;
  a.scale = table_scale;
  a.B = B;
  a.C = C;
  a.F_root = F_root;
  const int tiles = (B + 31) / 32;
  a.groups_per_root = (tiles + 3) / 4;
  const int roots_padded = (F_root + 7) / 8 * 8;
  dim3 grid(static_cast<unsigned>(roots_padded) * a.groups_per_root);
  const bool has_dense = w_dense != nullptr;
  return ck::dispatch(
      [=](hipStream_t s) {
        if (table_scale != nullptr) {
          if (w_layout == CK_W_ROWMAJOR) return launch_linear<CK_W_ROWMAJOR>(a, depth, grid, s);
          return launch_linear<CK_W_TILED_F32>(a, depth, grid, s);
        }
        if (w_layout == CK_W_ROWMAJOR) return launch_depth<CK_W_ROWMAJOR>(a, depth, has_dense, grid, s);
        return launch_depth<CK_W_TILED_F32>(a, depth, has_dense, grid, s);
      },
      stream);
}

}  // extern "C"
