// Cross-layer fusion of the leaf region of a circuit:
//
//     input layer (Categorical table gather)  ->  [dense sum layer]  ->  D levels of CP-T layers
//
// evaluated per (root fold, 32-row batch tile) by ONE wavefront, depth-first, entirely in
// registers.  None of the intermediate layers' (F, B, K) outputs ever reaches HBM: at the
// north-star config (784-var QuadTree, K=32, B=4096) that removes 3 x 411 MB of writes and
// 3 x 411 MB of reads per level-0/1/2 alone.
//
// Why this works in registers: `sum_lse_mfma`'s lane layout (ck_sum.hip) makes the OUTPUT
// registers of one 32x32x32 log-einsum-exp step the INPUT registers of the next, so a subtree is
// evaluated with a binary-counter walk: leaves are visited left to right; after leaf i, every
// level whose bit of i is set combines the saved left sibling with the fresh right sibling and
// applies that level's sum step.  The sibling stack is indexed statically (unrolled over levels),
// so it lives in VGPRs.
//
// Reference semantics per step (unchanged): TorchCategoricalLayer input.py:399-412, TorchSumLayer
// inner.py:266-273, TorchCPTLayer optimized.py:171-178, LSESumSemiring.apply_reduce
// semiring.py:383-408.
#include <algorithm>

#include "ck_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxDepth = 4;
constexpr int kK = 32;

struct SubtreeArgs {
  const float* table;    // (F0, C, K) leaf log-prob table (transposed)
  const int32_t* xt;     // (Dvars, B) staged batch
  const int64_t* scope;  // (F0) variable of each leaf fold
  const float* w_dense;  // (F0, K, K) linear-space weights of the dense layer, or nullptr
  const float* w[kMaxDepth];  // w[l-1]: (F_l, K, K) weights of CP-T level l = 1..D
  const int32_t* nodes;       // packed node tables, see node_off
  int node_off[kMaxDepth + 1];  // nodes + node_off[l] -> (F_root, 2^(D-l)) fold ids at level l
                                // (level 0 = folds of the dense layer, or of the input layer)
  int leaf_off;                 // nodes + leaf_off -> (F_root, 2^D) input-layer fold of each leaf
  float* out;                   // (F_root, B, K)
  int B, C, F_root, groups_per_root, tiles_per_wave;
  int ablate;  // debug: bit0 skip W loads, bit1 skip table gather, bit2 skip MFMA, bit3 skip exp/log
};

// One log-einsum-exp step on a 32-row tile held in registers (layout of sum_lse_mfma).
__device__ __forceinline__ void sum_step(const float* __restrict__ wf, float (&v)[16], int b_in,
                                         int kh, int ablate = 0) {
  float wa[16];
  const float* wrow = wf + b_in * kK + 4 * kh;
  if (ablate & 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) wa[j] = 0.03125f;
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t4 = *reinterpret_cast<const float4*>(wrow + 8 * g);
      wa[4 * g + 0] = t4.x;
      wa[4 * g + 1] = t4.y;
      wa[4 * g + 2] = t4.z;
      wa[4 * g + 3] = t4.w;
    }
  }
  float m = v[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) m = fmaxf(m, v[j]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  m = ck::clamp_finite(m);
  if (ablate & 8) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (v[j] - m) * 0.01f + 1.f;
  } else {
    // exp(v - m) = exp2(v * log2(e) - m * log2(e)): one FMA + one v_exp_f32 per element
    const float kL2E = 1.44269504088896340736f;
    const float nml = -m * kL2E;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
  }
  f32x16 acc;
  if (ablate & 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = v[r] * wa[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // Waves of one SIMD are in different phases (independent tiles): give the MFMA phase issue
    // priority so the matrix pipe never waits behind another wave's exp/log VALU stream.
    if (!(ablate & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], v[s], acc, 0, 0, 0);
    if (!(ablate & 16)) __builtin_amdgcn_s_setprio(0);
  }
  if (ablate & 8) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] * 0.5f + m;
  } else {
    // log(acc) + m = log2(acc) * ln(2) + m: one v_log_f32 + one FMA per element
    const float kLN2 = 0.69314718055994530942f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m);
  }
}

template <int D, bool HAS_DENSE>
__global__ void __launch_bounds__(256)
    subtree_cat_cpt_kernel(const SubtreeArgs a) {
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
  constexpr int kLeaves = 1 << D;
  const int32_t* leaf_ids = a.nodes + a.leaf_off + t * kLeaves;
  const int32_t* dense_ids = a.nodes + a.node_off[0] + t * kLeaves;
  // each wave walks `tiles_per_wave` 32-row tiles of the same root fold (amortises the wave launch
  // and the scalar table loads; weights and leaf tables stay hot in L1/L2)
  for (int tw = 0; tw < a.tiles_per_wave; ++tw) {
  const int b0 = ((tg * 4 + wave) * a.tiles_per_wave + tw) * 32;
  if (b0 >= a.B) return;
  const int b = b0 + b_in;
  const bool live = b < a.B;
  const int bl = live ? b : a.B - 1;

  // Batch values of all leaves first (one coalesced int per lane and leaf), then a one-leaf-ahead
  // prefetch of the gathered table rows: the dependent x -> row -> dense chain of leaf i+1 is in
  // flight while leaf i runs its MFMAs.  The leaf loop is fully unrolled so that the sibling stack,
  // the prefetch registers and the carry conditions are all static.
  int xv[kLeaves];
#pragma unroll
  for (int i = 0; i < kLeaves; ++i) {
    const int c = leaf_ids[i];
    const int v = a.xt[a.scope[c] * static_cast<int64_t>(a.B) + bl];
    xv[i] = min(max(v, 0), a.C - 1);
  }
  auto load_row = [&](int i, float (&dst)[16]) {
    const float* row = a.table + (static_cast<int64_t>(leaf_ids[i]) * a.C + xv[i]) * kK + 4 * kh;
    if (a.ablate & 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) dst[j] = -3.f - 0.01f * (xv[i] + j);
      return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t4 = *reinterpret_cast<const float4*>(row + 8 * g);
      dst[4 * g + 0] = t4.x;
      dst[4 * g + 1] = t4.y;
      dst[4 * g + 2] = t4.z;
      dst[4 * g + 3] = t4.w;
    }
  };
  float stack[D > 0 ? D : 1][16];
  float cur[16], nxt[16];
  load_row(0, nxt);
#pragma unroll
  for (int i = 0; i < kLeaves; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] = nxt[j];
    if (i + 1 < kLeaves) load_row(i + 1, nxt);
    if (HAS_DENSE) sum_step(a.w_dense + static_cast<int64_t>(dense_ids[i]) * kK * kK, cur, b_in, kh, a.ablate);
    // ---- carry: combine completed sibling pairs bottom-up ----
#pragma unroll
    for (int l = 0; l < D; ++l) {
      if (((i >> l) & 1) == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) stack[l][j] = cur[j];
        break;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) cur[j] += stack[l][j];
      const int fold = a.nodes[a.node_off[l + 1] + t * (kLeaves >> (l + 1)) + (i >> (l + 1))];
      sum_step(a.w[l] + static_cast<int64_t>(fold) * kK * kK, cur, b_in, kh, a.ablate);
    }
  }
  if (live) {
    float* dst = a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(dst + 8 * g) =
          make_float4(cur[4 * g + 0], cur[4 * g + 1], cur[4 * g + 2], cur[4 * g + 3]);
  }
  }  // tiles of this wave
}

template <int D>
hipError_t launch_depth(const SubtreeArgs& a, bool has_dense, dim3 grid, hipStream_t s) {
  if (has_dense)
    hipLaunchKernelGGL((subtree_cat_cpt_kernel<D, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((subtree_cat_cpt_kernel<D, false>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

int g_ablate = 0;

}  // namespace

extern "C" {

// Debug hook for kernel ablation studies (scripts/ablate_subtree.py); 0 restores normal operation.
int ck_debug_ablate(int mask) {
  g_ablate = mask;
  return CK_OK;
}

int ck_subtree_cat_cpt_fwd(const float* table, const int32_t* xt, const int64_t* scope,
                           const float* w_dense, const float* const* w_levels,
                           const int32_t* nodes, const int32_t* node_off, int leaf_off, float* out,
                           int depth, int F_root, int B, int K, int C, void* stream) {
  CK_REQUIRE(table && xt && scope && nodes && node_off && out, "ck_subtree_cat_cpt_fwd: null pointer");
  CK_REQUIRE(depth >= 0 && depth <= kMaxDepth, "ck_subtree_cat_cpt_fwd: depth %d outside [0, %d]", depth, kMaxDepth);
  CK_REQUIRE(depth == 0 || w_levels != nullptr, "ck_subtree_cat_cpt_fwd: w_levels is null");
  CK_REQUIRE(depth > 0 || w_dense != nullptr, "ck_subtree_cat_cpt_fwd: nothing to fuse (depth 0, no dense layer)");
  CK_REQUIRE(F_root > 0 && B > 0 && C > 0, "ck_subtree_cat_cpt_fwd: non-positive size");
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
  a.B = B;
  a.C = C;
  a.F_root = F_root;
  a.ablate = g_ablate;
  const int tiles = (B + 31) / 32;
  // tiles per wave: keep >= ~4 workgroups per CU-slot round while amortising per-wave setup
  int tpw = 1;
  if (g_ablate >> 8) tpw = (g_ablate >> 8) & 0xff;
  else while (tpw < 8 && static_cast<int64_t>(F_root) * ((tiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) >= 4096) tpw *= 2;
  a.tiles_per_wave = tpw;
  a.groups_per_root = (tiles + 4 * tpw - 1) / (4 * tpw);
  const int roots_padded = (F_root + 7) / 8 * 8;
  dim3 grid(static_cast<unsigned>(roots_padded) * a.groups_per_root);
  const bool has_dense = w_dense != nullptr;
  return ck::dispatch(
      [=](hipStream_t s) {
        switch (depth) {
          case 0:
            return launch_depth<0>(a, has_dense, grid, s);
          case 1:
            return launch_depth<1>(a, has_dense, grid, s);
          case 2:
            return launch_depth<2>(a, has_dense, grid, s);
          case 3:
            return launch_depth<3>(a, has_dense, grid, s);
          default:
            return launch_depth<4>(a, has_dense, grid, s);
        }
      },
      stream);
}

}  // extern "C"
