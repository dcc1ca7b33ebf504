// Backward of a circuit's trailing few-fold sum layers in ONE launch (the fused training step, cirkit_amd/training.py).
//
// The reference gets these gradients from autograd through TorchCPTLayer.forward (optimized.py:171-178) / the dense
// TorchSumLayer (inner.py:266-273) under LSESumSemiring.apply_reduce (semiring.py:383-408), one layer after the other.
// Layer by layer that is one launch per layer here too (ck_sum_lse_bwd: 9-18 us each for 1..24 folds at 4096 rows -- a launch
// whose 32 workgroups per fold all add their weight gradients to the same 1024 addresses -- six of them in a row: 71 us of a
// 0.7 ms step).  The layers only depend on each other WITHIN a batch tile, so one workgroup takes one 32-row tile through
// all of them, top down, a barrier per layer; the gradient tiles between the layers go through the (L2-resident) gradient
// arena as before.  Weight gradients are NOT added atomically: every (tile, fold) leaves its 32 x 32 contribution in its own
// slot of a (tiles, folds) buffer, and the launch that consumes the weight gradients (ck_param_softmax_bwd_batch) sums the
// slots -- 128 workgroups adding into the few folds of a level at the same moment is exactly the pattern float atomics are
// slowest at (~8 ns per atomic on one line: scripts/ubench/atomic_scatter.hip).
#include <algorithm>
#include <cstdlib>

#include "ck_bwd_tile.h"
#include "ck_internal.h"
#include "ck_tile.h"

namespace {

struct TailBwdFold {  // mirrors ck_tail_bwd_fold (cirkit_hip.h)
  const float* w;         // (Ko, 32) row-major LINEAR weights
  const float* gout;      // (B, Ko) gradient w.r.t. the fold's log-space output
  float* dw_part;         // this fold's (Ko, 32) slot of tile 0 in the partial buffer
  const float* child[4];  // (B, 32) log-space outputs of the children
  float* gchild[4];       // (B, 32) their gradient blocks (written)
  int32_t H, Ko;
};
static_assert(sizeof(TailBwdFold) == sizeof(ck_tail_bwd_fold), "TailBwdFold mirrors ck_tail_bwd_fold");

struct TailBwdArgs {
  const TailBwdFold* folds;    // top level first
  const int32_t* level_begin;  // (n_levels + 1)
  int n_folds, n_levels, B;
  int64_t part_stride;         // floats between the slots of consecutive tiles
};


// Pointers out of the descriptor table are generic: say "device memory" on every access (FLAT instructions count on vmcnt AND
// lgkmcnt -- every LDS wait behind one would also wait for the prefetched tiles; ck_internal.h gload4 / gstore4).
__device__ __forceinline__ void gtile_load(const float* row, float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t = ck::gload4(row + 8 * g);
    v[4 * g + 0] = t.x;
    v[4 * g + 1] = t.y;
    v[4 * g + 2] = t.z;
    v[4 * g + 3] = t.w;
  }
}
__device__ __forceinline__ void gtile_store(float* row, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) ck::gstore4(row + 8 * g, make_float4(v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]));
}

// kTbwWaves wavefronts per workgroup: a level of n folds takes ceil(n / kTbwWaves) rounds of one ~6 us unit chain each
template <int kTbwWaves>
__global__ void __launch_bounds__(kTbwWaves * 64) tail_bwd_kernel(const TailBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tbw_lds[];
  float* scratch = tbw_lds;                                                    // kTbwWaves x 2 x 1024: the operands of a wave's dW contraction
  TailBwdFold* s_fold = reinterpret_cast<TailBwdFold*>(scratch + kTbwWaves * 2048);
  int32_t* s_level = reinterpret_cast<int32_t*>(s_fold + a.n_folds);
  {
    const int n16 = a.n_folds * static_cast<int>(sizeof(TailBwdFold) / 16);
    const int4* src = reinterpret_cast<const int4*>(a.folds);
    int4* dst = reinterpret_cast<int4*>(s_fold);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i <= a.n_levels; i += blockDim.x) s_level[i] = a.level_begin[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b_in = lane & 31, kh = lane >> 5;
  const int tile = blockIdx.x;
  const int b = tile * 32 + b_in;
  const bool live = b < a.B;
  const int bl = live ? b : a.B - 1;
  float* s_gy = scratch + wave * 2048;
  float* s_e = s_gy + 1024;
  // What a fold needs that does NOT depend on the level above it -- its children's forward values and its weights in both
  // operand layouts -- travels while the wave's previous fold computes, across the level barrier too; only the fold's own
  // gradient tile has to wait for the barrier (and is requested a fold ahead inside a level).  The number of loads and
  // stores per fold is FIXED (a fold with one child names it twice; a wave without a next fold fetches its current one
  // again): with a data-dependent number of operations in flight the compiler's waits all become vmcnt(0), i.e. every first
  // use of a gradient tile would wait for the prefetch issued just before it.
  struct FoldRegs {
    float c0[16], c1[16];  // the children's log-space tiles
    WRegs wr;              // A operand of y = W e
    float wt[16];          // A operand of W^T gy: lane (n, kh) holds W[u(s, kh)][n]
  };
  auto prefetch = [&](int t, FoldRegs& r) {
    const TailBwdFold& fd = s_fold[t];
    const float* wf = fd.w;
    gtile_load(fd.child[0] + static_cast<int64_t>(bl) * kK + 4 * kh, r.c0);
    gtile_load(fd.child[1] + static_cast<int64_t>(bl) * kK + 4 * kh, r.c1);
#pragma unroll
    for (int g = 0; g < 4; ++g) r.wr.q[g] = ck::gload4(wf + (lane & 31) * kK + 4 * kh + 8 * g);  // (load_w<CK_W_ROWMAJOR>)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) r.wt[4 * g + q] = ck::as_global(wf)[(8 * g + 4 * kh + q) * kK + b_in];
  };
  auto first_fold_from = [&](int li) {  // this wave's first fold at or behind level li, or -1
    for (; li < a.n_levels; ++li)
      if (s_level[li] + wave < s_level[li + 1]) return s_level[li] + wave;
    return -1;
  };
  int li0 = 0;
  if (s_fold[0].Ko != kK) {
    // a scalar root (Ko = 1, one fold: checked by the host): y = sum_u w_u e_u per row, by wave 0; the weight gradient is
    // row 0 of gy^T e with gy in unit 0 only
    if (wave == 0) {
      const TailBwdFold& fd = s_fold[0];
      float c0[16], c1[16], e[16], gy[16], gv[16], wl[16];
      gtile_load(fd.child[0] + static_cast<int64_t>(bl) * kK + 4 * kh, c0);
      gtile_load(fd.child[1] + static_cast<int64_t>(bl) * kK + 4 * kh, c1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 w4 = ck::gload4(fd.w + 8 * g + 4 * kh);
        wl[4 * g + 0] = w4.x;
        wl[4 * g + 1] = w4.y;
        wl[4 * g + 2] = w4.z;
        wl[4 * g + 3] = w4.w;
      }
      const float go = ck::as_global(fd.gout)[bl];
#pragma unroll
      for (int j = 0; j < 16; ++j) e[j] = c0[j] + (fd.H > 1 ? c1[j] : 0.f);
      const float m = row_max16(e);
#pragma unroll
      for (int j = 0; j < 16; ++j) e[j] = live ? __builtin_amdgcn_exp2f((e[j] - m) * kL2E) : 0.f;
      float y = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) y = fmaf(wl[r], e[r], y);
      y += __shfl_xor(y, 32, 64);
      const float g1 = (live && y > 0.f && go != 0.f) ? go / y : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gy[r] = 0.f;
        gv[r] = e[r] * wl[r] * g1;
      }
      if (kh == 0) gy[0] = g1;  // (unit 0 lives in register 0 of the lanes with kh = 0)
      if (live) {
        gtile_store(fd.gchild[0] + static_cast<int64_t>(b) * kK + 4 * kh, gv);
        gtile_store(fd.gchild[1] + static_cast<int64_t>(b) * kK + 4 * kh, gv);
      }
      f32x16 dw;
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[r] = 0.f;
      dw_accumulate(dw, s_gy, s_e, b_in, kh, gy, e);
      if (kh == 0) ck::as_global(fd.dw_part + static_cast<int64_t>(tile) * a.part_stride)[b_in] = dw[0];  // (row o = 0)
    }
    li0 = 1;
  }
  FoldRegs cur, nxt;
  {
    const int t0 = first_fold_from(li0);
    if (t0 >= 0) prefetch(t0, cur);
  }
  if (li0 == 1) __syncthreads();
  for (int li = li0; li < a.n_levels; ++li) {
    const int t1 = s_level[li + 1];
    int t = s_level[li] + wave;
    float go[16], go_next[16];
    if (t < t1) gtile_load(s_fold[t].gout + static_cast<int64_t>(bl) * kK + 4 * kh, go);
    for (; t < t1; t += kTbwWaves) {
      const bool more = t + kTbwWaves < t1;
      int t_next = more ? t + kTbwWaves : first_fold_from(li + 1);
      if (t_next < 0) t_next = t;  // (nobody reads that)
      prefetch(t_next, nxt);
      gtile_load(s_fold[more ? t + kTbwWaves : t].gout + static_cast<int64_t>(bl) * kK + 4 * kh, go_next);
      const TailBwdFold& fd = s_fold[t];
      const int H = fd.H;
      float e[16], gy[16], gv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) e[j] = cur.c0[j] + (H > 1 ? cur.c1[j] : 0.f);
      const float m = row_max16(e);
#pragma unroll
      for (int j = 0; j < 16; ++j) e[j] = live ? __builtin_amdgcn_exp2f((e[j] - m) * kL2E) : 0.f;  // (v_exp_f32, as the forward: ~5e-7 relative)
      {
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = e[r];
        contract_linear<CK_W_ROWMAJOR>(cur.wr, y);
        grad_over_y(go, y, live, gy);
      }
      {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.wt[s2], gy[s2], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) gv[r] = acc[r] * e[r];
      }
      // (dead rows of the last tile write row B - 1's zero gradient to ... no: they are clamped away below)
      if (live) {
        gtile_store(fd.gchild[0] + static_cast<int64_t>(b) * kK + 4 * kh, gv);
        gtile_store(fd.gchild[1] + static_cast<int64_t>(b) * kK + 4 * kh, gv);  // (H = 1: the same block again)
      }
      f32x16 dw;
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[r] = 0.f;
      dw_accumulate(dw, s_gy, s_e, b_in, kh, gy, e);
      // D[o][i] in lane (i, hi) register r, o = 8 (r >> 2) + 4 hi + (r & 3): this tile's slot, row-major (32, 32)
      float* part = fd.dw_part + static_cast<int64_t>(tile) * a.part_stride;
#pragma unroll
      for (int r = 0; r < 16; ++r) ck::as_global(part)[(8 * (r >> 2) + 4 * kh + (r & 3)) * kK + b_in] = dw[r];
      cur = nxt;
#pragma unroll
      for (int r = 0; r < 16; ++r) go[r] = go_next[r];
    }
    __syncthreads();  // the level's gradient tiles are in memory (workgroup scope: the next level's waves are this workgroup's)
  }
}

}  // namespace

extern "C" int ck_tail_bwd(const ck_tail_bwd_fold* folds, int n_folds, const int32_t* level_begin, int n_levels, int B,
                           int64_t part_stride, void* stream) {
  CK_REQUIRE(folds != nullptr && level_begin != nullptr, "ck_tail_bwd: null pointer");
  CK_REQUIRE(n_folds > 0 && n_levels > 0 && B > 0 && part_stride > 0, "ck_tail_bwd: non-positive size");
  CK_REQUIRE(ck::aligned16(folds), "ck_tail_bwd: folds not 16-byte aligned");
  // (per fold -- DEVICE data, not checked here: H in {1, 2} with child[1] / gchild[1] naming the first child again when H = 1;
  //  Ko = 32 everywhere but, possibly, in a single first fold that is a level of its own)
  // (lab switch.  12 waves -- three per SIMD, 160 registers -- walk config 2's tail (1, 2, 4, 6, 11, 24 folds) in 7 rounds instead of
  //  9 and take the same 59 us: the launch is the levels' dependent round trips + the MFMA chains of one CU, not rounds;
  //  16 waves spill 38 registers: step 0.683 against 0.641 ms.  LAB_NOTES R5.4)
  static const int waves = [] {
    const char* e = getenv("CK_TAIL_BWD_WAVES");
    const int w = e ? atoi(e) : 8;
    return (w == 12 || w == 16) ? w : 8;
  }();
  const size_t lds = static_cast<size_t>(waves) * 2048 * sizeof(float) + static_cast<size_t>(n_folds) * sizeof(TailBwdFold) +
                     static_cast<size_t>(n_levels + 1) * sizeof(int32_t);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_tail_bwd: %d folds do not fit in LDS", n_folds);
  TailBwdArgs a{};
  a.folds = reinterpret_cast<const TailBwdFold*>(folds);
  a.level_begin = level_begin;
  a.n_folds = n_folds;
  a.n_levels = n_levels;
  a.B = B;
  a.part_stride = part_stride;
  const dim3 grid(static_cast<unsigned>((B + 31) / 32));
  auto kern = waves == 8 ? tail_bwd_kernel<8> : (waves == 16 ? tail_bwd_kernel<16> : tail_bwd_kernel<12>);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (lds > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
          if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(waves * 64), lds, s, a);
        return hipGetLastError();
      },
      stream);
}
