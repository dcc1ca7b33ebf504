// Fused tail: the last levels of a circuit have few folds (24, 11, 6, 4, 2, 1 at the north-star
// config), so one launch per level is pure launch latency.  Batch rows are independent, so ONE
// workgroup per 32-row batch tile walks all remaining layers: the folds of a layer are dealt
// round-robin to the workgroup's wavefronts, each fold is one register-tile log-einsum-exp step
// (ck_tile.h), outputs go to the arena (they are real layer outputs) and a workgroup barrier
// separates levels.  Children are addressed through the same arena offset tables as everywhere
// else, so any DAG shape (children from several earlier layers) is supported.
//
// Same arithmetic as ck_sum_lse_fwd in CK_SUM_PROD mode (TorchCPTLayer optimized.py:171-178 /
// dense TorchSumLayer inner.py:266-273 + semiring.py:383-408).
#include <algorithm>

#include "ck_internal.h"
#include "ck_tile.h"

namespace {

constexpr int kMaxTail = 12;
constexpr int kTailWaves = 16;

struct TailLayer {
  const int64_t* row_off;  // (F, H) arena element offsets
  const float* w;          // (F, Ko, 32) linear weights
  float* out;              // (F, B, Ko)
  int F, H, Ko;
};
struct TailArgs {
  TailLayer layer[kMaxTail];
  const float* arena;
  int n, B;
};

template <int LAYOUT>
__global__ void __launch_bounds__(kTailWaves * 64) tail_kernel(const TailArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = blockIdx.x * 32 + b_in;
  const bool live = b < a.B;
  const int bl = live ? b : a.B - 1;
  for (int li = 0; li < a.n; ++li) {
    const TailLayer& L = a.layer[li];
    for (int f = wave; f < L.F; f += kTailWaves) {
      float v[16];
      WRegs w;
      if (L.Ko == kK) load_w<LAYOUT>(L.w + static_cast<int64_t>(f) * kK * kK, lane, w);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = 0.f;
      for (int h = 0; h < L.H; ++h)
        tile_load_add(a.arena + L.row_off[static_cast<int64_t>(f) * L.H + h] + static_cast<int64_t>(bl) * kK + 4 * kh, v);
      if (L.Ko == kK) {
        sum_step<LAYOUT>(w, v);
        if (live) tile_store(L.out + (static_cast<int64_t>(f) * a.B + b) * kK + 4 * kh, v);
      } else {
        // Ko < 32 (the root: Ko = 1): plain dot products, lanes (b, 0) and (b, 1) each hold half a row
        // (these few-output layers always take ROW-MAJOR fp32 weights)
        const float m = row_max16(v);
        const float nml = exp_offset(m, 0.f);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
        for (int o = 0; o < L.Ko; ++o) {
          const float* wrow = L.w + (static_cast<int64_t>(f) * L.Ko + o) * kK + 4 * kh;
          float acc = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 w4 = *reinterpret_cast<const float4*>(wrow + 8 * g);
            acc = fmaf(w4.x, v[4 * g + 0], acc);
            acc = fmaf(w4.y, v[4 * g + 1], acc);
            acc = fmaf(w4.z, v[4 * g + 2], acc);
            acc = fmaf(w4.w, v[4 * g + 3], acc);
          }
          acc += __shfl_xor(acc, 32, 64);
          if (live && kh == 0)
            L.out[(static_cast<int64_t>(f) * a.B + b) * L.Ko + o] = fmaf(__builtin_amdgcn_logf(acc), kLN2, m);
        }
      }
    }
    __syncthreads();  // level boundary: this workgroup's stores are visible to its own loads
  }
}

}  // namespace

extern "C" {

int ck_tail_lse_fwd(const float* arena, int n_layers, const int64_t* const* row_off,
                    const float* const* w, float* const* out, const int32_t* F, const int32_t* H,
                    const int32_t* Ko, int B, int K, int w_layout, void* stream) {
  CK_REQUIRE(arena && row_off && w && out && F && H && Ko, "ck_tail_lse_fwd: null pointer");
  CK_REQUIRE(n_layers > 0 && n_layers <= kMaxTail, "ck_tail_lse_fwd: n_layers=%d outside [1, %d]", n_layers, kMaxTail);
  CK_REQUIRE(B > 0, "ck_tail_lse_fwd: B must be positive");
  if (K != kK) return ck::fail(CK_ERR_UNSUPPORTED, "ck_tail_lse_fwd: K=%d (only K=32)", K);
  CK_REQUIRE(w_layout == CK_W_ROWMAJOR || w_layout == CK_W_TILED_F32, "ck_tail_lse_fwd: unknown w_layout %d", w_layout);
  TailArgs a{};
  a.arena = arena;
  a.n = n_layers;
  a.B = B;
  for (int i = 0; i < n_layers; ++i) {
    CK_REQUIRE(row_off[i] && w[i] && out[i], "ck_tail_lse_fwd: null pointer in layer %d", i);
    CK_REQUIRE(F[i] > 0 && H[i] > 0 && Ko[i] > 0 && Ko[i] <= kK, "ck_tail_lse_fwd: bad sizes in layer %d", i);
    CK_REQUIRE(ck::aligned16(w[i]) && ck::aligned16(out[i]), "ck_tail_lse_fwd: layer %d buffers not 16-byte aligned", i);
    a.layer[i] = TailLayer{row_off[i], w[i], out[i], F[i], H[i], Ko[i]};
  }
  CK_REQUIRE(ck::aligned16(arena), "ck_tail_lse_fwd: arena not 16-byte aligned");
  dim3 grid((B + 31) / 32), block(kTailWaves * 64);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (w_layout == CK_W_ROWMAJOR)
          hipLaunchKernelGGL(tail_kernel<CK_W_ROWMAJOR>, grid, block, 0, s, a);
        else
          hipLaunchKernelGGL(tail_kernel<CK_W_TILED_F32>, grid, block, 0, s, a);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
