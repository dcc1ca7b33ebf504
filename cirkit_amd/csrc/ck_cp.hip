// CP sum-product block in one launch:  out[f, b, :] = sum_s G_s,
//     G_s = log(W_{f,s} . exp(v_s - max v_s)) + max v_s      (slot s carries a dense layer), or
//     G_s = v_s                                              (plain slot),
//     v_s = sum_h arena[row_off[f, s, h] + b*K + :]
//
// This is what RegionGraph.build_circuit emits for sum_product = 'cp'
// (cirkit/templates/region_graph/graph.py:424-456): one dense TorchSumLayer per child region
// (inner.py:266-273) feeding a TorchHadamardLayer (inner.py:126-127; a log-space product is a sum,
// semiring.py:375-376).  The reference materialises every dense output (F_dense, B, K) before the
// product reads it back; here the dense results live in registers only, which removes 4 of the 7
// (F, B, K) transfers of a two-child block.  S = 1 is the plain dense / CP-T layer (H children
// multiplied first), so `ck_sum_lse_fwd` routes its K = 64 case through the same kernel.
//
// Work decomposition (measured, scripts/ubench/sum64.hip): ONE 32-row tile per wavefront and the
// slot's weights shared by the workgroup through LDS.  A wave then needs < 128 VGPRs, so 4+ waves
// per SIMD are resident and the memory latency of a tile is hidden by the MFMA chains of the
// others; the register-resident-weights variant it replaces (124 VGPRs of which 64 were weights,
// several tiles per wave back to back) left the matrix pipe idle 63 % of the time.
//
// Numerics per slot are LSESumSemiring.apply_reduce (semiring.py:383-408) exactly as in ck_sum.hip:
// exact fp32 contraction on v_mfma_f32_32x32x2_f32, the lane layout of ck_tile.h.
#include "ck_internal.h"
#include "ck_tile.h"

namespace {

// Slots that read a TABLE instead of the arena: a dense layer applied to a Categorical layer only takes
// C distinct values per fold, so its output rows are precomputed per category (ck_param.hip kind-4 job:
// T'[d] = dense_d(log-table of the leaf fold)) and slot (f, s) gathers row x[b, var] of its table --
// a plain slot (no weights) whose input never existed as an (F, B, K) tensor.
struct GatherSlots {
  const int64_t* addr;  // (F, S...) device address of row 0 of the slot's (C+1, K) table, used where var >= 0
  const int32_t* var;   // (F, S...) variable of the slot, -1 = read the arena through row_off
  const int32_t* xt;    // (D, B) staged batch
  int C;                // categories (row C = the integral row, taken by negative values)
  int F;                // region_dma_kernel<.., BLOCK>: folds of the launch (its grid is one-dimensional)
};

__device__ __forceinline__ const float* slot_source(const GatherSlots& gs, const float* arena, int64_t off, int64_t e,
                                                    int bl, int B, int K, int kh) {
  if (gs.var != nullptr) {
    const int v = gs.var[e];
    if (v >= 0) {
      const int x = gs.xt[static_cast<int64_t>(v) * B + bl];
      const int c = x < 0 ? gs.C : min(x, gs.C - 1);
      return reinterpret_cast<const float*>(static_cast<uintptr_t>(gs.addr[e])) + static_cast<int64_t>(c) * K + 4 * kh;
    }
  }
  return arena + off + static_cast<int64_t>(bl) * K + 4 * kh;
}

// NK: K / 32.  WAVES: wavefronts (= 32-row tiles) per workgroup.  MULTI: more than one slot (the
// single-slot instance keeps no running sum and fits 7 waves per SIMD instead of 4).
template <int NK, int WAVES, bool MULTI>
__global__ void __launch_bounds__(WAVES * 64)
    cp_lse_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                  const int64_t* __restrict__ w_addr, const float* __restrict__ w_base,
                  const int64_t* __restrict__ w_post, const int64_t* __restrict__ out_off,
                  float* __restrict__ out, const GatherSlots gs, int S, int H, int B) {
  constexpr int K = 32 * NK;
  constexpr int WF4 = K * K / 4;  // float4 elements of one weight matrix
  // [buffer][p][q][g][lane] float4: the A operand of MFMA step (p, q, 4g .. 4g+3) for every lane
  __shared__ __attribute__((aligned(16))) float w_s[2][K * K];
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (blockIdx.x * WAVES + wave) * 32 + b_in;
  const bool live = b < B;
  const int bl = live ? b : B - 1;  // clamp loads; stores are masked
  const int64_t* ro = row_off + static_cast<int64_t>(f) * S * H;

  auto weights_of = [&](int s) -> const float* {
    if (w_addr == nullptr) return w_base + static_cast<int64_t>(f) * K * K;
    return reinterpret_cast<const float*>(static_cast<uintptr_t>(w_addr[static_cast<int64_t>(f) * S + s]));
  };
  // row-major (K, K) matrix -> operand layout
  auto stage = [&](const float* wf, int buf) {
    // consecutive threads fill consecutive LDS words (conflict-free); the global side reads 32 B
    // pieces of 32 rows per wave instruction, out of L2
    for (int i = threadIdx.x; i < WF4; i += WAVES * 64) {
      const int ln = i & 63, g = (i >> 6) & 3, pq = i >> 8, q = pq % NK, p = pq / NK;
      *reinterpret_cast<float4*>(&w_s[buf][4 * i]) = *reinterpret_cast<const float4*>(
          wf + static_cast<int64_t>(32 * p + (ln & 31)) * K + 32 * q + 8 * g + 4 * (ln >> 5));
    }
  };

  // a CP-T consumer: its own dense sum is applied to the product of the slots (step `nslots` of the pipeline)
  const float* w_last = w_post != nullptr ? reinterpret_cast<const float*>(static_cast<uintptr_t>(w_post[f])) : nullptr;
  float o[NK][16];
  const float* w_cur = weights_of(0);
  if (w_cur != nullptr) stage(w_cur, 0);
  const int nslots = MULTI ? S : 1;
  for (int s = 0; s < nslots; ++s) {
    // v_s: product (log-space sum) of the H children
    float v[NK][16];
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) v[q][j] = 0.f;
    for (int h = 0; h < H; ++h) {
      const float* src = slot_source(gs, arena, ro[s * H + h], (static_cast<int64_t>(f) * S + s) * H + h, bl, B, K, kh);
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t4 = *reinterpret_cast<const float4*>(src + 32 * q + 8 * g);
          v[q][4 * g + 0] += t4.x;
          v[q][4 * g + 1] += t4.y;
          v[q][4 * g + 2] += t4.z;
          v[q][4 * g + 3] += t4.w;
        }
    }
    // The next slot's weights land in the other buffer while this slot computes.  That buffer was
    // last read by slot s - 1, which slower waves may still be in: fence first (never taken for S <= 2).
    const float* w_next = MULTI && s + 1 < nslots ? weights_of(s + 1) : (s + 1 == nslots ? w_last : nullptr);
    if (w_next != nullptr) {
      if (s >= 1) __syncthreads();
      stage(w_next, (s + 1) & 1);
    }
    float m = 0.f;
    if (w_cur != nullptr) {  // uniform over the workgroup
      m = v[0][0];
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) m = fmaxf(m, v[q][j]);
      m = ck::xhalf_max(m);
      m = ck::clamp_finite(m);
      const float nml = exp_offset(m, 0.f);
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[q][j] = __builtin_amdgcn_exp2f(fmaf(v[q][j], kL2E, nml));
    }
    __syncthreads();  // buffer s & 1 holds W_s (staged one slot ago; just now for s = 0)
    if (w_cur != nullptr) {
      const float* wb = &w_s[s & 1][0];
#pragma unroll
      for (int p = 0; p < NK; ++p) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 w4 = *reinterpret_cast<const float4*>(wb + ((((p * NK + q) * 4 + g) * 64) + lane) * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, v[q][4 * g + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, v[q][4 * g + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, v[q][4 * g + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, v[q][4 * g + 3], acc, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float gs = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m);
          o[p][r] = (!MULTI || s == 0) ? gs : o[p][r] + gs;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[q][j] = (!MULTI || s == 0) ? v[q][j] : o[q][j] + v[q][j];
    }
    w_cur = w_next;
  }
  if (w_last != nullptr) {  // out = log(W_post . exp(P - max P)) + max P on the register tile P = o
    float m = o[0][0];
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) m = fmaxf(m, o[q][j]);
    m = ck::xhalf_max(m);
    m = ck::clamp_finite(m);
    const float nml = exp_offset(m, 0.f);
    float e[NK][16];
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) e[q][j] = __builtin_amdgcn_exp2f(fmaf(o[q][j], kL2E, nml));
    __syncthreads();  // W_post is staged (during the last slot)
    const float* wb = &w_s[nslots & 1][0];
#pragma unroll
    for (int p = 0; p < NK; ++p) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(wb + ((((p * NK + q) * 4 + g) * 64) + lane) * 4);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, e[q][4 * g + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, e[q][4 * g + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, e[q][4 * g + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, e[q][4 * g + 3], acc, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) o[p][r] = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m);
    }
  }
  if (live) {
    float* dst = out + (out_off != nullptr ? out_off[f] : static_cast<int64_t>(f) * B * K) + static_cast<int64_t>(b) * K + 4 * kh;
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = make_float4(o[p][4 * g], o[p][4 * g + 1], o[p][4 * g + 2], o[p][4 * g + 3]);
  }
}

// A whole region with several partitionings in one launch:
//     P_h = sum_s G_{h,s}                      (the CP block of partitioning h, as in cp_lse_kernel)
//     out = log(sum_h mw[:, h] * exp(P_h - M)) + M,   M = max over all (h, k)     (the mixing layer)
// i.e. RegionGraph.build_circuit's `mix_ins = [sum_prod_builder_(...) for ptn in region_inputs]`
// followed by the arity-H SumLayer with mixing weights (templates/region_graph/graph.py:556-583,
// symbolic/parameters.py:1007-1044, nodes.py:847-862).  The H Hadamard outputs the reference writes
// and reads back stay in registers.  The mixing sum is accumulated ONLINE over h: a running row
// maximum M and accumulators rescaled by exp(M_old - M_new) -- the same sum the reference forms
// with the final M, up to fp32 rounding of the rescaling.
//
// Weight matrices go through a ring of three LDS buffers (step t = h*S + s): W_{t+2} is staged
// after the barrier of step t, when every wave has left the MFMA chain of step t - 1 that read the
// buffer being overwritten -- one barrier per step.
template <int NK, int WAVES, int MINW>
__global__ void __launch_bounds__(WAVES * 64, MINW)  // MINW waves per SIMD: caps the VGPR budget
    region_lse_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                      const int64_t* __restrict__ w_addr, const float* __restrict__ mw,
                      float* __restrict__ out, const GatherSlots gs, int32_t* __restrict__ redo, int H, int S, int B) {
  constexpr int K = 32 * NK;
  constexpr int WF4 = K * K / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (redo != nullptr) {  // second launch of a region: only the workgroups the linear-space launch marked
    int32_t* flag = redo + (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x);
    if (*flag == 0) return;  // (uniform over the workgroup)
    __syncthreads();
    if (threadIdx.x == 0) *flag = 0;  // ready for the next replay of the program
  }
  float* w_s = smem;               // [3][K*K]
  float* mw_s = smem + 3 * K * K;  // [H][K]
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (blockIdx.x * WAVES + wave) * 32 + b_in;
  const bool live = b < B;
  const int bl = live ? b : B - 1;
  const int T = H * S;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * T;
  const int64_t* wa = w_addr + static_cast<int64_t>(f) * T;
  // W_t travels global -> LDS buffer t % 3 directly (global_load_lds_dwordx4: no registers, no store instruction),
  // issued right after the barrier of step t - 2 -- every wave has then left step t - 3, the last reader of that
  // buffer -- and covered by that step's MFMA chain; a wave waits for its own loads (vmcnt) before the barrier of
  // step t - 1.  Lane l of wave w writes 16 bytes at LDS word 4 (k * WAVES * 64 + w * 64 + l): the operand layout.
  constexpr int PF = (WF4 + WAVES * 64 - 1) / (WAVES * 64);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto stage_async = [&](int t) {
    const float* wf = reinterpret_cast<const float*>(static_cast<uintptr_t>(wa[t]));
    if (wf == nullptr) return;
    float* dstb = w_s + (t % 3) * K * K;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int i = threadIdx.x + k * (WAVES * 64);
      if (WF4 % (WAVES * 64) != 0 && i >= WF4) continue;
      const int ln = i & 63, g = (i >> 6) & 3, pq = i >> 8, q = pq % NK, p = pq / NK;
      const float* src = wf + static_cast<int64_t>(32 * p + (ln & 31)) * K + 32 * q + 8 * g + 4 * (ln >> 5);
      __builtin_amdgcn_global_load_lds((ck::gptr_t)src, (ck::lptr_t)(dstb + 4 * (k * WAVES * 64 + wave_u * 64)), 16, 0, 0);
    }
  };
  const float* mwf = mw + static_cast<int64_t>(f) * K * H;
  for (int i = threadIdx.x; i < K * H; i += WAVES * 64) {
    const int k = i / H, h = i - k * H;
    mw_s[h * K + k] = mwf[i];
  }
  stage_async(0);
  if (T > 1) stage_async(1);

  float A[NK][16];
#pragma unroll
  for (int p = 0; p < NK; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) A[p][r] = 0.f;
  float M = -INFINITY;
  int t = 0;
  for (int h = 0; h < H; ++h) {
    float P[NK][16];
    for (int s = 0; s < S; ++s, ++t) {
      float v[NK][16];
      const float* src = slot_source(gs, arena, ro[t], static_cast<int64_t>(f) * T + t, bl, B, K, kh);
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t4 = *reinterpret_cast<const float4*>(src + 32 * q + 8 * g);
          v[q][4 * g + 0] = t4.x;
          v[q][4 * g + 1] = t4.y;
          v[q][4 * g + 2] = t4.z;
          v[q][4 * g + 3] = t4.w;
        }
      const bool dense = wa[t] != 0;  // uniform over the workgroup
      float m = 0.f;
      if (dense) {
        m = v[0][0];
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
          for (int j = 0; j < 16; ++j) m = fmaxf(m, v[q][j]);
        m = ck::xhalf_max(m);
        m = ck::clamp_finite(m);
        const float nml = exp_offset(m, 0.f);
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
          for (int j = 0; j < 16; ++j) v[q][j] = __builtin_amdgcn_exp2f(fmaf(v[q][j], kL2E, nml));
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's share of W_t (and W_{t+1}) has landed in LDS
      __syncthreads();                      // W_t is in LDS; every wave has left the MFMA chain of step t - 1
      if (t + 2 < T) stage_async(t + 2);
      if (dense) {
        const float* wb = w_s + (t % 3) * K * K;
#pragma unroll
        for (int p = 0; p < NK; ++p) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int q = 0; q < NK; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 w4 = *reinterpret_cast<const float4*>(wb + ((((p * NK + q) * 4 + g) * 64) + lane) * 4);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, v[q][4 * g + 0], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, v[q][4 * g + 1], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, v[q][4 * g + 2], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, v[q][4 * g + 3], acc, 0, 0, 0);
            }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float gs = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m);
            P[p][r] = s == 0 ? gs : P[p][r] + gs;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
          for (int j = 0; j < 16; ++j) P[q][j] = s == 0 ? v[q][j] : P[q][j] + v[q][j];
      }
    }
    // mixing: fold P_h into the running sum
    float pm = P[0][0];
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) pm = fmaxf(pm, P[p][r]);
    pm = ck::xhalf_max(pm);
    const float Mn = ck::clamp_finite(fmaxf(M, pm));
    const float scale = __builtin_amdgcn_exp2f((M - Mn) * kL2E);  // 0 on the first partitioning (M = -inf)
    const float nml = exp_offset(Mn, 0.f);
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 c4 = *reinterpret_cast<const float4*>(mw_s + h * K + 32 * p + 8 * g + 4 * kh);
        A[p][4 * g + 0] = fmaf(c4.x, __builtin_amdgcn_exp2f(fmaf(P[p][4 * g + 0], kL2E, nml)), A[p][4 * g + 0] * scale);
        A[p][4 * g + 1] = fmaf(c4.y, __builtin_amdgcn_exp2f(fmaf(P[p][4 * g + 1], kL2E, nml)), A[p][4 * g + 1] * scale);
        A[p][4 * g + 2] = fmaf(c4.z, __builtin_amdgcn_exp2f(fmaf(P[p][4 * g + 2], kL2E, nml)), A[p][4 * g + 2] * scale);
        A[p][4 * g + 3] = fmaf(c4.w, __builtin_amdgcn_exp2f(fmaf(P[p][4 * g + 3], kL2E, nml)), A[p][4 * g + 3] * scale);
      }
    M = Mn;
  }
  if (live) {
    float* dst = out + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o4;
        o4.x = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 0]), kLN2, M);
        o4.y = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 1]), kLN2, M);
        o4.z = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 2]), kLN2, M);
        o4.w = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 3]), kLN2, M);
        *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = o4;
      }
  }
}

// The region launch with every operand staged through LDS by the DMA path (global_load_lds_dwordx4), for regions
// whose slots all read the arena (no table gathers):
//   * the 32-row input tile of step t + 1 travels global -> LDS while step t computes -- whole 128-byte lines per
//     request (a wave instruction of the register path touches 32 bytes of 32 rows), no registers held for data in
//     flight (the register path could not prefetch: > 168 VGPRs), one private 32 x K slot per wave;
//   * LDS reads -- the tile and the weights -- are inline ds_read_b128: for plain C++ loads from a buffer the DMA also
//     writes, the compiler waits for EVERY outstanding DMA (s_waitcnt vmcnt(0)) before the first read, which turned
//     the asynchronous weight staging of region_lse_kernel into a round trip per step.
// Tile slot layout: 16-byte chunk c of row r at chunk position c ^ swz(r) of row r (swz(r) = r & 15 for K = 64,
// (r >> 1) & 7 for K = 32): lane (b, kh) reading chunk 2g + kh (+ 8q) of row b then hits 16 distinct positions per 16
// lanes -- no bank conflicts -- and the DMA, whose LDS side is linear in the lane, applies the swizzle on its global side.
// Weights: ring of TWO buffers (W_{t+1} is staged after the barrier of step t, a whole MFMA chain ahead of its use).
// Same arithmetic, in the same order, as region_lse_kernel: the two agree bit for bit.
// BLOCK: the launch is a CP block on its own (H = 1, LINEAR = false; see below) -- the variants without it carry neither
// the CP-T step nor the table gathers, the variant with it no mixing sum.
// CT: 0 = the contractions in exact fp32 (the product); 3 / 6 = the labelled bf16-split VARIANTS ("bf16x3" / "bf16x6", ck_tile.h
// contract_bf16): a weight unit arrives by DMA as it always does and is cut IN PLACE into P = 2 / 3 bf16 pieces by the workgroup
// (thread (block of 16 inputs, lane) reads the two float4s of its lane that hold inputs 16 m .. 16 m + 15 and writes their pieces
// over them -- and, P = 3, into a third of a buffer behind the unit; one more barrier per unit), the exponentiated tile in
// registers, and the chain runs as 3 / 6 products per 16 inputs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
#ifndef CK_REGION_LABBITS
#define CK_REGION_LABBITS 0  // lab builds (scripts/exp_region.py, LAB_NOTES R4.5): 1 = two 16-deep MFMA chains per unit, 2 = two workgroups
                             // per CU, 4 = one barrier per slot (needs 2), 8 = operand reads of unit (t, 1) under the chain of (t, 0) (needs 4)
#endif
template <int NK, int WAVES, int MINW, bool LINEAR, bool BLOCK = false, int CT = 0>
__global__ void __launch_bounds__(WAVES * 64, MINW)
    region_dma_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                      const int64_t* __restrict__ w_addr, const float* __restrict__ mw, float* __restrict__ out,
                      int32_t* __restrict__ redo, int H, int S, int B, const float* __restrict__ w_cat,
                      const int64_t* __restrict__ w_post, const GatherSlots gs) {
  // gs.var != nullptr: slots with gs.var[f, t] >= 0 read rows of a (C+1, K) TABLE picked by the batch values instead of
  // an arena block (tabulated dense layers over Categorical inputs): the same whole-row DMAs with the row index taken
  // from the batch -- the wave's 32 values of every such slot are parked in LDS once, at the start.
  static_assert(!(BLOCK && LINEAR), "CP blocks on their own are evaluated in log space");
  // BLOCK (mw == nullptr, w_cat == nullptr, H = 1): a CP block on its own -- the product P of the slots is the
  // output, or (w_post[f] != 0, a CP-T layer: optimized.py:171-178) log(W_post . exp(P - max P)) + max P, one more step
  // of the weight pipeline on the register tile P.
  // w_cat != nullptr: a dense Sum layer over the CONCATENATION of H children (inner.py:266-273 with a full (K, H K) weight)
  // as a region with one slot per "partitioning" and unit mixing coefficients: W_h = columns h K .. h K + K - 1 of fold f's
  // row-major matrix (row stride H K), no w_addr / mw tables.  log sum_h exp(log(W_h e_h) + m_h) is the layer's value
  // whatever maxima are subtracted on the way.
  constexpr int K = 32 * NK;
  constexpr int UF4 = 32 * K / 4;    // float4 elements of one weight UNIT: the 32 output rows 32 p .. 32 p + 31 of a matrix
  constexpr int CH = K / 4;          // 16-byte chunks per row
  constexpr int TD = 32 * CH / 64;   // wave DMAs per tile
  constexpr int NP = CT == 0 ? 1 : CT / 3 + 1;
  constexpr int UB = CT == 6 ? 48 * K : 32 * K;  // floats of a ring buffer (bf16x6: the third pieces behind the unit)
  static_assert(CT == 0 || 2 * NK * 64 <= WAVES * 64, "one (block, lane) item per thread cuts a unit into pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // LAB bit 4: a ring of two whole matrices (four units) and ONE barrier per slot
  constexpr bool ONEB = (CK_REGION_LABBITS & 4) != 0 && NK == 2 && CT == 0 && !BLOCK && LINEAR;
  constexpr int RS = ONEB ? 3 : 1;
  float* w_s = smem;                           // [2][UB]: ring of two weight units
  float* tile_s = smem + (RS + 1) * UB;        // [WAVES][32*K]
  float* mw_s = tile_s + WAVES * 32 * K;       // [H][K]
  int32_t* xg_s = reinterpret_cast<int32_t*>(mw_s + H * K);  // gather slots: [WAVES][T][32] batch values, DMA order
  // BLOCK: a one-dimensional grid, XCD-aware (consecutive workgroup ids go to consecutive XCDs): every workgroup of a
  // fold gets the same id % 8, so the fold's tables and weights are fetched into ONE XCD's L2 and found there by the
  // fold's other row tiles (gridDim.y = 1; the F x row-tile grid otherwise)
  int f = blockIdx.y, bx = blockIdx.x;
  if constexpr (BLOCK) {
    const int gx = (B + WAVES * 32 - 1) / (WAVES * 32);
    const int n = blockIdx.x >> 3;
    f = (n / gx) * 8 + (blockIdx.x & 7);
    bx = n % gx;
    if (f >= gs.F) return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int b_in = lane & 31, kh = lane >> 5;
  const int b0 = (bx * WAVES + wave_u) * 32;
  const int b = b0 + b_in;
  const bool live = b < B;
  const int T = H * S;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * T;
  const int64_t* wa = w_addr + static_cast<int64_t>(f) * T;
  const int TU = (T + (BLOCK && w_post != nullptr ? 1 : 0)) * NK;  // weight units of the launch
  // weight unit u = t * NK + p goes to ring buffer u & 1, in the operand layout [(q * 4 + g) * 64 + lane] float4
  // CT != 0: thread (wave, lane) of the first 2 NK waves stages float4s 2 wave and 2 wave + 1 of its lane -- the two its own
  // item of the cut reads (see cut_unit: nobody waits for anybody else's requests before cutting)
  constexpr int PF = CT != 0 ? 2 : (UF4 + WAVES * 64 - 1) / (WAVES * 64);
  static_assert(NK == 1 || UF4 % (WAVES * 64) == 0, "the vmcnt bookkeeping of later units assumes every wave stages a share");
  // (addresses = a uniform base + a 32-bit lane offset: global_load_lds with an SGPR base, no 64-bit lane arithmetic)
  const int w_ld = w_cat != nullptr ? K * T : K;  // row stride of a weight matrix
  uint32_t w_off[PF];  // byte offset of this lane's 16 bytes inside a unit's 32 rows
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int i = CT != 0 ? (2 * wave + k) * 64 + lane : static_cast<int>(threadIdx.x) + k * (WAVES * 64);
    const int ln = i & 63, g = (i >> 6) & 3, q = i >> 8;
    w_off[k] = static_cast<uint32_t>(((ln & 31) * w_ld + 32 * q + 8 * g + 4 * (ln >> 5)) * 4);
  }
  auto stage_w = [&](int u) {
    const int t = u / NK, p = u % NK;
    const uint64_t wv = w_cat != nullptr
                            ? static_cast<uint64_t>(reinterpret_cast<uintptr_t>(w_cat + (static_cast<int64_t>(f) * K * T + t) * K))
                            : static_cast<uint64_t>(BLOCK && t == T ? w_post[f] : wa[t]);
    if (wv == 0) return;
    // (made uniform explicitly: the compiler otherwise carries the loaded address in vector registers)
    const uint64_t wu = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(wv >> 32)))) << 32) |
                        static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(wv)));
    const char* wf = reinterpret_cast<const char*>(static_cast<uintptr_t>(wu)) + static_cast<int64_t>(p) * (32 * 4) * w_ld;
    float* dstb = w_s + (u & RS) * UB;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      if constexpr (CT != 0) {
        if (wave_u >= 2 * NK) continue;  // (whole waves)
        uint32_t o = w_off[k];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((ck::gptr_t)(wf + o), (ck::lptr_t)(dstb + 4 * ((2 * wave_u + k) * 64)), 16, 0, 0);
        continue;
      }
      if (UF4 % (WAVES * 64) != 0 && static_cast<int>(threadIdx.x) + k * (WAVES * 64) >= UF4) continue;  // (whole waves)
      uint32_t o = w_off[k];
      asm volatile("" : "+v"(o));
      __builtin_amdgcn_global_load_lds((ck::gptr_t)(wf + o), (ck::lptr_t)(dstb + 4 * (k * WAVES * 64 + wave_u * 64)), 16, 0, 0);
    }
  };
  // the wave's tile slot; DMA k, lane l fills chunk position l % CH of row k * (64 / CH) + l / CH
  float* const my_tile = tile_s + wave_u * (32 * K);
  uint32_t src_off[TD];  // byte offset of this lane's chunk inside a slot's (B, K) block (< 2^32: checked on the host)
#pragma unroll
  for (int k = 0; k < TD; ++k) {
    const int r = k * (64 / CH) + lane / CH, pos = lane % CH;
    const int swz = CH == 16 ? (r & 15) : ((r >> 1) & 7);
    src_off[k] = static_cast<uint32_t>(min(b0 + r, B - 1) * K + 4 * (pos ^ swz)) * 4u;
  }
  const int32_t* gvar = BLOCK && gs.var != nullptr ? gs.var + static_cast<int64_t>(f) * T : nullptr;
  // a lane's TD values of slot t are contiguous: index (lane / CH) * TD + k holds the value of row k * (64 / CH) + lane / CH
  const uint32_t xg_rd = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xg_s)) + ((wave_u * T) * 32 + (lane / CH) * TD) * 4;
  auto stage_tile = [&](int t) {
    if (BLOCK && gvar != nullptr && gvar[t] >= 0) {  // (uniform) rows x[b] of the slot's table
      const char* base = reinterpret_cast<const char*>(static_cast<uintptr_t>(gs.addr[static_cast<int64_t>(f) * T + t]));
      int32_t xv[TD];
      static_for<0, TD / 4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        f32x4v r;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(xg_rd + t * 128), "n"(16 * j) : "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[4 * j + e] = __float_as_int(r[e]);
      });
#pragma unroll
      for (int k = 0; k < TD; ++k) {
        const int r = k * (64 / CH) + lane / CH, pos = lane % CH;
        const int swz = CH == 16 ? (r & 15) : ((r >> 1) & 7);
        const int c = xv[k] < 0 ? gs.C : min(xv[k], gs.C - 1);
        const uint32_t o = static_cast<uint32_t>(c * K + 4 * (pos ^ swz)) * 4u;
        __builtin_amdgcn_global_load_lds((ck::gptr_t)(base + o), (ck::lptr_t)(my_tile + k * 256), 16, 0, 0);
      }
      return;
    }
    const char* base = reinterpret_cast<const char*>(arena + ro[t]);
#pragma unroll
    for (int k = 0; k < TD; ++k) {
      uint32_t o = src_off[k];
      asm volatile("" : "+v"(o));  // (keeps the offset a 32-bit register: hoisted, it is widened to a 64-bit pair per request)
      __builtin_amdgcn_global_load_lds((ck::gptr_t)(base + o), (ck::lptr_t)(my_tile + k * 256), 16, 0, 0);
    }
  };
  // LDS byte address of chunk position (c ^ swz(b_in)) of row b_in is rd_row ^ (16 c)
  const uint32_t swz_b = CH == 16 ? (b_in & 15) : ((b_in >> 1) & 7);
  const uint32_t rd_row = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(my_tile)) + (b_in * CH + swz_b) * 16;
  const uint32_t w_rd = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(w_s)) + lane * 16;

  // CT != 0: unit u (fp32, operand layout [(4 q + g) * 64 + lane] float4) cut into pieces in place: item (blk = 2 q + m, lane) =
  // thread blk * 64 + lane owns float4s 2 blk and 2 blk + 1 of its lane -- inputs 32 q + 16 m + 8 s + 4 kh + t of row lane % 32,
  // s = 0 / 1, which this very thread requested (stage_w) -- and leaves piece 0 in the first, piece 1 in the second, piece 2 at
  // float4 (4 NK + blk) * 64 + lane.  Called BEFORE the barrier of the unit, behind the wait for the wave's own requests.
  auto cut_unit = [&](int u) {
    if (static_cast<int>(threadIdx.x) < 2 * NK * 64) {
      const uint32_t a0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(w_s)) + (u & RS) * (UB * 4) + (2 * wave_u * 64 + lane) * 16;
      f32x4v x0, x1;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(a0) : "memory");
      float r[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
        u32x4v d;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_amdgcn_perm(__float_as_uint(r[2 * j + 1]), __float_as_uint(r[2 * j]), 0x07060302u);
        const uint32_t ad = pc < 2 ? a0 + 1024 * pc : static_cast<uint32_t>(reinterpret_cast<uintptr_t>(w_s)) + (u & RS) * (UB * 4) + ((4 * NK + wave_u) * 64 + lane) * 16;
        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(d) : "memory");
        if (pc + 1 < NP) {
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] -= __uint_as_float(__float_as_uint(r[j]) & 0xffff0000u);  // exact
        }
      }
    }
  };
  // acc = W_u . x for the 32 outputs of unit u (in LDS; CT != 0: cut into pieces), x the exponentiated tile of this wave
  auto unit_product = [&](int u, const float (&x)[NK][16], const u32x4v (&xp)[NP][2 * NK], f32x16& acc) {
    const uint32_t wb = w_rd + (u & RS) * (UB * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (CT == 0 && NK == 2 && (CK_REGION_LABBITS & 1)) {  // LAB: two independent 16-deep chains, interleaved
      f32x4v w0, w1, w2, w3, u0, u1, u2, u3;
      lds_read4_off<0, 1024, 2048, 3072>(w0, w1, w2, w3, wb);
      lds_read4_off<4096, 4096 + 1024, 4096 + 2048, 4096 + 3072>(u0, u1, u2, u3, wb);
      const f32x4v* wg[4] = {&w0, &w1, &w2, &w3};
      const f32x4v* ug[4] = {&u0, &u1, &u2, &u3};
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[e], x[0][4 * g + e], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((*ug[g])[e], x[1][4 * g + e], acc2, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    } else if constexpr (CT == 0) {
      static_for<0, NK>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int o = q * 4096;
        f32x4v w0, w1, w2, w3;
        lds_read4_off<o, o + 1024, o + 2048, o + 3072>(w0, w1, w2, w3, wb);
        const f32x4v* wg[4] = {&w0, &w1, &w2, &w3};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[0], x[q][4 * g + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[1], x[q][4 * g + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[2], x[q][4 * g + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[3], x[q][4 * g + 3], acc, 0, 0, 0);
        }
      });
    } else {
      auto mm = [&](const f32x4v& w, const u32x4v& y) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, w), __builtin_bit_cast(bf16x8v, y), acc, 0, 0, 0);
      };
      static_for<0, NK>([&](auto qc) {  // blocks 2 q, 2 q + 1: pieces 0, 1 at float4s 4 q .. 4 q + 3, piece 2 behind the unit
        constexpr int q = decltype(qc)::value;
        constexpr int o = q * 4096;
        f32x4v w00, w01, w10, w11, w20, w21;
        lds_read4_off<o, o + 1024, o + 2048, o + 3072>(w00, w01, w10, w11, wb);  // (block 2 q: pieces 0, 1; block 2 q + 1: pieces 0, 1)
        if constexpr (NP == 3) {
          asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w20), "=&v"(w21)
                       : "v"(wb), "n"((4 * NK + 2 * q) * 1024), "n"((4 * NK + 2 * q + 1) * 1024)
                       : "memory");
          mm(w20, xp[0][2 * q]);  // (smallest terms first)
          mm(w01, xp[1][2 * q]);
          mm(w00, xp[2][2 * q]);
        }
        mm(w01, xp[0][2 * q]);
        mm(w00, xp[1][2 * q]);
        mm(w00, xp[0][2 * q]);
        if constexpr (NP == 3) {
          mm(w21, xp[0][2 * q + 1]);
          mm(w11, xp[1][2 * q + 1]);
          mm(w10, xp[2][2 * q + 1]);
        }
        mm(w11, xp[0][2 * q + 1]);
        mm(w10, xp[1][2 * q + 1]);
        mm(w10, xp[0][2 * q + 1]);
      });
    }
  };
  // the exponentiated tile cut into the B operands: piece p of registers 8 m .. 8 m + 7 of quarter q at [p][2 q + m]
  auto cut_tile = [&](float (&x)[NK][16], u32x4v (&xp)[NP][2 * NK]) {
    if constexpr (CT != 0) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc)
#pragma unroll
        for (int q = 0; q < NK; ++q) {
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int d = 0; d < 4; ++d)
              xp[pc][2 * q + mh][d] = __builtin_amdgcn_perm(__float_as_uint(x[q][8 * mh + 2 * d + 1]), __float_as_uint(x[q][8 * mh + 2 * d]), 0x07060302u);
          if (pc + 1 < NP) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x[q][j] -= __uint_as_float(__float_as_uint(x[q][j]) & 0xffff0000u);  // exact
          }
        }
    }
  };

  for (int i = threadIdx.x; i < K * H; i += WAVES * 64) {
    const int k = i / H, h = i - k * H;
    mw_s[h * K + k] = mw != nullptr ? mw[static_cast<int64_t>(f) * K * H + i] : 1.f;
  }
  if (BLOCK && gvar != nullptr && lane < 32) {  // row `lane` of this wave's tile: slot t's value at DMA-order index (see xg_rd)
    const int row = min(b0 + lane, B - 1);
    for (int t = 0; t < T; ++t) {
      const int v = gvar[t];
      if (v >= 0) xg_s[(wave_u * T + t) * 32 + (lane % (64 / CH)) * TD + lane / (64 / CH)] = gs.xt[static_cast<int64_t>(v) * B + row];
    }
  }
  stage_w(0);
  if constexpr (ONEB) stage_w(1);
  stage_tile(0);

  float A[NK][16];
#pragma unroll
  for (int p = 0; p < NK; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) A[p][r] = 0.f;
  float M = -INFINITY;
  int t = 0;
  bool bad = false;
  for (int h = 0; h < H; ++h) {
    // the product of the slots starts at the neutral element and every slot multiplies (adds) into it IN PLACE: with
    // `s == 0 ? x : P * x` the compiler keeps the result in fresh registers and copies all 32 back at the end of every slot
    float P[NK][16];
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) P[p][r] = LINEAR ? 1.f : 0.f;
    float sc = 0.f;  // LINEAR: log scale of the rows of P
    for (int s = 0; s < S; ++s, ++t) {
      float v[NK][16];
      // tile t (requested a step ago) and this wave's share of W_t (requested before it) have landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        f32x4v r0, r1, r2, r3;
        lds_read4(r0, r1, r2, r3, rd_row ^ (16u * (8 * q + kh)), rd_row ^ (16u * (8 * q + 2 + kh)),
                  rd_row ^ (16u * (8 * q + 4 + kh)), rd_row ^ (16u * (8 * q + 6 + kh)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[q][e] = r0[e];
          v[q][4 + e] = r1[e];
          v[q][8 + e] = r2[e];
          v[q][12 + e] = r3[e];
        }
      }
      const bool dense = w_cat != nullptr || wa[t] != 0;  // uniform over the workgroup
      float m = 0.f;
      if (dense || LINEAR) {  // (LINEAR: a plain slot enters the product as exp(v - m) with log scale m)
        m = v[0][0];
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
          for (int j = 0; j < 16; ++j) m = fmaxf(m, v[q][j]);
        m = ck::xhalf_max(m);
        m = ck::clamp_finite(m);
        const float nml = exp_offset(m, 0.f);
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
          for (int j = 0; j < 16; ++j) v[q][j] = __builtin_amdgcn_exp2f(fmaf(v[q][j], kL2E, nml));
      }
      u32x4v vp[NP][2 * NK];
      f32x4v wn[8];  // (LAB bit 8)
      if (dense) cut_tile(v, vp);
      static_for<0, NK>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        const int u = t * NK + p;
        // unit u is in LDS: this wave's share has landed (at p = 0 by the wait at the top of the step; later units were
        // requested BEFORE the next tile, whose TD requests may still be in flight) -- and so have the other waves'
        if constexpr (p > 0 && !ONEB) {
          if (t + 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TD) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (CT != 0) {
          if (dense) cut_unit(u);
        }
        if constexpr (!ONEB || p == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ... and every wave has left unit u - 1
        if constexpr (ONEB) {
          if constexpr (p == 0) {
            if (u + 2 < TU) stage_w(u + 2);
            if (u + 3 < TU) stage_w(u + 3);
          }
        } else if (u + 1 < TU) stage_w(u + 1);
        if constexpr (p == 0) {
          if (t + 1 < T) stage_tile(t + 1);  // (the slot is free since its reads returned)
        }
        if (dense) {
          f32x16 acc;
          if constexpr (ONEB && (CK_REGION_LABBITS & 8) != 0) {  // LAB: the operands of unit (t, 1) requested under the chain of (t, 0)
            auto chain = [&](const f32x4v (&a)[8]) {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
              for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + g][e], v[q][4 * g + e], acc, 0, 0, 0);
            };
            if constexpr (p == 0) {
              f32x4v a[8];
              const uint32_t wb0 = w_rd + (u & RS) * (UB * 4), wb1 = w_rd + ((u + 1) & RS) * (UB * 4);
              lds_read4_off<0, 1024, 2048, 3072>(a[0], a[1], a[2], a[3], wb0);
              lds_read4_off<4096, 4096 + 1024, 4096 + 2048, 4096 + 3072>(a[4], a[5], a[6], a[7], wb0);
              asm volatile(
                  "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                  "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168"
                  : "=&v"(wn[0]), "=&v"(wn[1]), "=&v"(wn[2]), "=&v"(wn[3]), "=&v"(wn[4]), "=&v"(wn[5]), "=&v"(wn[6]), "=&v"(wn[7])
                  : "v"(wb1)
                  : "memory");
              chain(a);
            } else {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              chain(wn);
            }
          } else
          unit_product(u, v, vp, acc);
          if constexpr (LINEAR) {  // P = prod_s G_s stays in linear space; the row's log scale is the sum of the m_s
#pragma unroll
            for (int r = 0; r < 16; ++r) P[p][r] *= acc[r];
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float gs = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m);
              P[p][r] += gs;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            if constexpr (LINEAR) P[p][j] *= v[p][j];  // (exp(v - m): see below)
            else P[p][j] += v[p][j];
          }
        }
      });
      if constexpr (LINEAR) sc += m;
    }
    if constexpr (BLOCK) {
      {  // (H = 1, no mixing layer behind the product)
        if (w_post != nullptr) {  // the CP-T sum on the register tile P: weight units T NK .. T NK + NK - 1
          float m = P[0][0];
#pragma unroll
          for (int q = 0; q < NK; ++q)
#pragma unroll
            for (int j = 0; j < 16; ++j) m = fmaxf(m, P[q][j]);
          m = ck::clamp_finite(ck::xhalf_max(m));
          const float nml = exp_offset(m, 0.f);
          float e[NK][16];
#pragma unroll
          for (int q = 0; q < NK; ++q)
#pragma unroll
            for (int j = 0; j < 16; ++j) e[q][j] = __builtin_amdgcn_exp2f(fmaf(P[q][j], kL2E, nml));
          u32x4v ep[NP][2 * NK];
          cut_tile(e, ep);
          static_for<0, NK>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            const int u = T * NK + p;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's share of unit u has landed
            if constexpr (CT != 0) cut_unit(u);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... everybody's; every wave has left unit u - 1
            if (u + 1 < TU) stage_w(u + 1);
            f32x16 acc;
            unit_product(u, e, ep, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) P[p][r] = fmaf(__builtin_amdgcn_logf(acc[r]), kLN2, m);
          });
        }
        if (live) {
          float* dst = out + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
          for (int p = 0; p < NK; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = make_float4(P[p][4 * g], P[p][4 * g + 1], P[p][4 * g + 2], P[p][4 * g + 3]);
        }
        return;
      }
    }
    // mixing: fold P_h into the running sum
    float pm = P[0][0];
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) pm = fmaxf(pm, P[p][r]);
    pm = ck::xhalf_max(pm);
    const uint32_t mw_rd = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(mw_s)) + (h * K + 4 * kh) * 4;
    if constexpr (LINEAR) {
      // the partition's rows are P * exp(sc): two exponentials per ROW (instead of one per element) rescale the
      // running sum and the new term to the running maximum.  A row whose largest product fell below 2^-80 (the
      // factors are <= 1 with different supports, or an input row was all -inf / non-finite) has lost elements the
      // log-space evaluation keeps: the tile is marked and evaluated again by region_lse_kernel.
      bad |= !(pm > kLinearFloor);
      const float Mh = fmaf(__builtin_amdgcn_logf(pm), kLN2, sc);
      const float Mn = ck::clamp_finite(fmaxf(M, Mh));
      const float c_old = __builtin_amdgcn_exp2f((M - Mn) * kL2E);  // 0 on the first partitioning (M = -inf)
      const float c_new = __builtin_amdgcn_exp2f((sc - Mn) * kL2E);
      static_for<0, NK>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        f32x4v c0, c1, c2, c3;
        lds_read4_off<128 * p, 128 * p + 32, 128 * p + 64, 128 * p + 96>(c0, c1, c2, c3, mw_rd);
        const f32x4v* cg[4] = {&c0, &c1, &c2, &c3};
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            A[p][4 * g + e] = fmaf((*cg[g])[e] * c_new, P[p][4 * g + e], A[p][4 * g + e] * c_old);
      });
      M = Mn;
    } else {
      const float Mn = ck::clamp_finite(fmaxf(M, pm));
      const float scale = __builtin_amdgcn_exp2f((M - Mn) * kL2E);  // 0 on the first partitioning (M = -inf)
      const float nml = exp_offset(Mn, 0.f);
      static_for<0, NK>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        f32x4v c0, c1, c2, c3;
        lds_read4_off<128 * p, 128 * p + 32, 128 * p + 64, 128 * p + 96>(c0, c1, c2, c3, mw_rd);
        const f32x4v* cg[4] = {&c0, &c1, &c2, &c3};
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            A[p][4 * g + e] = fmaf((*cg[g])[e], __builtin_amdgcn_exp2f(fmaf(P[p][4 * g + e], kL2E, nml)), A[p][4 * g + e] * scale);
      });
      M = Mn;
    }
  }
  if constexpr (LINEAR) {
    if (__any(bad) && lane == 0) atomicOr(redo + (static_cast<int64_t>(f) * gridDim.x + bx), 1);
  }
  if (live) {
    float* dst = out + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o4;
        o4.x = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 0]), kLN2, M);
        o4.y = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 1]), kLN2, M);
        o4.z = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 2]), kLN2, M);
        o4.w = fmaf(__builtin_amdgcn_logf(A[p][4 * g + 3]), kLN2, M);
        *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = o4;
      }
  }
}

// Dense sum layer over the CONCATENATION of H children (TorchSumLayer with arity > 1 and a full
// (K, H*K) weight, inner.py:266-273 -- e.g. the Sum -> Sum pair the reference collapses into one
// layer with a MatMul weight, optimization/layers.py:162-198):
//   m = max over all children, out = log(sum_h W[:, hK:(h+1)K] . exp(x_h - m)) + m.
// One maximum for the whole row (first pass over the children), then the H column blocks of the
// weight are contracted one after the other into the same accumulators.
template <int NK, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    cat_lse_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                   const float* __restrict__ w, float* __restrict__ out, int H, int B) {
  constexpr int K = 32 * NK;
  constexpr int WF4 = K * K / 4;
  __shared__ __attribute__((aligned(16))) float w_s[2][K * K];
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (blockIdx.x * WAVES + wave) * 32 + b_in;
  const bool live = b < B;
  const int bl = live ? b : B - 1;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int ldw = H * K;
  const float* wf = w + static_cast<int64_t>(f) * K * ldw;
  auto stage = [&](int h, int buf) {  // column block h of the row-major (K, H*K) matrix
    for (int i = threadIdx.x; i < WF4; i += WAVES * 64) {  // conflict-free LDS side, see cp_lse_kernel
      const int ln = i & 63, g = (i >> 6) & 3, pq = i >> 8, q = pq % NK, p = pq / NK;
      *reinterpret_cast<float4*>(&w_s[buf][4 * i]) = *reinterpret_cast<const float4*>(
          wf + static_cast<int64_t>(32 * p + (ln & 31)) * ldw + h * K + 32 * q + 8 * g + 4 * (ln >> 5));
    }
  };
  stage(0, 0);
  float m = -INFINITY;
  for (int h = 0; h < H; ++h) {
    const float* src = arena + ro[h] + static_cast<int64_t>(bl) * K + 4 * kh;
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t4 = *reinterpret_cast<const float4*>(src + 32 * q + 8 * g);
        m = fmaxf(m, fmaxf(fmaxf(t4.x, t4.y), fmaxf(t4.z, t4.w)));
      }
  }
  m = ck::xhalf_max(m);
  m = ck::clamp_finite(m);
  const float nml = exp_offset(m, 0.f);
  f32x16 acc[NK];
#pragma unroll
  for (int p = 0; p < NK; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  for (int h = 0; h < H; ++h) {
    float v[NK][16];
    const float* src = arena + ro[h] + static_cast<int64_t>(bl) * K + 4 * kh;
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t4 = *reinterpret_cast<const float4*>(src + 32 * q + 8 * g);
        v[q][4 * g + 0] = __builtin_amdgcn_exp2f(fmaf(t4.x, kL2E, nml));
        v[q][4 * g + 1] = __builtin_amdgcn_exp2f(fmaf(t4.y, kL2E, nml));
        v[q][4 * g + 2] = __builtin_amdgcn_exp2f(fmaf(t4.z, kL2E, nml));
        v[q][4 * g + 3] = __builtin_amdgcn_exp2f(fmaf(t4.w, kL2E, nml));
      }
    if (h + 1 < H) {
      if (h >= 1) __syncthreads();  // every wave is done with block h - 1, whose buffer is reused now
      stage(h + 1, (h + 1) & 1);
    }
    __syncthreads();  // block h is staged
    const float* wb = &w_s[h & 1][0];
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(wb + ((((p * NK + q) * 4 + g) * 64) + lane) * 4);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, v[q][4 * g + 0], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, v[q][4 * g + 1], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, v[q][4 * g + 2], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, v[q][4 * g + 3], acc[p], 0, 0, 0);
        }
  }
  if (live) {
    float* dst = out + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o4;
        o4.x = fmaf(__builtin_amdgcn_logf(acc[p][4 * g + 0]), kLN2, m);
        o4.y = fmaf(__builtin_amdgcn_logf(acc[p][4 * g + 1]), kLN2, m);
        o4.z = fmaf(__builtin_amdgcn_logf(acc[p][4 * g + 2]), kLN2, m);
        o4.w = fmaf(__builtin_amdgcn_logf(acc[p][4 * g + 3]), kLN2, m);
        *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = o4;
      }
  }
}

template <int NK>
int launch_cp(const float* arena, const int64_t* row_off, const int64_t* w_addr, const float* w_base,
              const int64_t* w_post, const int64_t* out_off, float* out, GatherSlots gs, int F, int S, int H, int B,
              void* stream) {
  const int tiles = (B + 31) / 32;
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern, int waves) {
          dim3 grid((tiles + waves - 1) / waves, F), block(waves * 64);
          hipLaunchKernelGGL(kern, grid, block, 0, s, arena, row_off, w_addr, w_base, w_post, out_off, out, gs, S, H, B);
        };
        if (S == 1 && w_post == nullptr)
          go(cp_lse_kernel<NK, 8, false>, 8);
        else
          go(cp_lse_kernel<NK, 8, true>, 8);  // 4 waves per workgroup measured the same on config 4
        return hipGetLastError();
      },
      stream);
}

}  // namespace

namespace ck {
// Dense layer over the concatenation of H children, contiguous (F, K, H*K) weights, K in {32, 64}.
int cat_dense(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int K,
              void* stream, int contraction) {
  constexpr int WAVES = 8;
  const int tiles = (B + 31) / 32;
  {  // as a region of H one-slot "partitionings" on the DMA-staged kernel (inputs and weights prefetched a step ahead)
    const int waves = K == 64 ? 4 : 8;
    const size_t lds_dma = (static_cast<size_t>(2) * (contraction == 6 ? 48 : 32) * K + static_cast<size_t>(waves) * 32 * K + static_cast<size_t>(H) * K) * sizeof(float);
    if (lds_dma <= 80 * 1024 && static_cast<int64_t>(B) * K < (int64_t{1} << 30) && static_cast<int64_t>(H) * K * 32 * 4 < (int64_t{1} << 31) &&
        !ck::debug_force_generic()) {
      const dim3 grid((tiles + waves - 1) / waves, F), block(waves * 64);
      return ck::dispatch(
          [=](hipStream_t s) {
            auto go = [&](auto kern) {
              hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds_dma));
              if (e != hipSuccess) return e;
              hipLaunchKernelGGL(kern, grid, block, lds_dma, s, arena, row_off, static_cast<const int64_t*>(nullptr),
                                 static_cast<const float*>(nullptr), out, static_cast<int32_t*>(nullptr), H, 1, B, w,
                                 static_cast<const int64_t*>(nullptr), GatherSlots{});
              return hipGetLastError();
            };
            if (contraction == 3) return K == 64 ? go(region_dma_kernel<2, 4, 3, false, false, 3>) : go(region_dma_kernel<1, 8, 2, false, false, 3>);
            if (contraction == 6) return K == 64 ? go(region_dma_kernel<2, 4, 2, false, false, 6>) : go(region_dma_kernel<1, 8, 2, false, false, 6>);
            return K == 64 ? go(region_dma_kernel<2, 4, 3, false>) : go(region_dma_kernel<1, 8, 2, false>);
          },
          stream);
    }
  }
  dim3 grid((tiles + WAVES - 1) / WAVES, F), block(WAVES * 64);  // (exact fp32 whatever `contraction` says)
  return ck::dispatch(
      [=](hipStream_t s) {
        if (K == 64)
          hipLaunchKernelGGL((cat_lse_kernel<2, WAVES>), grid, block, 0, s, arena, row_off, w, out, H, B);
        else
          hipLaunchKernelGGL((cat_lse_kernel<1, WAVES>), grid, block, 0, s, arena, row_off, w, out, H, B);
        return hipGetLastError();
      },
      stream);
}

// K = 64 dense / CP-T layers of ck_sum_lse_fwd (one slot, contiguous (F, K, K) weights).
int cp_single_slot(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int K,
                   void* stream) {
  if (K == 64) return launch_cp<2>(arena, row_off, nullptr, w, nullptr, nullptr, out, GatherSlots{}, F, 1, H, B, stream);
  return launch_cp<1>(arena, row_off, nullptr, w, nullptr, nullptr, out, GatherSlots{}, F, 1, H, B, stream);
}
}  // namespace ck

extern "C" int ck_cp_lse_fwd_v(const float* arena, const int64_t* row_off, const int64_t* w_addr, const int64_t* w_post,
                               const int64_t* out_off, float* out, const int64_t* g_addr, const int32_t* g_var,
                               const int32_t* xt, int C, int F, int S, int H, int B, int K, int contraction, void* stream) {
  CK_REQUIRE(contraction == 0 || contraction == 3 || contraction == 6, "ck_cp_lse_fwd_v: contraction %d (0, 3 or 6)", contraction);
  CK_REQUIRE(g_var == nullptr || (g_addr && xt && C > 0), "ck_cp_lse_fwd: gather slots need g_addr, xt and C");
  const GatherSlots gs{g_addr, g_var, xt, C, F};
  CK_REQUIRE(arena && row_off && w_addr && out, "ck_cp_lse_fwd: null pointer");
  CK_REQUIRE(F > 0 && S > 0 && H > 0 && B > 0, "ck_cp_lse_fwd: non-positive size F=%d S=%d H=%d B=%d", F, S, H, B);
  CK_REQUIRE(K == 32 || K == 64, "ck_cp_lse_fwd: K must be 32 or 64, found %d", K);
  CK_REQUIRE(F <= 65535, "ck_cp_lse_fwd: F=%d exceeds grid.y", F);
  CK_REQUIRE(ck::aligned16(arena) && ck::aligned16(out), "ck_cp_lse_fwd: buffers must be 16-byte aligned");
  // a block of one child per slot with contiguous output on the DMA-staged kernel of the regions: every operand (the CP-T
  // matrix included) prefetched a step ahead through LDS
  {
    const int waves = K == 64 ? 4 : 8;
    const size_t lds_dma = (static_cast<size_t>(2) * (contraction == 6 ? 48 : 32) * K + static_cast<size_t>(waves) * 32 * K + static_cast<size_t>(K) +
                            (g_var != nullptr ? static_cast<size_t>(waves) * S * 32 : 0)) * sizeof(float);
    if (H == 1 && (g_var == nullptr || S <= 8) && out_off == nullptr && !ck::debug_force_generic() &&
        static_cast<int64_t>(B) * K < (int64_t{1} << 30)) {
      const int tiles = (B + 31) / 32, gx = (tiles + waves - 1) / waves;
      const dim3 grid(static_cast<unsigned>((F + 7) / 8 * 8 * gx)), block(waves * 64);  // (XCD-aware: see the kernel)
      return ck::dispatch(
          [=](hipStream_t s) {
            auto go = [&](auto kern) {
              hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds_dma));
              if (e != hipSuccess) return e;
              hipLaunchKernelGGL(kern, grid, block, lds_dma, s, arena, row_off, w_addr, static_cast<const float*>(nullptr), out,
                                 static_cast<int32_t*>(nullptr), 1, S, B, static_cast<const float*>(nullptr), w_post, gs);
              return hipGetLastError();
            };
            if (contraction == 3) return K == 64 ? go(region_dma_kernel<2, 4, 3, false, true, 3>) : go(region_dma_kernel<1, 8, 2, false, true, 3>);
            if (contraction == 6) return K == 64 ? go(region_dma_kernel<2, 4, 2, false, true, 6>) : go(region_dma_kernel<1, 8, 2, false, true, 6>);
            return K == 64 ? go(region_dma_kernel<2, 4, 3, false, true>) : go(region_dma_kernel<1, 8, 2, false, true>);
          },
          stream);
    }
  }
  // (blocks the DMA-staged launch does not take are evaluated in exact fp32 whatever `contraction` says)
  if (K == 64) return launch_cp<2>(arena, row_off, w_addr, nullptr, w_post, out_off, out, gs, F, S, H, B, stream);
  return launch_cp<1>(arena, row_off, w_addr, nullptr, w_post, out_off, out, gs, F, S, H, B, stream);
}

extern "C" int ck_cp_lse_fwd(const float* arena, const int64_t* row_off, const int64_t* w_addr, const int64_t* w_post,
                             const int64_t* out_off, float* out, const int64_t* g_addr, const int32_t* g_var,
                             const int32_t* xt, int C, int F, int S, int H, int B, int K, void* stream) {
  return ck_cp_lse_fwd_v(arena, row_off, w_addr, w_post, out_off, out, g_addr, g_var, xt, C, F, S, H, B, K, 0, stream);
}

extern "C" int ck_region_lse_fwd_v(const float* arena, const int64_t* row_off, const int64_t* w_addr, const float* mw,
                                   float* out, const int64_t* g_addr, const int32_t* g_var, const int32_t* xt, int C,
                                   int32_t* redo, int F, int H, int S, int B, int K, int contraction, void* stream) {
  CK_REQUIRE(g_var == nullptr || (g_addr && xt && C > 0), "ck_region_lse_fwd: gather slots need g_addr, xt and C");
  CK_REQUIRE(contraction == 0 || contraction == 3 || contraction == 6, "ck_region_lse_fwd_v: contraction %d (0, 3 or 6)", contraction);
  const GatherSlots gs{g_addr, g_var, xt, C, F};
  CK_REQUIRE(arena && row_off && w_addr && mw && out, "ck_region_lse_fwd: null pointer");
  CK_REQUIRE(F > 0 && S > 0 && H > 0 && B > 0, "ck_region_lse_fwd: non-positive size F=%d H=%d S=%d B=%d", F, H, S, B);
  CK_REQUIRE(K == 32 || K == 64, "ck_region_lse_fwd: K must be 32 or 64, found %d", K);
  CK_REQUIRE(F <= 65535, "ck_region_lse_fwd: F=%d exceeds grid.y", F);
  CK_REQUIRE(ck::aligned16(arena) && ck::aligned16(out), "ck_region_lse_fwd: buffers must be 16-byte aligned");
  const size_t lds = (static_cast<size_t>(3) * K * K + static_cast<size_t>(H) * K) * sizeof(float);
  CK_REQUIRE(lds <= 64 * 1024, "ck_region_lse_fwd: H=%d mixing coefficients do not fit in LDS", H);
  const int tiles = (B + 31) / 32;
  const int waves = K == 64 ? 4 : 8;  // (both kernels: one 32-row tile per wave, the same workgroup <-> tiles mapping)
  const dim3 grid((tiles + waves - 1) / waves, F), block(waves * 64);
  // every operand through the LDS DMA path (region_dma_kernel) unless a slot gathers table rows
  size_t lds_dma = (static_cast<size_t>(2) * (contraction == 6 ? 48 : 32) * K + static_cast<size_t>(waves) * 32 * K + static_cast<size_t>(H) * K) * sizeof(float);
  if ((CK_REGION_LABBITS & 4) != 0 && K == 64 && contraction == 0) lds_dma += 2 * 32 * K * sizeof(float);  // LAB
  const bool dma = g_var == nullptr && !ck::debug_force_generic() && lds_dma <= 80 * 1024 &&
                   static_cast<int64_t>(B) * K < (int64_t{1} << 30);
  auto exact = [=](hipStream_t s, int32_t* redo_ws) {  // region_lse_kernel: everything, or (redo_ws) the marked workgroups
    // measured on config 4 (MI355X): 4 waves per workgroup at <= 168 VGPRs (no spills, three
    // workgroups per CU) 2.85 ms; 8 waves capped at 128 VGPRs (spills) 3.28 ms; 2 waves 4.3 ms
    if (K == 64)
      hipLaunchKernelGGL((region_lse_kernel<2, 4, 3>), grid, block, lds, s, arena, row_off, w_addr, mw, out, gs, redo_ws, H, S, B);
    else
      hipLaunchKernelGGL((region_lse_kernel<1, 8, 4>), grid, block, lds, s, arena, row_off, w_addr, mw, out, gs, redo_ws, H, S, B);
    return hipGetLastError();
  };
  if (!dma) return ck::dispatch([=](hipStream_t s) { return exact(s, nullptr); }, stream);
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds_dma));
          if (e != hipSuccess) return e;
          hipLaunchKernelGGL(kern, grid, block, lds_dma, s, arena, row_off, w_addr, mw, out, redo, H, S, B, static_cast<const float*>(nullptr),
                             static_cast<const int64_t*>(nullptr), GatherSlots{});
          return hipGetLastError();
        };
        // K = 64: 48 KiB + H x 256 B of LDS and <= 168 VGPRs: three workgroups (12 waves) per CU while H <= 20
        // (the variants: launches that gather table rows, and the tiles the linear-space launch marks, stay exact fp32)
        if (contraction == 3) {
          if (redo == nullptr) return K == 64 ? go(region_dma_kernel<2, 4, 3, false, false, 3>) : go(region_dma_kernel<1, 8, 2, false, false, 3>);
          const hipError_t e = K == 64 ? go(region_dma_kernel<2, 4, 3, true, false, 3>) : go(region_dma_kernel<1, 8, 2, true, false, 3>);
          return e != hipSuccess ? e : exact(s, redo);
        }
        if (contraction == 6) {  // (K = 64: 56 KiB + H x 256 B of LDS: two workgroups per CU)
          if (redo == nullptr) return K == 64 ? go(region_dma_kernel<2, 4, 2, false, false, 6>) : go(region_dma_kernel<1, 8, 2, false, false, 6>);
          const hipError_t e = K == 64 ? go(region_dma_kernel<2, 4, 2, true, false, 6>) : go(region_dma_kernel<1, 8, 2, true, false, 6>);
          return e != hipSuccess ? e : exact(s, redo);
        }
        if constexpr ((CK_REGION_LABBITS & 2) != 0) {  // LAB: two workgroups per CU (256 registers)
          if (K == 64 && redo != nullptr) {
            const hipError_t e2 = go(region_dma_kernel<2, 4, 2, true>);
            return e2 != hipSuccess ? e2 : exact(s, redo);
          }
        }
        if (redo == nullptr) return K == 64 ? go(region_dma_kernel<2, 4, 3, false>) : go(region_dma_kernel<1, 8, 2, false>);
        const hipError_t e = K == 64 ? go(region_dma_kernel<2, 4, 3, true>) : go(region_dma_kernel<1, 8, 2, true>);
        if (e != hipSuccess) return e;
        return exact(s, redo);  // the workgroups the linear-space launch marked, in log space (none, normally: they exit at once)
      },
      stream);
}

extern "C" int ck_region_lse_fwd(const float* arena, const int64_t* row_off, const int64_t* w_addr, const float* mw,
                                 float* out, const int64_t* g_addr, const int32_t* g_var, const int32_t* xt, int C,
                                 int32_t* redo, int F, int H, int S, int B, int K, void* stream) {
  return ck_region_lse_fwd_v(arena, row_off, w_addr, mw, out, g_addr, g_var, xt, C, redo, F, H, S, B, K, 0, stream);
}
