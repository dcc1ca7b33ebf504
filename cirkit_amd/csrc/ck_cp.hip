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
                      float* __restrict__ out, const GatherSlots gs, int H, int S, int B) {
  constexpr int K = 32 * NK;
  constexpr int WF4 = K * K / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
              void* stream) {
  constexpr int WAVES = 8;
  const int tiles = (B + 31) / 32;
  dim3 grid((tiles + WAVES - 1) / WAVES, F), block(WAVES * 64);
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

extern "C" int ck_cp_lse_fwd(const float* arena, const int64_t* row_off, const int64_t* w_addr, const int64_t* w_post,
                             const int64_t* out_off, float* out, const int64_t* g_addr, const int32_t* g_var,
                             const int32_t* xt, int C, int F, int S, int H, int B, int K, void* stream) {
  CK_REQUIRE(g_var == nullptr || (g_addr && xt && C > 0), "ck_cp_lse_fwd: gather slots need g_addr, xt and C");
  const GatherSlots gs{g_addr, g_var, xt, C};
  CK_REQUIRE(arena && row_off && w_addr && out, "ck_cp_lse_fwd: null pointer");
  CK_REQUIRE(F > 0 && S > 0 && H > 0 && B > 0, "ck_cp_lse_fwd: non-positive size F=%d S=%d H=%d B=%d", F, S, H, B);
  CK_REQUIRE(K == 32 || K == 64, "ck_cp_lse_fwd: K must be 32 or 64, found %d", K);
  CK_REQUIRE(F <= 65535, "ck_cp_lse_fwd: F=%d exceeds grid.y", F);
  CK_REQUIRE(ck::aligned16(arena) && ck::aligned16(out), "ck_cp_lse_fwd: buffers must be 16-byte aligned");
  if (K == 64) return launch_cp<2>(arena, row_off, w_addr, nullptr, w_post, out_off, out, gs, F, S, H, B, stream);
  return launch_cp<1>(arena, row_off, w_addr, nullptr, w_post, out_off, out, gs, F, S, H, B, stream);
}

extern "C" int ck_region_lse_fwd(const float* arena, const int64_t* row_off, const int64_t* w_addr, const float* mw,
                                 float* out, const int64_t* g_addr, const int32_t* g_var, const int32_t* xt, int C,
                                 int F, int H, int S, int B, int K, void* stream) {
  CK_REQUIRE(g_var == nullptr || (g_addr && xt && C > 0), "ck_region_lse_fwd: gather slots need g_addr, xt and C");
  const GatherSlots gs{g_addr, g_var, xt, C};
  CK_REQUIRE(arena && row_off && w_addr && mw && out, "ck_region_lse_fwd: null pointer");
  CK_REQUIRE(F > 0 && S > 0 && H > 0 && B > 0, "ck_region_lse_fwd: non-positive size F=%d H=%d S=%d B=%d", F, H, S, B);
  CK_REQUIRE(K == 32 || K == 64, "ck_region_lse_fwd: K must be 32 or 64, found %d", K);
  CK_REQUIRE(F <= 65535, "ck_region_lse_fwd: F=%d exceeds grid.y", F);
  CK_REQUIRE(ck::aligned16(arena) && ck::aligned16(out), "ck_region_lse_fwd: buffers must be 16-byte aligned");
  const size_t lds = (static_cast<size_t>(3) * K * K + static_cast<size_t>(H) * K) * sizeof(float);
  CK_REQUIRE(lds <= 64 * 1024, "ck_region_lse_fwd: H=%d mixing coefficients do not fit in LDS", H);
  const int tiles = (B + 31) / 32;
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern, int waves) {
          dim3 grid((tiles + waves - 1) / waves, F), block(waves * 64);
          hipLaunchKernelGGL(kern, grid, block, lds, s, arena, row_off, w_addr, mw, out, gs, H, S, B);
        };
        // measured on config 4 (MI355X): 4 waves per workgroup at <= 168 VGPRs (no spills, three
        // workgroups per CU) 2.85 ms; 8 waves capped at 128 VGPRs (spills) 3.28 ms; 2 waves 4.3 ms
        if (K == 64)
          go(region_lse_kernel<2, 4, 3>, 4);
        else
          go(region_lse_kernel<1, 8, 4>, 8);
        return hipGetLastError();
      },
      stream);
}
