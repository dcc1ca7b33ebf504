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

// NK: K / 32.  WAVES: wavefronts (= 32-row tiles) per workgroup.  MULTI: more than one slot (the
// single-slot instance keeps no running sum and fits 7 waves per SIMD instead of 4).
template <int NK, int WAVES, bool MULTI>
__global__ void __launch_bounds__(WAVES * 64)
    cp_lse_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                  const int64_t* __restrict__ w_addr, const float* __restrict__ w_base,
                  const int64_t* __restrict__ out_off, float* __restrict__ out, int S, int H, int B) {
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
  // coalesced read of the row-major (K, K) matrix, scattered into the operand layout
  auto stage = [&](const float* wf, int buf) {
    for (int i = threadIdx.x; i < WF4; i += WAVES * 64) {
      const int o = i / (K / 4), k = 4 * (i - o * (K / 4));
      const int p = o >> 5, q = k >> 5, g = (k >> 3) & 3, ln = (o & 31) + 32 * ((k >> 2) & 1);
      *reinterpret_cast<float4*>(&w_s[buf][((((p * NK + q) * 4 + g) * 64) + ln) * 4]) =
          *reinterpret_cast<const float4*>(wf + 4 * static_cast<int64_t>(i));
    }
  };

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
      const float* src = arena + ro[s * H + h] + static_cast<int64_t>(bl) * K + 4 * kh;
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
    const float* w_next = MULTI && s + 1 < nslots ? weights_of(s + 1) : nullptr;
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
      m = fmaxf(m, __shfl_xor(m, 32, 64));
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
  if (live) {
    float* dst = out + (out_off != nullptr ? out_off[f] : static_cast<int64_t>(f) * B * K) + static_cast<int64_t>(b) * K + 4 * kh;
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = make_float4(o[p][4 * g], o[p][4 * g + 1], o[p][4 * g + 2], o[p][4 * g + 3]);
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
    for (int i = threadIdx.x; i < WF4; i += WAVES * 64) {
      const int o = i / (K / 4), k = 4 * (i - o * (K / 4));
      const int p = o >> 5, q = k >> 5, g = (k >> 3) & 3, ln = (o & 31) + 32 * ((k >> 2) & 1);
      *reinterpret_cast<float4*>(&w_s[buf][((((p * NK + q) * 4 + g) * 64) + ln) * 4]) =
          *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(o) * ldw + h * K + k);
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
  m = fmaxf(m, __shfl_xor(m, 32, 64));
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
              const int64_t* out_off, float* out, int F, int S, int H, int B, void* stream) {
  constexpr int WAVES = 8;
  const int tiles = (B + 31) / 32;
  dim3 grid((tiles + WAVES - 1) / WAVES, F), block(WAVES * 64);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (S == 1)
          hipLaunchKernelGGL((cp_lse_kernel<NK, WAVES, false>), grid, block, 0, s, arena, row_off, w_addr, w_base, out_off, out, S, H, B);
        else
          hipLaunchKernelGGL((cp_lse_kernel<NK, WAVES, true>), grid, block, 0, s, arena, row_off, w_addr, w_base, out_off, out, S, H, B);
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
  if (K == 64) return launch_cp<2>(arena, row_off, nullptr, w, nullptr, out, F, 1, H, B, stream);
  return launch_cp<1>(arena, row_off, nullptr, w, nullptr, out, F, 1, H, B, stream);
}
}  // namespace ck

extern "C" int ck_cp_lse_fwd(const float* arena, const int64_t* row_off, const int64_t* w_addr, const int64_t* out_off,
                             float* out, int F, int S, int H, int B, int K, void* stream) {
  CK_REQUIRE(arena && row_off && w_addr && out, "ck_cp_lse_fwd: null pointer");
  CK_REQUIRE(F > 0 && S > 0 && H > 0 && B > 0, "ck_cp_lse_fwd: non-positive size F=%d S=%d H=%d B=%d", F, S, H, B);
  CK_REQUIRE(K == 32 || K == 64, "ck_cp_lse_fwd: K must be 32 or 64, found %d", K);
  CK_REQUIRE(F <= 65535, "ck_cp_lse_fwd: F=%d exceeds grid.y", F);
  CK_REQUIRE(ck::aligned16(arena) && ck::aligned16(out), "ck_cp_lse_fwd: buffers must be 16-byte aligned");
  if (K == 64) return launch_cp<2>(arena, row_off, w_addr, nullptr, out_off, out, F, S, H, B, stream);
  return launch_cp<1>(arena, row_off, w_addr, nullptr, out_off, out, F, S, H, B, stream);
}
