// Dense / CP-T log-einsum-exp layers with MANY units (Ki, Ko multiples of 32 beyond the 32 / 64 tiles of
// ck_sum.hip / ck_cp.hip): per fold a (B x N) . (N x Ko) product in linear space between the exp and
// the log of LSESumSemiring.apply_reduce (semiring.py:383-408), N = Ki (TorchCPTLayer / arity-1
// TorchSumLayer: the children are multiplied first) or H*Ki (TorchSumLayer over the concatenation of
// its children, inner.py:266-273).
//
// Decomposition: a workgroup = 4 wavefronts = 4 x 32 batch rows of one fold.  Each wave keeps the
// WHOLE exponentiated row block e = exp(v - m) of its 32 rows in registers (N / 32 tiles of ck_tile.h:
// the row maximum is then register-local + one cross-lane exchange, and e is reused for every output
// block).  The weights are streamed one 32-output block at a time through two LDS buffers shared by the
// four waves (one barrier per block: the next block is staged while the current one is contracted), as
// MFMA A operands; v_mfma_f32_32x32x2_f32 keeps the contraction exact fp32.  Per barrier a wave issues
// 16 * N/32 MFMAs (8192 cycles at N = 256), so the kernel is bound by the matrix pipe, not by the
// staging.
#include "ck_internal.h"
#include "ck_tile.h"

namespace {

// NQ: N / 32 (1..8).  CAT: the N inputs are the concatenation of H children of Ki units (else their product).
template <int NQ, bool CAT>
__global__ void __launch_bounds__(256)
    sum_lse_gemm_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                        const float* __restrict__ w, float* __restrict__ out, int H, int B, int Ki, int Ko) {
  constexpr int N = 32 * NQ;
  extern __shared__ __attribute__((aligned(16))) float w_s[];  // [2][NQ][4][64] float4
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (blockIdx.x * 4 + wave) * 32 + b_in;
  const bool live = b < B;
  const int bl = live ? b : B - 1;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* wf = w + static_cast<int64_t>(f) * Ko * N;
  const int npb = Ko >> 5;  // output blocks

  auto stage = [&](int p, int buf) {  // rows 32p .. 32p+31 of the (Ko, N) matrix -> operand layout
    float* dst = w_s + buf * (NQ * 1024);
    for (int i = threadIdx.x; i < NQ * 256; i += 256) {
      const int ln = i & 63, g = (i >> 6) & 3, q = i >> 8;
      *reinterpret_cast<float4*>(dst + 4 * i) = *reinterpret_cast<const float4*>(
          wf + static_cast<int64_t>(32 * p + (ln & 31)) * N + 32 * q + 8 * g + 4 * (ln >> 5));
    }
  };
  stage(0, 0);

  float e[NQ][16];
  if (CAT) {
    const int qpc = Ki >> 5;  // 32-unit blocks per child
#pragma unroll
    for (int q = 0; q < NQ; ++q) tile_load(arena + ro[q / qpc] + static_cast<int64_t>(bl) * Ki + 32 * (q % qpc) + 4 * kh, e[q]);
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) tile_load(arena + ro[0] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, e[q]);
    for (int h = 1; h < H; ++h)
#pragma unroll
      for (int q = 0; q < NQ; ++q) tile_load_add(arena + ro[h] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, e[q]);
  }
  float m = e[0][0];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, e[q][j]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  m = ck::clamp_finite(m);
  const float nml = exp_offset(m, 0.f);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) e[q][j] = __builtin_amdgcn_exp2f(fmaf(e[q][j], kL2E, nml));

  float* dst = out + (static_cast<int64_t>(f) * B + bl) * Ko + 4 * kh;
  for (int p = 0; p < npb; ++p) {
    __syncthreads();  // block p is staged; every wave has left the contraction of block p - 1
    if (p + 1 < npb) stage(p + 1, (p + 1) & 1);
    const float* wb = w_s + (p & 1) * (NQ * 1024);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 w4 = *reinterpret_cast<const float4*>(wb + ((q * 4 + g) * 64 + lane) * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, e[q][4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, e[q][4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, e[q][4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, e[q][4 * g + 3], acc, 0, 0, 0);
      }
    if (live) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o4;
        o4.x = fmaf(__builtin_amdgcn_logf(acc[4 * g + 0]), kLN2, m);
        o4.y = fmaf(__builtin_amdgcn_logf(acc[4 * g + 1]), kLN2, m);
        o4.z = fmaf(__builtin_amdgcn_logf(acc[4 * g + 2]), kLN2, m);
        o4.w = fmaf(__builtin_amdgcn_logf(acc[4 * g + 3]), kLN2, m);
        *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = o4;
      }
    }
  }
}

template <int NQ>
hipError_t launch_nq(bool cat, dim3 grid, size_t lds, hipStream_t s, const float* arena, const int64_t* row_off,
                     const float* w, float* out, int H, int B, int Ki, int Ko) {
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, arena, row_off, w, out, H, B, Ki, Ko);
    return hipGetLastError();
  };
  return cat ? go(sum_lse_gemm_kernel<NQ, true>) : go(sum_lse_gemm_kernel<NQ, false>);
}

}  // namespace

namespace ck {

bool gemm_applies(int H, int Ki, int Ko, int mode) {
  const bool cat = mode == CK_SUM_CAT && H > 1;
  const int n = cat ? H * Ki : Ki;
  return (mode == CK_SUM_CAT || mode == CK_SUM_PROD) && Ki % 32 == 0 && Ko % 32 == 0 && n / 32 >= 1 && n / 32 <= 8;
}

// Real dense / CP-T layer with Ki, Ko multiples of 32 and at most 256 contracted inputs.
int sum_lse_gemm(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int Ki,
                 int Ko, int mode, void* stream) {
  const bool cat = mode == CK_SUM_CAT && H > 1;
  const int nq = (cat ? H * Ki : Ki) / 32;
  const size_t lds = static_cast<size_t>(2) * nq * 1024 * sizeof(float);
  const int tiles = (B + 31) / 32;
  dim3 grid((tiles + 3) / 4, F);
  return ck::dispatch(
      [=](hipStream_t s) {
        switch (nq) {
          case 1: return launch_nq<1>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 2: return launch_nq<2>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 3: return launch_nq<3>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 4: return launch_nq<4>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 5: return launch_nq<5>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 6: return launch_nq<6>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 7: return launch_nq<7>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          default: return launch_nq<8>(cat, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
        }
      },
      stream);
}

}  // namespace ck
