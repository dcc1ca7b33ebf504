// Micro-benchmark: where does the K = 64 dense log-sum-exp layer spend its time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form scripts/ubench/sum64.hip -o scripts/ubench/sum64
//   scripts/ubench/sum64 [F] [B]
// Variants of the production kernel (cirkit_amd/csrc/ck_sum.hip: sum_lse_mfma<2,2>) with parts removed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kL2E = 1.4426950408889634f, kLN2 = 0.6931471805599453f;

// FLAGS: 1 = skip MFMA, 2 = skip global loads of activations, 4 = skip exp/log, 8 = skip stores
template <int FLAGS>
__global__ void __launch_bounds__(256)
    k64(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int B, int tpw, int never) {
  constexpr int NK = 2, K = 64;
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  float wa[NK][NK][16];
  const float* wf = w + static_cast<int64_t>(f) * K * K;
#pragma unroll
  for (int p = 0; p < NK; ++p)
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t4 = *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(32 * p + b_in) * K + 32 * q + 8 * g + 4 * kh);
        wa[p][q][4 * g + 0] = t4.x; wa[p][q][4 * g + 1] = t4.y; wa[p][q][4 * g + 2] = t4.z; wa[p][q][4 * g + 3] = t4.w;
      }
  const int tile0 = (blockIdx.x * 4 + wave) * tpw;
  for (int tt = 0; tt < tpw; ++tt) {
    const int b0 = (tile0 + tt) * 32;
    if (b0 >= B) break;
    const int b = b0 + b_in;
    float v[NK][16];
    const float* src = x + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 t4;
        if (FLAGS & 2) t4 = make_float4(-1.f - g, -2.f - q, -3.f - lane, -0.5f * tt);
        else t4 = *reinterpret_cast<const float4*>(src + 32 * q + 8 * g);
        v[q][4 * g + 0] = t4.x; v[q][4 * g + 1] = t4.y; v[q][4 * g + 2] = t4.z; v[q][4 * g + 3] = t4.w;
      }
    float m = v[0][0];
    if (!(FLAGS & 4)) {
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) m = fmaxf(m, v[q][j]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const float nml = -m * kL2E;
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[q][j] = __builtin_amdgcn_exp2f(fmaf(v[q][j], kL2E, nml));
    }
    float* dst = out + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
    for (int p = 0; p < NK; ++p) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (FLAGS & 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = v[p][r] + wa[p][0][r];
      } else {
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
          for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[p][q][s], v[q][s], acc, 0, 0, 0);
      }
      if (!(FLAGS & 8) || never) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 o4;
          if (FLAGS & 4) {
            o4 = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
          } else {
            o4.x = fmaf(__builtin_amdgcn_logf(acc[4 * g + 0]), kLN2, m);
            o4.y = fmaf(__builtin_amdgcn_logf(acc[4 * g + 1]), kLN2, m);
            o4.z = fmaf(__builtin_amdgcn_logf(acc[4 * g + 2]), kLN2, m);
            o4.w = fmaf(__builtin_amdgcn_logf(acc[4 * g + 3]), kLN2, m);
          }
          *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = o4;
        }
      } else {  // keep the work alive without HBM writes
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += (FLAGS & 4) ? acc[r] : __builtin_amdgcn_logf(acc[r]);
        if (s == 1234.5678f) dst[0] = s;
      }
    }
  }
}

// Software-pipelined variant: the loads of tile t+1 are issued before the MFMA chain of tile t.
// WLDS: weights staged once per workgroup in LDS (A operand read with ds_read_b128) instead of registers.
template <bool WLDS>
__global__ void __launch_bounds__(256)
    k64_pipe(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int B, int tpw) {
  constexpr int NK = 2, K = 64;
  __shared__ __attribute__((aligned(16))) float w_s[WLDS ? 64 * 64 : 4];
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  float wa[WLDS ? 1 : NK][WLDS ? 1 : NK][16];
  const float* wf = w + static_cast<int64_t>(f) * K * K;
  if (WLDS) {
    // LDS layout: [p][q][g][lane][4]  -> one ds_read_b128 per (p, q, g), conflict-free
    for (int i = threadIdx.x; i < 1024; i += 256) {
      const int ln = i & 63, g = (i >> 6) & 3, q = (i >> 8) & 1, p = i >> 9;
      const float4 t4 = *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(32 * p + (ln & 31)) * K + 32 * q + 8 * g + 4 * (ln >> 5));
      *reinterpret_cast<float4*>(&w_s[((((p * 2 + q) * 4 + g) * 64) + ln) * 4]) = t4;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int p = 0; p < NK; ++p)
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t4 = *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(32 * p + b_in) * K + 32 * q + 8 * g + 4 * kh);
          wa[p][q][4 * g + 0] = t4.x; wa[p][q][4 * g + 1] = t4.y; wa[p][q][4 * g + 2] = t4.z; wa[p][q][4 * g + 3] = t4.w;
        }
  }
  const int tile0 = (blockIdx.x * 4 + wave) * tpw;
  const int ntiles = min(tpw, (B + 31) / 32 - tile0);
  if (ntiles <= 0) return;
  float4 nx[NK][4];
  auto issue = [&](int t) {
    const int b = min((tile0 + t) * 32 + b_in, B - 1);
    const float* src = x + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) nx[q][g] = *reinterpret_cast<const float4*>(src + 32 * q + 8 * g);
  };
  issue(0);
  for (int tt = 0; tt < ntiles; ++tt) {
    const int b = (tile0 + tt) * 32 + b_in;
    float v[NK][16];
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[q][4 * g + 0] = nx[q][g].x; v[q][4 * g + 1] = nx[q][g].y; v[q][4 * g + 2] = nx[q][g].z; v[q][4 * g + 3] = nx[q][g].w;
      }
    if (tt + 1 < ntiles) issue(tt + 1);
    float m = v[0][0];
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) m = fmaxf(m, v[q][j]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float nml = -m * kL2E;
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) v[q][j] = __builtin_amdgcn_exp2f(fmaf(v[q][j], kL2E, nml));
    float* dst = out + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
#pragma unroll
    for (int p = 0; p < NK; ++p) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 w4;
          if (WLDS) w4 = *reinterpret_cast<const float4*>(&w_s[((((p * 2 + q) * 4 + g) * 64) + lane) * 4]);
          else w4 = make_float4(wa[WLDS ? 0 : p][WLDS ? 0 : q][4 * g], wa[WLDS ? 0 : p][WLDS ? 0 : q][4 * g + 1], wa[WLDS ? 0 : p][WLDS ? 0 : q][4 * g + 2], wa[WLDS ? 0 : p][WLDS ? 0 : q][4 * g + 3]);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, v[q][4 * g + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, v[q][4 * g + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, v[q][4 * g + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, v[q][4 * g + 3], acc, 0, 0, 0);
        }
      if (b < B) {
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
}

// Wave specialisation probe: waves 0-1 of a block only move data (2 tiles each per slot), waves 2-3
// only run the MFMA chains of 2 tiles each: can the memory system and the matrix pipe overlap at all?
__global__ void __launch_bounds__(256)
    k64_split(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int B, int tpw) {
  constexpr int K = 64;
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int tile0 = blockIdx.x * 4 * tpw;
  if (wave < 2) {
    for (int tt = wave; tt < 4 * tpw; tt += 2) {
      const int b = (tile0 + tt) * 32 + b_in;
      if (b >= B) break;
      const float* src = x + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
      float* dst = out + (static_cast<int64_t>(f) * B + b) * K + 4 * kh;
      float4 t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<const float4*>(src + 8 * i);
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(dst + 8 * i) = t[i];
    }
    return;
  }
  float wa[2][2][16];
  const float* wf = w + static_cast<int64_t>(f) * K * K;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t4 = *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(32 * p + b_in) * K + 32 * q + 8 * g + 4 * kh);
        wa[p][q][4 * g + 0] = t4.x; wa[p][q][4 * g + 1] = t4.y; wa[p][q][4 * g + 2] = t4.z; wa[p][q][4 * g + 3] = t4.w;
      }
  float s = 0.f;
  for (int tt = wave - 2; tt < 4 * tpw; tt += 2) {
    float v[2][16];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) v[q][j] = -1.f - 0.01f * (j + q + tt + lane);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[p][q][j], v[q][j], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[r];
    }
  }
  if (s == 1234.5678f) out[0] = s;
}

// Thread-level parallelism instead of per-wave pipelining: ONE tile per wave, weights shared by the
// workgroup through LDS, so a wave needs ~70 VGPRs and 6+ waves per SIMD hide the memory latency.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    k64_tlp(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int B) {
  constexpr int K = 64;
  __shared__ __attribute__((aligned(16))) float w_s[64 * 64];
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (blockIdx.x * WAVES + wave) * 32 + b_in;
  const int bl = min(b, B - 1);
  const float* src = x + (static_cast<int64_t>(f) * B + bl) * K + 4 * kh;
  float v[2][16];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t4 = *reinterpret_cast<const float4*>(src + 32 * q + 8 * g);
      v[q][4 * g + 0] = t4.x; v[q][4 * g + 1] = t4.y; v[q][4 * g + 2] = t4.z; v[q][4 * g + 3] = t4.w;
    }
  const float* wf = w + static_cast<int64_t>(f) * K * K;
  for (int i = threadIdx.x; i < 1024; i += WAVES * 64) {
    const int ln = i & 63, g = (i >> 6) & 3, q = (i >> 8) & 1, p = i >> 9;
    const float4 t4 = *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(32 * p + (ln & 31)) * K + 32 * q + 8 * g + 4 * (ln >> 5));
    *reinterpret_cast<float4*>(&w_s[((((p * 2 + q) * 4 + g) * 64) + ln) * 4]) = t4;
  }
  float m = v[0][0];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, v[q][j]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float nml = -m * kL2E;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) v[q][j] = __builtin_amdgcn_exp2f(fmaf(v[q][j], kL2E, nml));
  __syncthreads();
  float* dst = out + (static_cast<int64_t>(f) * B + bl) * K + 4 * kh;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 w4 = *reinterpret_cast<const float4*>(&w_s[((((p * 2 + q) * 4 + g) * 64) + lane) * 4]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, v[q][4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, v[q][4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, v[q][4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, v[q][4 * g + 3], acc, 0, 0, 0);
      }
    if (b < B) {
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

// pure streaming copy with the same access pattern and a plain coalesced one, for reference
__global__ void __launch_bounds__(256) copy_tile(const float* __restrict__ x, float* __restrict__ out, int B, int tpw) {
  const int f = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b_in = lane & 31, kh = lane >> 5;
  const int tile0 = (blockIdx.x * 4 + wave) * tpw;
  for (int tt = 0; tt < tpw; ++tt) {
    const int b = (tile0 + tt) * 32 + b_in;
    if (b >= B) break;
    const float* src = x + (static_cast<int64_t>(f) * B + b) * 64 + 4 * kh;
    float* dst = out + (static_cast<int64_t>(f) * B + b) * 64 + 4 * kh;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(dst + 8 * i) = *reinterpret_cast<const float4*>(src + 8 * i);
  }
}
__global__ void __launch_bounds__(256) copy_flat(const float4* __restrict__ x, float4* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) out[i] = x[i];
}

template <typename L>
float time_ms(L launch, int iters = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main(int argc, char** argv) {
  const int F = argc > 1 ? atoi(argv[1]) : 1060, B = argc > 2 ? atoi(argv[2]) : 4096;
  const size_t n = static_cast<size_t>(F) * B * 64;
  float *x, *w, *o;
  hipMalloc(&x, n * 4); hipMalloc(&o, n * 4); hipMalloc(&w, static_cast<size_t>(F) * 4096 * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = -static_cast<float>((i * 2654435761u) % 1000) * 0.01f;
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<float> hw(static_cast<size_t>(F) * 4096, 1.f / 64);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const double gb = 2.0 * n * 4 / 1e9, gflop = 2.0 * F * B * 64 * 64 / 1e9;
  for (int tpw : {2, 4, 8, 16}) {
    const int tiles = (B + 31) / 32;
    dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), F), block(256);
    auto rep = [&](const char* name, float ms) {
      printf("tpw=%d %-34s %8.3f ms  %7.1f GB/s  %6.1f TFLOP/s\n", tpw, name, ms, gb / ms * 1e3, gflop / ms);
    };
    rep("full", time_ms([&] { hipLaunchKernelGGL(k64<0>, grid, block, 0, 0, x, w, o, B, tpw, 0); }));
    rep("no mfma", time_ms([&] { hipLaunchKernelGGL(k64<1>, grid, block, 0, 0, x, w, o, B, tpw, 0); }));
    rep("no loads/stores (mfma+exp/log)", time_ms([&] { hipLaunchKernelGGL(k64<10>, grid, block, 0, 0, x, w, o, B, tpw, 0); }));
    rep("mfma only (no mem, no exp/log)", time_ms([&] { hipLaunchKernelGGL(k64<14>, grid, block, 0, 0, x, w, o, B, tpw, 0); }));
    rep("mem + mfma, no exp/log", time_ms([&] { hipLaunchKernelGGL(k64<4>, grid, block, 0, 0, x, w, o, B, tpw, 0); }));
    rep("mem only (no mfma, no exp/log)", time_ms([&] { hipLaunchKernelGGL(k64<5>, grid, block, 0, 0, x, w, o, B, tpw, 0); }));
    rep("pipelined, W in registers", time_ms([&] { hipLaunchKernelGGL(k64_pipe<false>, grid, block, 0, 0, x, w, o, B, tpw); }));
    rep("pipelined, W in LDS", time_ms([&] { hipLaunchKernelGGL(k64_pipe<true>, grid, block, 0, 0, x, w, o, B, tpw); }));
    rep("split waves: 2 copy + 2 mfma", time_ms([&] { hipLaunchKernelGGL(k64_split, grid, block, 0, 0, x, w, o, B, tpw); }));
    rep("copy, tile access pattern", time_ms([&] { hipLaunchKernelGGL(copy_tile, grid, block, 0, 0, x, o, B, tpw); }));
  }
  {
    const int tiles = (B + 31) / 32;
    auto rep = [&](const char* name, float ms) { printf("%-40s %8.3f ms  %7.1f GB/s  %6.1f TFLOP/s\n", name, ms, gb / ms * 1e3, gflop / ms); };
    rep("tlp, 4 waves/block", time_ms([&] { hipLaunchKernelGGL(k64_tlp<4>, dim3((tiles + 3) / 4, F), dim3(256), 0, 0, x, w, o, B); }));
    rep("tlp, 8 waves/block", time_ms([&] { hipLaunchKernelGGL(k64_tlp<8>, dim3((tiles + 7) / 8, F), dim3(512), 0, 0, x, w, o, B); }));
    rep("tlp, 16 waves/block", time_ms([&] { hipLaunchKernelGGL(k64_tlp<16>, dim3((tiles + 15) / 16, F), dim3(1024), 0, 0, x, w, o, B); }));
  }
  float ms = time_ms([&] { hipLaunchKernelGGL(copy_flat, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(o), static_cast<int64_t>(n / 4)); });
  printf("flat copy %8.3f ms  %7.1f GB/s\n", ms, gb / ms * 1e3);
  return 0;
}
