// Micro-benchmark: does fp32 VALU work co-execute with v_mfma_f32_32x32x2_f32 on a gfx950 SIMD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form scripts/ubench/mfma_valu.hip -o scripts/ubench/mfma_valu.bin
// One iteration = a dependent chain of 16 MFMAs (1024 cycles of matrix-pipe time) with NV independent VALU instructions
// placed NV/16 after each MFMA in program order (sched_group_barrier).  Printed: shader cycles per iteration and SIMD for
// 1 and 2 waves per SIMD and three kinds of filler (v_fma_f32, v_exp_f32, v_max_f32 + v_mul_f32).  If the two pipes
// overlap, the time stays at ~1024 (x waves per SIMD) until the filler alone exceeds it; if they share the fp32 lanes the
// times add.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ void __launch_bounds__(256) mix_loop(float* out, long long* t, int iters, float a, float b) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int PER = NV / 16;
  float x[PER > 0 ? PER : 1];
#pragma unroll
  for (int k = 0; k < (PER > 0 ? PER : 1); ++k) x[k] = a * (k + 1) + threadIdx.x * 1e-6f;
  const long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if (KIND == 0) x[k] = fmaf(x[k], 0.999f, 1e-3f);
        if (KIND == 1) x[k] = __builtin_amdgcn_exp2f(x[k]) * 0.5f;
        if (KIND == 2) x[k] = fmaxf(x[k] * 0.999f, 1e-3f);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (PER > 0) __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? PER : 2 * PER, 0);
    }
  }
  const long long c1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
#pragma unroll
  for (int k = 0; k < (PER > 0 ? PER : 1); ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}

template <int NV, int KIND>
void run(int wps, int iters) {
  const int wgs = 256 * wps;
  float* out;
  long long* t;
  hipMalloc(&out, sizeof(float) * wgs * 256);
  hipMalloc(&t, sizeof(long long) * wgs);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_loop<NV, KIND>), dim3(wgs), dim3(256), 0, 0, out, t, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(wgs);
  hipMemcpy(h.data(), t, sizeof(long long) * wgs, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (int i = 0; i < wgs; ++i) cyc += h[i];
  cyc /= wgs;
  static const char* names[] = {"v_fma", "v_exp+v_mul", "v_mul+v_max"};
  printf("waves/SIMD %d  filler %-12s NV %4d : %7.1f cycles per (16-MFMA chain + NV VALU) per wave, %7.1f per SIMD-wave-slot, kernel %.3f ms\n",
         wps, names[KIND], NV * (KIND == 0 ? 1 : 2), cyc / iters, cyc / iters / wps, ms);
  hipFree(out);
  hipFree(t);
}

int main() {
  const int iters = 4000;
  for (int wps : {1, 2}) {
    run<0, 0>(wps, iters);
    run<64, 0>(wps, iters);
    run<128, 0>(wps, iters);
    run<256, 0>(wps, iters);
    run<512, 0>(wps, iters);
    run<64, 1>(wps, iters);
    run<128, 1>(wps, iters);
    run<64, 2>(wps, iters);
    run<128, 2>(wps, iters);
    run<256, 2>(wps, iters);
  }
  return 0;
}
