// Micro-benchmark: sustained shader clock and issue rate of v_mfma_f32_32x32x2_f32 on every SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form scripts/ubench/mfma_clock.hip -o scripts/ubench/mfma_clock.bin
//   scripts/ubench/mfma_clock.bin [waves per SIMD] [chains per wave]
// Prints, per configuration: cycles per MFMA per SIMD (shader clock, s_memtime), the shader clock implied by the
// constant 100 MHz wall clock (s_memrealtime), and the fp32 matrix TFLOP/s this corresponds to.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma_loop(float* out, long long* t, int iters, float a, float b) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    t[2 * blockIdx.x] = c1 - c0;
    t[2 * blockIdx.x + 1] = w1 - w0;
  }
}

template <int CHAINS>
void run(int wps, int iters) {
  const int wgs = 256 * wps;  // one 4-wave workgroup per CU and wave-per-SIMD
  float* out;
  long long* t;
  hipMalloc(&out, sizeof(float) * wgs * 256);
  hipMalloc(&t, sizeof(long long) * 2 * wgs);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(wgs), dim3(256), 0, 0, out, t, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(2 * wgs);
  hipMemcpy(h.data(), t, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < wgs; ++i) {
    cyc += h[2 * i];
    wall += h[2 * i + 1];
  }
  cyc /= wgs;
  wall /= wgs;
  const double mfmas = 16.0 * CHAINS * iters;  // per wave
  const double ghz = cyc / (wall / 100e6) / 1e9;
  const double tflops = 4096.0 * mfmas * wgs * 4 / (ms * 1e-3) / 1e12;
  printf("waves/SIMD %d chains %d: %.1f shader cycles per MFMA and SIMD, shader clock %.2f GHz, kernel %.3f ms, %.1f TFLOP/s\n", wps,
         CHAINS, cyc / (mfmas * wps), ghz, ms, tflops);
  hipFree(out);
  hipFree(t);
}

int main(int argc, char** argv) {
  const int iters = 20000;
  for (int wps : {1, 2, 4}) {
    run<1>(wps, iters / wps);
    run<2>(wps, iters / wps / 2);
  }
  return 0;
}
