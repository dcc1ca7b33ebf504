// Micro-benchmark: one wave of every SIMD issues ONLY dependent v_mfma_f32_32x32x2_f32 chains, the other wave of the same
// SIMD ONLY fp32 VALU work (8 independent v_fma chains) -- do they run concurrently?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form scripts/ubench/mfma_valu_split.hip -o scripts/ubench/mfma_valu_split.bin
// Workgroups of 8 waves, one per CU: waves 0..3 (one per SIMD) run role A, waves 4..7 role B.  Roles: M = MFMA chains,
// V = VALU, - = exit at once.  Printed: time per role when run alone and together.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float run_mfma(int iters, float a, float b) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  return s;
}
template <int KIND>
__device__ __forceinline__ float run_valu(int iters, float a) {
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = a * (k + 1) + threadIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (KIND == 0) x[k] = fmaf(x[k], 0.999f, 1e-3f);
        if (KIND == 1) x[k] = fmaxf(x[k] * 0.999f, 1e-3f);  // v_mul + v_max
      }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  return s;
}

// roles: 0 = exit, 1 = MFMA, 2 = VALU fma, 3 = VALU mul+max
__global__ void __launch_bounds__(512) split(float* out, long long* t, int roleA, int roleB, int it_m, int it_v) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;
  const long long w0 = wall_clock64();
  float s = 0.f;
  if (role == 1) s = run_mfma(it_m, 1.0f + threadIdx.x * 1e-3f, 0.5f);
  if (role == 2) s = run_valu<0>(it_v, 1.0f);
  if (role == 3) s = run_valu<1>(it_v, 1.0f);
  const long long w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + wave] = w1 - w0;
}

void run(int roleA, int roleB, int it_m, int it_v) {
  float* out;
  long long* t;
  hipMalloc(&out, sizeof(float) * 256 * 512);
  hipMalloc(&t, sizeof(long long) * 256 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(split, dim3(256), dim3(512), 0, 0, out, t, roleA, roleB, it_m, it_v);
    hipDeviceSynchronize();
  }
  std::vector<long long> h(256 * 8);
  hipMemcpy(h.data(), t, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int g = 0; g < 256; ++g)
    for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[g * 8 + w];
  static const char* nm[] = {"-", "MFMA", "VALU fma", "VALU mul+max"};
  printf("waves 0-3: %-12s waves 4-7: %-12s  ->  %8.1f us | %8.1f us  (100 MHz wall clock per wave)\n", nm[roleA], nm[roleB], a / 1024 / 100.0,
         b / 1024 / 100.0);
  hipFree(out);
  hipFree(t);
}

int main() {
  const int it_m = 4000, it_v = 4000;  // 64000 MFMAs (4.1 M cycles) | 1.02 M VALU instructions
  run(1, 0, it_m, it_v);
  run(0, 2, it_m, it_v);
  run(0, 3, it_m, it_v);
  run(1, 2, it_m, it_v);
  run(1, 3, it_m, it_v);
  run(1, 1, it_m, it_v);
  run(2, 2, it_m, it_v);
  return 0;
}
