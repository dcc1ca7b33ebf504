// Micro-benchmark: one wave of every SIMD issues ONLY dependent MFMA chains (v_mfma_f32_32x32x2_f32 or v_mfma_f32_32x32x16_bf16),
// the other wave of the same SIMD ONLY VALU work (8 independent chains of v_fma / v_mul + v_max / v_exp / v_perm + v_and + v_sub)
// -- do they run concurrently?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form scripts/ubench/mfma_bf16_valu_split.hip -o scripts/ubench/mfma_bf16_valu_split.bin
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
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
template <int CHAINS>
__device__ __forceinline__ float run_mfma_bf16(int iters, unsigned a, unsigned b) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const u32x4v av = {a, a + 1, a + 2, a + 3}, bv = {b, b, b, b};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, av), __builtin_bit_cast(bf16x8v, bv), acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
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
        if (KIND == 2) x[k] = __builtin_amdgcn_exp2f(x[k]);  // v_exp_f32 (quarter rate)
        if (KIND == 3) x[k] = x[k] - __uint_as_float(__float_as_uint(x[k]) & 0xffff0000u) + __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(x[k]), __float_as_uint(x[(k + 1) & 7]), 0x07060302u));  // and, sub, perm, add
      }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  return s;
}

// roles: 0 = exit, 1 = fp32 MFMA, 2 = VALU fma, 3 = VALU mul+max, 4 = bf16 MFMA (one chain), 5 = bf16 MFMA (two chains), 6 = v_exp, 7 = split
__global__ void __launch_bounds__(512) split(float* out, long long* t, int roleA, int roleB, int it_m, int it_v) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;
  const long long w0 = wall_clock64();
  float s = 0.f;
  if (role == 1) s = run_mfma(it_m, 1.0f + threadIdx.x * 1e-3f, 0.5f);
  if (role == 2) s = run_valu<0>(it_v, 1.0f);
  if (role == 3) s = run_valu<1>(it_v, 1.0f);
  if (role == 4) s = run_mfma_bf16<1>(it_m, 0x3f803f80u + threadIdx.x, 0x3f003f00u);
  if (role == 5) s = run_mfma_bf16<2>(it_m, 0x3f803f80u + threadIdx.x, 0x3f003f00u);
  if (role == 6) s = run_valu<2>(it_v, 1.0f);
  if (role == 7) s = run_valu<3>(it_v / 4, 1.0f);
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
  static const char* nm[] = {"-", "MFMA f32", "VALU fma", "VALU mul+max", "MFMA bf16", "MFMA bf16 x2", "VALU exp", "VALU split/4"};
  printf("waves 0-3: %-12s waves 4-7: %-12s  ->  %8.1f us | %8.1f us  (100 MHz wall clock per wave)\n", nm[roleA], nm[roleB], a / 1024 / 100.0,
         b / 1024 / 100.0);
  hipFree(out);
  hipFree(t);
}

int main() {
  const int it_m = 4000, it_v = 4000;  // 64000 MFMAs (4.1 M cycles) | 1.02 M VALU instructions
  // 64000 MFMAs: fp32 32x32x2 = 64 cycles each (4.1 M cycles), bf16 32x32x16 = 32 cycles each if back to back (2.05 M)
  run(1, 0, it_m, it_v);
  run(4, 0, it_m, it_v);
  run(5, 0, it_m, it_v);
  run(0, 2, it_m, it_v);
  run(0, 6, it_m, it_v);
  run(0, 7, it_m, it_v);
  run(1, 2, it_m, it_v);
  run(4, 2, it_m, it_v);
  run(5, 2, it_m, it_v);
  run(4, 6, it_m, it_v);
  run(4, 7, it_m, it_v);
  run(4, 4, it_m, it_v);
  run(5, 5, it_m, it_v);
  run(2, 2, it_m, it_v);
  return 0;
}
