// Micro-benchmark: what does ds_add_f32 cost when a wave adds two 32-float rows (lanes 0-31 one row, 32-63 another, rows picked
// at random among 256) of an LDS-resident (256, 32) table -- the pattern of a Categorical scatter kept on chip?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/lds_atomic.hip -o scripts/ubench/lds_atomic.bin
// Prints shader cycles per wave instruction for 1, 2, 4 waves per SIMD adding into the SAME table (4 tables of 32 KB per
// workgroup), next to plain ds_write_b32 of the same pattern.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <bool ATOMIC>
__global__ void __launch_bounds__(1024) k(long long* t, float* sink, int iters, int waves) {
  extern __shared__ float tab[];  // 4 x 256 x 32 floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 8192; i += blockDim.x) tab[i] = 0.f;
  __syncthreads();
  if (wave >= waves) return;
  unsigned s = 1234567u * (blockIdx.x * 64 + wave + 1);
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = __builtin_amdgcn_readfirstlane(s >> 16);
      const int row = (lane < 32 ? r : r >> 8) & 255;
      float* a = tab + (p & 3) * 8192 + row * 32 + (lane & 31);
      if (ATOMIC) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(a), 1.0f + p, 0, 0, false);
      else *reinterpret_cast<volatile float*>(a) = 1.0f + p;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const long long c1 = clock64();
  if (lane == 0) t[blockIdx.x * 16 + wave] = c1 - c0;
  if (iters < 0) sink[threadIdx.x] = tab[threadIdx.x];
}

int main() {
  long long* t;
  float* sink;
  hipMalloc(&t, 256 * 16 * 8);
  hipMalloc(&sink, 4096);
  const int iters = 2000;
  for (int atomic = 1; atomic >= 0; --atomic)
    for (int waves : {4, 8, 16}) {
      hipMemset(t, 0, 256 * 16 * 8);
      if (atomic) hipLaunchKernelGGL(k<true>, dim3(256), dim3(1024), 4 * 32768, 0, t, sink, iters, waves);
      else hipLaunchKernelGGL(k<false>, dim3(256), dim3(1024), 4 * 32768, 0, t, sink, iters, waves);
      hipDeviceSynchronize();
      std::vector<long long> h(256 * 16);
      hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
      double sum = 0;
      int n = 0;
      for (int b = 0; b < 256; ++b)
        for (int w = 0; w < waves; ++w) {
          sum += h[b * 16 + w];
          ++n;
        }
      printf("%s, %2d waves per workgroup (%d per SIMD): %.1f cycles per wave instruction (%.1f per instruction and CU)\n", atomic ? "ds_add_f32 " : "ds_write_b32",
             waves, waves / 4, sum / n / (16.0 * iters), sum / n / (16.0 * iters) / waves);
    }
  return 0;
}
