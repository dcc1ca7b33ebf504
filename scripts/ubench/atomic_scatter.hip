// Micro-benchmark: what does the scatter-add of the leaf gradients of a training step cost as float atomics?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/atomic_scatter.hip -o scripts/ubench/atomic_scatter.bin
// The fused backward of the north-star circuit leaves, per (leaf fold f, batch row b), 32 floats that belong to row
// x[b, f] of the (784, 257, 32) table gradient: 784 x 4096 x 32 = 103 M float additions into 25.8 MB.  Variants:
//   0  lane (b, kh) adds its 16 registers to row x_b at units 8g + 4kh + t   (register layout of the MFMA tile: one
//      instruction touches 32 rows, 8 bytes of each)
//   1  the tile transposed first (through LDS in the real kernel; here just re-indexed): lane l adds unit l & 31 of rows
//      2p + (l >> 5): one instruction touches 2 rows, 128 bytes of each
//   2  as 1, rows of a tile sorted by category first is NOT done; instead every workgroup owns a private copy?  no -- too big
//   3  plain stores instead of atomics in the pattern of 1 (the floor: the same traffic without read-modify-write)
// Tiles are dealt to 256 x 8 waves like the leaf walk: workgroup w walks the leaves of roots w % 49 ... (16 leaves per root).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int F = 784, C1 = 257, K = 32, B = 4096;

template <int VARIANT, bool FOLDMAJOR>
__global__ void __launch_bounds__(512) scatter(float* __restrict__ dt, const unsigned char* __restrict__ x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  // unit of work: (leaf fold f, tile): workgroup wg takes root = wg * 49 / 256 ..., simply: global wave id strides over (f, tile)
  const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = 1.0f + j + lane * 0.001f;
  auto unit = [&](int f, int tile) {
    const int b = tile * 32 + b_in;
    if (VARIANT == 0) {
      const int row = x[f * B + b];
      float* p = dt + (static_cast<size_t>(f) * C1 + row) * K + 4 * kh;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) atomicAdd(p + 8 * g + t, v[4 * g + t]);
    } else {
#pragma unroll
      for (int p2 = 0; p2 < 16; ++p2) {
        const int r = 2 * p2 + kh;  // row of the tile this lane serves in this instruction
        const int row = x[f * B + tile * 32 + r];
        float* p = dt + (static_cast<size_t>(f) * C1 + row) * K + b_in;
        if (VARIANT == 1) atomicAdd(p, v[p2]);
        else *p = v[p2];
      }
    }
  };
  if (FOLDMAJOR) {  // a workgroup stays on one fold (its 8 waves share the 128 tiles): all additions to a fold's 32 KB come from one CU
    for (int f = blockIdx.x; f < F; f += gridDim.x)
      for (int tile = wave; tile < B / 32; tile += 8) unit(f, tile);
  } else {
    const int n_units = F * (B / 32);
    for (int u = gw; u < n_units; u += nw) unit(u % F, u / F);
  }
}

template <int VARIANT, bool FOLDMAJOR>
void run(float* dt, unsigned char* x, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((scatter<VARIANT, FOLDMAJOR>), dim3(256), dim3(512), 0, 0, dt, x);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double n = static_cast<double>(F) * B * K;
  printf("%-40s %.3f ms   %.1f G float adds/s   (%.0f M adds)\n", name, best, n / best / 1e6, n / 1e6);
}

int main() {
  float* dt;
  unsigned char* x;
  hipMalloc(&dt, sizeof(float) * F * C1 * K);
  hipMemset(dt, 0, sizeof(float) * F * C1 * K);
  hipMalloc(&x, F * B);
  std::vector<unsigned char> hx(F * B);
  unsigned s = 12345;
  for (auto& c : hx) {
    s = s * 1664525u + 1013904223u;
    c = static_cast<unsigned char>(s >> 24);
  }
  hipMemcpy(x, hx.data(), F * B, hipMemcpyHostToDevice);
  run<0, false>(dt, x, "atomics, register layout, folds spread");
  run<1, false>(dt, x, "atomics, transposed, folds spread");
  run<3, false>(dt, x, "plain stores, transposed, folds spread");
  run<0, true>(dt, x, "atomics, register layout, fold per CU");
  run<1, true>(dt, x, "atomics, transposed, fold per CU");
  run<3, true>(dt, x, "plain stores, transposed, fold per CU");
  return 0;
}
