// Micro-benchmark: what does a hand-over of a 64 KB block between two workgroups cost INSIDE one launch (a dataflow step list
// instead of one launch per level)?  Workgroup i (ticket order) waits for flag[i - 1], reads block i - 1 (written by another
// workgroup, as a rule on another XCD), writes block i = block i - 1 + 1, publishes it, raises flag[i].
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/flag_chain.hip -o scripts/ubench/flag_chain.bin
// Variants: FENCE  = plain loads / stores, release fence (agent) before the flag, acquire fence after the wait;
//           ATOMIC = agent-scope atomic loads / stores of the data (write-through, past the non-coherent caches), no fences.
// Prints microseconds per hop and checks the last block.  Then the same with LEVELS of W workgroups (every workgroup of level l
// waits until all W of level l - 1 are done and reads one of their blocks): the cost of a level boundary without a launch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kBlock = 16384;  // floats per block (64 KB)

typedef float v4 __attribute__((ext_vector_type(4)));
template <bool ATOMIC>
__device__ __forceinline__ void copy_plus_one(const float* src, float* dst, int n4) {
  // all loads first (16 per thread for a 64 KB block), then the stores
  v4 t[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = threadIdx.x + 256 * u;
    if (i < n4) {
      if (ATOMIC) {
        for (int k = 0; k < 4; ++k) t[u][k] = __hip_atomic_load(src + 4 * i + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        t[u] = reinterpret_cast<const v4*>(src)[i];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = threadIdx.x + 256 * u;
    if (i < n4) {
      if (ATOMIC) {
        for (int k = 0; k < 4; ++k) __hip_atomic_store(dst + 4 * i + k, t[u][k] + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        reinterpret_cast<v4*>(dst)[i] = t[u] + 1.f;
      }
    }
  }
}

template <bool ATOMIC>
__global__ void __launch_bounds__(256) chain(float* blocks, unsigned* flags, unsigned* ticket, unsigned epoch, int width, int n4) {
  __shared__ unsigned me;
  if (threadIdx.x == 0) me = atomicAdd(ticket, 1u);
  __syncthreads();
  const unsigned i = me;
  const unsigned level = i / width;
  if (level > 0) {
    if (threadIdx.x == 0) {
      // wait for every workgroup of the level below (width 1: the chain)
      const unsigned* f = flags + (level - 1);
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * width) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (!ATOMIC) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    copy_plus_one<ATOMIC>(blocks + static_cast<size_t>(i - width) * kBlock, blocks + static_cast<size_t>(i) * kBlock, n4);
  } else {
    for (int k = threadIdx.x; k < kBlock; k += 256) {
      if (ATOMIC) __hip_atomic_store(blocks + static_cast<size_t>(i) * kBlock + k, static_cast<float>(epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else blocks[static_cast<size_t>(i) * kBlock + k] = static_cast<float>(epoch);
    }
  }
  if (ATOMIC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + level, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool ATOMIC>
void run(int n, int width, int n4, const char* name) {
  float* blocks;
  unsigned *flags, *ticket;
  hipMalloc(&blocks, static_cast<size_t>(n) * kBlock * 4);
  hipMalloc(&flags, 4096 * 4);
  hipMalloc(&ticket, 4);
  hipMemset(flags, 0, 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (unsigned epoch = 1; epoch <= 6; ++epoch) {
    hipMemset(ticket, 0, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<ATOMIC>, dim3(n), dim3(256), 0, 0, blocks, flags, ticket, epoch, width, n4);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (epoch > 1 && ms < best) best = ms;
  }
  std::vector<float> h(kBlock);
  hipMemcpy(h.data(), blocks + static_cast<size_t>(n - 1) * kBlock, kBlock * 4, hipMemcpyDeviceToHost);
  const float want = 6.f + (n - 1) / width;
  int bad = 0;
  for (float v : h) bad += v != want;
  printf("%-7s %6d B blocks, %4d workgroups, levels of %3d: %8.2f us total, %6.2f us per level   (last block %s: %g, want %g)\n", name, 16 * n4, n, width,
         best * 1e3f, best * 1e3f / (n / width), bad ? "WRONG" : "ok", h[0], want);
  hipFree(blocks);
  hipFree(flags);
  hipFree(ticket);
}

int main() {
  for (int n4 : {64, 4096})
    for (int width : {1, 8, 64}) {
      run<false>(width == 1 ? 256 : 512, width, n4, "fence");
      run<true>(width == 1 ? 256 : 512, width, n4, "atomic");
    }
  return 0;
}
