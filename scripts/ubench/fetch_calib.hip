// Calibration of rocprofv3's FETCH_SIZE on gfx950 per ACCESS KIND: every kernel below reads each byte of a 2 GiB
// buffer (8 x the Infinity Cache) exactly once, so bytes requested = 2 GiB and  factor = 2 GiB / (FETCH_SIZE x 1024).
//   stream16        coalesced 16 bytes per lane (the guide's calibration: factor 2)
//   gather128       random 128-byte rows, 8 lanes x global_load_dwordx4 per row  (K = 32 table rows, registers)
//   gather128_lds   the same rows through global_load_lds_dwordx4 (the leaf launch's row gathers, LDS DMA)
//   gather256       random 256-byte rows, 16 lanes x global_load_dwordx4        (K = 64 table rows)
//   stream4         coalesced 4 bytes per lane
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/fetch_calib.hip -o scripts/ubench/fetch_calib.bin
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p -- scripts/ubench/fetch_calib.bin     (scripts/fetch_calib.sh)
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
constexpr size_t kBytes = 2ull << 30;

__global__ void __launch_bounds__(256) stream16(const v4* __restrict__ src, float* __restrict__ sink, size_t n16) {
  v4 acc = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = 1.f;
}
__global__ void __launch_bounds__(256) stream4(const float* __restrict__ src, float* __restrict__ sink, size_t n4) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += gridDim.x * 256ull) acc += src[i];
  if (acc == 123.456f) sink[0] = 1.f;
}
// rows of ROW16 x 16 bytes; row r of the walk = (r * mult + 12345) mod n_rows (n_rows a power of two, mult odd: a permutation)
template <int ROW16, bool LDS>
__global__ void __launch_bounds__(256) gather(const v4* __restrict__ src, float* __restrict__ sink, size_t n_rows) {
  __shared__ v4 buf[256];
  const int lane_in_row = threadIdx.x % ROW16;
  v4 acc = 0.f;
  const size_t rows_per_pass = static_cast<size_t>(gridDim.x) * (256 / ROW16);
  for (size_t r = blockIdx.x * (256 / ROW16) + threadIdx.x / ROW16; r < n_rows; r += rows_per_pass) {
    const size_t row = (r * 2654435761ull + 12345ull) & (n_rows - 1);
    const v4* p = src + row * ROW16 + lane_in_row;
    if (LDS) {
      // (the destination of an LDS DMA is wave-uniform base + lane * 16: this wave's 64 slots)
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(buf + (threadIdx.x & ~63)), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += buf[threadIdx.x];
    } else {
      acc += *p;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = 1.f;
}

int main() {
  v4* src;
  float* sink;
  if (hipMalloc(&src, kBytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  (void)hipMemset(src, 0, kBytes);
  const size_t n16 = kBytes / 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, src, sink, n16);
    hipLaunchKernelGGL(stream4, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float*>(src), sink, kBytes / 4);
    hipLaunchKernelGGL((gather<8, false>), dim3(4096), dim3(256), 0, 0, src, sink, kBytes / 128);
    hipLaunchKernelGGL((gather<8, true>), dim3(4096), dim3(256), 0, 0, src, sink, kBytes / 128);
    hipLaunchKernelGGL((gather<16, false>), dim3(4096), dim3(256), 0, 0, src, sink, kBytes / 256);
  }
  (void)hipDeviceSynchronize();
  printf("done: every kernel read %zu bytes once\n", kBytes);
  return 0;
}
