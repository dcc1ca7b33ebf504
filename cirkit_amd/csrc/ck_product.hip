// Product-type inner layers in log space: Hadamard (elementwise sum over the arity axis) and
// Kronecker (outer sum).  Pure streaming kernels; children are addressed through arena offsets so
// the reference's materialising gather (circuits.py:42-47) never happens.
#include <algorithm>

#include "ck_internal.h"

namespace {

// W = words (floats) per logical row = K * esize.  Vector path: W % 4 == 0.
__global__ void __launch_bounds__(256)
    hadamard_vec(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                 float* __restrict__ out, int H, int64_t words_per_fold /* B*W */, int esize) {
  const int f = blockIdx.y;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t n4 = words_per_fold >> 2;
  float4* dst = reinterpret_cast<float4*>(out + static_cast<int64_t>(f) * words_per_fold);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    float4 acc = reinterpret_cast<const float4*>(arena + ro[0] * esize)[i];
    for (int h = 1; h < H; ++h) {
      const float4 t = reinterpret_cast<const float4*>(arena + ro[h] * esize)[i];
      acc.x += t.x;
      acc.y += t.y;
      acc.z += t.z;
      acc.w += t.w;
    }
    dst[i] = acc;
  }
}

__global__ void __launch_bounds__(256)
    hadamard_scalar(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                    float* __restrict__ out, int H, int64_t words_per_fold, int esize) {
  const int f = blockIdx.y;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  float* dst = out + static_cast<int64_t>(f) * words_per_fold;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < words_per_fold;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    float acc = arena[ro[0] * esize + i];
    for (int h = 1; h < H; ++h) acc += arena[ro[h] * esize + i];
    dst[i] = acc;
  }
}

// out[f,b,i*K+j] = x0[f,b,i] + x1[f,b,j]; esize words per element.
__global__ void __launch_bounds__(256)
    kronecker_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                     float* __restrict__ out, int B, int K, int esize) {
  const int f = blockIdx.y;
  const int64_t o0 = row_off[2 * f] * esize, o1 = row_off[2 * f + 1] * esize;
  const int64_t kk = static_cast<int64_t>(K) * K;
  const int64_t n = static_cast<int64_t>(B) * kk;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t b = i / kk;
    const int r = static_cast<int>(i - b * kk);
    const int ii = r / K, jj = r - ii * K;
    for (int c = 0; c < esize; ++c)
      out[(static_cast<int64_t>(f) * n + i) * esize + c] =
          arena[o0 + (b * K + ii) * esize + c] + arena[o1 + (b * K + jj) * esize + c];
  }
}

}  // namespace

extern "C" {

int ck_hadamard_fwd(const float* arena, const int64_t* row_off, float* out, int F, int H, int B,
                    int K, int esize, void* stream) {
  CK_REQUIRE(arena && row_off && out, "ck_hadamard_fwd: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && K > 0, "ck_hadamard_fwd: non-positive size");
  CK_REQUIRE(esize == 1 || esize == 2, "ck_hadamard_fwd: esize must be 1 or 2");
  CK_REQUIRE(F <= 65535, "ck_hadamard_fwd: F=%d exceeds grid.y", F);
  const int64_t words = static_cast<int64_t>(B) * K * esize;
  // row_off counts activation ELEMENTS (esize words each); the kernels index 4-byte words.
  const bool vec = (words % 4 == 0) && ck::aligned16(arena) && ck::aligned16(out);
  if (vec) {
    dim3 grid(static_cast<unsigned>(std::min<int64_t>((words / 4 + 255) / 256, 2048)), F), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(hadamard_vec, grid, block, 0, s, arena, row_off, out, H, words, esize);
          return hipGetLastError();
        },
        stream);
  }
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((words + 255) / 256, 2048)), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(hadamard_scalar, grid, block, 0, s, arena, row_off, out, H, words, esize);
        return hipGetLastError();
      },
      stream);
}

int ck_kronecker_fwd(const float* arena, const int64_t* row_off, float* out, int F, int B, int K,
                     int esize, void* stream) {
  CK_REQUIRE(arena && row_off && out, "ck_kronecker_fwd: null pointer");
  CK_REQUIRE(F > 0 && B > 0 && K > 0, "ck_kronecker_fwd: non-positive size");
  CK_REQUIRE(esize == 1 || esize == 2, "ck_kronecker_fwd: esize must be 1 or 2");
  CK_REQUIRE(F <= 65535, "ck_kronecker_fwd: F=%d exceeds grid.y", F);
  const int64_t n = static_cast<int64_t>(B) * K * K;
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 2048)), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(kronecker_kernel, grid, block, 0, s, arena, row_off, out, B, K, esize);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
