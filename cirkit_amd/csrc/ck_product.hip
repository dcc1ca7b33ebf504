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

// out[f, b, r] = sum_h x_h[f, b, digit_h(r)], r = sum_h digit_h K^(H-1-h) (child 0 is the most significant digit:
// flatten(y0[..., None] + x_i[..., None, :]) repeated, inner.py:178-187); esize words per element.
__global__ void __launch_bounds__(256)
    kronecker_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                     float* __restrict__ out, int H, int B, int K, int64_t kk, int esize) {
  const int f = blockIdx.y;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int64_t n = static_cast<int64_t>(B) * kk;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t b = i / kk;
    int64_t r = i - b * kk;
    float acc[2] = {0.f, 0.f};
    for (int h = H - 1; h >= 0; --h) {
      const int64_t q = r / K;
      const int d = static_cast<int>(r - q * K);
      r = q;
      const float* src = arena + (ro[h] + b * K + d) * esize;
      acc[0] += src[0];
      if (esize == 2) acc[1] += src[1];
    }
    for (int c = 0; c < esize; ++c) out[(static_cast<int64_t>(f) * n + i) * esize + c] = acc[c];
  }
}

}  // namespace

extern "C" {

int ck_hadamard_fwd(const float* arena, const int64_t* row_off, float* out, int F, int H, int B,
                    int K, int esize, void* stream) {
  CK_REQUIRE(arena && row_off && out, "ck_hadamard_fwd: null pointer");
  CK_REQUIRE(F > 0 && H > 0 && B > 0 && K > 0, "ck_hadamard_fwd: non-positive size");
  CK_REQUIRE(esize == 1 || esize == 2, "ck_hadamard_fwd: esize must be 1 or 2");
  const int64_t words = static_cast<int64_t>(B) * K * esize;
  if (F > ck::kMaxFoldsPerLaunch)
    return ck::chunk_folds(F, [&](int f0, int n) {
      return ck_hadamard_fwd(arena, row_off + static_cast<int64_t>(f0) * H, out + f0 * words, n, H, B, K, esize, stream);
    });
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

int ck_kronecker_fwd(const float* arena, const int64_t* row_off, float* out, int F, int H, int B, int K,
                     int esize, void* stream) {
  CK_REQUIRE(arena && row_off && out, "ck_kronecker_fwd: null pointer");
  CK_REQUIRE(F > 0 && H >= 2 && B > 0 && K > 0, "ck_kronecker_fwd: non-positive size or arity < 2");
  CK_REQUIRE(esize == 1 || esize == 2, "ck_kronecker_fwd: esize must be 1 or 2");
  int64_t kk = 1;
  for (int h = 0; h < H; ++h) {
    kk *= K;
    CK_REQUIRE(kk <= (int64_t{1} << 31), "ck_kronecker_fwd: K^H = %d^%d output units", K, H);
  }
  if (F > ck::kMaxFoldsPerLaunch)
    return ck::chunk_folds(F, [&](int f0, int n) {
      return ck_kronecker_fwd(arena, row_off + static_cast<int64_t>(f0) * H, out + static_cast<int64_t>(f0) * B * kk * esize, n, H, B, K,
                              esize, stream);
    });
  const int64_t n = static_cast<int64_t>(B) * kk;
  dim3 grid(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 2048)), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(kronecker_kernel, grid, block, 0, s, arena, row_off, out, H, B, K, kk, esize);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
