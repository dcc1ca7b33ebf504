// TorchEinsumParameter (parameters/optimized.py:282-284): an einsum over the per-fold values of up to four parameter
// nodes, real or complex -- out[f, o...] = sum over the contracted indices of prod_k x_k[f, idx_k...].  The two-real-matrix
// patterns of the squared circuits go to ck_param_bmm; this is every other pattern (three or four operands, repeated or
// batch indices, complex operands -- products of more than two circuits, complex parameters).  Parameters are small: one
// thread per output element walks the contracted index space (correct for any pattern, not tuned).
#include "ck_internal.h"

namespace {

using ck::c32;
constexpr int kMaxOps = CK_EINSUM_MAX_OPERANDS, kMaxIdx = CK_EINSUM_MAX_INDICES;

struct EinsumArgs {
  const float* x[kMaxOps];
  float* out;
  int n_ops, n_out, n_red, out_complex;
  int cplx[kMaxOps];
  int ext_out[kMaxIdx], ext_red[kMaxIdx];
  int64_t s_out[kMaxOps][kMaxIdx], s_red[kMaxOps][kMaxIdx];  // element strides of operand k along output / contracted index i
  int64_t fold_stride[kMaxOps];
  int64_t per_fold_out, total;  // output elements per fold, F * per_fold_out
};

__global__ void __launch_bounds__(256) einsum_kernel(const EinsumArgs a) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= a.total) return;
  const int64_t f = t / a.per_fold_out;
  int64_t rem = t - f * a.per_fold_out;
  int64_t base[kMaxOps];
  for (int k = 0; k < a.n_ops; ++k) base[k] = f * a.fold_stride[k];
  for (int i = a.n_out - 1; i >= 0; --i) {
    const int64_t v = rem % a.ext_out[i];
    rem /= a.ext_out[i];
    for (int k = 0; k < a.n_ops; ++k) base[k] += v * a.s_out[k][i];
  }
  int64_t n_red = 1;
  for (int i = 0; i < a.n_red; ++i) n_red *= a.ext_red[i];
  float acc_re = 0.f, acc_im = 0.f;
  for (int64_t r = 0; r < n_red; ++r) {
    int64_t rr = r;
    int64_t off[kMaxOps];
    for (int k = 0; k < a.n_ops; ++k) off[k] = base[k];
    for (int i = a.n_red - 1; i >= 0; --i) {
      const int64_t v = rr % a.ext_red[i];
      rr /= a.ext_red[i];
      for (int k = 0; k < a.n_ops; ++k) off[k] += v * a.s_red[k][i];
    }
    float pr = 1.f, pi = 0.f;
    for (int k = 0; k < a.n_ops; ++k) {
      float xr, xi = 0.f;
      if (a.cplx[k]) {
        xr = a.x[k][2 * off[k]];
        xi = a.x[k][2 * off[k] + 1];
      } else {
        xr = a.x[k][off[k]];
      }
      const float nr = pr * xr - pi * xi, ni = pr * xi + pi * xr;
      pr = nr;
      pi = ni;
    }
    acc_re += pr;
    acc_im += pi;
  }
  if (a.out_complex) {
    a.out[2 * t] = acc_re;
    a.out[2 * t + 1] = acc_im;
  } else {
    a.out[t] = acc_re;
  }
}

}  // namespace

extern "C" int ck_param_einsum(const ck_einsum_desc* d, void* stream) {
  CK_REQUIRE(d != nullptr && d->out != nullptr, "ck_param_einsum: null descriptor or output");
  CK_REQUIRE(d->n_ops >= 1 && d->n_ops <= kMaxOps, "ck_param_einsum: %d operands (1 .. %d)", d->n_ops, kMaxOps);
  CK_REQUIRE(d->n_idx >= 0 && d->n_idx <= kMaxIdx && d->n_out >= 0 && d->n_out <= d->n_idx, "ck_param_einsum: %d indices of which %d kept (at most %d)",
             d->n_idx, d->n_out, kMaxIdx);
  CK_REQUIRE(d->F > 0, "ck_param_einsum: F must be positive");
  EinsumArgs a{};
  a.n_ops = d->n_ops;
  a.n_out = d->n_out;
  a.n_red = d->n_idx - d->n_out;
  a.out = d->out;
  a.out_complex = d->out_complex;
  a.per_fold_out = 1;
  // indices 0 .. n_out - 1 are the output's, in its order; the rest are contracted
  for (int i = 0; i < d->n_idx; ++i) {
    CK_REQUIRE(d->extent[i] > 0, "ck_param_einsum: index %d has extent %d", i, d->extent[i]);
    if (i < d->n_out) {
      a.ext_out[i] = d->extent[i];
      a.per_fold_out *= d->extent[i];
    } else {
      a.ext_red[i - d->n_out] = d->extent[i];
    }
  }
  bool any_complex = false;
  for (int k = 0; k < d->n_ops; ++k) {
    CK_REQUIRE(d->x[k] != nullptr, "ck_param_einsum: operand %d is null", k);
    a.x[k] = d->x[k];
    a.cplx[k] = d->is_complex[k];
    any_complex = any_complex || d->is_complex[k] != 0;
    a.fold_stride[k] = d->fold_stride[k];
    for (int i = 0; i < d->n_idx; ++i) {
      if (i < d->n_out) a.s_out[k][i] = d->stride[k][i];
      else a.s_red[k][i - d->n_out] = d->stride[k][i];
    }
  }
  CK_REQUIRE(!any_complex || d->out_complex, "ck_param_einsum: complex operands need a complex output");
  a.total = a.per_fold_out * d->F;
  const dim3 grid(static_cast<unsigned>((a.total + 255) / 256)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(einsum_kernel, grid, block, 0, s, a);
        return hipGetLastError();
      },
      stream);
}
