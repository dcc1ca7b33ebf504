// Micro-benchmark of the TensorDot launches of a squared circuit's partition function (ck_tensordot2_lse_fwd / _bwd on ONE row,
// K = 32, complex): microseconds per launch for F folds (HIP events over a chain of dependent launches) and, in a lab build, the
// wall-clock stamps of workgroup (0, 0) inside one launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCK_TD_STAMPS -Iinclude -Icirkit_amd/csrc scripts/ubench/td_stamps.hip \
//         cirkit_amd/csrc/ck_runtime.hip -o scripts/ubench/td_stamps.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ck_backward_c.hip"

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                               \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  const int K = 32;
  const int folds[] = {1, 2, 6, 24, 98, 392};
  const int reps = 50;
  for (int F : folds) {
    const size_t blk = static_cast<size_t>(K) * K;  // complex values per (fold, row) block
    std::vector<float> hx(2 * blk * F), hw(static_cast<size_t>(F) * K * K), hg(2 * blk * F);
    srand(7);
    for (auto& v : hx) v = -3.f + 6.f * (rand() / static_cast<float>(RAND_MAX));
    for (auto& v : hw) v = -1.f + 2.f * (rand() / static_cast<float>(RAND_MAX));
    for (auto& v : hg) v = -1.f + 2.f * (rand() / static_cast<float>(RAND_MAX));
    float *x, *gx, *mid, *gmid, *out, *gout, *w1, *w2, *dw1, *dw2;
    int64_t* ro;
    CHECK(hipMalloc(&x, hx.size() * 4));
    CHECK(hipMalloc(&gx, hx.size() * 4));
    CHECK(hipMalloc(&mid, hx.size() * 4));
    CHECK(hipMalloc(&gmid, hx.size() * 4));
    CHECK(hipMalloc(&out, hx.size() * 4));
    CHECK(hipMalloc(&gout, hx.size() * 4));
    CHECK(hipMalloc(&w1, hw.size() * 4));
    CHECK(hipMalloc(&w2, hw.size() * 4));
    CHECK(hipMalloc(&dw1, hw.size() * 4));
    CHECK(hipMalloc(&dw2, hw.size() * 4));
    CHECK(hipMalloc(&ro, F * sizeof(int64_t)));
    std::vector<int64_t> hro(F);
    for (int f = 0; f < F; ++f) hro[f] = static_cast<int64_t>(f) * blk;  // (complex elements)
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(gout, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w1, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w2, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ro, hro.data(), F * sizeof(int64_t), hipMemcpyHostToDevice));
    CHECK(hipMemset(dw1, 0, hw.size() * 4));
    CHECK(hipMemset(dw2, 0, hw.size() * 4));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto fwd = [&] { return ck_tensordot2_lse_fwd(x, ro, 1, w1, mid, w2, out, F, 1, K, K, K, K, 1, s); };
    auto bwd = [&] { return ck_tensordot2_lse_bwd(x, gx, ro, 1, w1, mid, gmid, w2, out, gout, dw1, dw2, F, 1, K, K, K, K, 1, s); };
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = 0; i < 5; ++i) {
        if ((pass == 0 ? fwd() : bwd()) != 0) {
          printf("launch failed: %s\n", ck_last_error());
          return 1;
        }
      }
      CHECK(hipStreamSynchronize(s));
      CHECK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; ++i) pass == 0 ? fwd() : bwd();
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("F = %3d  %s  %.2f us per launch", F, pass == 0 ? "fwd" : "bwd", 1e3f * ms / reps);
#ifdef CK_TD_STAMPS
      long long* st;
      CHECK(hipMalloc(&st, 32 * sizeof(long long)));
      CHECK(hipMemset(st, 0, 32 * sizeof(long long)));
      int zero = 0;
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_td_stamps), &st, sizeof(st)));
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_td_n), &zero, sizeof(int)));
      pass == 0 ? fwd() : bwd();
      CHECK(hipStreamSynchronize(s));
      long long hs[32];
      int n = 0;
      CHECK(hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost));
      CHECK(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_td_n), sizeof(int)));
      long long* null = nullptr;
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_td_stamps), &null, sizeof(null)));
      printf("   stamps (us after the first):");
      for (int i = 1; i < n && i < 32; ++i) printf(" %.2f", (hs[i] - hs[0]) * 0.01);  // (100 MHz)
      CHECK(hipFree(st));
#endif
      printf("\n");
    }
    {  // checksums after ONE forward + ONE backward on zeroed weight gradients (compare CK_TD_GENERIC=1 against the default)
      CHECK(hipMemset(dw1, 0, hw.size() * 4));
      CHECK(hipMemset(dw2, 0, hw.size() * 4));
      fwd();
      bwd();
      CHECK(hipStreamSynchronize(s));
      std::vector<float> ho(hx.size()), hgx(hx.size()), hd1(hw.size()), hd2(hw.size());
      CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hgx.data(), gx, hgx.size() * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hd1.data(), dw1, hd1.size() * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hd2.data(), dw2, hd2.size() * 4, hipMemcpyDeviceToHost));
      auto sums = [](const std::vector<float>& v) {
        double a = 0., b = 0.;
        for (size_t i = 0; i < v.size(); ++i) {
          a += v[i];
          b += static_cast<double>(v[i]) * ((i % 7) + 1);
        }
        printf(" %.9g %.9g |", a, b);
      };
      printf("         sums out, gx, dw1, dw2:");
      sums(ho);
      sums(hgx);
      sums(hd1);
      sums(hd2);
      printf("\n");
    }
    hipFree(x); hipFree(gx); hipFree(mid); hipFree(gmid); hipFree(out); hipFree(gout); hipFree(w1); hipFree(w2); hipFree(dw1); hipFree(dw2); hipFree(ro);
  }
  return 0;
}
