// Micro-benchmark: what one linear-domain CP-T step of the fused leaf kernel (ck_leaf.hip / ck_fused.hip) costs a SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Icirkit_amd/csrc -Iinclude scripts/ubench/leaf_step.hip -o scripts/ubench/leaf_step.bin
// A wave repeats `iters` times a "tile" of 15 steps on register data (no gathers, no stores):
//   MODE 0: the 16-MFMA contraction only, weights read from LDS once
//   MODE 1: + the step's 4 x ds_read_b128 of the weights
//   MODE 2: + the VALU block of the step (products, row maximum, division, scaling, log)      = the kernel's step
//   MODE 3: MODE 2 with v_rcp instead of the IEEE division and a raw v_log (no denormal handling)
// Printed: shader cycles per step and SIMD for 1, 2 and 3 waves per SIMD (workgroups of 4 x waves wavefronts, one per CU).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "ck_internal.h"
#include "ck_tile.h"

template <int MODE, int WPS, int PRIO = 0>
__global__ void __launch_bounds__(256 * WPS) step_loop(float* out, long long* t, int iters, const float* wsrc) {
  // PRIO 1: the second wave of every SIMD (waves 4..7 of the workgroup) runs at a higher static priority;
  // PRIO 2: every wave raises its priority for the duration of its MFMA chain; PRIO 3: both
  if ((PRIO & 1) && (threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(2);
  __shared__ __attribute__((aligned(16))) float w_lds[15 * 1024];
  for (int i = threadIdx.x; i < 15 * 1024; i += blockDim.x) w_lds[i] = wsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float cur[16], stack[4][16], cs = 0.f;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cur[j] = 0.5f + 0.01f * ((lane * 7 + j * 3) % 31);
#pragma unroll
    for (int l = 0; l < 4; ++l) stack[l][j] = 0.3f + 0.02f * ((lane * 5 + j + l) % 29);
  }
  WRegs w0;
#pragma unroll
  for (int q = 0; q < 4; ++q) w0.q[q] = *reinterpret_cast<const float4*>(w_lds + q * 256 + lane * 4);
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    int woff = 0;
    asm volatile("" : "+v"(woff));  // the weights are re-read every iteration (as for every tile in the kernel)
    const float* wl = w_lds + woff;
#pragma unroll
    for (int step = 0; step < 15; ++step) {
      const int l = step & 3;
      WRegs wcur = w0;
      if (MODE >= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wcur.q[q] = *reinterpret_cast<const float4*>(wl + step * 1024 + q * 256 + lane * 4);
      }
      if (MODE == 2 || MODE == 3) {
        float p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) p[j] = cur[j] * stack[l][j];
        float mx = p[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, p[j]);
        mx = ck::xhalf_max(mx);
        if (__builtin_expect(__any(!(mx > 1e-30f)), 0)) {
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = __expf(__logf(cur[j]) + __logf(stack[l][j]));
        } else if (MODE == 2) {
          const float inv = 1.f / mx;
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = p[j] * inv;
          cs += __logf(mx);
        } else if (MODE == 3) {
          const float inv = __builtin_amdgcn_rcpf(mx);
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = p[j] * inv;
          cs = fmaf(__builtin_amdgcn_logf(mx), kLN2, cs);
        }
      }
      if (MODE >= 4) {
        // power-of-two scaling: e = p * 2^-k, k = exponent of the row maximum (exact), scale += k ln 2; tree maximum;
        // MODE 5: no branch (the underflow flag is accumulated and would be checked once per tile)
        float p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) p[j] = cur[j] * stack[l][j];
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(p[0], p[1]), p[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(p[3], p[4]), p[5]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(p[6], p[7]), p[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(p[9], p[10]), p[11]);
        const float m4 = __builtin_fmaxf(__builtin_fmaxf(p[12], p[13]), p[14]);
        float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), p[15]));
        mx = ck::xhalf_max(mx);
        const int k = __builtin_amdgcn_frexp_expf(mx);
        const float sc = __builtin_amdgcn_ldexpf(1.f, -k);
        if (MODE == 4) {
          if (__builtin_expect(__any(!(mx > 1e-30f)), 0)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[j] = __expf(__logf(cur[j]) + __logf(stack[l][j]));
          } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[j] = p[j] * sc;
          }
        } else {
          bad |= !(mx > 1e-30f);
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = p[j] * sc;
        }
        cs = fmaf(static_cast<float>(k), kLN2, cs);
      }
      if (PRIO == 4) {  // asymmetric: the first wave of a SIMD wins every arbitration while both are in their chains
        if ((threadIdx.x >> 6) >= 4)
          __builtin_amdgcn_s_setprio(2);
        else
          __builtin_amdgcn_s_setprio(3);
      } else if (PRIO & 2) {
        __builtin_amdgcn_s_setprio(3);
      }
      contract_linear<CK_W_TILED_F32>(wcur, cur);
      if (PRIO == 4) {
        asm volatile("" ::"v"(cur[0]));
        __builtin_amdgcn_s_setprio(0);
      }
      if (PRIO & 2) {
        if ((PRIO & 1) && (threadIdx.x >> 6) >= 4)
          __builtin_amdgcn_s_setprio(2);
        else
          __builtin_amdgcn_s_setprio(0);
      }
      if (MODE >= 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) stack[l][j] = fminf(cur[j], 1.f) + 0.25f;  // keep the values in range (1 VALU per element: the stack push)
      }
    }
  }
  const long long c1 = clock64();
  float s = cs + (bad ? 1.f : 0.f);
#pragma unroll
  for (int j = 0; j < 16; ++j) s += cur[j] + stack[0][j] + stack[1][j] + stack[2][j] + stack[3][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = c1 - c0;
}


// ---- prototype of the software-pipelined stream -----------------------------------------------------------------
// Two independent chains X and Y alternate inside ONE wave: the 16 MFMAs of one chain's contraction are issued with the
// VALU block (products, tree maximum, power-of-two scaling) and the weight reads of the OTHER chain's next step placed
// between them in program order (pinned with sched_barrier), ~3 VALU per MFMA.
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

struct Prep {  // VALU block state of one chain
  float p[16];
  float mx, sc;
  float m[5];
  int k;
};

// piece J of the VALU block: cur (x) stk -> e (in p), scale exponent in k
template <int J>
__device__ __forceinline__ void prep_piece(Prep& P, const float (&cur)[16], const float (&stk)[16], float& cs, bool& bad) {
  if constexpr (J >= 2 && J <= 5) {
#pragma unroll
    for (int j = 4 * (J - 2); j < 4 * (J - 1); ++j) P.p[j] = cur[j] * stk[j];
  } else if constexpr (J == 6) {
    P.m[0] = __builtin_fmaxf(__builtin_fmaxf(P.p[0], P.p[1]), P.p[2]);
    P.m[1] = __builtin_fmaxf(__builtin_fmaxf(P.p[3], P.p[4]), P.p[5]);
    P.m[2] = __builtin_fmaxf(__builtin_fmaxf(P.p[6], P.p[7]), P.p[8]);
    P.m[3] = __builtin_fmaxf(__builtin_fmaxf(P.p[9], P.p[10]), P.p[11]);
  } else if constexpr (J == 7) {
    P.m[4] = __builtin_fmaxf(__builtin_fmaxf(P.p[12], P.p[13]), P.p[14]);
    P.mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(P.m[0], P.m[1]), P.m[2]), __builtin_fmaxf(__builtin_fmaxf(P.m[3], P.m[4]), P.p[15]));
  } else if constexpr (J == 8) {
    P.mx = ck::xhalf_max(P.mx);
  } else if constexpr (J == 9) {
    P.k = __builtin_amdgcn_frexp_expf(P.mx);
    P.sc = __builtin_amdgcn_ldexpf(1.f, -P.k);
    bad |= !(P.mx > 1e-30f);
    cs = fmaf(static_cast<float>(P.k), kLN2, cs);
  } else if constexpr (J >= 10 && J <= 13) {
#pragma unroll
    for (int j = 4 * (J - 10); j < 4 * (J - 9); ++j) P.p[j] *= P.sc;
  }
}

template <int WPS>
__global__ void __launch_bounds__(256 * WPS) pipe_loop(float* out, long long* t, int iters, const float* wsrc) {
  __shared__ __attribute__((aligned(16))) float w_lds[15 * 1024];
  for (int i = threadIdx.x; i < 15 * 1024; i += blockDim.x) w_lds[i] = wsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float cx[16], cy[16], stack[4][16], csx = 0.f, csy = 0.f;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cx[j] = 0.5f + 0.01f * ((lane * 7 + j * 3) % 31);
    cy[j] = 0.4f + 0.01f * ((lane * 3 + j * 5) % 29);
#pragma unroll
    for (int l = 0; l < 4; ++l) stack[l][j] = 0.3f + 0.02f * ((lane * 5 + j + l) % 29);
  }
  Prep PX, PY;
  WRegs wx, wy;
  f32x16 ax, ay;
  const long long c0 = clock64();
  // prologue: e of chain X for its first step
  sfor<2, 14>([&](auto jc) { prep_piece<decltype(jc)::value>(PX, cx, stack[0], csx, bad); });
#pragma unroll
  for (int q = 0; q < 4; ++q) wx.q[q] = *reinterpret_cast<const float4*>(w_lds + q * 256 + lane * 4);
  for (int it = 0; it < iters; ++it) {
    int woff = 0;
    asm volatile("" : "+v"(woff));
    const float* wl = w_lds + woff;
#pragma unroll
    for (int step = 0; step < 15; ++step) {
      const int l = step & 3;
      // contraction of X (e in PX.p, weights wx) with the block of Y (cy (x) stack[l] -> PY.p, weights -> wy) in between
      sfor<0, 16>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const float a = J % 4 == 0 ? wx.q[J / 4].x : J % 4 == 1 ? wx.q[J / 4].y : J % 4 == 2 ? wx.q[J / 4].z : wx.q[J / 4].w;
        if constexpr (J == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) ax[r] = 0.f;
        }
        ax = __builtin_amdgcn_mfma_f32_32x32x2f32(a, PX.p[J], ax, 0, 0, 0);
        if constexpr (J == 0) {
          wy.q[0] = *reinterpret_cast<const float4*>(wl + ((step + 7) % 15) * 1024 + 0 * 256 + lane * 4);
          wy.q[1] = *reinterpret_cast<const float4*>(wl + ((step + 7) % 15) * 1024 + 1 * 256 + lane * 4);
        } else if constexpr (J == 1) {
          wy.q[2] = *reinterpret_cast<const float4*>(wl + ((step + 7) % 15) * 1024 + 2 * 256 + lane * 4);
          wy.q[3] = *reinterpret_cast<const float4*>(wl + ((step + 7) % 15) * 1024 + 3 * 256 + lane * 4);
        } else {
          prep_piece<J>(PY, cy, stack[l], csy, bad);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // contraction of Y with the block of X's next step in between (X's result ax is complete two MFMA slots later)
      sfor<0, 16>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const float a = J % 4 == 0 ? wy.q[J / 4].x : J % 4 == 1 ? wy.q[J / 4].y : J % 4 == 2 ? wy.q[J / 4].z : wy.q[J / 4].w;
        if constexpr (J == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) ay[r] = 0.f;
        }
        ay = __builtin_amdgcn_mfma_f32_32x32x2f32(a, PY.p[J], ay, 0, 0, 0);
        if constexpr (J == 0) {
          wx.q[0] = *reinterpret_cast<const float4*>(wl + ((step + 1) % 15) * 1024 + 0 * 256 + lane * 4);
          wx.q[1] = *reinterpret_cast<const float4*>(wl + ((step + 1) % 15) * 1024 + 1 * 256 + lane * 4);
        } else if constexpr (J == 1) {
          wx.q[2] = *reinterpret_cast<const float4*>(wl + ((step + 1) % 15) * 1024 + 2 * 256 + lane * 4);
          wx.q[3] = *reinterpret_cast<const float4*>(wl + ((step + 1) % 15) * 1024 + 3 * 256 + lane * 4);
        } else if constexpr (J == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) cx[r] = ax[r];
          prep_piece<J>(PX, cx, stack[(l + 1) & 3], csx, bad);
        } else {
          prep_piece<J>(PX, cx, stack[(l + 1) & 3], csx, bad);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int r = 0; r < 16; ++r) cy[r] = ay[r];
    }
  }
  const long long c1 = clock64();
  float s = csx + csy + (bad ? 1.f : 0.f);
#pragma unroll
  for (int j = 0; j < 16; ++j) s += cx[j] + cy[j] + PX.p[j] + stack[0][j] + stack[1][j] + stack[2][j] + stack[3][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = c1 - c0;
}

template <int WPS>
void run_pipe(int iters, const float* w) {
  const int wgs = 256, threads = 256 * WPS;
  float* out;
  long long* t;
  hipMalloc(&out, sizeof(float) * wgs * threads);
  hipMalloc(&t, sizeof(long long) * wgs * 4 * WPS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((pipe_loop<WPS>), dim3(wgs), dim3(threads), 0, 0, out, t, iters, w);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // 2 x 15 x iters steps per wave
  printf("pipelined waves/SIMD %d: kernel %.3f ms = %.1f ns per step and SIMD (MFMA floor ~435)\n", WPS, ms, ms * 1e6 / (30.0 * iters * WPS));
  hipFree(out);
  hipFree(t);
}

// MODE 2 with s_memtime stamps around the phases of a step (1 wave per SIMD is the informative case):
// phase 0 = weights ds_read + products + row maximum, 1 = branch + division + scaling + log, 2 = MFMA chain, 3 = stack push
__global__ void __launch_bounds__(768) step_stamped(float* out, long long* t, int iters, const float* wsrc) {
  __shared__ __attribute__((aligned(16))) float w_lds[15 * 1024];
  for (int i = threadIdx.x; i < 15 * 1024; i += blockDim.x) w_lds[i] = wsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float cur[16], stack[4][16], cs = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cur[j] = 0.5f + 0.01f * ((lane * 7 + j * 3) % 31);
#pragma unroll
    for (int l = 0; l < 4; ++l) stack[l][j] = 0.3f + 0.02f * ((lane * 5 + j + l) % 29);
  }
  long long ph[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int step = 0; step < 15; ++step) {
      const int l = step & 3;
      const long long s0 = clock64();
      WRegs wcur;
#pragma unroll
      for (int q = 0; q < 4; ++q) wcur.q[q] = *reinterpret_cast<const float4*>(w_lds + step * 1024 + q * 256 + lane * 4);
      float p[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) p[j] = cur[j] * stack[l][j];
      float mx = p[0];
#pragma unroll
      for (int j = 1; j < 16; ++j) mx = fmaxf(mx, p[j]);
      mx = ck::xhalf_max(mx);
      asm volatile("" ::"v"(mx), "v"(wcur.q[3].w));
      const long long s1 = clock64();
      if (__builtin_expect(__any(!(mx > 1e-30f)), 0)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) cur[j] = __expf(__logf(cur[j]) + __logf(stack[l][j]));
      } else {
        const float inv = 1.f / mx;
#pragma unroll
        for (int j = 0; j < 16; ++j) cur[j] = p[j] * inv;
        cs += __logf(mx);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("" ::"v"(cur[j]));
      asm volatile("" ::"v"(cs));
      const long long s2 = clock64();
      contract_linear<CK_W_TILED_F32>(wcur, cur);
      asm volatile("" ::"v"(cur[0]));
      const long long s3 = clock64();
#pragma unroll
      for (int j = 0; j < 16; ++j) stack[l][j] = fminf(cur[j], 1.f) + 0.25f;
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("" ::"v"(stack[l][j]));
      const long long s4 = clock64();
      ph[0] += s1 - s0;
      ph[1] += s2 - s1;
      ph[2] += s3 - s2;
      ph[3] += s4 - s3;
    }
  }
  float s = cs;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += cur[j] + stack[0][j] + stack[1][j] + stack[2][j] + stack[3][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 64)
    for (int k = 0; k < 4; ++k) t[blockIdx.x * 4 + k] = ph[k];
}

void run_stamped(int wps, int iters, const float* w) {
  const int wgs = 256, threads = 256 * wps;
  float* out;
  long long* t;
  hipMalloc(&out, sizeof(float) * wgs * threads);
  hipMalloc(&t, sizeof(long long) * 64 * 4);
  hipLaunchKernelGGL(step_stamped, dim3(wgs), dim3(threads), 0, 0, out, t, iters, w);
  hipDeviceSynchronize();
  std::vector<long long> h(64 * 4);
  hipMemcpy(h.data(), t, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double ph[4] = {0, 0, 0, 0};
  for (int b = 0; b < 64; ++b)
    for (int k = 0; k < 4; ++k) ph[k] += h[b * 4 + k];
  printf("stamped, waves/SIMD %d: per step (s_memtime ticks, 100 MHz x ?): W read + products + max %.1f | branch + division + scale + log %.1f | MFMA chain %.1f | push %.1f\n",
         wps, ph[0] / (64.0 * 15 * iters), ph[1] / (64.0 * 15 * iters), ph[2] / (64.0 * 15 * iters), ph[3] / (64.0 * 15 * iters));
  hipFree(out);
  hipFree(t);
}

template <int MODE, int WPS, int PRIO = 0>
void run(int iters, const float* w) {
  const int wps = WPS;
  const int wgs = 256, threads = 256 * wps;
  float* out;
  long long* t;
  hipMalloc(&out, sizeof(float) * wgs * threads);
  hipMalloc(&t, sizeof(long long) * wgs * 4 * wps);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((step_loop<MODE, WPS, PRIO>), dim3(wgs), dim3(threads), 0, 0, out, t, iters, w);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(wgs * 4 * wps);
  hipMemcpy(h.data(), t, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double cyc = 0;
  for (auto v : h) cyc += v;
  cyc /= h.size();
  printf("prio %d mode %d waves/SIMD %d: kernel %.3f ms = %.1f ns per step and SIMD (MFMA floor ~435)\n", PRIO, MODE, wps, ms, ms * 1e6 / (15.0 * iters * wps));
  hipFree(out);
  hipFree(t);
}

int main() {
  std::vector<float> hw(15 * 1024);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 1.f / 32.f + 1e-4f * (i % 17);
  float* w;
  hipMalloc(&w, hw.size() * 4);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const int iters = 300;
  run<0, 1>(iters, w); run<1, 1>(iters, w); run<2, 1>(iters, w); run<4, 1>(iters, w); run<5, 1>(iters, w);
  run<0, 2>(iters, w); run<1, 2>(iters, w); run<2, 2>(iters, w); run<4, 2>(iters, w); run<5, 2>(iters, w);
  run<0, 3>(iters, w); run<1, 3>(iters, w); run<2, 3>(iters, w); run<4, 3>(iters, w); run<5, 3>(iters, w);
  run<2, 2, 1>(iters, w); run<2, 2, 2>(iters, w); run<2, 2, 3>(iters, w);
  run<4, 2, 1>(iters, w); run<4, 2, 2>(iters, w); run<4, 2, 3>(iters, w); run<4, 2, 4>(iters, w); run<2, 2, 4>(iters, w);
  run_pipe<1>(iters, w);
  run_pipe<2>(iters, w);
  return 0;
}
