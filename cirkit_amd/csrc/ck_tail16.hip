// Fused tail on 16-row tiles: the last levels of a circuit (24, 11, 6, 4, 2, 1 folds at the north-star config) are a
// chain of tiny dependent steps -- pure latency.  ck_tail.hip walks them with one workgroup per 32-row tile (128
// workgroups at batch 4096: half the chip idle, 24 folds x 1024 MFMA cycles on one CU for the first level, and an L2
// round trip between levels).  Here:
//   * one workgroup of 16 wavefronts per SIXTEEN batch rows (ck_tile16.h): twice the workgroups (every CU busy at
//     batch 4096), half the MFMA cycles per fold step, four waves per SIMD to deal a level's folds to;
//   * every fold output of the tail stays in LDS for the rest of the walk (2 KB per fold; it is ALSO written to the
//     arena, it is a real layer output): a level reads its in-tail children from LDS, not from memory it has just
//     written -- children produced before the tail come from the arena as before;
//   * a wave loads the weights of its next fold before the level barrier (they do not depend on the level below);
//   * optionally the circuit's log-likelihood sum (ck_ll_sum) is folded in: each workgroup adds up its 16 root values
//     in a fixed order, the last workgroup to arrive adds up the per-workgroup partial sums in index order
//     (deterministic, no extra launch).
// Same arithmetic per fold as ck_sum_lse_fwd in CK_SUM_PROD mode (TorchCPTLayer optimized.py:171-178 / dense TorchSumLayer
// inner.py:266-273 + LSESumSemiring.apply_reduce semiring.py:383-408).
#include <algorithm>

#include "ck_internal.h"
#include "ck_tile16.h"

namespace {

constexpr int kTail16Waves = 16;
constexpr int kTail16MaxFolds = 64;   // 2 KB of LDS each
constexpr int kTail16MaxLevels = 15;
constexpr int kTail16MaxArity = 4;

// one fold of the tail, as the host lays it out (ck_tail16_fold in cirkit_hip.h)
struct FoldDesc {
  const float* w;                         // (Ko, 32) linear weights of this fold
  float* out;                             // (B, Ko) output block of this fold
  const float* child[kTail16MaxArity];    // (B, 32) blocks of the children read from memory (when child_src < 0)
  int32_t child_src[kTail16MaxArity];     // index of the child among the tail's folds (its tile is in LDS), or -1
  int32_t H, Ko;
  int32_t skip_store;  // != 0: a 32-unit fold nobody outside the tail reads -- its tile stays in LDS, `out` is not written
  int32_t pad;
};
static_assert(sizeof(FoldDesc) == 80, "FoldDesc layout");
static_assert(sizeof(FoldDesc) == sizeof(ck_tail16_fold), "FoldDesc mirrors ck_tail16_fold");

struct Tail16Args {
  const FoldDesc* folds;     // (n_folds) in level order
  const int32_t* level_begin;  // (n_levels + 1) first fold of each level
  double* ll;          // nullptr, or [sum_b log p, B]: the last fold must then be the scalar root
  double* ll_partial;  // (gridDim.x) partial sums
  unsigned int* ll_ticket;
  const int32_t* bad_input;  // nullptr, or the sticky input-validation flag (ck_stage_categories): nonzero -> NaN outputs
  int n_folds, n_levels, B;
};

// SIGNED: a real-valued circuit under complex-lse-sum (semiring.py:441-476): memory blocks are (B, Ko) complex64 holding
// (log|v|, 0 or pi); inside the walk a value is (log|v|, sign bit) -- one word of sign bits per lane and fold in LDS next
// to the fold's tile.  A product adds the logarithms and xors the signs, a sum step exponentiates with the sign.
template <int LAYOUT, bool SIGNED>
__global__ void __launch_bounds__(kTail16Waves * 64) tail16_kernel(const Tail16Args a) {
  // [tail fold][beta][lane] float4 tiles (2 KB each), (SIGNED: [tail fold][lane] sign words,) then the fold descriptors and
  // the level table
  extern __shared__ __attribute__((aligned(16))) float tiles[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x * 16 + b_in;
  const bool live = b < a.B;
  const int bl = live ? b : a.B - 1;
  // The descriptors of all folds and the level table go to LDS first -- one 16-byte load per thread, all in flight at
  // once: afterwards nothing in the walk waits for an index or a pointer from memory.
  uint32_t* s_sign = reinterpret_cast<uint32_t*>(tiles + a.n_folds * 512);
  FoldDesc* s_fold = reinterpret_cast<FoldDesc*>(tiles + a.n_folds * (SIGNED ? 576 : 512));
  int32_t* s_level = reinterpret_cast<int32_t*>(s_fold + a.n_folds);
  {
    const int n16 = a.n_folds * static_cast<int>(sizeof(FoldDesc) / 16);
    const int4* src = reinterpret_cast<const int4*>(a.folds);
    int4* dst = reinterpret_cast<int4*>(s_fold);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i <= a.n_levels; i += blockDim.x) s_level[i] = a.level_begin[i];
  }
  __syncthreads();
  const bool poison = a.bad_input != nullptr && *a.bad_input != 0;
  WRegs16 w;
  int w_for = -1;  // fold whose 32-output weights are in `w`
  auto prefetch = [&](int t) {
    w_for = -1;
    if (t >= 0 && s_fold[t].Ko == kK) {
      load_w16<LAYOUT>(s_fold[t].w, lane, w);
      w_for = t;
    }
  };
  auto first_fold_of = [&](int li) { return li < a.n_levels && s_level[li] + wave < s_level[li + 1] ? s_level[li] + wave : -1; };
  auto gather = [&](int t, float (&v)[8], uint32_t& sg) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    sg = 0;
    const int H = s_fold[t].H;
    for (int h = 0; h < H; ++h) {
      const int src = s_fold[t].child_src[h];
      if (src >= 0) {
        const float* tl = tiles + src * 512 + lane * 4;
#pragma unroll
        for (int beta = 0; beta < 2; ++beta) {
          const float4 t4 = *reinterpret_cast<const float4*>(tl + beta * 256);
          v[4 * beta + 0] += t4.x;
          v[4 * beta + 1] += t4.y;
          v[4 * beta + 2] += t4.z;
          v[4 * beta + 3] += t4.w;
        }
        if constexpr (SIGNED) sg ^= s_sign[src * 64 + lane];
      } else if constexpr (SIGNED) {
        // (B, 32) complex64: units 16 beta + 4 kq + r of row b are the 8 floats at 2 (32 b + 16 beta + 4 kq)
        const float* row = s_fold[t].child[h] + (static_cast<int64_t>(bl) * kK + 4 * kq) * 2;
#pragma unroll
        for (int beta = 0; beta < 2; ++beta) {
          const float4 c0 = ck::gload4(row + 32 * beta), c1 = ck::gload4(row + 32 * beta + 4);
          v[4 * beta + 0] += c0.x;
          v[4 * beta + 1] += c0.z;
          v[4 * beta + 2] += c1.x;
          v[4 * beta + 3] += c1.z;
          sg ^= ((c0.y > 1.5f ? 1u : 0u) | (c0.w > 1.5f ? 2u : 0u) | (c1.y > 1.5f ? 4u : 0u) | (c1.w > 1.5f ? 8u : 0u)) << (4 * beta);
        }
      } else {
        tile16_load_add(s_fold[t].child[h] + static_cast<int64_t>(bl) * kK + 4 * kq, v);
      }
    }
  };
  prefetch(first_fold_of(0));
  for (int li = 0; li < a.n_levels; ++li) {
    const int t1 = s_level[li + 1];
    // children of the wave's SECOND fold of the level (the widest level has more folds than waves) are requested together
    // with those of the first one
    const int t2 = s_level[li] + wave + kTail16Waves;
    float v2[8];
    uint32_t sg2 = 0;
    if (t2 < t1) gather(t2, v2, sg2);
    for (int t = s_level[li] + wave; t < t1; t += kTail16Waves) {
      float v[8];
      uint32_t sg = 0;
      if (t == t2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v2[j];
        sg = sg2;
      } else {
        gather(t, v, sg);
      }
      const int Ko = s_fold[t].Ko;
      float* out = s_fold[t].out;
      const int t_next = t + kTail16Waves < t1 ? t + kTail16Waves : first_fold_of(li + 1);
      if (Ko == kK) {
        if (w_for != t) load_w16<LAYOUT>(s_fold[t].w, lane, w);
        if constexpr (SIGNED) {
          sum_step16_signed(w, v, sg);
          prefetch(t_next);
          if (live && s_fold[t].skip_store == 0) tile16_store_clog(out + (static_cast<int64_t>(b) * kK + 4 * kq) * 2, v, sg);
          s_sign[t * 64 + lane] = sg;
        } else {
          sum_step16(w, v);
          prefetch(t_next);
          if (live && s_fold[t].skip_store == 0) tile16_store(out + static_cast<int64_t>(b) * kK + 4 * kq, v);
        }
        float* tl = tiles + t * 512 + lane * 4;
#pragma unroll
        for (int beta = 0; beta < 2; ++beta)
          *reinterpret_cast<float4*>(tl + beta * 256) = make_float4(v[4 * beta + 0], v[4 * beta + 1], v[4 * beta + 2], v[4 * beta + 3]);
      } else {
        // Ko < 32 (the root: Ko = 1): plain dot products; the four lanes (b, 0..3) each hold a quarter of the row
        // (these few-output layers always take ROW-MAJOR fp32 weights); their outputs are never children inside the tail
        const float* wf = s_fold[t].w;
        const float m = ck::clamp_finite(row_max8(v));
        const float nml = exp_offset(m, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
          if constexpr (SIGNED) v[j] = (sg >> j) & 1u ? -v[j] : v[j];
        }
        for (int o = 0; o < Ko; ++o) {
          const float* wrow = wf + o * kK + 4 * kq;
          float acc = 0.f;
#pragma unroll
          for (int beta = 0; beta < 2; ++beta) {
            const float4 w4 = ck::gload4(wrow + 16 * beta);
            acc = fmaf(w4.x, v[4 * beta + 0], acc);
            acc = fmaf(w4.y, v[4 * beta + 1], acc);
            acc = fmaf(w4.z, v[4 * beta + 2], acc);
            acc = fmaf(w4.w, v[4 * beta + 3], acc);
          }
          acc = xquad_sum(acc);
          float y = fmaf(__builtin_amdgcn_logf(SIGNED ? __builtin_fabsf(acc) : acc), kLN2, m);
          if (poison) y = __builtin_nanf("");  // an out-of-range category somewhere in the batch (the reference raises)
          if constexpr (SIGNED) {
            if (live && kq == 0)
              *reinterpret_cast<float2*>(out + (static_cast<int64_t>(b) * Ko + o) * 2) = make_float2(y, acc < 0.f ? 3.14159265358979323846f : 0.f);
          } else {
            if (live && kq == 0) ck::as_global(out)[static_cast<int64_t>(b) * Ko + o] = y;
          }
          if (a.ll != nullptr && t == a.n_folds - 1) {
            // sum of this workgroup's (up to) 16 root values, rows in order, in double precision
            double s = 0.0;
            for (int r = 0; r < 16; ++r) {
              const float yr = __shfl(y, r, 64);
              if (blockIdx.x * 16 + r < a.B) s += static_cast<double>(yr);
            }
            // 8-byte agent-scope atomics on both sides (write-through store, completed before the ticket; loads past L1):
            // MI355X_MICROARCH.md, inter-workgroup visibility -- no cache-wide release / acquire needed for one granule
            unsigned int ticket = 0;
            if (lane == 0) {
              __hip_atomic_store(a.ll_partial + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the 8-byte write-through store has completed
              ticket = __hip_atomic_fetch_add(a.ll_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ticket = __shfl(ticket, 0, 64);
            if (ticket == gridDim.x - 1) {  // last workgroup: every partial sum has been published
              // lane l adds partials l, l + 64, ... in order; then a fixed shuffle tree: deterministic.  The loads go past
              // the caches (sc0 sc1, as the atomic loads they replace) but are ordinary buffer loads: four of them in
              // flight per lane instead of one round trip after the other (ordered atomics: ~4 us at 256 workgroups)
              double tot = 0.0;
              const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.ll_partial, 0, 0x7fffffff, 0x27000);
              for (unsigned int g0 = 0; g0 < gridDim.x; g0 += 256) {
                typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
                u32x2v p[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const unsigned int g = g0 + lane + 64 * k;
                  p[k] = g < gridDim.x ? __builtin_amdgcn_raw_buffer_load_b64(rp, g * 8u, 0, 17) : u32x2v{0u, 0u};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  if (g0 + lane + 64 * k < gridDim.x) tot += __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(p[k].y) << 32) | p[k].x));
              }
#pragma unroll
              for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
              if (lane == 0) {
                a.ll[0] = tot;
                a.ll[1] = static_cast<double>(a.B);
                __hip_atomic_store(a.ll_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
              }
            }
          }
        }
        prefetch(t_next);
      }
    }
    if (w_for < 0 && s_level[li] + wave >= t1) prefetch(first_fold_of(li + 1));  // (no fold in this level: get ready for the next one)
    // level boundary: the level's tiles are in LDS.  Not __syncthreads(): that also waits for the level's stores to memory
    // and for the weights requested for the next level (s_waitcnt vmcnt(0)), a memory round trip per level
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

}  // namespace

extern "C" {

int ck_tail16_lse_fwd(const ck_tail16_fold* folds, int n_folds, const int32_t* level_begin, int n_levels, int B, int K,
                      int w_layout, double* ll, double* ll_partial, uint32_t* ll_ticket, const int32_t* bad_input,
                      int signed_values, void* stream) {
  CK_REQUIRE(folds && level_begin, "ck_tail16_lse_fwd: null pointer");
  CK_REQUIRE(n_folds > 0 && n_folds <= kTail16MaxFolds, "ck_tail16_lse_fwd: n_folds=%d outside [1, %d]", n_folds, kTail16MaxFolds);
  CK_REQUIRE(n_levels > 0 && n_levels <= kTail16MaxLevels, "ck_tail16_lse_fwd: n_levels=%d outside [1, %d]", n_levels, kTail16MaxLevels);
  CK_REQUIRE(B > 0, "ck_tail16_lse_fwd: B must be positive");
  if (K != kK) return ck::fail(CK_ERR_UNSUPPORTED, "ck_tail16_lse_fwd: K=%d (only K=32)", K);
  if (w_layout != CK_W_ROWMAJOR && w_layout != CK_W_TILED_F32)
    return ck::fail(CK_ERR_UNSUPPORTED, "ck_tail16_lse_fwd: w_layout %d (row-major or tiled fp32 only)", w_layout);
  CK_REQUIRE(ll == nullptr || (ll_partial != nullptr && ll_ticket != nullptr), "ck_tail16_lse_fwd: ll needs ll_partial and ll_ticket");
  CK_REQUIRE(ck::aligned16(folds), "ck_tail16_lse_fwd: folds not 16-byte aligned");
  CK_REQUIRE(!signed_values || ll == nullptr, "ck_tail16_lse_fwd: no log-likelihood sum of signed (complex) outputs");
  Tail16Args a{};
  a.folds = reinterpret_cast<const FoldDesc*>(folds);
  a.level_begin = level_begin;
  a.n_folds = n_folds;
  a.n_levels = n_levels;
  a.B = B;
  a.ll = ll;
  a.ll_partial = ll_partial;
  a.ll_ticket = ll_ticket;
  a.bad_input = bad_input;
  const size_t lds = static_cast<size_t>(n_folds) * ((signed_values ? 576 : 512) * sizeof(float) + sizeof(FoldDesc)) + (n_levels + 1) * sizeof(int32_t) + 16;
  dim3 grid((B + 15) / 16), block(kTail16Waves * 64);
  return ck::dispatch(
      [=](hipStream_t s) {
        auto go = [&](auto kern) {
          if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
          }
          hipLaunchKernelGGL(kern, grid, block, lds, s, a);
          return hipGetLastError();
        };
        if (signed_values) return w_layout == CK_W_ROWMAJOR ? go(tail16_kernel<CK_W_ROWMAJOR, true>) : go(tail16_kernel<CK_W_TILED_F32, true>);
        return w_layout == CK_W_ROWMAJOR ? go(tail16_kernel<CK_W_ROWMAJOR, false>) : go(tail16_kernel<CK_W_TILED_F32, false>);
      },
      stream);
}

}  // extern "C"
