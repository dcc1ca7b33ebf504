// The walk of a circuit's trailing few-fold levels on 16-row tiles (ck_tail16.hip's arithmetic) as a device function, for
// the launches that do more than that walk: the persistent leaf launch (ck_leaf.hip, leaf_tail_phase) and the launch that
// evaluates the next forward's parameters beside it (ck_tailp.hip).  Reference per fold: TorchCPTLayer optimized.py:171-178
// / dense TorchSumLayer inner.py:266-273 + LSESumSemiring.apply_reduce semiring.py:383-408.
#pragma once

#include "ck_internal.h"
#include "ck_tile.h"
#include "ck_tile16.h"

namespace {

// one fold of the tail, as the host lays it out (ck_tail16_fold in cirkit_hip.h; ck_tail16.hip's FoldDesc)
struct TailFold {
  const float* w;
  float* out;
  const float* child[4];
  int32_t child_src[4];
  int32_t H, Ko;
  int32_t skip_store;
  int32_t slot;  // LDS slot of this fold's tile (tail_walk); the stand-alone 16-row kernel uses the fold index itself
};
static_assert(sizeof(TailFold) == sizeof(ck_tail16_fold), "TailFold mirrors ck_tail16_fold");

// 16-byte accesses that other CUs (other XCDs) must see / that must see other CUs' stores: write-through stores and loads
// past the non-coherent caches, sc0 sc1 on both sides (MI355X_MICROARCH.md, inter-workgroup visibility) -- no cache-wide
// release / acquire.  Buffer instructions through the compiler's intrinsics (cache-policy bits in `aux`), NOT inline asm:
// an asm load's destination registers are fair game for the register allocator before the data has arrived, and the
// hazard recogniser does not see an asm store's data registers (both observed: memory faults, not wrong digits).
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
constexpr int kAuxWriteThrough = 17;  // gfx940+: bit 0 = sc0, bit 4 = sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_buffer(const void* uniform_base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_base), 0, 0x7fffffff, 0x27000);  // raw dwords, byte offsets
}
__device__ __forceinline__ void store4_wt(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, float x, float y, float z, float w) {
  const u32x4v v = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, kAuxWriteThrough);
}
__device__ __forceinline__ f32x4v load4_wt(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, kAuxWriteThrough);
  return f32x4v{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
// a register tile into the (rows, 32) block at `uniform_block` (32-bit byte offsets inside the block: checked on the host)
__device__ __forceinline__ void tile_store_wt(const float* uniform_block, uint32_t row_byte_off, const float (&v)[16]) {
  const __amdgpu_buffer_rsrc_t r = wt_buffer(uniform_block);
#pragma unroll
  for (int g = 0; g < 4; ++g) store4_wt(r, row_byte_off + 32 * g, v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}


// LDS of the tail phase (the leaf walk's two arrays, free by then): the [beta][lane] float4 tiles of the tail's folds, 2 KB
// each, fill the gather slots (`n_main` folds) and go on behind the first kTailCtlFloats floats of the weight array, which
// hold the fold descriptors, the level table and four control words.
constexpr int kTailCtlFloats = 2048;
struct TailTiles {
  float* main;
  float* more;
  int n_main;
  __device__ __forceinline__ float* of(int t) const { return t < n_main ? main + t * 512 : more + (t - n_main) * 512; }
};

struct TailWalkArgs {
  int B, n_levels, n_folds;
  int w_rowmajor;  // layout of the 32-output weights (else CK_W_TILED_F32)
  int write;       // store the 32-unit fold outputs (unless a fold says skip_store)
  double* ll;      // nullptr, or [sum_b log p, B]
  double* ll_partial;
  unsigned int* ll_ticket;
};

// One 16-row tile of the batch through the levels of the tail, by the WAVES waves of a workgroup (all of them call this
// together: a workgroup barrier per level).  Fold t's tile lives in LDS slot s_fold[t].slot (`tiles.of`); child_src[h] >= 0
// names the SLOT of a child that is a fold of the tail.  WT: children in memory were written by other workgroups of the
// SAME launch (write-through): read them past the caches; otherwise they come from earlier launches (plain loads).
// the barrier of a whole workgroup whose LDS data is to be published (not __syncthreads(): no wait for memory operations)
struct WorkgroupBarrier {
  __device__ __forceinline__ void operator()() const { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
};
// A barrier among `n` of a workgroup's waves only -- the hardware barrier takes all of them -- through a counter in LDS:
// a wave's LDS writes have completed (lgkmcnt) before it arrives; lane 0 arrives and polls, the wave follows it.  ~0.1 us.
struct WaveGroupBarrier {
  unsigned int* ctr;  // in LDS, zero at the start; one per group
  unsigned int n;
  unsigned int round;
  __device__ __forceinline__ void operator()() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ++round;
    if ((threadIdx.x & 63) == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < round * n) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  }
};

#ifdef CK_TAILP_STAMPS
__shared__ long long* g_tail_level_stamps;  // (ck_tailp.hip, CK_TAILP_STAMPS: where thread 0 notes the end of each level)
#endif
// `wave`: this wave's number among the WAVES that walk the tile; `barrier`: their barrier.
template <int WAVES, bool WT, class Barrier>
__device__ __forceinline__ void tail_walk(const TailWalkArgs& a, int tile, const TailTiles& tiles, const TailFold* s_fold,
                                          const int32_t* s_level, bool poison, int wave, Barrier&& barrier) {
  const int lane = threadIdx.x & 63;
  const int b_in = lane & 15, kq = lane >> 4;
  const int b = tile * 16 + b_in;
  const bool live = b < a.B;
  const int bl = live ? b : a.B - 1;
  const uint32_t row_off = static_cast<uint32_t>(bl * kK + 4 * kq) * 4u;  // byte offset of the lane's 16 bytes in a (B, 32) block
  struct Children {
    f32x4v m[4][2];
  };
  // the children a fold reads from memory (roots of the leaf walk, written by any workgroup of this launch: past the caches)
  auto fetch = [&](int t, Children& c) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h < s_fold[t].H && s_fold[t].child_src[h] < 0) {  // (uniform: t is)
        const uint64_t p = reinterpret_cast<uint64_t>(s_fold[t].child[h]);
        // (readfirstlane returns a signed int: through uint32_t, or a set bit 31 of the low half smears into the high half)
        const uint32_t p_lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(p))));
        const uint32_t p_hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(p >> 32))));
        const uint64_t pu = (static_cast<uint64_t>(p_hi) << 32) | p_lo;
        const __amdgpu_buffer_rsrc_t r = wt_buffer(reinterpret_cast<const void*>(pu));
        if constexpr (WT) {
          c.m[h][0] = load4_wt(r, row_off);
          c.m[h][1] = load4_wt(r, row_off + 64);
        } else {
          const float* row = reinterpret_cast<const float*>(pu) + (row_off >> 2);
          const float4 x0 = ck::gload4(row), x1 = ck::gload4(row + 16);
          c.m[h][0] = f32x4v{x0.x, x0.y, x0.z, x0.w};
          c.m[h][1] = f32x4v{x1.x, x1.y, x1.z, x1.w};
        }
      }
    }
  };
  // the weights of a wave's NEXT fold are requested while the current one computes -- across the level barrier too: they do
  // not depend on the level below (a weight fetch from L2 per level, exposed, was a third of this walk's time)
  auto load_w = [&](int t, WRegs16& w) {
    if (s_fold[t].Ko == kK) {  // (parameters: written by an earlier launch, plain loads)
      if (a.w_rowmajor) load_w16<CK_W_ROWMAJOR>(s_fold[t].w, lane, w);
      else load_w16<CK_W_TILED_F32>(s_fold[t].w, lane, w);
    }
  };
  auto first_fold_from = [&](int li) {  // this wave's first fold at or behind level li, or -1
    for (; li < a.n_levels; ++li)
      if (s_level[li] + wave < s_level[li + 1]) return s_level[li] + wave;
    return -1;
  };
  WRegs16 w, wn;
  {
    const int t0 = first_fold_from(0);
    if (t0 >= 0) load_w(t0, w);
  }
  for (int li = 0; li < a.n_levels; ++li) {
    const int t1 = s_level[li + 1];
    int t = s_level[li] + wave;
    Children cur, nxt;
    if (t < t1) fetch(t, cur);
    for (; t < t1; t += WAVES) {
      if (t + WAVES < t1) fetch(t + WAVES, nxt);  // the next fold's children are on their way while this one computes
      const int t_next = t + WAVES < t1 ? t + WAVES : first_fold_from(li + 1);
      if (t_next >= 0) load_w(t_next, wn);
      const int H = s_fold[t].H, Ko = s_fold[t].Ko;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (h >= H) break;
        const int src = s_fold[t].child_src[h];
        if (src >= 0) {
          const float* tl = tiles.of(src) + lane * 4;
#pragma unroll
          for (int beta = 0; beta < 2; ++beta) {
            const float4 t4 = *reinterpret_cast<const float4*>(tl + beta * 256);
            v[4 * beta + 0] += t4.x;
            v[4 * beta + 1] += t4.y;
            v[4 * beta + 2] += t4.z;
            v[4 * beta + 3] += t4.w;
          }
        } else {
#pragma unroll
          for (int beta = 0; beta < 2; ++beta)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * beta + r] += cur.m[h][beta][r];
        }
      }
      float* out = s_fold[t].out;
      if (Ko == kK) {
        sum_step16(w, v);
        if (a.write && live && s_fold[t].skip_store == 0) tile16_store(out + static_cast<int64_t>(b) * kK + 4 * kq, v);
        float* tl = tiles.of(s_fold[t].slot) + lane * 4;
#pragma unroll
        for (int beta = 0; beta < 2; ++beta)
          *reinterpret_cast<float4*>(tl + beta * 256) = make_float4(v[4 * beta + 0], v[4 * beta + 1], v[4 * beta + 2], v[4 * beta + 3]);
      } else {
        // Ko < 32 (the root: Ko = 1): plain dot products, row-major fp32 weights (as tail16_kernel)
        const float* wf = s_fold[t].w;
        const float m = ck::clamp_finite(row_max8(v));
        const float nml = exp_offset(m, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], kL2E, nml));
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
          float y = fmaf(__builtin_amdgcn_logf(acc), kLN2, m);
          if (poison) y = __builtin_nanf("");
          if (live && kq == 0) ck::as_global(out)[static_cast<int64_t>(b) * Ko + o] = y;
          if (a.ll != nullptr && t == a.n_folds - 1) {
            // this tile's (up to) 16 root values, rows in order, in double precision; the last tile of the launch to get
            // here adds up the per-tile sums in index order (deterministic; ck_tail16.hip)
            double sacc = 0.0;
            for (int r = 0; r < 16; ++r) {
              const float yr = __shfl(y, r, 64);
              if (tile * 16 + r < a.B) sacc += static_cast<double>(yr);
            }
            const unsigned int n_tiles = static_cast<unsigned int>((a.B + 15) >> 4);
            unsigned int ticket = 0;
            if (lane == 0) {
              __hip_atomic_store(a.ll_partial + tile, sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              ticket = __hip_atomic_fetch_add(a.ll_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ticket = __shfl(ticket, 0, 64);
            if (ticket == n_tiles - 1) {
              double tot = 0.0;  // (four past-the-cache loads in flight per lane, as ck_tail16.hip)
              const __amdgpu_buffer_rsrc_t rp = wt_buffer(a.ll_partial);
              for (unsigned int g0 = 0; g0 < n_tiles; g0 += 256) {
                typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
                u32x2v pv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const unsigned int g = g0 + lane + 64 * k;
                  pv[k] = g < n_tiles ? __builtin_amdgcn_raw_buffer_load_b64(rp, g * 8u, 0, kAuxWriteThrough) : u32x2v{0u, 0u};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  if (g0 + lane + 64 * k < n_tiles) tot += __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(pv[k].y) << 32) | pv[k].x));
              }
#pragma unroll
              for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
              if (lane == 0) {
                a.ll[0] = tot;
                a.ll[1] = static_cast<double>(a.B);
                __hip_atomic_store(a.ll_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          }
        }
      }
      cur = nxt;
      w = wn;
    }
    barrier();  // the level's tiles are in LDS (no wait for the weights just requested for the next level)
#ifdef CK_TAILP_STAMPS
    if (threadIdx.x == 0 && g_tail_level_stamps != nullptr && li < 12) g_tail_level_stamps[li] = static_cast<long long>(wall_clock64());
#endif
  }
}


}  // namespace
