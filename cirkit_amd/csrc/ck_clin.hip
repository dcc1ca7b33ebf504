// Complex-lse-sum circuits with COMPLEX values (complex Embedding weights and / or complex sum weights, or real parameters
// forced onto this path): CP-T / dense layers of 32 units chained in LINEAR space on (re, im) tile pairs.
//
// The reference evaluates such a layer as  z = sum_h x_h;  m = max Re z;  y = W exp(z - m);  out = log y + m  on complex64
// (ComplexLSESumSemiring.apply_reduce, semiring.py:441-476; ComplexSafeLog utils.py:32-50; TorchCPTLayer optimized.py:171-178):
// per element and layer an exponential, a sine / cosine pair, a logarithm and an arctangent -- 46 vector instructions per MFMA
// in the layer-wise kernel (sum_clse_tile32, profiles/r05_c_cfg5_complex.txt) -- and 8 bytes per unit through memory.
// Here a value is carried between levels as  v = (re + i im) 2^e : a 32 x 32 tile of re, one of im (the MFMA register layout
// of ck_tile.h) and ONE integer exponent per row.  A product of children is a complex multiply and a power-of-two
// renormalisation by the row's largest |re|, |im| (v_frexp_exp / v_ldexp: exact), a sum is two (real weights) or four
// (complex weights) fp32 MFMA chains, the logarithm and the phase are taken ONCE, where a value leaves the chain
// (c_log_shift_tile).  Phases of the reference are sums of atan2 values re-wrapped by every sum layer; here the phase is the
// argument of the final complex number: equal modulo 2 pi.
//
//   clin_table_kernel   Embedding weights (F, 32, C) real or complex -> table rows (F, C + 1, 64) = [re 32 | im 32] normalised
//                       to max(|re|, |im|) in [0.5, 1) + an exponent per row; row C is the sum over c (a negative category;
//                       the reference's Embedding layer has no integral, HipEmbeddingLayer refuses `integrate_vars`)
//   clin_leaf_kernel<D> Embedding -> D CP-T levels in one launch: a wave walks a (root, 32-row tile) unit depth-first, sibling
//                       tiles in registers, leaves gathered from the table; writes the root's tile + exponents
//   clin_layer_kernel   one CP-T / dense layer on tile blocks (children from any earlier block), optionally writing the
//                       reference's (log|v|, arg v) pairs instead (the layer the circuit outputs)
// Blocks are tile-native: (fold, tile) -> 1024 dwords re + 1024 dwords im, dword 256 g + 4 lane + t = register 4 g + t of the
// lane: a wave load is one contiguous KiB.
#include "ck_tile.h"

namespace {

using ck::c32;

constexpr int kTileFloats = 2048;  // re + im of one (fold, 32-row tile)

struct CT {
  float re[16], im[16];
};

__device__ __forceinline__ void load_native(const float* __restrict__ base, int lane, float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t = ck::gload4(base + 256 * g + 4 * lane);
    v[4 * g + 0] = t.x;
    v[4 * g + 1] = t.y;
    v[4 * g + 2] = t.z;
    v[4 * g + 3] = t.w;
  }
}
__device__ __forceinline__ void store_native(float* __restrict__ base, int lane, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) ck::gstore4(base + 256 * g + 4 * lane, make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]));
}

// p <- p * q (complex, element-wise); e_p += e_q; then the row is renormalised by a power of two.
// fp32-input MFMA and the VALU share the SIMD's fp32 lanes: every vector instruction here adds to the
// chains' time, so the products and the rescale are PACKED (v_pk_mul_f32 / v_pk_fma_f32: two elements per instruction).
__device__ __forceinline__ void cmul_renorm(CT& p, const CT& q, int& e, int eq) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x2v ar = {p.re[2 * j], p.re[2 * j + 1]}, ai = {p.im[2 * j], p.im[2 * j + 1]};
    const f32x2v br = {q.re[2 * j], q.re[2 * j + 1]}, bi = {q.im[2 * j], q.im[2 * j + 1]};
    const f32x2v t0 = ai * bi, t1 = ai * br;
    const f32x2v re = __builtin_elementwise_fma(ar, br, -t0);
    const f32x2v im = __builtin_elementwise_fma(ar, bi, t1);
    p.re[2 * j] = re.x;
    p.re[2 * j + 1] = re.y;
    p.im[2 * j] = im.x;
    p.im[2 * j + 1] = im.y;
  }
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(p.re[j]), __builtin_fabsf(p.im[j])));  // v_max3_f32 |a| |b|
  mx = ck::xhalf_max(mx);
  // (mx == 0: the whole row is zero, k = 0; NaN / inf: the values stay as they are and reach the output)
  const int k = (mx > 0.f && mx < 3.0e38f) ? __builtin_amdgcn_frexp_expf(mx) : 0;
  const float sc = __builtin_amdgcn_ldexpf(1.f, -k);
  tile_scale(p.re, sc);
  tile_scale(p.im, sc);
  e += eq + k;
}

// real weights W (32, 32) row-major: y = W p on both parts
struct WReal {
  WRegs w;
};
struct WCplx {
  WRegs re, im;
};
// (Ko < 32 -- a circuit's scalar top layer: the rows beyond Ko are zero, their outputs are never stored)
__device__ __forceinline__ void load_weights(const float* __restrict__ wp, int lane, WReal& w, int Ko = 32) {
  if ((lane & 31) < Ko) {
    load_w<CK_W_ROWMAJOR>(wp, lane, w.w);
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) w.w.q[g] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void load_weights(const float* __restrict__ wp, int lane, WCplx& w, int Ko = 32) {
  // complex64 (Ko, 32) row-major = 64 floats per row, (re, im) interleaved; lane (o, kh) takes inputs 8 g + 4 kh + t
  const float* row = wp + (lane & 31) * 64 + 8 * (lane >> 5);
  const bool live = (lane & 31) < Ko;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = live ? *reinterpret_cast<const float4*>(row + 16 * g) : z;
    const float4 b = live ? *reinterpret_cast<const float4*>(row + 16 * g + 4) : z;
    w.re.q[g] = make_float4(a.x, a.z, b.x, b.z);
    w.im.q[g] = make_float4(a.y, a.w, b.y, b.w);
  }
}
__device__ __forceinline__ void chain(const WRegs& w, const float (&v)[16], f32x16& acc) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].x, v[4 * g + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].y, v[4 * g + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].z, v[4 * g + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.q[g].w, v[4 * g + 3], acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void contract(const WReal& w, CT& p) {
  f32x16 yr, yi;
#pragma unroll
  for (int r = 0; r < 16; ++r) yr[r] = yi[r] = 0.f;
  chain(w.w, p.re, yr);
  chain(w.w, p.im, yi);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    p.re[r] = yr[r];
    p.im[r] = yi[r];
  }
}
__device__ __forceinline__ void contract(const WCplx& w, CT& p) {
  f32x16 yr, yi;
#pragma unroll
  for (int r = 0; r < 16; ++r) yr[r] = yi[r] = 0.f;
  float nim[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) nim[j] = -p.im[j];
  chain(w.re, p.re, yr);  // Re y = Wr pr - Wi pi
  chain(w.im, nim, yr);
  chain(w.re, p.im, yi);  // Im y = Wr pi + Wi pr
  chain(w.im, p.re, yi);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    p.re[r] = yr[r];
    p.im[r] = yi[r];
  }
}

// ---- the Embedding table in linear form --------------------------------------------------------------------------------
// One workgroup (256 threads) per fold.  w: (F, 32, C) real, or complex64 interleaved.  Row c of the table: the 32 units'
// weights of category c, divided by 2^e with e the exponent of the largest |re|, |im| of the row; row C: sum over c.
// (real weights: rows of 32 floats, no imaginary half -- half the bytes the leaf launch gathers)
template <bool WC>
__global__ void __launch_bounds__(256) clin_table_kernel(const float* __restrict__ w, float* __restrict__ table, int32_t* __restrict__ table_e, int C) {
  extern __shared__ float lds[];  // (32, C + 1) re, then the same for im; column C holds the integral
  const int f = blockIdx.x;
  const int stride = C + 1;
  float* sre = lds;
  float* sim = lds + 32 * stride;
  const float* wf = w + static_cast<int64_t>(f) * 32 * C * (WC ? 2 : 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a wave per unit (eight units per wave), lanes along the categories: coalesced reads, the unit's integral by a wave sum
  for (int k = wave; k < 32; k += 4) {
    float ar = 0.f, ai = 0.f;
    for (int c = lane; c < C; c += 64) {
      float vr, vi = 0.f;
      if (WC) {
        const float2 v = *reinterpret_cast<const float2*>(wf + 2 * (static_cast<int64_t>(k) * C + c));
        vr = v.x;
        vi = v.y;
      } else {
        vr = wf[static_cast<int64_t>(k) * C + c];
      }
      sre[k * stride + c] = vr;
      if (WC) sim[k * stride + c] = vi;
      ar += vr;
      ai += vi;
    }
    ar = ck::wave_sum(ar);  // (row C: the sum over the states -- what a negative category selects, as in the other gather tables)
    if (WC) ai = ck::wave_sum(ai);
    if (lane == 0) {
      sre[k * stride + C] = ar;
      if (WC) sim[k * stride + C] = ai;
    }
  }
  __syncthreads();
  // a half-wave per row: 32 lanes = 32 units (stride C + 1 is odd: consecutive units hit consecutive banks)
  const int k = threadIdx.x & 31;
  for (int c = threadIdx.x >> 5; c <= C; c += 8) {
    const float vr = sre[k * stride + c], vi = WC ? sim[k * stride + c] : 0.f;
    float mx = __builtin_fmaxf(__builtin_fabsf(vr), __builtin_fabsf(vi));
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, d, 32));
    const int e = (mx > 0.f && mx < 3.0e38f) ? __builtin_amdgcn_frexp_expf(mx) : 0;
    const float sc = __builtin_amdgcn_ldexpf(1.f, -e);
    float* row = table + (static_cast<int64_t>(f) * stride + c) * (WC ? 64 : 32);
    row[k] = vr * sc;
    if (WC) row[32 + k] = vi * sc;
    if (k == 0) table_e[static_cast<int64_t>(f) * stride + c] = e;
  }
}

// Consecutive workgroup ids go round-robin over the 8 XCDs, each with an L2 of its own: the (tile group, fold) pairs are dealt
// so that ALL tile groups of a fold run on one XCD -- its table rows / weight matrices are fetched into one L2, not eight.
// id -> (fold, tile group); folds beyond F (the grid is padded to a multiple of 8 folds) return false.
__device__ __forceinline__ bool xcd_unit(int id, int tile_groups, int F, int& fold, int& tg) {
  const int x = id & 7, q = id >> 3;
  fold = (q / tile_groups) * 8 + x;
  tg = q % tile_groups;
  return fold < F;
}

// ---- Embedding -> D CP-T levels ----------------------------------------------------------------------------------------
struct LeafArgs {
  const float* table;
  const int32_t* table_e;
  const int32_t* xt;         // (D_vars, B) staged categories (negative: row C), or null: the raw batch below
  const int64_t* x64;        // (B, D_vars) the caller's int64 batch, read -- and validated -- by the launch itself
  int32_t* bad_flag;         // raw batch: raised when a category is >= C (the row's values become NaN); null: no validation
  int n_vars;
  const int32_t* leaf_fold;  // (R, 2^D) Embedding fold of every leaf, walk order
  const int32_t* leaf_var;   // (R, 2^D) its variable
  const float* const* wnode; // (R, 2^D - 1) weight matrices in the order the walk contracts them
  float* out;                // (R, tiles, 2048)
  int32_t* out_e;            // (R, tiles * 32)
  int R, B, tiles, C;
};

template <int D, class W, bool TC, bool XRAW>
__global__ void __launch_bounds__(256) clin_leaf_kernel(const LeafArgs a) {
  constexpr int kLeaves = 1 << D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int root, tg;
  if (!xcd_unit(blockIdx.x, (a.tiles + 3) >> 2, a.R, root, tg)) return;
  const int tile = tg * 4 + wave;
  if (tile >= a.tiles) return;
  const int b_in = lane & 31, kh = lane >> 5;
  const int bl = min(tile * 32 + b_in, a.B - 1);
  const int32_t* lf = a.leaf_fold + static_cast<int64_t>(root) * kLeaves;
  const int32_t* lv = a.leaf_var + static_cast<int64_t>(root) * kLeaves;
  const float* const* wn = a.wnode + static_cast<int64_t>(root) * (kLeaves - 1);
  CT stack[D];
  int estack[D];
  CT cur;
  int e = 0;
  bool row_bad = false;  // (raw batch: a category outside the layer's range -- TorchEmbeddingLayer's indexing raises there)
  static_for<0, kLeaves>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    {  // leaf i: a table row
      int c;
      if constexpr (XRAW) {
        const int64_t xv = a.x64[static_cast<int64_t>(bl) * a.n_vars + lv[i]];
        row_bad |= xv >= a.C;
        c = xv < 0 ? a.C : (xv >= a.C ? a.C - 1 : static_cast<int>(xv));
      } else {
        const int xv = a.xt[static_cast<int64_t>(lv[i]) * a.B + bl];
        c = xv < 0 ? a.C : min(xv, a.C - 1);
      }
      const int64_t r = static_cast<int64_t>(lf[i]) * (a.C + 1) + c;
      const float* row = a.table + r * (TC ? 64 : 32) + 4 * kh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 tr = ck::gload4(row + 8 * g), ti = TC ? ck::gload4(row + 32 + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        cur.re[4 * g + 0] = tr.x; cur.re[4 * g + 1] = tr.y; cur.re[4 * g + 2] = tr.z; cur.re[4 * g + 3] = tr.w;
        cur.im[4 * g + 0] = ti.x; cur.im[4 * g + 1] = ti.y; cur.im[4 * g + 2] = ti.z; cur.im[4 * g + 3] = ti.w;
      }
      e = a.table_e[r];
    }
    static_for<0, steps_after(i)>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      W w;
      load_weights(wn[steps_before(i) + l], lane, w);
      cmul_renorm(cur, stack[l], e, estack[l]);
      contract(w, cur);
    });
    if constexpr (steps_after(i) < D) {
      stack[steps_after(i)] = cur;
      estack[steps_after(i)] = e;
    }
  });
  if constexpr (XRAW) {
    if (a.bad_flag != nullptr) {  // a bad ROW becomes NaN (it travels through every product and contraction above) and raises the flag
      if (row_bad) {
#pragma unroll
        for (int j = 0; j < 16; ++j) cur.re[j] = __builtin_nanf("");
      }
      if (__ballot(row_bad) != 0 && lane == 0) atomicOr(a.bad_flag, 1);
    }
  }
  float* dst = a.out + (static_cast<int64_t>(root) * a.tiles + tile) * kTileFloats;
  store_native(dst, lane, cur.re);
  store_native(dst + 1024, lane, cur.im);
  if (kh == 0) a.out_e[(static_cast<int64_t>(root) * a.tiles + tile) * 32 + b_in] = e;
}

// ---- one layer on tile blocks ------------------------------------------------------------------------------------------
struct LayerArgs {
  const float* lin;            // the arena of tile blocks
  const int32_t* lin_e;        // the arena of exponents
  const int64_t* child_off;    // (F, H) float offset of the child fold's tile 0 in `lin`
  const int64_t* child_eoff;   // (F, H) offset of its exponents in `lin_e`
  const float* const* w;       // (F) weight matrices (32, 32), rows beyond Ko zero
  float* out;                  // (F, tiles, 2048) or null
  int32_t* out_e;
  float* out_log;              // (F, B, Ko) complex64 or null: the reference's (log|v|, arg v)
  int F, H, Ko, B, tiles;
};

// one (fold, tile) unit of a layer: children -> product -> contraction -> tile block and / or (log|v|, arg v) rows
template <class W>
__device__ __forceinline__ void clin_fold(const float* __restrict__ lin, const int32_t* __restrict__ lin_e, const int64_t* __restrict__ co,
                                          const int64_t* __restrict__ ce, int H, const float* __restrict__ wp, float* __restrict__ out,
                                          int32_t* __restrict__ out_e, float* __restrict__ out_log, int Ko, int B, int tile, int lane) {
  const int b_in = lane & 31, kh = lane >> 5;
  W w;
  load_weights(wp, lane, w, Ko);
  CT cur;
  int e;
  {
    const float* src = lin + co[0] + static_cast<int64_t>(tile) * kTileFloats;
    load_native(src, lane, cur.re);
    load_native(src + 1024, lane, cur.im);
    e = lin_e[ce[0] + tile * 32 + b_in];
  }
  for (int h = 1; h < H; ++h) {
    CT sib;
    const float* src = lin + co[h] + static_cast<int64_t>(tile) * kTileFloats;
    load_native(src, lane, sib.re);
    load_native(src + 1024, lane, sib.im);
    const int es = lin_e[ce[h] + tile * 32 + b_in];
    cmul_renorm(cur, sib, e, es);
  }
  contract(w, cur);
  if (out != nullptr) {  // (the fold's block: tile 0 at `out`)
    float* dst = out + static_cast<int64_t>(tile) * kTileFloats;
    store_native(dst, lane, cur.re);
    store_native(dst + 1024, lane, cur.im);
    if (kh == 0) out_e[tile * 32 + b_in] = e;
  }
  if (out_log != nullptr) {  // (the fold's (B, Ko) complex64 rows)
    const int b = tile * 32 + b_in;
    if (b < B) {
      const float m = static_cast<float>(e) * kLN2;
      float* dst = out_log + static_cast<int64_t>(b) * Ko * 2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 8 * (r >> 2) + 4 * kh + (r & 3);  // the output unit register r of lane (b, kh) holds
        if (o < Ko) {
          const c32 z = ck::c_log_shift_tile(c32{cur.re[r], cur.im[r]}, m);
          dst[2 * o] = z.re;
          dst[2 * o + 1] = z.im;
        }
      }
    }
  }
}

template <class W>
__global__ void __launch_bounds__(256) clin_layer_kernel(const LayerArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int f, tg;
  if (!xcd_unit(blockIdx.x, (a.tiles + 3) >> 2, a.F, f, tg)) return;
  const int tile = tg * 4 + wave;
  if (tile >= a.tiles) return;
  clin_fold<W>(a.lin, a.lin_e, a.child_off + static_cast<int64_t>(f) * a.H, a.child_eoff + static_cast<int64_t>(f) * a.H, a.H, a.w[f],
               a.out != nullptr ? a.out + static_cast<int64_t>(f) * a.tiles * kTileFloats : nullptr,
               a.out != nullptr ? a.out_e + static_cast<int64_t>(f) * a.tiles * 32 : nullptr,
               a.out_log != nullptr ? a.out_log + static_cast<int64_t>(f) * a.B * a.Ko * 2 : nullptr, a.Ko, a.B, tile, lane);
}

// ---- the few-fold top of a circuit in ONE launch -----------------------------------------------------------------------
// The last layers of a tree-shaped circuit have a handful of folds each (config 5: 24, 11, 6, 4, 2, 1): as launches they cost
// their latency six times.  Here a workgroup of eight waves owns a 32-row tile and walks the layers in order, a wave per fold,
// the folds' blocks going through memory as between launches (they are read by other waves of the SAME workgroup: a
// workgroup-scope fence + the workgroup barrier between layers).
struct TailFold {
  int64_t co[2], ce[2];  // children: float / int32 offsets of their folds' tile 0 (H <= 2)
  const float* w;        // (Ko, 32) weights
  int64_t out, oute;     // this fold's block (offsets of tile 0), -1: not kept
  float* out_log;        // this fold's (B, Ko) complex64 rows, or null
  int32_t H, Ko;
};
struct TailArgs {
  float* lin;
  int32_t* lin_e;
  const TailFold* folds;
  const int32_t* level_off;  // (n_levels + 1) fold ranges of the layers
  int n_levels, B, tiles;
};
constexpr int kTailWaves = 8;

template <class W>
__global__ void __launch_bounds__(kTailWaves * 64) clin_tail_kernel(const TailArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x;
  for (int lv = 0; lv < a.n_levels; ++lv) {
    const int f0 = a.level_off[lv], f1 = a.level_off[lv + 1];
    for (int i = f0 + wave; i < f1; i += kTailWaves) {
      const TailFold* d = a.folds + i;
      const int64_t out = d->out, oute = d->oute;
      clin_fold<W>(a.lin, a.lin_e, d->co, d->ce, d->H, d->w, out >= 0 ? a.lin + out : nullptr, out >= 0 ? a.lin_e + oute : nullptr,
                   d->out_log, d->Ko, a.B, tile, lane);
    }
    if (lv + 1 < a.n_levels) {
      // The next layer's readers are waves of THIS workgroup, i.e. of this compute unit: workgroup scope.  (A device-scope fence
      // writes back and invalidates the whole L2 of the XCD on gfx950 -- measured: 255 us for this launch instead of ~20.)  The
      // stores have left the CU when vmcnt reaches 0 (the vector cache is write-through); the blocks a layer reads were never
      // read before in this launch, so no stale line of them can sit in the CU's cache.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
}

}  // namespace

extern "C" {

int ck_clin_table(const float* w, int w_is_complex, float* table, int32_t* table_e, int F, int C, void* stream) {
  CK_REQUIRE(w && table && table_e, "ck_clin_table: null pointer");
  CK_REQUIRE(F > 0 && C > 0, "ck_clin_table: non-positive size");
  const size_t lds = static_cast<size_t>(w_is_complex ? 2 : 1) * 32 * (C + 1) * sizeof(float);
  if (lds > 150 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_clin_table: %d categories do not fit in LDS", C);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (w_is_complex)
          hipLaunchKernelGGL(clin_table_kernel<true>, dim3(F), dim3(256), lds, s, w, table, table_e, C);
        else
          hipLaunchKernelGGL(clin_table_kernel<false>, dim3(F), dim3(256), lds, s, w, table, table_e, C);
        return hipGetLastError();
      },
      stream);
}

int ck_clin_leaf_fwd(const float* table, const int32_t* table_e, const int32_t* xt, const int64_t* x_rows, int x_input, int n_vars,
                     int32_t* bad_flag, const int32_t* leaf_fold, const int32_t* leaf_var, const float* const* wnode, int w_is_complex,
                     int table_is_complex, float* out, int32_t* out_e, int R, int depth, int B, int C, void* stream) {
  CK_REQUIRE(table && table_e && leaf_fold && leaf_var && wnode && out && out_e, "ck_clin_leaf_fwd: null pointer");
  CK_REQUIRE(R > 0 && B > 0 && C > 0, "ck_clin_leaf_fwd: non-positive size");
  if (depth < 1 || depth > 4) return ck::fail(CK_ERR_UNSUPPORTED, "ck_clin_leaf_fwd: depth %d (1..4)", depth);
  const bool raw = xt == nullptr;
  const void* const* slot = nullptr;
  if (raw) {
    CK_REQUIRE(x_rows != nullptr || x_input >= 0, "ck_clin_leaf_fwd: neither a staged batch (xt) nor the raw batch (x_rows / x_input)");
    CK_REQUIRE(n_vars > 0, "ck_clin_leaf_fwd: the raw batch needs its row length (n_vars)");
    if (x_input >= 0) {
      slot = ck::program_input_slot(x_input);
      CK_REQUIRE(slot != nullptr, "ck_clin_leaf_fwd: x_input=%d names a program input, but no program is being recorded on this thread "
                                  "(or the index is out of range)", x_input);
    }
  }
  LeafArgs a{table, table_e, xt, x_rows, raw ? bad_flag : nullptr, n_vars, leaf_fold, leaf_var, wnode, out, out_e, R, B, (B + 31) / 32, C};
  const dim3 grid(static_cast<unsigned>(((a.tiles + 3) / 4) * ((R + 7) / 8 * 8))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        LeafArgs b = a;
        if (slot != nullptr) b.x64 = static_cast<const int64_t*>(*slot);  // the batch of THIS replay (ck_program_set_input)
        if (raw && b.x64 == nullptr) return hipErrorInvalidValue;
#define CK_CLIN_LEAF_X(DD, WW, TT)                                                       \
  if (raw)                                                                              \
    hipLaunchKernelGGL((clin_leaf_kernel<DD, WW, TT, true>), grid, block, 0, s, b);     \
  else                                                                                  \
    hipLaunchKernelGGL((clin_leaf_kernel<DD, WW, TT, false>), grid, block, 0, s, b);
#define CK_CLIN_LEAF(DD)                                            \
  case DD:                                                          \
    if (w_is_complex && table_is_complex) {                         \
      CK_CLIN_LEAF_X(DD, WCplx, true)                               \
    } else if (w_is_complex) {                                      \
      CK_CLIN_LEAF_X(DD, WCplx, false)                              \
    } else if (table_is_complex) {                                  \
      CK_CLIN_LEAF_X(DD, WReal, true)                               \
    } else {                                                        \
      CK_CLIN_LEAF_X(DD, WReal, false)                              \
    }                                                               \
    break;
        switch (depth) {
          CK_CLIN_LEAF(1)
          CK_CLIN_LEAF(2)
          CK_CLIN_LEAF(3)
          CK_CLIN_LEAF(4)
        }
#undef CK_CLIN_LEAF
#undef CK_CLIN_LEAF_X
        return hipGetLastError();
      },
      stream);
}

int ck_clin_layer_fwd(const float* lin, const int32_t* lin_e, const int64_t* child_off, const int64_t* child_eoff, const float* const* w,
                      int w_is_complex, float* out, int32_t* out_e, float* out_log, int F, int H, int Ko, int B, void* stream) {
  CK_REQUIRE(lin && lin_e && child_off && child_eoff && w, "ck_clin_layer_fwd: null pointer");
  CK_REQUIRE((out != nullptr && out_e != nullptr) || out_log != nullptr, "ck_clin_layer_fwd: no output");
  CK_REQUIRE(F > 0 && B > 0 && H >= 1 && Ko >= 1 && Ko <= 32, "ck_clin_layer_fwd: bad sizes (H >= 1, 1 <= Ko <= 32)");
  LayerArgs a{lin, lin_e, child_off, child_eoff, w, out, out_e, out_log, F, H, Ko, B, (B + 31) / 32};
  const dim3 grid(static_cast<unsigned>(((a.tiles + 3) / 4) * ((F + 7) / 8 * 8))), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (w_is_complex)
          hipLaunchKernelGGL(clin_layer_kernel<WCplx>, grid, block, 0, s, a);
        else
          hipLaunchKernelGGL(clin_layer_kernel<WReal>, grid, block, 0, s, a);
        return hipGetLastError();
      },
      stream);
}

int ck_clin_tail_fwd(float* lin, int32_t* lin_e, const void* folds, const int32_t* level_off, int n_levels, int w_is_complex, int B,
                     void* stream) {
  CK_REQUIRE(lin && lin_e && folds && level_off, "ck_clin_tail_fwd: null pointer");
  CK_REQUIRE(n_levels > 0 && B > 0, "ck_clin_tail_fwd: non-positive size");
  static_assert(sizeof(TailFold) == 72, "ck_clin_tail_fold of include/cirkit_hip.h");
  TailArgs a{lin, lin_e, static_cast<const TailFold*>(folds), level_off, n_levels, B, (B + 31) / 32};
  const dim3 grid(a.tiles), block(kTailWaves * 64);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (w_is_complex)
          hipLaunchKernelGGL(clin_tail_kernel<WCplx>, grid, block, 0, s, a);
        else
          hipLaunchKernelGGL(clin_tail_kernel<WReal>, grid, block, 0, s, a);
        return hipGetLastError();
      },
      stream);
}

}  // extern "C"
