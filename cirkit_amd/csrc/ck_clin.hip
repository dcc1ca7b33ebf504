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
//                       to max(|re|, |im|) in [0.5, 1) + an exponent per row; row C is the layer's integral (sum over c)
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

// p <- p * q (complex, element-wise); e_p += e_q; then the row is renormalised by a power of two
__device__ __forceinline__ void cmul_renorm(CT& p, const CT& q, int& e, int eq) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float ar = p.re[j], ai = p.im[j];
    p.re[j] = fmaf(ar, q.re[j], -(ai * q.im[j]));
    p.im[j] = fmaf(ar, q.im[j], ai * q.re[j]);
  }
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(p.re[j]), __builtin_fabsf(p.im[j])));
  mx = ck::xhalf_max(mx);
  // (mx == 0: the whole row is zero, k = 0; NaN / inf: frexp_exp returns 0, the values stay as they are and reach the output)
  const int k = (mx > 0.f && mx < 3.0e38f) ? __builtin_amdgcn_frexp_expf(mx) : 0;
  const float sc = __builtin_amdgcn_ldexpf(1.f, -k);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    p.re[j] *= sc;
    p.im[j] *= sc;
  }
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
  extern __shared__ float lds[];  // (32, C + 1) re, then the same for im; column C accumulates the integral
  const int f = blockIdx.x;
  const int stride = C + 1;
  float* sre = lds;
  float* sim = lds + 32 * stride;
  const float* wf = w + static_cast<int64_t>(f) * 32 * C * (WC ? 2 : 1);
  for (int i = threadIdx.x; i < 32 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    sre[k * stride + c] = WC ? wf[2 * i] : wf[i];
    sim[k * stride + c] = WC ? wf[2 * i + 1] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 32) {  // the integral of unit k (TorchEmbeddingLayer.integrate, input.py:280-282: the sum over the states)
    float ar = 0.f, ai = 0.f;
    for (int c = 0; c < C; ++c) {
      ar += sre[threadIdx.x * stride + c];
      ai += sim[threadIdx.x * stride + c];
    }
    sre[threadIdx.x * stride + C] = ar;
    sim[threadIdx.x * stride + C] = ai;
  }
  __syncthreads();
  // a half-wave per row: 32 lanes = 32 units
  const int k = threadIdx.x & 31;
  for (int c = threadIdx.x >> 5; c <= C; c += 8) {
    const float vr = sre[k * stride + c], vi = sim[k * stride + c];
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

// ---- Embedding -> D CP-T levels ----------------------------------------------------------------------------------------
struct LeafArgs {
  const float* table;
  const int32_t* table_e;
  const int32_t* xt;         // (D_vars, B) staged categories (negative: integrate the variable)
  const int32_t* leaf_fold;  // (R, 2^D) Embedding fold of every leaf, walk order
  const int32_t* leaf_var;   // (R, 2^D) its variable
  const float* const* wnode; // (R, 2^D - 1) weight matrices in the order the walk contracts them
  float* out;                // (R, tiles, 2048)
  int32_t* out_e;            // (R, tiles * 32)
  int R, B, tiles, C;
};

template <int D, class W, bool TC>
__global__ void __launch_bounds__(256) clin_leaf_kernel(const LeafArgs a) {
  constexpr int kLeaves = 1 << D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int root = blockIdx.y;
  const int tile = blockIdx.x * 4 + wave;
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
  static_for<0, kLeaves>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    {  // leaf i: a table row
      const int xv = a.xt[static_cast<int64_t>(lv[i]) * a.B + bl];
      const int c = xv < 0 ? a.C : min(xv, a.C - 1);
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
  int H, Ko, B, tiles;
};

template <class W>
__global__ void __launch_bounds__(256) clin_layer_kernel(const LayerArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.y;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= a.tiles) return;
  const int b_in = lane & 31, kh = lane >> 5;
  W w;
  load_weights(a.w[f], lane, w, a.Ko);
  CT cur;
  int e;
  {
    const float* src = a.lin + a.child_off[static_cast<int64_t>(f) * a.H] + static_cast<int64_t>(tile) * kTileFloats;
    load_native(src, lane, cur.re);
    load_native(src + 1024, lane, cur.im);
    e = a.lin_e[a.child_eoff[static_cast<int64_t>(f) * a.H] + tile * 32 + b_in];
  }
  for (int h = 1; h < a.H; ++h) {
    CT sib;
    const float* src = a.lin + a.child_off[static_cast<int64_t>(f) * a.H + h] + static_cast<int64_t>(tile) * kTileFloats;
    load_native(src, lane, sib.re);
    load_native(src + 1024, lane, sib.im);
    const int es = a.lin_e[a.child_eoff[static_cast<int64_t>(f) * a.H + h] + tile * 32 + b_in];
    cmul_renorm(cur, sib, e, es);
  }
  contract(w, cur);
  if (a.out != nullptr) {
    float* dst = a.out + (static_cast<int64_t>(f) * a.tiles + tile) * kTileFloats;
    store_native(dst, lane, cur.re);
    store_native(dst + 1024, lane, cur.im);
    if (kh == 0) a.out_e[(static_cast<int64_t>(f) * a.tiles + tile) * 32 + b_in] = e;
  }
  if (a.out_log != nullptr) {
    const int b = tile * 32 + b_in;
    if (b < a.B) {
      const float m = static_cast<float>(e) * kLN2;
      float* dst = a.out_log + (static_cast<int64_t>(f) * a.B + b) * a.Ko * 2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 8 * (r >> 2) + 4 * kh + (r & 3);  // the output unit register r of lane (b, kh) holds
        if (o < a.Ko) {
          const c32 z = ck::c_log_shift_tile(c32{cur.re[r], cur.im[r]}, m);
          dst[2 * o] = z.re;
          dst[2 * o + 1] = z.im;
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int ck_clin_table(const float* w, int w_is_complex, float* table, int32_t* table_e, int F, int C, void* stream) {
  CK_REQUIRE(w && table && table_e, "ck_clin_table: null pointer");
  CK_REQUIRE(F > 0 && C > 0, "ck_clin_table: non-positive size");
  const size_t lds = static_cast<size_t>(2) * 32 * (C + 1) * sizeof(float);
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

int ck_clin_leaf_fwd(const float* table, const int32_t* table_e, const int32_t* xt, const int32_t* leaf_fold, const int32_t* leaf_var,
                     const float* const* wnode, int w_is_complex, int table_is_complex, float* out, int32_t* out_e, int R, int depth,
                     int B, int C, void* stream) {
  CK_REQUIRE(table && table_e && xt && leaf_fold && leaf_var && wnode && out && out_e, "ck_clin_leaf_fwd: null pointer");
  CK_REQUIRE(R > 0 && B > 0 && C > 0, "ck_clin_leaf_fwd: non-positive size");
  CK_REQUIRE(R <= 65535, "ck_clin_leaf_fwd: %d roots exceed grid.y", R);
  if (depth < 1 || depth > 4) return ck::fail(CK_ERR_UNSUPPORTED, "ck_clin_leaf_fwd: depth %d (1..4)", depth);
  LeafArgs a{table, table_e, xt, leaf_fold, leaf_var, wnode, out, out_e, R, B, (B + 31) / 32, C};
  const dim3 grid((a.tiles + 3) / 4, R), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
#define CK_CLIN_LEAF(DD)                                                                      \
  case DD:                                                                                    \
    if (w_is_complex && table_is_complex)                                                     \
      hipLaunchKernelGGL((clin_leaf_kernel<DD, WCplx, true>), grid, block, 0, s, a);          \
    else if (w_is_complex)                                                                    \
      hipLaunchKernelGGL((clin_leaf_kernel<DD, WCplx, false>), grid, block, 0, s, a);         \
    else if (table_is_complex)                                                                \
      hipLaunchKernelGGL((clin_leaf_kernel<DD, WReal, true>), grid, block, 0, s, a);          \
    else                                                                                      \
      hipLaunchKernelGGL((clin_leaf_kernel<DD, WReal, false>), grid, block, 0, s, a);         \
    break;
        switch (depth) {
          CK_CLIN_LEAF(1)
          CK_CLIN_LEAF(2)
          CK_CLIN_LEAF(3)
          CK_CLIN_LEAF(4)
        }
#undef CK_CLIN_LEAF
        return hipGetLastError();
      },
      stream);
}

int ck_clin_layer_fwd(const float* lin, const int32_t* lin_e, const int64_t* child_off, const int64_t* child_eoff, const float* const* w,
                      int w_is_complex, float* out, int32_t* out_e, float* out_log, int F, int H, int Ko, int B, void* stream) {
  CK_REQUIRE(lin && lin_e && child_off && child_eoff && w, "ck_clin_layer_fwd: null pointer");
  CK_REQUIRE((out != nullptr && out_e != nullptr) || out_log != nullptr, "ck_clin_layer_fwd: no output");
  CK_REQUIRE(F > 0 && B > 0 && H >= 1 && Ko >= 1 && Ko <= 32, "ck_clin_layer_fwd: bad sizes (H >= 1, 1 <= Ko <= 32)");
  CK_REQUIRE(F <= 65535, "ck_clin_layer_fwd: %d folds exceed grid.y", F);
  LayerArgs a{lin, lin_e, child_off, child_eoff, w, out, out_e, out_log, H, Ko, B, (B + 31) / 32};
  const dim3 grid((a.tiles + 3) / 4, F), block(256);
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

}  // extern "C"
