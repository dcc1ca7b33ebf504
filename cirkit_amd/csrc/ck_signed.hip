// Signed-log sum layers (K = 32): the training path of squared circuits with REAL parameters.
//
// A circuit compiled under complex-lse-sum whose parameters are all real -- Embedding inputs, CP-T / sum layers with real
// weights: the c(x) of a squared circuit, BASELINE config 5 -- is real-valued: the reference carries every value as
// (log|v|, 0 or pi) (ComplexLSESumSemiring, semiring.py:441-476; csafelog, utils.py:32-50; TorchEmbeddingLayer.forward,
// layers/input.py:258-266; TorchCPTLayer.forward, optimized.py:171-178).  Here a value is an fp32 log|v| plus ONE BIT: blocks
// are (rows, 32) fp32 with one sign word per row (bit k set: unit k is negative) -- 4.1 bytes per value instead of 8, one MFMA
// contraction per step instead of two, and the gradient of a loss that reads log|c(x)| (the squared circuit's
// 2 Re c(x) - Re Z) is real.  With x_j = s_j exp(v_j) the product of the children, a_j = x_j exp(-m), y = W a:
//     out_o = log|y_o| + m, sign y_o        t_o = G_o exp(m) / y_o        g v_j = a_j (W^T t)_j        dW = t^T a
// -- the three contractions of the real backward (ck_bwd_tile.h) on signed operands.  Children that are folds of an
// Embedding layer are read from the signed-log form of its weight table (ck_slse_table: (F0, C + 1, 32) log|w| and a sign
// word per row) by the batch values: the layer's output is never stored.  The H children of a product receive the SAME
// gradient: a fold writes it once, into its own (B, 32) block, and its children (ck_slse_bwd's gout_off, ck_embedding_bwd's
// gfold) read it there.
// Layout: a fold's block is TILE-NATIVE -- ceil(B / 32) tiles of 1024 floats, dword (g, lane, t) of a tile = unit 8 g + 4 (lane >> 5) + t
// of row 32 tile + (lane & 31): the MFMA register order, so that a wave instruction loads or stores ONE contiguous KiB (a row-major
// block is 32 lines of 32 bytes per instruction: 3.3 TB/s where these layers stream).  Blocks are Bp = 32 ceil(B / 32) rows long
// (the padding rows hold anything); sign words stay one per row.  Row-major are only what other kernels read: the top layer's
// (B, Ko <= 4) outputs and a gathering layer's gradient block (ck_embedding_bwd scatters rows).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "ck_bwd_tile.h"
#include "ck_internal.h"
#include "ck_tile.h"

namespace {

// the 16 values lane (b, kh) holds of a row: units 8 g + 4 kh + t in register 4 g + t
__device__ __forceinline__ void load16(const float* __restrict__ row, int kh, float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 x = *reinterpret_cast<const float4*>(row + 8 * g + 4 * kh);
    v[4 * g + 0] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
  }
}
// tile `tile` of a tile-native block: register 4 g + t of lane l <-> dword 1024 tile + 256 g + 4 l + t
__device__ __forceinline__ void load_tile_native(const float* __restrict__ blk, int tile, int lane, float (&v)[16]) {
  const float* p = blk + static_cast<int64_t>(tile) * 1024 + 4 * lane;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 x = *reinterpret_cast<const float4*>(p + 256 * g);
    v[4 * g + 0] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
  }
}
__device__ __forceinline__ void store_tile_native(float* __restrict__ blk, int tile, int lane, const float (&v)[16]) {
  float* p = blk + static_cast<int64_t>(tile) * 1024 + 4 * lane;
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(p + 256 * g) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
// dword of (row, unit) inside a tile-native block
__device__ __forceinline__ int64_t native_index(int64_t row, int unit) {
  return (row >> 5) * 1024 + (unit >> 3) * 256 + ((row & 31) + 32 * ((unit >> 2) & 1)) * 4 + (unit & 3);
}
__device__ __forceinline__ int padded_rows(int B) { return ((B + 31) >> 5) << 5; }
__device__ __forceinline__ bool bit_of(uint32_t word, int j, int kh) { return (word >> (8 * (j >> 2) + 4 * kh + (j & 3))) & 1u; }

// exp(d), d <= 0: v_exp_f32 on d log2(e) with the rounding error of that product put back (1 + lo ln 2): ~1e-7 relative, as expf,
// at a quarter of its instructions (the layers are otherwise bound by these: 48 transcendentals per lane and tile)
__device__ __forceinline__ float exp_fast(float d) {
#ifdef CK_SLSE_LIBM
  return expf(d);
#else
  const float t = d * kL2E;
  const float lo = fmaf(d, kL2E, -t);
  const float e = __builtin_amdgcn_exp2f(t);
  return fabsf(t) < 1e30f ? fmaf(e, lo * kLN2, e) : e;  // (d = -inf: 0, not the NaN of inf - inf)
#endif
}
__device__ __forceinline__ float log_abs(float y) {
#ifdef CK_SLSE_LIBM
  return logf(fabsf(y));
#else
  return __builtin_amdgcn_logf(fabsf(y)) * kLN2;
#endif
}

struct Gather {
  const float* table;         // (F0, C + 1, 32) log|w| of the Embedding weights, row C the integral row; nullptr: children in the arena
  const uint32_t* tsigns;     // (F0, C + 1) their sign words
  const int32_t* child_fold;  // (F, H) Embedding fold of each child
  const int32_t* child_var;   // (F, H) its variable
  const int32_t* xt;          // (D, B) staged batch
  int C;
};

// v <- sum over the children of log|x_h|, sign <- the product of their signs (a word; bits of all 32 units)
__device__ __forceinline__ uint32_t load_children(const float* __restrict__ arena, const uint32_t* __restrict__ signs,
                                                  const int64_t* __restrict__ ro, const Gather& ga, int f, int H, int B, int64_t bl, int kh,
                                                  int tile, int lane, float (&v)[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = 0.f;
  uint32_t sw = 0;
  for (int h = 0; h < H; ++h) {
    float x[16];
    if (ga.table != nullptr) {
      const int64_t e = static_cast<int64_t>(f) * H + h;
      const int xv = ga.xt[static_cast<int64_t>(ga.child_var[e]) * B + bl];
      const int c = xv < 0 ? ga.C : min(xv, ga.C - 1);
      const int64_t row = static_cast<int64_t>(ga.child_fold[e]) * (ga.C + 1) + c;
      load16(ga.table + row * 32, kh, x);
      sw ^= ga.tsigns[row];
    } else {
      load_tile_native(arena + ro[h], tile, lane, x);
      sw ^= signs[(ro[h] >> 5) + bl];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] += x[j];
  }
  return sw;
}

__global__ void __launch_bounds__(256)
    slse_tile32_fwd(const float* __restrict__ arena, const uint32_t* __restrict__ signs, const int64_t* __restrict__ row_off,
                    const float* __restrict__ w, float* __restrict__ out, uint32_t* __restrict__ sout, int H, int B, int tiles_per_wave,
                    Gather ga, int F, int nx) {
  // consecutive workgroups go to consecutive XCDs: the nx workgroups of a fold are spaced 8 apart so that they share one L2
  // (a gathering fold re-reads its children's 2 x 33 KB table rows there instead of in the MALL)
  const int seq = blockIdx.x >> 3;
  const int f = (seq / nx) * 8 + (blockIdx.x & 7), bx = seq % nx;
  if (f >= F) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  WRegs wr;
  load_w<CK_W_ROWMAJOR>(w + static_cast<int64_t>(f) * kK * kK, lane, wr);
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const int tile0 = (bx * 4 + wave) * tiles_per_wave;
  const int Bp = padded_rows(B);
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const int b0 = (tile0 + tt) * 32;
    if (b0 >= B) break;
    const int b = b0 + b_in;
    const bool live = b < B;
    const int64_t bl = live ? b : B - 1;
    float v[16];
    const uint32_t sw = load_children(arena, signs, ro, ga, f, H, B, bl, kh, tile0 + tt, lane, v);
    const float m = row_max16(v);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float e = exp_fast(v[j] - m);
      v[j] = bit_of(sw, j, kh) ? -e : e;
    }
    contract_linear<CK_W_ROWMAJOR>(wr, v);
    uint32_t so = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      so |= (v[j] < 0.f ? 1u : 0u) << (8 * (j >> 2) + 4 * kh + (j & 3));
      v[j] = log_abs(v[j]) + m;
    }
    so |= __shfl_xor(so, 32, 64);
    store_tile_native(out + static_cast<int64_t>(f) * Bp * kK, tile0 + tt, lane, v);  // (rows past B: padding)
    if (live && kh == 0) sout[static_cast<int64_t>(f) * Bp + b] = so;
  }
}

// Four waves walk the row tiles of one fold with dW in registers; one float atomic per weight entry and workgroup at the end
// (the skeleton of sum_clse_bwd_tile32, ck_backward_c.hip, with half its contractions).
template <bool READ_Y>  // (lab, CK_SLSE_READ_Y=1: the stored outputs instead of y = W a again)
__global__ void __launch_bounds__(256)
    slse_tile32_bwd(const float* __restrict__ arena, const uint32_t* __restrict__ signs, float* __restrict__ gx,
                    const int64_t* __restrict__ row_off, const float* __restrict__ w, const float* __restrict__ out,
                    const uint32_t* __restrict__ sout, const float* __restrict__ gout, const int64_t* __restrict__ gout_off,
                    float* __restrict__ dw, int H, int B, Gather ga, int F, int nx, int gx_rowmajor) {
  __shared__ __attribute__((aligned(16))) float wt_s[1024];        // W^T, "transposed tiled" (child_gradient)
  __shared__ __attribute__((aligned(16))) float scr_s[4][2][1024];  // per wave: the two operands of dw_accumulate
  const int seq = blockIdx.x >> 3;  // (the workgroups of a fold on one XCD, as in the forward)
  const int f = (seq / nx) * 8 + (blockIdx.x & 7), bx = seq % nx;
  if (f >= F) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* wf = w + static_cast<int64_t>(f) * 1024;
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int q = idx >> 8, ln = (idx >> 2) & 63, t = idx & 3;
    wt_s[idx] = wf[(8 * q + 4 * (ln >> 5) + t) * 32 + (ln & 31)];
  }
  WRegs wr;  // (A operand of y = W a)
  load_w<CK_W_ROWMAJOR>(wf, lane, wr);
  __syncthreads();
  f32x16 dacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
  const int tiles = (B + 31) / 32;
  const int Bp = padded_rows(B);
  const float* gf = gout + (gout_off != nullptr ? gout_off[f] : static_cast<int64_t>(f) * Bp * 32);
  for (int tile = bx * 4 + wave; tile < tiles; tile += nx * 4) {
    const int b = tile * 32 + b_in;
    const bool live = b < B;
    const int64_t bl = live ? b : B - 1;
    float a[16];
    const uint32_t sw = load_children(arena, signs, ro, ga, f, H, B, bl, kh, tile, lane, a);
    const float m = row_max16(a);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float e = exp_fast(a[j] - m);
      a[j] = bit_of(sw, j, kh) ? -e : e;
    }
    float t[16];
    {
      // y = W a again instead of the stored output (16 MFMAs against 4 KB per tile of a launch that waits for memory: the
      // first layer of config 5 reads 205 MB less): t_o = G_o exp(m) / (y_o exp(m)) = G_o / (W a)_o, signed as it comes; an
      // output without gradient contributes nothing (also where y = 0: out = -inf)
      float y[16], g[16];
      load_tile_native(gf, tile, lane, g);
      if constexpr (READ_Y) {
        load_tile_native(out + static_cast<int64_t>(f) * Bp * 32, tile, lane, y);
        const uint32_t so = sout[static_cast<int64_t>(f) * Bp + bl];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float r = exp_fast(m - y[j]) * g[j];  // exp(m) / y_o = sign exp(m - out_o)
          t[j] = (live && g[j] != 0.f) ? (bit_of(so, j, kh) ? -r : r) : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) y[j] = a[j];
        contract_linear<CK_W_ROWMAJOR>(wr, y);
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = (live && g[j] != 0.f && y[j] != 0.f) ? g[j] * __builtin_amdgcn_rcpf(y[j]) : 0.f;
      }
    }
    float gv[16];
    child_gradient(wt_s, lane, t, a, gv);
    if (!gx_rowmajor) {
      store_tile_native(gx + static_cast<int64_t>(f) * Bp * 32, tile, lane, gv);
    } else if (live) {  // (a gathering layer: rows of (B, 32), what ck_embedding_bwd scatters)
      float* dst = gx + (static_cast<int64_t>(f) * B + b) * 32 + 4 * kh;
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(gv[4 * g], gv[4 * g + 1], gv[4 * g + 2], gv[4 * g + 3]);
    }
    dw_accumulate(dacc, scr_s[wave][0], scr_s[wave][1], b_in, kh, t, a);
  }
  // D[o][i] sits in lane (i, hi) register r with o = 8 (r >> 2) + 4 hi + (r & 3): the four waves' sums through LDS
  __syncthreads();
  float* red = &scr_s[0][0][0];  // [4][1024]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + (8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + b_in] = dacc[r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const float v = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
    if (v != 0.f) atomicAdd(dw + static_cast<int64_t>(f) * 1024 + idx, v);
  }
}

// Few outputs (Ko <= 4: the scalar sum fold at the top of the circuit) over 32 product-type inputs: a half-wave per batch row,
// lane = input unit, the dot products are lane reductions.  BWD: also the children's gradient and dW (registers over the rows
// a half-wave walks, float atomics at the end).
template <bool BWD>
__global__ void __launch_bounds__(256)
    slse_few_kernel(const float* __restrict__ arena, const uint32_t* __restrict__ signs, float* __restrict__ gxo,
                    const int64_t* __restrict__ row_off, const float* __restrict__ w, float* __restrict__ out, uint32_t* __restrict__ sout,
                    const float* __restrict__ gout, float* __restrict__ dw, int H, int B, int Ko) {
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, sub = lane >> 5;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  float wv[4], dacc[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    wv[o] = o < Ko ? w[(static_cast<int64_t>(f) * Ko + o) * 32 + n] : 0.f;
    dacc[o] = 0.f;
  }
  for (int b0 = (blockIdx.x * 4 + wave) * 2; b0 < B; b0 += gridDim.x * 8) {
    const int b = b0 + sub;
    const bool live = b < B;
    const int64_t bl = live ? b : B - 1;
    float v = 0.f;
    uint32_t sw = 0;
    for (int h = 0; h < H; ++h) {
      v += arena[ro[h] + native_index(bl, n)];
      sw ^= signs[(ro[h] >> 5) + bl];
    }
    float m = v;
#pragma unroll
    for (int s2 = 1; s2 < 32; s2 <<= 1) m = fmaxf(m, __shfl_xor(m, s2, 64));
    m = ck::clamp_finite(m);
    float e = exp_fast(v - m);
    if ((sw >> n) & 1u) e = -e;
    float gsum = 0.f;
    uint32_t so = 0;
    if constexpr (BWD) so = sout[static_cast<int64_t>(f) * B + bl];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o >= Ko) break;
      if constexpr (!BWD) {
        float y = wv[o] * e;
#pragma unroll
        for (int s2 = 1; s2 < 32; s2 <<= 1) y += __shfl_xor(y, s2, 64);
        if (y < 0.f) so |= 1u << o;
        if (live && n == 0) out[(static_cast<int64_t>(f) * B + b) * Ko + o] = log_abs(y) + m;
      } else {
        const float g = gout[(static_cast<int64_t>(f) * B + bl) * Ko + o];
        const float r = exp_fast(m - out[(static_cast<int64_t>(f) * B + bl) * Ko + o]) * g;
        const float t = (live && g != 0.f) ? (((so >> o) & 1u) ? -r : r) : 0.f;
        gsum = fmaf(wv[o], t, gsum);
        dacc[o] = fmaf(t, e, dacc[o]);
      }
    }
    if constexpr (!BWD) {
      if (live && n == 0) sout[static_cast<int64_t>(f) * B + b] = so;
    } else if (live) {
      gxo[static_cast<int64_t>(f) * padded_rows(B) * 32 + native_index(b, n)] = e * gsum;
    }
  }
  if constexpr (BWD) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o >= Ko) break;
      const float d = dacc[o] + __shfl_xor(dacc[o], 32, 64);
      if (sub == 0 && d != 0.f) atomicAdd(dw + (static_cast<int64_t>(f) * Ko + o) * 32 + n, d);
    }
  }
}

// rows of 32 weights -> log|w| and one sign word per row (a half-wave per row)
__global__ void __launch_bounds__(256)
    slse_table_kernel(const float* __restrict__ table, float* __restrict__ ltab, uint32_t* __restrict__ tsigns, int64_t rows) {
  const int64_t row = (blockIdx.x * 256ll + threadIdx.x) >> 5;
  const int k = threadIdx.x & 31;
  if (row >= rows) return;
  const float w = table[row * 32 + k];
  ltab[row * 32 + k] = logf(fabsf(w));
  uint32_t bit = w < 0.f ? (1u << k) : 0u;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) bit |= __shfl_xor(bit, o, 64);
  if (k == 0) tsigns[row] = bit;
}

// The three tables of an Embedding fold from its weight (32, C) in one pass through LDS: the gather table (C + 1, 32) the
// backward divides by (row C: the integral row of ones, input.py:280-282), its signed-log form and the sign words.
__global__ void __launch_bounds__(256)
    slse_tables_kernel(const float* __restrict__ weight, float* __restrict__ table, float* __restrict__ ltab, uint32_t* __restrict__ tsigns,
                       int C) {
  extern __shared__ float tw_s[];  // [32][C + 1]
  const int f = blockIdx.x, ld = C + 1;
  const float* wf = weight + static_cast<int64_t>(f) * 32 * C;
  for (int k = threadIdx.x >> 6; k < 32; k += 4)  // (a wave per unit: no division by C)
    for (int c = threadIdx.x & 63; c < C; c += 64) tw_s[k * ld + c] = wf[k * C + c];
  __syncthreads();
  const int64_t row0 = static_cast<int64_t>(f) * (C + 1);
  for (int i = threadIdx.x; i < (C + 1) * 32; i += 256) {  // (a half-wave per row of 32 units)
    const int c = i >> 5, k = i & 31;
    const float w = c < C ? tw_s[k * ld + c] : 1.f;
    table[row0 * 32 + i] = w;
    if (ltab == nullptr) continue;  // (uniform: the linear table alone -- the leaf region reads no other)
    ltab[row0 * 32 + i] = logf(fabsf(w));
    uint32_t bit = w < 0.f ? (1u << k) : 0u;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) bit |= __shfl_xor(bit, o, 64);
    if (k == 0) tsigns[row0 + c] = bit;
  }
}

int check(const void* arena, const void* signs, const int64_t* row_off, const float* w, const void* out, const void* sout, int F, int H, int B,
          int Ko, const Gather& ga, const char* who) {
  CK_REQUIRE(w && out && sout, "%s: null pointer", who);
  CK_REQUIRE(F > 0 && F <= 65535 && H > 0 && B > 0, "%s: bad sizes (F=%d, H=%d, B=%d)", who, F, H, B);
  CK_REQUIRE(Ko == 32 || (Ko >= 1 && Ko <= 4), "%s: 32 output units, or 1 .. 4 (Ko=%d)", who, Ko);
  if (ga.table != nullptr) {
    CK_REQUIRE(ga.tsigns && ga.child_fold && ga.child_var && ga.xt && ga.C > 0 && Ko == 32,
               "%s: a gathering layer needs table_signs, child_fold, child_var, xt, C and 32 outputs", who);
    CK_REQUIRE(ck::aligned16(ga.table), "%s: the table must be 16-byte aligned", who);
  } else {
    CK_REQUIRE(arena && signs && row_off, "%s: null pointer", who);
    CK_REQUIRE(ck::aligned16(arena), "%s: the arena must be 16-byte aligned", who);
  }
  CK_REQUIRE(ck::aligned16(w) && ck::aligned16(out), "%s: buffers must be 16-byte aligned", who);
  return CK_OK;
}

}  // namespace

extern "C" int ck_slse_table(const float* table, float* log_table, uint32_t* table_signs, int64_t rows, void* stream) {
  CK_REQUIRE(table && log_table && table_signs && rows > 0, "ck_slse_table: bad arguments");
  const dim3 grid(static_cast<unsigned>((rows * 32 + 255) / 256)), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(slse_table_kernel, grid, block, 0, s, table, log_table, table_signs, rows);
        return hipGetLastError();
      },
      stream);
}

extern "C" int ck_slse_tables(const float* weight, float* table, float* log_table, uint32_t* table_signs, int F, int C, void* stream) {
  CK_REQUIRE(weight && table && (log_table != nullptr) == (table_signs != nullptr) && F > 0 && C > 0, "ck_slse_tables: bad arguments");
  const size_t lds = static_cast<size_t>(32) * (C + 1) * sizeof(float);
  if (lds > 160 * 1024) return ck::fail(CK_ERR_UNSUPPORTED, "ck_slse_tables: %d states do not fit in LDS", C);
  return ck::dispatch(
      [=](hipStream_t s) {
        if (lds > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(slse_tables_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
          if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(slse_tables_kernel, dim3(F), dim3(256), lds, s, weight, table, log_table, table_signs, C);
        return hipGetLastError();
      },
      stream);
}

extern "C" int ck_slse_fwd(const float* arena, const uint32_t* signs, const int64_t* row_off, const float* w, float* out, uint32_t* sout,
                           int F, int H, int B, int Ko, const float* log_table, const uint32_t* table_signs, const int32_t* child_fold,
                           const int32_t* child_var, const int32_t* xt, int C, void* stream) {
  const Gather ga{log_table, table_signs, child_fold, child_var, xt, C};
  if (int st = check(arena, signs, row_off, w, out, sout, F, H, B, Ko, ga, "ck_slse_fwd")) return st;
  if (Ko == 32) {
    const int tiles = (B + 31) / 32;
    int tpw = 1;
    while (tpw < 4 && static_cast<int64_t>(F) * ((tiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) >= 2048) tpw *= 2;
    const int nx = (tiles + 4 * tpw - 1) / (4 * tpw);
    const dim3 grid(static_cast<unsigned>((F + 7) / 8 * 8 * nx)), block(256);
    return ck::dispatch(
        [=](hipStream_t s) {
          hipLaunchKernelGGL(slse_tile32_fwd, grid, block, 0, s, arena, signs, row_off, w, out, sout, H, B, tpw, ga, F, nx);
          return hipGetLastError();
        },
        stream);
  }
  const dim3 grid(static_cast<unsigned>(std::max(1, std::min((B + 7) / 8, 1024))), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(slse_few_kernel<false>, grid, block, 0, s, arena, signs, static_cast<float*>(nullptr), row_off, w, out, sout,
                           static_cast<const float*>(nullptr), static_cast<float*>(nullptr), H, B, Ko);
        return hipGetLastError();
      },
      stream);
}

extern "C" int ck_slse_bwd(const float* arena, const uint32_t* signs, const int64_t* row_off, const float* w, const float* out,
                           const uint32_t* sout, const float* gout, const int64_t* gout_off, float* gx, float* dw, int F, int H, int B, int Ko,
                           const float* log_table, const uint32_t* table_signs, const int32_t* child_fold, const int32_t* child_var,
                           const int32_t* xt, int C, void* stream) {
  const Gather ga{log_table, table_signs, child_fold, child_var, xt, C};
  if (int st = check(arena, signs, row_off, w, out, sout, F, H, B, Ko, ga, "ck_slse_bwd")) return st;
  CK_REQUIRE(gx && gout && dw && ck::aligned16(gx) && ck::aligned16(gout), "ck_slse_bwd: null or misaligned pointer");
  if (Ko == 32) {
    const int tiles = (B + 31) / 32;
    const int nx = std::max(1, std::min((tiles + 3) / 4, 16));
    const dim3 grid(static_cast<unsigned>((F + 7) / 8 * 8 * nx)), block(256);
    const char* env = getenv("CK_SLSE_READ_Y");
    const bool read_y = env != nullptr && atoi(env) != 0;
    return ck::dispatch(
        [=](hipStream_t s) {
          if (read_y)
            hipLaunchKernelGGL(slse_tile32_bwd<true>, grid, block, 0, s, arena, signs, gx, row_off, w, out, sout, gout, gout_off, dw, H, B, ga, F,
                               nx, ga.table != nullptr ? 1 : 0);
          else
            hipLaunchKernelGGL(slse_tile32_bwd<false>, grid, block, 0, s, arena, signs, gx, row_off, w, out, sout, gout, gout_off, dw, H, B, ga, F,
                               nx, ga.table != nullptr ? 1 : 0);
          return hipGetLastError();
        },
        stream);
  }
  CK_REQUIRE(gout_off == nullptr, "ck_slse_bwd: gout_off is read by 32-output layers only");
  const dim3 grid(static_cast<unsigned>(std::max(1, std::min((B + 7) / 8, 256))), F), block(256);
  return ck::dispatch(
      [=](hipStream_t s) {
        hipLaunchKernelGGL(slse_few_kernel<true>, grid, block, 0, s, arena, signs, gx, row_off, w, const_cast<float*>(out),
                           const_cast<uint32_t*>(sout), gout, dw, H, B, Ko);
        return hipGetLastError();
      },
      stream);
}
