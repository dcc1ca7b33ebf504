// Dense / CP-T log-einsum-exp layers with MANY units (Ki, Ko multiples of 32 beyond the 32 / 64 tiles of
// ck_sum.hip / ck_cp.hip): per fold a (B x N) . (N x Ko) product in linear space between the exp and
// the log of LSESumSemiring.apply_reduce (semiring.py:383-408), N = Ki (TorchCPTLayer / arity-1
// TorchSumLayer: the children are multiplied first) or H*Ki (TorchSumLayer over the concatenation of
// its children, inner.py:266-273).
//
// Decomposition: a workgroup = 4 wavefronts = 4 x 32 batch rows of one fold.  Each wave keeps the
// WHOLE exponentiated row block e = exp(v - m) of its 32 rows in registers (N / 32 tiles of ck_tile.h:
// the row maximum is then register-local + one cross-lane exchange, and e is reused for every output
// block).  The weights are streamed one 32-output block at a time through two LDS buffers shared by the
// four waves (one barrier per block: the next block is staged while the current one is contracted), as
// MFMA A operands; v_mfma_f32_32x32x2_f32 keeps the contraction exact fp32.  Per barrier a wave issues
// 16 * N/32 MFMAs (8192 cycles at N = 256), so the kernel is bound by the matrix pipe, not by the
// staging.
#include "ck_internal.h"
#include "ck_tile.h"

namespace {

// NQ: N / 32 (1..8).  CAT: the N inputs are the concatenation of H children of Ki units (else their product).
// CT: 0 = exact fp32 (the product); 3 / 6 = the labelled bf16-split VARIANTS (ck_tile.h contract_bf16; ck_cp.hip region_dma_kernel
// has the same scheme): a staged block of weights is cut in LDS, in place, into P = 2 / 3 bf16 pieces by the threads that requested
// it (thread (block of 16 inputs, lane) requests the two float4s of its lane it cuts; the third pieces go behind the block) while
// the previous block is being contracted by nobody else's leave -- i.e. behind the wave's own chain; the exponentiated row block is
// cut ONCE per row tile, and every 16 inputs cost 3 / 6 v_mfma_f32_32x32x16_bf16 instead of 8 v_mfma_f32_32x32x2_f32.
template <int NQ, bool CAT, int CT = 0>
__global__ void __launch_bounds__(256)
    sum_lse_gemm_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                        const float* __restrict__ w, float* __restrict__ out, int H, int B, int Ki, int Ko) {
  constexpr int N = 32 * NQ;
  constexpr int P = CT == 0 ? 1 : CT / 3 + 1;
  constexpr int BUF = CT == 6 ? NQ * 1536 : NQ * 1024;  // floats of one block buffer
  constexpr int ITEMS = (NQ + 1) / 2;                   // CT != 0: (block of 16 inputs, lane) items per thread
  extern __shared__ __attribute__((aligned(16))) float w_s[];  // [2][NQ][4][64] float4 (CT = 6: + [2 NQ][64] third pieces)
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (blockIdx.x * 4 + wave) * 32 + b_in;
  const bool live = b < B;
  const int bl = live ? b : B - 1;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* wf = w + static_cast<int64_t>(f) * Ko * N;
  const int npb = Ko >> 5;  // output blocks

  // rows 32p .. 32p+31 of the (Ko, N) matrix -> operand layout, global -> LDS directly (global_load_lds_dwordx4: lane l
  // of wave w writes 16 bytes at LDS word 4 (256 k + 64 w + l)), one block ahead of its use: the loads are in flight
  // during a whole block of MFMAs and a wave waits for its own share (vmcnt) before the barrier that publishes it
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto stage_async = [&](int p, int buf) {
    float* dst = w_s + buf * BUF;
    if constexpr (CT != 0) {
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const int blk = wave_u + 4 * r;  // (whole waves)
        if (blk >= 2 * NQ) continue;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int q = blk >> 1, g = 2 * (blk & 1) + k;
          const float* src = wf + static_cast<int64_t>(32 * p + (lane & 31)) * N + 32 * q + 8 * g + 4 * (lane >> 5);
          __builtin_amdgcn_global_load_lds((ck::gptr_t)src, (ck::lptr_t)(dst + 4 * ((2 * blk + k) * 64)), 16, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int ln = i & 63, g = (i >> 6) & 3, q = i >> 8;
      const float* src = wf + static_cast<int64_t>(32 * p + (ln & 31)) * N + 32 * q + 8 * g + 4 * (ln >> 5);
      __builtin_amdgcn_global_load_lds((ck::gptr_t)src, (ck::lptr_t)(dst + 4 * (256 * k + 64 * wave_u)), 16, 0, 0);
    }
  };
  // CT != 0: this thread's items of the block in buffer `buf` (landed: behind the wave's own vmcnt wait), cut in place
  auto cut_block = [&](int buf) {
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(w_s)) + buf * (BUF * 4) + lane * 16;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const int blk = wave_u + 4 * r;
      if (blk >= 2 * NQ) continue;
      const uint32_t a0 = base + 2 * blk * 1024;
      f32x4v x0, x1;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(a0) : "memory");
      float rr[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
      for (int pc = 0; pc < P; ++pc) {
        u32x4v d;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_amdgcn_perm(__float_as_uint(rr[2 * j + 1]), __float_as_uint(rr[2 * j]), 0x07060302u);
        const uint32_t ad = pc < 2 ? a0 + 1024 * pc : base + (4 * NQ + blk) * 1024;
        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(d) : "memory");
        if (pc + 1 < P) {
#pragma unroll
          for (int j = 0; j < 8; ++j) rr[j] -= __uint_as_float(__float_as_uint(rr[j]) & 0xffff0000u);  // exact
        }
      }
    }
  };
  stage_async(0, 0);

  float e[NQ][16];
  if (CAT) {
    const int qpc = Ki >> 5;  // 32-unit blocks per child
#pragma unroll
    for (int q = 0; q < NQ; ++q) tile_load(arena + ro[q / qpc] + static_cast<int64_t>(bl) * Ki + 32 * (q % qpc) + 4 * kh, e[q]);
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) tile_load(arena + ro[0] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, e[q]);
    for (int h = 1; h < H; ++h)
#pragma unroll
      for (int q = 0; q < NQ; ++q) tile_load_add(arena + ro[h] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, e[q]);
  }
  float m = e[0][0];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, e[q][j]);
  m = ck::xhalf_max(m);
  m = ck::clamp_finite(m);
  const float nml = exp_offset(m, 0.f);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) e[q][j] = __builtin_amdgcn_exp2f(fmaf(e[q][j], kL2E, nml));

  // CT != 0: the B operands, piece pc of registers 8 m .. 8 m + 7 of input block q at [pc][2 q + m] (cut once per row tile)
  u32x4v ep[P][2 * NQ];
  if constexpr (CT != 0) {
#pragma unroll
    for (int pc = 0; pc < P; ++pc)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int d = 0; d < 4; ++d)
            ep[pc][2 * q + mh][d] = __builtin_amdgcn_perm(__float_as_uint(e[q][8 * mh + 2 * d + 1]), __float_as_uint(e[q][8 * mh + 2 * d]), 0x07060302u);
        if (pc + 1 < P) {
#pragma unroll
          for (int j = 0; j < 16; ++j) e[q][j] -= __uint_as_float(__float_as_uint(e[q][j]) & 0xffff0000u);  // exact
        }
      }
  }

  if constexpr (CT != 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cut_block(0);
  }
  float* dst = out + (static_cast<int64_t>(f) * B + bl) * Ko + 4 * kh;
  for (int p = 0; p < npb; ++p) {
    // this wave's share of block p has landed in LDS (CT != 0: and is cut); block p is staged when every wave has; every wave has
    // left block p - 1
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (p + 1 < npb) stage_async(p + 1, (p + 1) & 1);
    // (inline LDS reads: with plain loads the compiler waits for the block requested two lines above, ck_tile.h)
    const uint32_t wb = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(w_s)) + (p & 1) * (BUF * 4) + lane * 16;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (CT != 0) {
      auto mm = [&](const f32x4v& ww, const u32x4v& y) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, ww), __builtin_bit_cast(bf16x8v, y), acc, 0, 0, 0);
      };
      static_for<0, NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int o = q * 4096;
        f32x4v w00, w01, w10, w11, w20, w21;
        lds_read4_off<o, o + 1024, o + 2048, o + 3072>(w00, w01, w10, w11, wb);
        if constexpr (P == 3) {
          asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w20), "=&v"(w21)
                       : "v"(wb), "n"((4 * NQ + 2 * q) * 1024), "n"((4 * NQ + 2 * q + 1) * 1024)
                       : "memory");
          mm(w20, ep[0][2 * q]);  // (smallest terms first)
          mm(w01, ep[1][2 * q]);
          mm(w00, ep[2][2 * q]);
        }
        mm(w01, ep[0][2 * q]);
        mm(w00, ep[1][2 * q]);
        mm(w00, ep[0][2 * q]);
        if constexpr (P == 3) {
          mm(w21, ep[0][2 * q + 1]);
          mm(w11, ep[1][2 * q + 1]);
          mm(w10, ep[2][2 * q + 1]);
        }
        mm(w11, ep[0][2 * q + 1]);
        mm(w10, ep[1][2 * q + 1]);
        mm(w10, ep[0][2 * q + 1]);
      });
    } else
    static_for<0, NQ>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int o = q * 4096;
      f32x4v w0, w1, w2, w3;
      lds_read4_off<o, o + 1024, o + 2048, o + 3072>(w0, w1, w2, w3, wb);
      const f32x4v* wg[4] = {&w0, &w1, &w2, &w3};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[0], e[q][4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[1], e[q][4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[2], e[q][4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((*wg[g])[3], e[q][4 * g + 3], acc, 0, 0, 0);
      }
    });
    if (live) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o4;
        o4.x = fmaf(__builtin_amdgcn_logf(acc[4 * g + 0]), kLN2, m);
        o4.y = fmaf(__builtin_amdgcn_logf(acc[4 * g + 1]), kLN2, m);
        o4.z = fmaf(__builtin_amdgcn_logf(acc[4 * g + 2]), kLN2, m);
        o4.w = fmaf(__builtin_amdgcn_logf(acc[4 * g + 3]), kLN2, m);
        *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = o4;
      }
    }
    if constexpr (CT != 0) {  // block p + 1 (requested at the top of this step: in flight during the chain above) cut behind the chain
      if (p + 1 < npb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cut_block((p + 1) & 1);
      }
    }
  }
}

// The same layer with 257..1024 contracted inputs: the exponentiated row block no longer fits the registers of one
// wave, so S waves share a tile of 32 rows and each keeps 1 / S of the inputs (NQ blocks of 32).  Per block of 32
// outputs every wave contracts its inputs against its slice of the weight rows and the S partial sums are added
// through LDS in a fixed order.  Weights travel global -> registers -> LDS one step ahead of their use (steps of
// QS input blocks per wave, two LDS buffers), as in the Tucker kernel below.
// CT: 0 = exact fp32; 3 / 6 = the labelled bf16-split variants: a thread requests the two float4s of one lane that hold 16 inputs
// (an item = (slice, input block, half, lane)), cuts their 8 weights into P = 2 / 3 bf16 pieces on the way to LDS (layout
// [(slice, block, half)][piece][lane] x 8 bf16: one ds_read_b128 per operand), the exponentiated row block once per row tile.
template <int NQ, int S, bool CAT, int CT = 0>
__global__ void __launch_bounds__(256)
    sum_lse_gemm_split_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                              const float* __restrict__ w, float* __restrict__ out, int H, int B, int Ki, int Ko) {
  constexpr int N = 32 * NQ * S;
  constexpr int R = 4 / S;                                    // row tiles per workgroup
  constexpr int QS = S == 4 ? 2 : (NQ % 2 == 0 ? NQ / 2 : NQ);  // input blocks per wave and step
  constexpr int STEPS = NQ / QS;
  constexpr int P = CT == 0 ? 1 : CT / 3 + 1;
  constexpr int CHUNK = CT == 6 ? S * QS * 1536 : S * QS * 1024;  // floats per buffer
  constexpr int PF = S * QS * 1024 / 4 / 256;                 // float4s of a step per thread
  static_assert(CT == 0 || PF % 2 == 0, "items of two float4s");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w_s = smem;                    // [2][S][QS][4][64] float4
  float* red_s = smem + 2 * CHUNK;      // [4 waves][16][64] partial sums; first used for the row maxima
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rt = wave / S, sp = wave % S;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (blockIdx.x * R + rt) * 32 + b_in;
  const bool live = b < B;
  const int bl = live ? b : B - 1;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * H;
  const float* wf = w + static_cast<int64_t>(f) * Ko * N;
  const int nsteps = (Ko >> 5) * STEPS;

  float pre[PF][4];  // (a float4 array copied whole stays in scratch memory: SROA does not split it)
  auto fetch = [&](int c) {  // step c = (output block p, input blocks h * QS .. of every wave)
    const int p = c / STEPS, h = c % STEPS;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      // CT != 0: float4s 2 r and 2 r + 1 of a thread are the two halves (g = 2 m, 2 m + 1) of item j = thread + 256 r
      const int i = CT == 0 ? static_cast<int>(threadIdx.x) + 256 * k
                            : ((((static_cast<int>(threadIdx.x) + 256 * (k >> 1)) >> 6) * 2 + (k & 1)) << 6) + lane;
      const int ln = i & 63, g = (i >> 6) & 3, qq = (i >> 8) % QS, s2 = (i >> 8) / QS;
      const float4 v = *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(32 * p + (ln & 31)) * N +
                                                        32 * (s2 * NQ + h * QS + qq) + 8 * g + 4 * (ln >> 5));
      pre[k][0] = v.x;
      pre[k][1] = v.y;
      pre[k][2] = v.z;
      pre[k][3] = v.w;
    }
  };
  auto commit = [&](int buf) {
    float* dst = w_s + buf * CHUNK;
    if constexpr (CT != 0) {
#pragma unroll
      for (int r = 0; r < PF / 2; ++r) {
        const int j = static_cast<int>(threadIdx.x) + 256 * r;  // item: (slice, block, half) = j >> 6, lane = j & 63
        float rr[8] = {pre[2 * r][0], pre[2 * r][1], pre[2 * r][2], pre[2 * r][3], pre[2 * r + 1][0], pre[2 * r + 1][1], pre[2 * r + 1][2], pre[2 * r + 1][3]};
#pragma unroll
        for (int pc = 0; pc < P; ++pc) {
          u32x4v d;
#pragma unroll
          for (int t = 0; t < 4; ++t) d[t] = __builtin_amdgcn_perm(__float_as_uint(rr[2 * t + 1]), __float_as_uint(rr[2 * t]), 0x07060302u);
          *reinterpret_cast<u32x4v*>(dst + 4 * (((j >> 6) * P + pc) * 64 + lane)) = d;
          if (pc + 1 < P) {
#pragma unroll
            for (int t = 0; t < 8; ++t) rr[t] -= __uint_as_float(__float_as_uint(rr[t]) & 0xffff0000u);  // exact
          }
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) *reinterpret_cast<float4*>(dst + 4 * (threadIdx.x + 256 * k)) = make_float4(pre[k][0], pre[k][1], pre[k][2], pre[k][3]);
  };
  fetch(0);

  float e[NQ][16];
  if (CAT) {
    const int qpc = Ki >> 5;  // 32-unit blocks per child
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int Q = sp * NQ + q;
      tile_load(arena + ro[Q / qpc] + static_cast<int64_t>(bl) * Ki + 32 * (Q % qpc) + 4 * kh, e[q]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) tile_load(arena + ro[0] + static_cast<int64_t>(bl) * Ki + 32 * (sp * NQ + q) + 4 * kh, e[q]);
    for (int h = 1; h < H; ++h)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        tile_load_add(arena + ro[h] + static_cast<int64_t>(bl) * Ki + 32 * (sp * NQ + q) + 4 * kh, e[q]);
  }
  float m = e[0][0];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, e[q][j]);
  m = ck::xhalf_max(m);
  red_s[wave * 64 + lane] = m;
  __syncthreads();
#pragma unroll
  for (int s2 = 0; s2 < S; ++s2) m = fmaxf(m, red_s[(rt * S + s2) * 64 + lane]);
  m = ck::clamp_finite(m);
  const float nml = exp_offset(m, 0.f);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) e[q][j] = __builtin_amdgcn_exp2f(fmaf(e[q][j], kL2E, nml));
  __syncthreads();  // the maxima have been read: red_s is free for the partial sums
  u32x4v ep[P][2 * NQ];  // CT != 0: the B operands, piece pc of registers 8 m .. 8 m + 7 of input block q at [pc][2 q + m]
  if constexpr (CT != 0) {
#pragma unroll
    for (int pc = 0; pc < P; ++pc)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int d = 0; d < 4; ++d)
            ep[pc][2 * q + mh][d] = __builtin_amdgcn_perm(__float_as_uint(e[q][8 * mh + 2 * d + 1]), __float_as_uint(e[q][8 * mh + 2 * d]), 0x07060302u);
        if (pc + 1 < P) {
#pragma unroll
          for (int j = 0; j < 16; ++j) e[q][j] -= __uint_as_float(__float_as_uint(e[q][j]) & 0xffff0000u);  // exact
        }
      }
  }

  float* dst = out + (static_cast<int64_t>(f) * B + bl) * Ko + 4 * kh;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  commit(0);
  fetch(nsteps > 1 ? 1 : 0);
  for (int c = 0; c < nsteps; ++c) {
    __syncthreads();  // step c is in LDS; every wave has left step c - 1
    if (c + 1 < nsteps) commit((c + 1) & 1);
    fetch(c + 2 < nsteps ? c + 2 : nsteps - 1);  // (the last two fetches are redundant re-reads, never committed)
    const float* wb = w_s + (c & 1) * CHUNK + sp * (CT == 0 ? QS * 1024 : QS * 512 * P);
    const int h = c % STEPS;
#pragma unroll
    for (int hh = 0; hh < STEPS; ++hh)  // static register indices: the body of the taken h only
      if (hh == h) {
        if constexpr (CT != 0) {
          const u32x4v* wp = reinterpret_cast<const u32x4v*>(wb) + lane;
          auto mm = [&](const u32x4v& x, const u32x4v& y) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, x), __builtin_bit_cast(bf16x8v, y), acc, 0, 0, 0);
          };
#pragma unroll
          for (int bq = 0; bq < 2 * QS; ++bq) {  // block bq = 2 qq + m of this slice
            u32x4v wq[P];
#pragma unroll
            for (int pc = 0; pc < P; ++pc) wq[pc] = wp[(bq * P + pc) * 64];
            const int eb = 2 * (hh * QS) + bq;
            if constexpr (P == 3) {  // (smallest terms first)
              mm(wq[2], ep[0][eb]);
              mm(wq[1], ep[1][eb]);
              mm(wq[0], ep[2][eb]);
            }
            mm(wq[1], ep[0][eb]);
            mm(wq[0], ep[1][eb]);
            mm(wq[0], ep[0][eb]);
          }
        } else
#pragma unroll
        for (int qq = 0; qq < QS; ++qq)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 w4 = *reinterpret_cast<const float4*>(wb + ((qq * 4 + g) * 64 + lane) * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, e[hh * QS + qq][4 * g + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, e[hh * QS + qq][4 * g + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, e[hh * QS + qq][4 * g + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, e[hh * QS + qq][4 * g + 3], acc, 0, 0, 0);
          }
      }
    if (h == STEPS - 1) {  // output block p is complete: add the S partial sums, wave 0 of the tile writes
      const int p = c / STEPS;
#pragma unroll
      for (int r = 0; r < 16; ++r) red_s[(wave * 16 + r) * 64 + lane] = acc[r];
      __syncthreads();
      if (sp == 0 && live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = red_s[((rt * S) * 16 + r) * 64 + lane];
#pragma unroll
          for (int s2 = 1; s2 < S; ++s2) t += red_s[((rt * S + s2) * 16 + r) * 64 + lane];
          acc[r] = t;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 o4;
          o4.x = fmaf(__builtin_amdgcn_logf(acc[4 * g + 0]), kLN2, m);
          o4.y = fmaf(__builtin_amdgcn_logf(acc[4 * g + 1]), kLN2, m);
          o4.z = fmaf(__builtin_amdgcn_logf(acc[4 * g + 2]), kLN2, m);
          o4.w = fmaf(__builtin_amdgcn_logf(acc[4 * g + 3]), kLN2, m);
          *reinterpret_cast<float4*>(dst + 32 * p + 8 * g) = o4;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // the barrier at the top of the next step separates these reads from the next writes of red_s: a wave
      // reaches its next write only after a whole step, i.e. after that barrier
    }
  }
}

template <int NQ, int S>
hipError_t launch_split(bool cat, int ct, hipStream_t s, const float* arena, const int64_t* row_off, const float* w, float* out,
                        int F, int H, int B, int Ki, int Ko) {
  constexpr int QS = S == 4 ? 2 : (NQ % 2 == 0 ? NQ / 2 : NQ);
  const size_t lds = (2 * S * QS * (ct == 6 ? 1536 : 1024) + 4 * 16 * 64) * sizeof(float);
  const int tiles = (B + 31) / 32, R = 4 / S;
  const dim3 grid((tiles + R - 1) / R, F);
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, arena, row_off, w, out, H, B, Ki, Ko);
    return hipGetLastError();
  };
  if (ct == 3) return cat ? go(sum_lse_gemm_split_kernel<NQ, S, true, 3>) : go(sum_lse_gemm_split_kernel<NQ, S, false, 3>);
  if (ct == 6) return cat ? go(sum_lse_gemm_split_kernel<NQ, S, true, 6>) : go(sum_lse_gemm_split_kernel<NQ, S, false, 6>);
  return cat ? go(sum_lse_gemm_split_kernel<NQ, S, true>) : go(sum_lse_gemm_split_kernel<NQ, S, false>);
}

// Tucker layer of arity 2 (TorchTuckerLayer.forward, optimized.py:89-103: einsum "fbi,fbj,foij->fbo" between the exp and
// the log): per fold a (B x Ki^2) . (Ki^2 x Ko) product whose left operand is the outer product
// e_l (x) e_r of the two exponentiated children.  The outer product is never materialised: a wave keeps
// e_r of its 32 rows as a register tile, reads e_l[b, i] from LDS, and forms e_l[b, i] * e_r[b, j] with one
// multiply per MFMA operand.  The (Ko, Ki^2) weight matrix (1 MiB per fold at K = 64 -- the dominant
// traffic at small batches) is streamed once per workgroup through two LDS buffers, CI left indices i
// and NB blocks of 32 outputs at a time, and shared by the four waves.
// NK = Ki / 32, NB = blocks of 32 outputs accumulated together (independent MFMA chains).
// SPLIT = false: the four waves are four tiles of 32 rows and every wave contracts the whole chunk.
// SPLIT = true (few folds x rows: the launch would not fill the chip and its time is the serial length of ONE
//   workgroup): the four waves share ONE tile of 32 rows, wave w contracts left index i0 + w of every chunk, and the
//   four partial sums are added through LDS in a fixed order -- a quarter of the serial length, 4x the workgroups.
// Ko need not be a multiple of 32 (the root layer has Ko = 1): rows >= Ko of the last block are staged as zeros.
template <int NK, int NB, bool SPLIT>
__global__ void __launch_bounds__(256)
    tucker_lse_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off, const float* __restrict__ w,
                      float* __restrict__ out, int F, int B, int Ko, int gx, int gsplit) {
  constexpr int Ki = 32 * NK;
  constexpr int N = Ki * Ki;
  constexpr int CI = SPLIT ? 4 : (NK == 1 ? 2 : 1);   // left indices per staged chunk
  constexpr int CHUNK = CI * NB * NK * 1024;          // floats per buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w_s = smem;                                  // [2][CI][NB][NK][4][64] float4
  float* el_s = smem + 2 * CHUNK;                     // [4 waves][Ki][32]
  float* red_s = el_s + 4 * Ki * 32;                  // SPLIT: [4 waves][NB][16][64] partial sums
  // XCD-aware mapping (consecutive workgroup ids go to consecutive XCDs): every workgroup of one fold -- its
  // row tiles and output splits -- gets the same id % 8, so the fold's weights are fetched into ONE XCD's L2.
  const int xcd = blockIdx.x & 7, i_in = blockIdx.x >> 3;
  const int per_fold = gx * gsplit;
  const int f = (i_in / per_fold) * 8 + xcd;
  if (f >= F) return;
  const int tg = i_in % per_fold, tx = tg % gx, z = tg / gx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int b = (SPLIT ? tx : tx * 4 + wave) * 32 + b_in;
  const bool live = b < B;
  const int bl = live ? b : B - 1;
  const int64_t* ro = row_off + static_cast<int64_t>(f) * 2;
  const int nblocks = (Ko + 31) / 32;
  const int ngroups = nblocks / NB / gsplit;          // groups of NB output blocks handled by this workgroup
  const int nchunks = ngroups * (Ki / CI);
  const int o_base = z * ngroups * 32 * NB;           // first output of this workgroup
  const float* wf = w + static_cast<int64_t>(f) * Ko * N;

  // chunk c = (output group, left indices i0 .. i0 + CI - 1).  Its weights travel global -> registers one
  // iteration ahead of the registers -> LDS copy, so the HBM latency is covered by a whole chunk of MFMAs.
  // 4 consecutive threads read 64 contiguous bytes of one weight row; 16 rows per 64 threads.
  constexpr int PF = CHUNK / 4 / 256;
  float4 pre[PF];
  auto fetch = [&](int c) {
    const int grp = c / (Ki / CI), i0 = (c % (Ki / CI)) * CI;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int idx = threadIdx.x + 256 * k;
      const int c4 = ((idx >> 6) % (2 * NK)) * 4 + (idx & 3);
      const int rest = (idx >> 6) / (2 * NK);
      const int o = o_base + 32 * NB * grp + (rest % (2 * NB)) * 16 + ((idx >> 2) & 15);
      const int ci = rest / (2 * NB);
      pre[k] = o < Ko ? *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(o) * N + (i0 + ci) * Ki + 4 * c4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int buf) {
    float* dst = w_s + buf * CHUNK;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int idx = threadIdx.x + 256 * k;
      const int c4 = ((idx >> 6) % (2 * NK)) * 4 + (idx & 3);
      const int rest = (idx >> 6) / (2 * NK);
      const int o = (rest % (2 * NB)) * 16 + ((idx >> 2) & 15);
      const int ci = rest / (2 * NB);
      const int col = 4 * c4, q = col >> 5, g = (col >> 3) & 3, k2 = (col >> 2) & 1;
      *reinterpret_cast<float4*>(dst + ((((ci * NB + (o >> 5)) * NK + q) * 4 + g) * 64 + (o & 31) + 32 * k2) * 4) = pre[k];
    }
  };
  fetch(0);

  // exponentiated children: e_r as a register tile, e_l through LDS (read back by left index)
  float er[NK][16];
  float m;
  {
    float el[NK][16];
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      tile_load(arena + ro[0] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, el[q]);
      tile_load(arena + ro[1] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, er[q]);
    }
    float ml = el[0][0], mr = er[0][0];
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        ml = fmaxf(ml, el[q][j]);
        mr = fmaxf(mr, er[q][j]);
      }
    ml = ck::clamp_finite(ck::xhalf_max(ml));
    mr = ck::clamp_finite(ck::xhalf_max(mr));
    const float nl = exp_offset(ml, 0.f), nr = exp_offset(mr, 0.f);
    float* eb = el_s + wave * (Ki * 32) + b_in;
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        er[q][j] = __builtin_amdgcn_exp2f(fmaf(er[q][j], kL2E, nr));
        eb[(32 * q + 8 * (j >> 2) + 4 * kh + (j & 3)) * 32] = __builtin_amdgcn_exp2f(fmaf(el[q][j], kL2E, nl));
      }
    m = ml + mr;
  }

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  const float* el_w = el_s + wave * (Ki * 32) + b_in;
  float* dst = out + (static_cast<int64_t>(f) * B + bl) * Ko;
  commit(0);
  if (nchunks > 1) fetch(1);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // chunk c (and, the first time, e_l) is in LDS; every wave has left chunk c - 1
    if (c + 1 < nchunks) commit((c + 1) & 1);
    if (c + 2 < nchunks) fetch(c + 2);
    const float* wb = w_s + (c & 1) * CHUNK;
    const int i0 = (c % (Ki / CI)) * CI;
#pragma unroll
    for (int cc = 0; cc < (SPLIT ? 1 : CI); ++cc) {
      const int ci = SPLIT ? wave : cc;
      const float eli = el_w[(i0 + ci) * 32];
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 w4[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            w4[nb] = *reinterpret_cast<const float4*>(wb + ((((ci * NB + nb) * NK + q) * 4 + g) * 64 + lane) * 4);
          const float p0 = eli * er[q][4 * g + 0], p1 = eli * er[q][4 * g + 1];
          const float p2 = eli * er[q][4 * g + 2], p3 = eli * er[q][4 * g + 3];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[nb].x, p0, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[nb].y, p1, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[nb].z, p2, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[nb].w, p3, acc[nb], 0, 0, 0);
        }
    }
    if (i0 + CI == Ki) {  // the output group is complete
      const int grp = c / (Ki / CI);
      if constexpr (SPLIT) {  // add the four partial sums (fixed order), wave 0 writes the result
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) red_s[((wave * NB + nb) * 16 + r) * 64 + lane] = acc[nb][r];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[nb][r] = (red_s[((0 * NB + nb) * 16 + r) * 64 + lane] + red_s[((1 * NB + nb) * 16 + r) * 64 + lane]) +
                           (red_s[((2 * NB + nb) * 16 + r) * 64 + lane] + red_s[((3 * NB + nb) * 16 + r) * 64 + lane]);
        }
        __syncthreads();
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int o0 = o_base + 32 * (NB * grp + nb) + 4 * kh;
        if (live && (!SPLIT || wave == 0)) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float o4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) o4[t] = fmaf(__builtin_amdgcn_logf(acc[nb][4 * g + t]), kLN2, m);
            if ((Ko & 31) == 0) {
              *reinterpret_cast<float4*>(dst + o0 + 8 * g) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            } else {
#pragma unroll
              for (int t = 0; t < 4; ++t)
                if (o0 + 8 * g + t < Ko) dst[o0 + 8 * g + t] = o4[t];
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
      }
    }
  }
}

template <int NK, int NB, bool SPLIT>
hipError_t launch_tucker(hipStream_t s, const float* arena, const int64_t* row_off, const float* w, float* out, int F,
                         int B, int Ko, int gx, int gsplit) {
  constexpr int CI = SPLIT ? 4 : (NK == 1 ? 2 : 1);
  const size_t lds = (2 * CI * NB * NK * 1024 + 4 * 32 * NK * 32 + (SPLIT ? 4 * NB * 1024 : 0)) * sizeof(float);
  auto kern = tucker_lse_kernel<NK, NB, SPLIT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  const dim3 grid(static_cast<unsigned>((F + 7) / 8 * 8 * gx * gsplit));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, arena, row_off, w, out, F, B, Ko, gx, gsplit);
  return hipGetLastError();
}

// ---- stream-K Tucker launch --------------------------------------------------------------------------------------
// tucker_lse_kernel gives a workgroup one (fold, 32 outputs, 128 rows) tile: at batch 128 a layer of F folds is 2 F
// workgroups of 55 us each over 768 resident slots -- 2.04 rounds for F = 784 (a third of the machine idle in the
// last one), 1.02 for F = 392, and whole layers of 2..42 folds that take one workgroup's serial time (36 us).  Here the
// work is the flat list of CHUNKS (tile, left index i) and each of G persistent workgroups takes a contiguous
// 1/G of it: every workgroup does the same number of MFMAs whatever F is.  A tile whose chunks straddle workgroups
// is a sum of partial accumulators (they share the row maxima, which only depend on the inputs): each contributor
// stores its 128 x 32 partial sum in its own workspace slot and takes a ticket; the last one adds the slots in
// contributor order -- its own included, so the order never depends on who arrives last -- and writes log(sum) + m.
// Nobody waits for anybody.  Exact fp32 MFMA as tucker_lse_kernel; the split points change the order of the adds, so
// the two agree to fp32 rounding.
// reductions over the 8 lanes that share a weight row (DPP: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror)
__device__ __forceinline__ float oct_max(float v) {
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));
  return fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true)));
}
__device__ __forceinline__ float oct_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true));
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true));
  return v + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xF, 0xF, true));
}

#ifndef CK_TUCKER_LABBITS
#define CK_TUCKER_LABBITS 0
#endif
// lab builds (scripts/exp_tucker_bf16.sh; wrong results on purpose): 1 = every chunk read from the same 32 KiB, 2 = no LDS stores
// of the staged chunk, 4 = no LDS reads of the A operand, 8 = no MFMAs, 16 = no exponentials, 32 = no cut into pieces, 64 = no barrier
constexpr int kLab = CK_TUCKER_LABBITS;
struct StreamKArgs {
  const float* arena;
  const int64_t* row_off;
  const float* w;
  float* out;
  float* ws;            // (G, 2, 128 x 32) partial accumulators: slot 0 = a workgroup's first tile, slot 1 = its last
  float* stats;         // LOGITS: (G, 2, 64) running maximum (log2 units) and sum of the 32 weight rows of a partial tile
  uint32_t* tickets;    // (tiles) zero on entry, zero again afterwards
  int F, B, Ko, nblk, rgroups;
  int64_t total;        // chunks = F * nblk * rgroups * Ki
};

// LOGITS: `w` holds logits theta and the weights are softmax(theta) over the (i, j) axis (parameters/nodes.py:764-772),
// normalised ONLINE: a workgroup stages exp(theta - m) with a running row maximum m (raised, with the accumulators and the
// running sum rescaled, only when a chunk's maximum exceeds it by more than 2^24 -- in practice on no chunk but the
// first), keeps sum exp(theta - m) beside the accumulators and subtracts log(sum) once per output; partial tiles carry
// (m, sum) per weight row into the combine.  The logits are read ONCE per forward and no normalised copy or normaliser
// ever exists in memory.
//
// CT: 0 = the contraction in exact fp32 (the product); 3 / 6 = the labelled bf16-split VARIANTS ("bf16x3" / "bf16x6", ck_tile.h
// contract_bf16): a chunk's weights are cut by truncation into P = 2 / 3 bf16 pieces while they are staged (after the online
// softmax's exponential), e_r once per tile, and the chain over j runs as 3 / 6 products per 16 right indices on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  A chunk in LDS is [piece][block of 16 right indices][slot] x 8 bf16: what
// lane o + 32 kh of the A operand reads with one ds_read_b128, slot = lane ^ 8 kh so that the 8-byte stores of a staging wave
// (8 rows x 8 float4 columns) fall on every bank twice, the minimum.
// DEPTH: weight chunks a thread has requested ahead of the one being staged (registers; the exact launch is bound by its
// instruction stream and keeps one, the variants are bound by the memory's latency: with D chunks of 8 KiB in flight per
// workgroup the launch moves (workgroups x D x 8 KiB) / latency).  OCC: workgroups per CU the registers are budgeted for.
template <int NK, bool LOGITS, int CT = 0, int DEPTH = 1, int OCC = 3>
__global__ void __launch_bounds__(256, OCC) tucker_streamk_kernel(const StreamKArgs a) {
  constexpr int Ki = 32 * NK;
  constexpr int N = Ki * Ki;
  constexpr int P = CT == 0 ? 1 : CT / 3 + 1;
  // floats of one chunk: 32 outputs x Ki right indices as [c4][o_local ^ 8 (c4 & 1)] float4s -- the operand layout with
  // the rows of odd float4 columns swizzled, so that the 64 float4s a wave stages per instruction (8 rows x 8 columns)
  // fall on 16 bank groups x 4, the minimum
  constexpr int CHUNK = CT == 0 ? NK * 1024 : P * NK * 512;
  constexpr int PF = NK;  // float4s of a chunk per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w_s = smem;               // [2][CHUNK]
  float* el_s = smem + 2 * CHUNK;  // [4 waves][Ki][32]
  float* alpha_s = el_s + 4 * Ki * 32;                             // LOGITS: [2][32] accumulator rescaling of chunk buffer b
  uint32_t* flag_s = reinterpret_cast<uint32_t*>(alpha_s + 2 * 32); // [2][4] "some row of wave w was rescaled"
  float* stat_s = alpha_s + 2 * 32 + 2 * 4;                        // [2][32] final (m, sum) of the tile part
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b_in = lane & 31, kh = lane >> 5;
  const int64_t G = gridDim.x, g = blockIdx.x;
  auto start_of = [&](int64_t gg) { return gg * a.total / G; };
  auto owner_of = [&](int64_t c) {  // the workgroup whose range holds chunk c
    int64_t gg = (c * G + G - 1) / a.total;  // >= the answer - 1 ... narrow down
    if (gg > G - 1) gg = G - 1;
    while (gg > 0 && start_of(gg) > c) --gg;
    while (gg + 1 < G && start_of(gg + 1) <= c) ++gg;
    return gg;
  };
  const int64_t c0 = start_of(g), c1 = start_of(g + 1);
  const int64_t first_tile = c0 / Ki;
  float4 pre[DEPTH][PF];

  for (int64_t c = c0; c < c1;) {
    const int64_t tile = c / Ki;
    const int i_begin = static_cast<int>(c - tile * Ki);
    const int i_end = static_cast<int>(min<int64_t>(Ki, i_begin + (c1 - c)));
    const int rg = static_cast<int>(tile % a.rgroups);
    const int nb = static_cast<int>((tile / a.rgroups) % a.nblk);
    const int f = static_cast<int>(tile / (static_cast<int64_t>(a.rgroups) * a.nblk));
    const int b = (rg * 4 + wave) * 32 + b_in;
    const bool live = b < a.B;
    const int bl = live ? b : a.B - 1;
    const int64_t* ro = a.row_off + static_cast<int64_t>(f) * 2;
    const float* wf = a.w + static_cast<int64_t>(f) * a.Ko * N;
    const int o_base = 32 * nb;
    // weight row o_local = 8 wave + lane / 8 (one row per 8 lanes: its maximum is three DPP steps away), float4 column
    // c4 = lane % 8 + 8 k: 8 lanes read 128 contiguous bytes; the operand layout in LDS is [c4][o_local] float4s
    const int o_local = wave * 8 + (lane >> 3);
    const bool o_live = o_base + o_local < a.Ko;
    auto fetch = [&](int i, auto slot) {
      constexpr int sl = decltype(slot)::value;
      const float fill = LOGITS ? -INFINITY : 0.f;  // (an absent row: weight 0 whatever the running maximum is)
#pragma unroll
      for (int k = 0; k < PF; ++k)
        if constexpr (DEPTH > 1)  // (never around a branch: a row past Ko reads the last row -- its outputs are not stored)
          pre[sl][k] = *reinterpret_cast<const float4*>(((kLab & 1) ? a.w : wf) + static_cast<int64_t>((kLab & 1) ? o_local % a.Ko : min(o_base + o_local, a.Ko - 1)) * N + ((kLab & 1) ? (i & 3) : i) * Ki + 4 * ((lane & 7) + 8 * k));
        else
        pre[sl][k] = o_live ? *reinterpret_cast<const float4*>(wf + static_cast<int64_t>(o_base + o_local) * N + i * Ki + 4 * ((lane & 7) + 8 * k))
                        : make_float4(fill, fill, fill, fill);
    };
    float mrun = 0.f, srun = 0.f;  // LOGITS: running maximum (log2 units) of the row, this lane's share of sum exp2(t - mrun)
    auto commit = [&](int i, int buf, auto slot) {
      constexpr int sl = decltype(slot)::value;
      float* dst = w_s + buf * CHUNK;
      float4 v[PF];
#pragma unroll
      for (int k = 0; k < PF; ++k) v[k] = pre[sl][k];
      if constexpr (LOGITS) {
        // The row maximum of a chunk is only computed when it is needed: on the first chunk of a tile part, and when the
        // exponentials against the running maximum come out larger than 2^24 for some lane (then the maximum is raised and
        // they are computed again) -- otherwise eight maxima, three DPP steps and a comparison per chunk would buy nothing.
        float alpha = 1.f;
        bool raised = false;
        auto row_max = [&] {
          float cm = v[0].x;
#pragma unroll
          for (int k = 0; k < PF; ++k) cm = fmaxf(fmaxf(fmaxf(cm, v[k].x), fmaxf(v[k].y, v[k].z)), v[k].w);
          return ck::clamp_finite(oct_max(cm) * kL2E);
        };
        if (i == i_begin) {
          mrun = row_max();
          srun = 0.f;
        }
        float4 e[PF];
        float part = 0.f;
        auto exps = [&] {
          part = 0.f;
#pragma unroll
          for (int k = 0; k < PF; ++k) {
            e[k].x = __builtin_amdgcn_exp2f(fmaf(v[k].x, kL2E, -mrun));
            e[k].y = __builtin_amdgcn_exp2f(fmaf(v[k].y, kL2E, -mrun));
            e[k].z = __builtin_amdgcn_exp2f(fmaf(v[k].z, kL2E, -mrun));
            e[k].w = __builtin_amdgcn_exp2f(fmaf(v[k].w, kL2E, -mrun));
            part += (e[k].x + e[k].y) + (e[k].z + e[k].w);
          }
        };
        if (!(kLab & 16)) exps(); else {
#pragma unroll
          for (int k = 0; k < PF; ++k) e[k] = v[k];
          part = 1.f;
        }
        if (__any(!(part <= 134217728.f))) {  // 2^27 = 8 values of 2^24 (or a NaN): rare, uniform over the wave
          const float cm = row_max();
          if (cm > mrun) {
            raised = true;
            alpha = __builtin_amdgcn_exp2f(mrun - cm);
            srun *= alpha;
            mrun = cm;
          }
          exps();
        }
        srun += part;
#pragma unroll
        for (int k = 0; k < PF; ++k) v[k] = e[k];
        if ((lane & 7) == 0) alpha_s[buf * 32 + o_local] = alpha;
        const bool any_raised = __any(raised);
        if (lane == 0) flag_s[buf * 4 + wave] = any_raised ? 1u : 0u;
      }
      if constexpr (CT == 0) {
#pragma unroll
        for (int k = 0; k < PF; ++k)
          *reinterpret_cast<float4*>(dst + (((lane & 7) + 8 * k) * 32 + (o_local ^ ((lane & 1) * 8))) * 4) = v[k];
      } else {
        // float4 column c4 = lane % 8 + 8 k holds right indices 32 q + 16 m + 8 s + 4 kh + t, t = 0..3 (q = k, m, s, kh = bits 2, 1, 0
        // of lane % 8): dwords 2 s, 2 s + 1 of slot (o_local + 32 kh) ^ 8 kh of block 2 q + m, once per piece
        const int kh_w = lane & 1, s_w = (lane >> 1) & 1, m_w = (lane >> 2) & 1;
        const int slot = (o_local ^ (8 * kh_w)) + 32 * kh_w;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          float r[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
          for (int p = 0; p < P; ++p) {
            uint2 d;
            if (kLab & 32) { d.x = __float_as_uint(r[0]); d.y = __float_as_uint(r[2]); } else {
            d.x = __builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u);
            d.y = __builtin_amdgcn_perm(__float_as_uint(r[3]), __float_as_uint(r[2]), 0x07060302u);
            }
            if (!(kLab & 2)) *reinterpret_cast<uint2*>(dst + ((p * 2 * NK + 2 * k + m_w) * 64 + slot) * 4 + 2 * s_w) = d;
            if (p + 1 < P && !(kLab & 32)) {
#pragma unroll
              for (int t = 0; t < 4; ++t) r[t] -= __uint_as_float(__float_as_uint(r[t]) & 0xffff0000u);  // exact
            }
          }
        }
      }
    };
    // chunk c of this tile part travels in ring slot (c - i_begin) % DEPTH.  With DEPTH > 1 every step requests exactly one chunk
    // (past the end: the last one again) -- a number of requests in flight that depends on the path makes the compiler wait for
    // all of them (vmcnt(0)) where it only needs the oldest
    auto request = [&](int c, auto slot) {
      if constexpr (DEPTH == 1) {
        if (c < i_end) fetch(c, slot);
      } else {
        fetch(min(c, i_end - 1), slot);
      }
    };
    fetch(i_begin, std::integral_constant<int, 0>{});
    static_for<1, DEPTH>([&](auto d) { request(i_begin + decltype(d)::value, d); });
    // exponentiated children of (fold f, this wave's 32 rows): e_r as a register tile, e_l through LDS
    float er[NK][16];
    float m = 0.f;
    __syncthreads();  // every wave has left the previous tile (its e_l rows and weight buffers)
    {
      float el[NK][16];
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        tile_load(a.arena + ro[0] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, el[q]);
        tile_load(a.arena + ro[1] + static_cast<int64_t>(bl) * Ki + 32 * q + 4 * kh, er[q]);
      }
      float ml = el[0][0], mr = er[0][0];
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          ml = fmaxf(ml, el[q][j]);
          mr = fmaxf(mr, er[q][j]);
        }
      ml = ck::clamp_finite(ck::xhalf_max(ml));
      mr = ck::clamp_finite(ck::xhalf_max(mr));
      const float nl = exp_offset(ml, 0.f), nr = exp_offset(mr, 0.f);
      float* eb = el_s + wave * (Ki * 32) + b_in;
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          er[q][j] = __builtin_amdgcn_exp2f(fmaf(er[q][j], kL2E, nr));
          eb[(32 * q + 8 * (j >> 2) + 4 * kh + (j & 3)) * 32] = __builtin_amdgcn_exp2f(fmaf(el[q][j], kL2E, nl));
        }
      m = ml + mr;
    }
    u32x4v erp[P][2 * NK];  // CT != 0: the B operands, piece p of e_r's registers 8 m .. 8 m + 7 of quarter q at [p][2 q + m]
    if constexpr (CT != 0) {
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < NK; ++q) {
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int d = 0; d < 4; ++d)
              erp[p][2 * q + mh][d] = __builtin_amdgcn_perm(__float_as_uint(er[q][8 * mh + 2 * d + 1]), __float_as_uint(er[q][8 * mh + 2 * d]), 0x07060302u);
          if (p + 1 < P) {
#pragma unroll
            for (int j = 0; j < 16; ++j) er[q][j] -= __uint_as_float(__float_as_uint(er[q][j]) & 0xffff0000u);  // exact
          }
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* el_w = el_s + wave * (Ki * 32) + b_in;
    // after the barrier of chunk i: chunk i + 1 (ring slot `slot`) into the other buffer, chunk i + 1 + DEPTH requested
    auto stage_ahead = [&](int i, auto slot) {
      if (i + 1 < i_end) commit(i + 1, (i + 1 - i_begin) & 1, slot);
      request(i + 1 + DEPTH, slot);
    };
    auto contract = [&](int i) {  // chunk i against the outer-product row e_l[i] * e_r[.]
      const float* wb = w_s + ((i - i_begin) & 1) * CHUNK;
      if constexpr (LOGITS) {  // chunk i was staged against a higher maximum than the accumulators hold (rare)
        const int buf = (i - i_begin) & 1;
        const uint4 fl = *reinterpret_cast<const uint4*>(flag_s + 4 * buf);
        if (__builtin_amdgcn_readfirstlane(fl.x | fl.y | fl.z | fl.w) != 0) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const float4 al = *reinterpret_cast<const float4*>(alpha_s + 32 * buf + 8 * gq + 4 * kh);
            acc[4 * gq + 0] *= al.x;
            acc[4 * gq + 1] *= al.y;
            acc[4 * gq + 2] *= al.z;
            acc[4 * gq + 3] *= al.w;
          }
        }
      }
      // acc[b, o] += e_l[b, i] * sum_j W[o, i, j] e_r[b, j]: the chain runs on e_r as it is and its result enters the
      // accumulators with ONE multiply-add per output (16 per chunk instead of the 32 products e_l[i] * e_r[j])
      const float eli = el_w[i * 32];
      f32x16 part;
      if constexpr (CT != 0) {
        const u32x4v* wp = reinterpret_cast<const u32x4v*>(wb) + (lane ^ (8 * kh));
        bool first = true;
        auto mm = [&](u32x4v x, u32x4v y) {
          if (first) {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            part = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, x), __builtin_bit_cast(bf16x8v, y), zero, 0, 0, 0);
            first = false;
          } else {
            part = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, x), __builtin_bit_cast(bf16x8v, y), part, 0, 0, 0);
          }
        };
#pragma unroll
        for (int blk = 0; blk < 2 * NK; ++blk) {
          u32x4v w[P];
#pragma unroll
          for (int p = 0; p < P; ++p) w[p] = (kLab & 4) ? erp[p][blk] : wp[(p * 2 * NK + blk) * 64];
          if (kLab & 8) { part[blk] = __uint_as_float(w[0][0] ^ w[P - 1][1]); continue; }
          if constexpr (P == 3) {  // (smallest terms first)
            mm(w[2], erp[0][blk]);
            mm(w[1], erp[1][blk]);
            mm(w[0], erp[2][blk]);
          }
          mm(w[1], erp[0][blk]);
          mm(w[0], erp[1][blk]);
          mm(w[0], erp[0][blk]);
        }
      } else {
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 w4 = *reinterpret_cast<const float4*>(wb + ((q * 4 + gq) * 64 + kh * 32 + (b_in ^ (8 * kh))) * 4);
          if (q == 0 && gq == 0) {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            part = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, er[q][0], zero, 0, 0, 0);
          } else {
            part = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, er[q][4 * gq + 0], part, 0, 0, 0);
          }
          part = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, er[q][4 * gq + 1], part, 0, 0, 0);
          part = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, er[q][4 * gq + 2], part, 0, 0, 0);
          part = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, er[q][4 * gq + 3], part, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaf(part[r], eli, acc[r]);
    };
    commit(i_begin, 0, std::integral_constant<int, 0>{});
    request(i_begin + DEPTH, std::integral_constant<int, 0>{});
    if constexpr (DEPTH == 1) {
      for (int i = i_begin; i < i_end; ++i) {
        __syncthreads();  // chunk i (and, the first time, e_l) is in LDS; every wave has left chunk i - 1
        stage_ahead(i, std::integral_constant<int, 0>{});
        contract(i);
      }
    } else {
      // whole turns of the ring without a branch around a request, then the rest of the part (nothing left to request)
      int i = i_begin;
      for (; i + DEPTH <= i_end; i += DEPTH) {
        static_for<0, DEPTH>([&](auto d) {
          const int ii = i + decltype(d)::value;
          if (!(kLab & 64)) __syncthreads();
          stage_ahead(ii, std::integral_constant<int, (decltype(d)::value + 1) % DEPTH>{});
          contract(ii);
        });
      }
      static_for<0, DEPTH - 1>([&](auto d) {
        const int ii = i + decltype(d)::value;
        if (ii < i_end) {
          __syncthreads();
          if (ii + 1 < i_end) commit(ii + 1, (ii + 1 - i_begin) & 1, std::integral_constant<int, decltype(d)::value + 1>{});
          contract(ii);
        }
      });
    }
    float* dst = a.out + (static_cast<int64_t>(f) * a.B + bl) * a.Ko;
    const int o0 = o_base + 4 * kh;
    if constexpr (LOGITS) {  // (m, sum) of the 32 weight rows of this tile part
      const float st = oct_sum(srun);
      if ((lane & 7) == 0) {
        stat_s[o_local] = mrun;
        stat_s[32 + o_local] = st;
      }
      __syncthreads();
    }
    // y: sums of weights * products; sw (LOGITS): the sums of the weights of the rows, log(y / sw) + m is the output
    auto write_out = [&](const f32x16& y, const f32x16& sw) {
      if (!live) return;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float o4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float l2 = __builtin_amdgcn_logf(y[4 * gq + t]);
          if constexpr (LOGITS) l2 -= __builtin_amdgcn_logf(sw[4 * gq + t]);
          o4[t] = fmaf(l2, kLN2, m);
        }
        if ((a.Ko & 31) == 0) {
          *reinterpret_cast<float4*>(dst + o0 + 8 * gq) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (o0 + 8 * gq + t < a.Ko) dst[o0 + 8 * gq + t] = o4[t];
        }
      }
    };
    if (i_begin == 0 && i_end == Ki) {  // the whole tile is this workgroup's
      f32x16 sw;
#pragma unroll
      for (int r = 0; r < 16; ++r) sw[r] = LOGITS ? stat_s[32 + 8 * (r >> 2) + 4 * kh + (r & 3)] : 1.f;
      write_out(acc, sw);
    } else {
      // partial: slot (g, tile == first tile ? 0 : 1); lane-major so that a wave writes 16 contiguous KiB-quarters
      const int64_t my_slot = g * 2 + (tile == first_tile ? 0 : 1);
      float* slot = a.ws + (my_slot * 4 + wave) * 1024;
      // agent-scope atomic stores and loads (write-through / past the non-coherent caches) instead of release / acquire
      // fences: a fence writes back and invalidates a whole L2 -- with 768 workgroups doing it, a sixth of the launch
#pragma unroll
      for (int r = 0; r < 16; ++r) __hip_atomic_store(slot + r * 64 + lane, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (LOGITS) {
        if (wave == 0) __hip_atomic_store(a.stats + my_slot * 64 + lane, stat_s[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have completed
      __syncthreads();  // ... those of all four waves
      const int64_t t0 = tile * Ki;
      const int64_t g_first = owner_of(t0), g_last = owner_of(t0 + Ki - 1);
      unsigned int ticket = 0;
      if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __shared__ unsigned int s_ticket;
      if (threadIdx.x == 0) s_ticket = ticket;
      __syncthreads();
      if (s_ticket == static_cast<unsigned int>(g_last - g_first)) {  // the last contributor: everybody's slot is visible
        f32x16 y, sw;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          y[r] = 0.f;
          sw[r] = 1.f;
        }
        auto slot_of = [&](int64_t gg) { return gg * 2 + ((start_of(gg) / Ki == tile) ? 0 : 1); };
        auto load_part = [&](int64_t gg, f32x16& p) {
          const float* src = a.ws + (slot_of(gg) * 4 + wave) * 1024;
#pragma unroll
          for (int r = 0; r < 16; ++r) p[r] = __hip_atomic_load(src + r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // one pass over the parts in contributor order, the next part's loads in flight while this one is added (each
        // is a round trip to L2)
        f32x16 cur, nxt;
        load_part(g_first, cur);
        // LOGITS: the parts were staged against different row maxima m_p.  With M = max_p m_p the factors
        // e_p = exp2(m_p - M) bring them to one scale: y = sum_p e_p acc_p, sum of weights = sum_p e_p s_p.  All (m_p, s_p)
        // are fetched at once (a thread per part and row), the factors go through LDS (the chunk buffers are free now).
        float* ef_s = w_s;                                                 // [parts][32] m_p, then e_p
        float* sp_s = w_s + 32 * static_cast<int>(g_last - g_first + 1);   // [parts][32] s_p; row 0 becomes the total
        if constexpr (LOGITS) {
          const int np = static_cast<int>(g_last - g_first + 1);  // <= Ki / 8 + 1: a workgroup has at least 8 chunks
          for (int idx = threadIdx.x; idx < np * 32; idx += 256) {
            const float* st = a.stats + slot_of(g_first + (idx >> 5)) * 64 + (idx & 31);
            ef_s[idx] = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sp_s[idx] = __hip_atomic_load(st + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __syncthreads();
          if (threadIdx.x < 32) {
            float mm = ef_s[threadIdx.x];
            for (int q = 1; q < np; ++q) mm = fmaxf(mm, ef_s[q * 32 + threadIdx.x]);
            float tot = 0.f;
            for (int q = 0; q < np; ++q) {
              const float e = __builtin_amdgcn_exp2f(ef_s[q * 32 + threadIdx.x] - mm);
              ef_s[q * 32 + threadIdx.x] = e;
              tot = fmaf(sp_s[q * 32 + threadIdx.x], e, tot);
            }
            sp_s[threadIdx.x] = tot;
          }
          __syncthreads();
#pragma unroll
          for (int r = 0; r < 16; ++r) sw[r] = sp_s[8 * (r >> 2) + 4 * kh + (r & 3)];
        }
        for (int64_t gg = g_first; gg <= g_last; ++gg) {
          if (gg < g_last) load_part(gg + 1, nxt);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (LOGITS) y[r] = fmaf(cur[r], ef_s[static_cast<int>(gg - g_first) * 32 + 8 * (r >> 2) + 4 * kh + (r & 3)], y[r]);
            else y[r] += cur[r];
          }
          cur = nxt;
        }
        write_out(y, sw);
        if (threadIdx.x == 0) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    c += i_end - i_begin;
  }
}

template <int NQ>
hipError_t launch_nq(bool cat, int ct, dim3 grid, size_t lds, hipStream_t s, const float* arena, const int64_t* row_off,
                     const float* w, float* out, int H, int B, int Ki, int Ko) {
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, arena, row_off, w, out, H, B, Ki, Ko);
    return hipGetLastError();
  };
  if (ct == 3) return cat ? go(sum_lse_gemm_kernel<NQ, true, 3>) : go(sum_lse_gemm_kernel<NQ, false, 3>);
  if (ct == 6) return cat ? go(sum_lse_gemm_kernel<NQ, true, 6>) : go(sum_lse_gemm_kernel<NQ, false, 6>);
  return cat ? go(sum_lse_gemm_kernel<NQ, true>) : go(sum_lse_gemm_kernel<NQ, false>);
}

}  // namespace

namespace ck {

bool gemm_applies(int H, int Ki, int Ko, int mode) {
  const bool cat = mode == CK_SUM_CAT && H > 1;
  const int n = cat ? H * Ki : Ki;
  if (!((mode == CK_SUM_CAT || mode == CK_SUM_PROD) && Ki % 32 == 0 && Ko % 32 == 0)) return false;
  return (n >= 32 && n <= 256) || (n > 256 && n <= 512 && n % 64 == 0) || n == 768 || n == 1024;
}

bool tucker_applies(int H, int Ki, int Ko, int mode) { return mode == CK_SUM_KRON && H == 2 && (Ki == 32 || Ki == 64); }

// Tucker layer of arity 2 with 32 or 64 units per child.
int tucker_lse(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int B, int Ki, int Ko,
               void* stream, bool logits, int contraction) {
  const int tiles = (B + 31) / 32, nblocks = (Ko + 31) / 32;
  // Many workgroups: two blocks of outputs share every e_l * e_r product.  Fewer: one block per workgroup, so that a
  // fold's work is spread over Ko / 32 workgroups.  Fewer than the chip has CUs: the waves of a workgroup split the
  // contraction instead of the rows (SPLIT).
  const int64_t wg1 = static_cast<int64_t>(F) * ((tiles + 3) / 4) * nblocks;  // workgroups at one block each
  // Few workgroups per resident slot: the flat chunk list dealt evenly to persistent workgroups (tucker_streamk_kernel),
  // if the caller lent a workspace for the partial tiles
  {
    const int slots = ck::num_cus() * 3;  // 48 KiB of LDS per workgroup
    // (the bf16 variants: two workgroups per CU with four chunks each in flight)
    const int resident = ck::num_cus() * (contraction != 0 ? 2 : 3);
    const ck::Workspace ws = ck::workspace();
    const int rgroups = (tiles + 3) / 4;
    const int64_t total = wg1 * Ki;
    // (at least 8 chunks per workgroup: each pays the exponentials of its tile's children once; measured 4 / 8 / 16 / 32:
    // 22 / 22 / 30 / 48 us for the layers of 2..12 folds, no difference for the large ones)
    const int64_t G = std::max<int64_t>(1, std::min<int64_t>(total / 8, resident));
    // workspace layout (the same for every launch that shares it):
    // [slots x 2 partial tiles of 16 KiB][slots x 2 x (32 maxima, 32 sums)][ticket per tile]
    const int64_t slot_bytes = static_cast<int64_t>(slots) * 2 * 4 * 1024 * static_cast<int64_t>(sizeof(float));
    const int64_t stat_bytes = static_cast<int64_t>(slots) * 2 * 64 * static_cast<int64_t>(sizeof(float));
    const int64_t need = slot_bytes + stat_bytes + wg1 * 4;
    // (the bf16 variants exist in this launch only and take it at any size: a chunk costs them a third to a half of the exact
    // chain, the exponentials of a logits launch repeated per 128 rows included)
    if ((wg1 <= 8 * static_cast<int64_t>(slots) || contraction != 0) && ws.ptr != nullptr && ws.bytes >= need && !ck::debug_force_generic()) {
      StreamKArgs a{};
      a.arena = arena;
      a.row_off = row_off;
      a.w = w;
      a.out = out;
      a.ws = static_cast<float*>(ws.ptr);
      a.stats = reinterpret_cast<float*>(static_cast<char*>(ws.ptr) + slot_bytes);
      a.tickets = reinterpret_cast<uint32_t*>(static_cast<char*>(ws.ptr) + slot_bytes + stat_bytes);
      a.F = F;
      a.B = B;
      a.Ko = Ko;
      a.nblk = nblocks;
      a.rgroups = rgroups;
      a.total = total;
      return ck::dispatch(
          [=](hipStream_t s) {
            auto go = [&](auto kern, size_t lds) {
              hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
              if (e != hipSuccess) return e;
              hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(G)), dim3(256), lds, s, a);
              return hipGetLastError();
            };
            constexpr size_t extra = 2 * 32 + 2 * 4 + 2 * 32;  // (alpha_s, flag_s, stat_s)
            const int nk = Ki / 32, chunk = contraction == 0 ? nk * 1024 : (contraction / 3 + 1) * nk * 512;
            const size_t lds = (2 * chunk + 4 * Ki * 32 + extra) * sizeof(float);
            if (contraction == 3) {
              if (logits) return Ki == 32 ? go(tucker_streamk_kernel<1, true, 3, 4, 2>, lds) : go(tucker_streamk_kernel<2, true, 3, 4, 2>, lds);
              return Ki == 32 ? go(tucker_streamk_kernel<1, false, 3, 4, 2>, lds) : go(tucker_streamk_kernel<2, false, 3, 4, 2>, lds);
            }
            if (contraction == 6) {
              if (logits) return Ki == 32 ? go(tucker_streamk_kernel<1, true, 6, 4, 2>, lds) : go(tucker_streamk_kernel<2, true, 6, 4, 2>, lds);
              return Ki == 32 ? go(tucker_streamk_kernel<1, false, 6, 4, 2>, lds) : go(tucker_streamk_kernel<2, false, 6, 4, 2>, lds);
            }
            if (logits) return Ki == 32 ? go(tucker_streamk_kernel<1, true>, lds) : go(tucker_streamk_kernel<2, true>, lds);
            return Ki == 32 ? go(tucker_streamk_kernel<1, false>, lds) : go(tucker_streamk_kernel<2, false>, lds);
          },
          stream);
    }
  }
  if (contraction != 0)
    return ck::fail(CK_ERR_UNSUPPORTED, "Tucker launch, contraction variant %d: %lld tiles need the stream-K launch (a workspace and at most %d tiles)",
                    contraction, static_cast<long long>(wg1), 8 * ck::num_cus() * 3);
  if (logits)  // (one workgroup per tile would apply the exponential once per 128 rows: the caller normalises first)
    return ck::fail(CK_ERR_UNSUPPORTED, "Tucker launch on logits: %lld tiles need the stream-K launch (a workspace and at most %d tiles)",
                    static_cast<long long>(wg1), 8 * ck::num_cus() * 3);
  const bool two = nblocks % 2 == 0 && wg1 > 4096;
  const bool split = wg1 <= 128;
  const int gx = split ? tiles : (tiles + 3) / 4;
  return ck::dispatch(
      [=](hipStream_t s) {
        if (two) return Ki == 32 ? launch_tucker<1, 2, false>(s, arena, row_off, w, out, F, B, Ko, gx, 1)
                                 : launch_tucker<2, 2, false>(s, arena, row_off, w, out, F, B, Ko, gx, 1);
        if (split) return Ki == 32 ? launch_tucker<1, 1, true>(s, arena, row_off, w, out, F, B, Ko, gx, nblocks)
                                   : launch_tucker<2, 1, true>(s, arena, row_off, w, out, F, B, Ko, gx, nblocks);
        return Ki == 32 ? launch_tucker<1, 1, false>(s, arena, row_off, w, out, F, B, Ko, gx, nblocks)
                        : launch_tucker<2, 1, false>(s, arena, row_off, w, out, F, B, Ko, gx, nblocks);
      },
      stream);
}

// Real dense / CP-T layer with Ki, Ko multiples of 32 and at most 1024 contracted inputs.
int sum_lse_gemm(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int H, int B, int Ki,
                 int Ko, int mode, void* stream, int contraction) {
  int ct = contraction;
  const bool cat = mode == CK_SUM_CAT && H > 1;
  const int nq = (cat ? H * Ki : Ki) / 32;
  if (nq > 8) {  // the inputs of a row tile are split over 2 or 4 waves
    // (the variants: two-wave splits whose pieces fit in LDS; the four-wave splits -- 768, 1024 inputs, one wave per SIMD at 256
    // registers -- measured slower on pieces than in exact fp32 and stay exact)
    if (ct != 0) {
      const int nqw = nq / 2, qs = nqw % 2 == 0 ? nqw / 2 : nqw;
      const size_t need = (static_cast<size_t>(2) * 2 * qs * (ct == 6 ? 1536 : 1024) + 4 * 16 * 64) * sizeof(float);
      if (nq > 16 || need > 160 * 1024) ct = 0;
    }
    return ck::dispatch(
        [=](hipStream_t s) {
          switch (nq) {
            case 10: return launch_split<5, 2>(cat, ct, s, arena, row_off, w, out, F, H, B, Ki, Ko);
            case 12: return launch_split<6, 2>(cat, ct, s, arena, row_off, w, out, F, H, B, Ki, Ko);
            case 14: return launch_split<7, 2>(cat, ct, s, arena, row_off, w, out, F, H, B, Ki, Ko);
            case 16: return launch_split<8, 2>(cat, ct, s, arena, row_off, w, out, F, H, B, Ki, Ko);
            case 24: return launch_split<6, 4>(cat, ct, s, arena, row_off, w, out, F, H, B, Ki, Ko);
            default: return launch_split<8, 4>(cat, ct, s, arena, row_off, w, out, F, H, B, Ki, Ko);
          }
        },
        stream);
  }
  const size_t lds = static_cast<size_t>(2) * nq * (ct == 6 ? 1536 : 1024) * sizeof(float);
  const int tiles = (B + 31) / 32;
  dim3 grid((tiles + 3) / 4, F);
  return ck::dispatch(
      [=](hipStream_t s) {
        switch (nq) {
          case 1: return launch_nq<1>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 2: return launch_nq<2>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 3: return launch_nq<3>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 4: return launch_nq<4>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 5: return launch_nq<5>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 6: return launch_nq<6>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          case 7: return launch_nq<7>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
          default: return launch_nq<8>(cat, ct, grid, lds, s, arena, row_off, w, out, H, B, Ki, Ko);
        }
      },
      stream);
}

}  // namespace ck
