// The tail of forward k and the parameters of forward k + 1 in ONE launch.
//
// A forward of the north-star circuit is three launches of which two are latency-bound: the parameter prologue (21 us:
// 784 table jobs that load, compute and store in step; ~5 us of latency + 62 MB) and the 16-row tail (19 us: six dependent
// levels).  Neither fills a compute unit, they use different resources, and they do not depend on each other if the
// parameters are evaluated for the NEXT forward: the reference re-evaluates its parameter graphs once per forward
// (parameters/parameter.py:180-188) -- WHEN inside the forward is immaterial as long as nobody changes the parameters in
// between, and that the host knows (TensorStore.data_version; a changed store re-evaluates at the start of the forward).
// So the launch that ends forward k carries, beside the tail's workgroups, the workgroups of the prologue:
//
//   blocks [0, n_tail)                 one 16-row tile of the batch each through the tail's levels (ck_tailwalk.h: the walk
//                                      of ck_tail16.hip on 8 waves, fold tiles in LDS slots assigned by the host -- a slot
//                                      is reused once its fold has been read for the last time --, log-likelihood sum
//                                      folded in);
//   blocks [n_tail, n_tail + n_pair)   two table jobs each (ck_softmax.h: the Categorical log-table of a fold pushed
//                                      through its dense fold, 4 waves per job, both jobs' logits requested up front);
//   the rest                           32-wide softmaxes (weights of the sum layers), kRowsPerBlock matrices per block.
//
// 512 threads and <= 80 KB of LDS per block: two blocks per compute unit, so a tail block and a parameter block share a CU
// and the parameter blocks' loads, MFMAs and stores fill the tail's waits.  Same device functions as the launches this one
// replaces: bit-identical tables, weights, layer outputs and log-likelihood sum.
//
// Measured (MI355X, north-star shape, instrumented pass): 31 us, against 16.8 + 21.8 us for the two launches it replaced
// -- 29 us against 16.7 + 17.9 us once the table jobs had lost half their instructions (ck_softmax.h); the tail part alone
// 20 us (six levels: 10.2 / 4.3 / 1.8 / 2.1 / 1.8 / 2.1 us by the cycle counter), the parameter part alone 21 -> 18 us.  What limits the overlap is instruction issue, not memory: a
// table job is ~3 us of dependent VALU/MFMA/DPP issue on its busiest wave, three jobs per compute unit, and with the tail
// block holding half of a CU's LDS and registers only two jobs per CU are resident.  Tried and dropped: ONE 16-wave block per
// CU (waves 0-7 the tail, waves 8-15 two table-job groups looping over their jobs, LDS-counter barriers among the waves
// concerned) -- 36 us: the counter barrier costs 0.3 us where s_barrier costs nothing, and two resident jobs per CU take
// 28 us for the parameter part alone; requesting a group's next job's logits before it computes the current one changed
// nothing (29 us), which is how the "latency" reading of the prologue was ruled out.
//
// The parameter part REWRITES buffers the tail part reads (the tail layers' weights).  It writes what they already hold --
// the same raw parameters through the same deterministic arithmetic (the host evaluates the parameters on their own before
// a forward whose store has changed) -- so a concurrent reader sees the one value either way.
#include <algorithm>

#include "ck_internal.h"
#include "ck_softmax.h"
#include "ck_tailwalk.h"

namespace {

constexpr int kRowsPerBlock = 4;  // 32 x 32 softmaxes per block of the last kind

struct TailParamsArgs {
  // tail
  const TailFold* folds;
  const int32_t* level_begin;
  TailWalkArgs walk;
  const int32_t* bad_input;
  int n_slots;     // LDS slots (2 KB each) the fold tiles need
  int n_tail;      // blocks of the first kind = ceil(B / 16)
  // table jobs
  const float* cat_logits;
  const int64_t* cat_idx;
  const float* dense_logits;
  float* table;
  float* scale;
  int n_tables, C;
  int n_pair;      // blocks of the second kind = ceil(n_tables / 2)
  // 32-wide softmaxes
  const ck_rows32_job* rows;
  int n_rows;
#ifdef CK_TAILP_STAMPS
  long long* stamps;  // scripts/tailp_stamps.py: 16 wall-clock stamps (100 MHz) per block
#endif
};

// CK_TAILP_STAMPS (a copy of the library built by scripts/tailp_stamps.py; never defined in the product): when each block of
// the launch enters and leaves, and when a tail block has finished each of its levels (ck_tailwalk.h, CK_TAIL_LEVEL_STAMP).
#ifdef CK_TAILP_STAMPS
#define CK_TP_STAMP(k)                                                                                          \
  do {                                                                                                          \
    if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[16 * blockIdx.x + (k)] = static_cast<long long>(wall_clock64()); \
  } while (0)
#else
#define CK_TP_STAMP(k)
#endif

__global__ void __launch_bounds__(512, 4) tail_params_kernel(const TailParamsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = blockIdx.x;
  CK_TP_STAMP(0);
  if (bid < a.n_tail) {
    // ---- one 16-row tile of the tail
    TailFold* s_fold = reinterpret_cast<TailFold*>(smem);
    int32_t* s_level = reinterpret_cast<int32_t*>(s_fold + a.walk.n_folds);
    float* tiles_base = smem + kTailCtlFloats;
    {
      const int n16 = a.walk.n_folds * static_cast<int>(sizeof(TailFold) / 16);
      const int4* src = reinterpret_cast<const int4*>(a.folds);
      int4* dst = reinterpret_cast<int4*>(s_fold);
      for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
      for (int i = threadIdx.x; i <= a.walk.n_levels; i += blockDim.x) s_level[i] = a.level_begin[i];
    }
    __syncthreads();
    const bool poison = a.bad_input != nullptr && *a.bad_input != 0;
    const TailTiles tiles{tiles_base, tiles_base, a.n_slots};
#ifdef CK_TAILP_STAMPS
    g_tail_level_stamps = a.stamps != nullptr ? a.stamps + 16 * blockIdx.x + 2 : nullptr;
#endif
    CK_TP_STAMP(1);
    tail_walk<8, false>(a.walk, bid, tiles, s_fold, s_level, poison, wave, WorkgroupBarrier{});
    CK_TP_STAMP(15);
    return;
  }
  const int v = bid - a.n_tail;
  if (v < a.n_pair) {
    // ---- two table jobs, four waves each
    const int half = wave >> 2, w4 = wave & 3, kh = lane >> 5;
    const int C = a.C;
    float* tile = smem + half * (32 * (C + 4) + 1024);
    const int d = 2 * v + half;
    const float *theta = nullptr, *theta_w = nullptr;
    if (d < a.n_tables) {
      const int64_t f = a.cat_idx != nullptr ? a.cat_idx[d] : d;
      theta = a.cat_logits + f * 32 * C;
      theta_w = a.dense_logits + static_cast<int64_t>(d) * 1024;
    }
    float* dst = a.table + static_cast<int64_t>(d) * (C + 1) * 32;
    float* dsc = a.scale + static_cast<int64_t>(d) * (C + 1);
    table_dense_rows<4, true>(theta, theta_w, C, tile, w4, lane, [] { __syncthreads(); }, [&](int c, const float (&val)[16], float m) {
      if (c <= C && kh == 0) dsc[c] = m;
      if (c <= C) tile_store(dst + static_cast<int64_t>(c) * 32 + 4 * kh, val);
    });
    CK_TP_STAMP(15);
    return;
  }
  // ---- 32-wide softmaxes: kRowsPerBlock (rows <= 32, 32) blocks, two rows of each per wave pass.  ALL their logits are
  // requested before the first is evaluated: one after the other they are sixteen memory round trips in a row, 16 us for a
  // block that enters at 12 us -- these blocks, not the tail, were what the launch ended on (scripts/tailp_stamps.py).
  // (descriptors first, then every block's logits, then the arithmetic: three memory round trips per workgroup)
  const int x0 = (v - a.n_pair) * kRowsPerBlock;
  const int n = min(kRowsPerBlock, a.n_rows - x0);
  ck_rows32_job xj[kRowsPerBlock];
#pragma unroll
  for (int i = 0; i < kRowsPerBlock; ++i) xj[i] = a.rows[x0 + min(i, n - 1)];
  float xr[kRowsPerBlock][2];
#pragma unroll
  for (int i = 0; i < kRowsPerBlock; ++i) softmax_rows32_load<2>(xj[i].in, xj[i].rows, wave, 8, lane, xr[i]);
#pragma unroll
  for (int i = 0; i < kRowsPerBlock; ++i) {
    if (i < n) {
      float* const out = xj[i].out;
      const bool tiled = xj[i].tiled != 0;
      softmax_rows32_apply<2>(xr[i], xj[i].rows, wave, 8, lane, [&](int row, int l, float p) { ck::as_global(out)[w32_index(row, l, tiled)] = p; });
    }
  }
  CK_TP_STAMP(15);
}

}  // namespace

extern "C" int ck_tail_params_fwd(const ck_tail_params_launch* d, void* stream) {
  CK_REQUIRE(d != nullptr, "ck_tail_params_fwd: null descriptor");
  CK_REQUIRE(d->folds && d->level_begin && d->n_folds > 0 && d->n_levels > 0 && d->n_levels <= 15 && d->B > 0,
             "ck_tail_params_fwd: bad tail description");
  CK_REQUIRE(ck::aligned16(d->folds), "ck_tail_params_fwd: folds not 16-byte aligned");
  CK_REQUIRE(d->ll == nullptr || (d->ll_partial != nullptr && d->ll_ticket != nullptr), "ck_tail_params_fwd: ll needs ll_partial and ll_ticket");
  CK_REQUIRE(d->w_layout == CK_W_TILED_F32 || d->w_layout == CK_W_ROWMAJOR, "ck_tail_params_fwd: tail weights must be CK_W_TILED_F32 or row-major");
  CK_REQUIRE(d->n_tables == 0 || (d->cat_logits && d->dense_logits && d->table && d->table_scale), "ck_tail_params_fwd: table jobs need "
             "cat_logits, dense_logits, table and table_scale");
  CK_REQUIRE(d->n_tables == 0 || (d->C > 0 && d->C <= 256 && (d->C & 3) == 0), "ck_tail_params_fwd: table jobs need C <= 256, C %% 4 == 0");
  CK_REQUIRE(d->n_rows == 0 || d->rows != nullptr, "ck_tail_params_fwd: rows is null");
  const size_t ctl = static_cast<size_t>(d->n_folds) * sizeof(TailFold) + (d->n_levels + 1) * sizeof(int32_t);
  const size_t lds_tail = kTailCtlFloats * sizeof(float) + static_cast<size_t>(d->n_slots) * 2048;
  const size_t lds_pair = d->n_tables > 0 ? static_cast<size_t>(2) * (32 * (d->C + 4) + 1024) * sizeof(float) : 0;
  const size_t lds = std::max(lds_tail, lds_pair);
  if (ctl > kTailCtlFloats * sizeof(float) || d->n_slots <= 0 || lds > 80 * 1024)
    return ck::fail(CK_ERR_UNSUPPORTED, "ck_tail_params_fwd: %d folds in %d slots do not fit two blocks per compute unit", d->n_folds, d->n_slots);
  TailParamsArgs a{};  // (`a0` below: the copy the recorded launch keeps)
  a.folds = reinterpret_cast<const TailFold*>(d->folds);
  a.level_begin = d->level_begin;
  a.walk.B = d->B;
  a.walk.n_levels = d->n_levels;
  a.walk.n_folds = d->n_folds;
  a.walk.w_rowmajor = d->w_layout == CK_W_ROWMAJOR ? 1 : 0;
  a.walk.write = 1;
  a.walk.ll = d->ll;
  a.walk.ll_partial = d->ll_partial;
  a.walk.ll_ticket = d->ll_ticket;
  a.bad_input = d->bad_input;
  a.n_slots = d->n_slots;
  a.n_tail = (d->B + 15) / 16;
  a.cat_logits = d->cat_logits;
  a.cat_idx = d->cat_idx;
  a.dense_logits = d->dense_logits;
  a.table = d->table;
  a.scale = d->table_scale;
  a.n_tables = d->n_tables;
  a.C = d->C;
  a.n_pair = (d->n_tables + 1) / 2;
  a.rows = d->rows;
  a.n_rows = d->n_rows;
#ifdef CK_TAILP_STAMPS
  a.stamps = getenv("CK_STAMP_PTR") != nullptr ? reinterpret_cast<long long*>(strtoull(getenv("CK_STAMP_PTR"), nullptr, 0)) : nullptr;
#endif
  const int blocks = a.n_tail + a.n_pair + (d->n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
  const TailParamsArgs a0 = a;
  const void* const* ll_slot = nullptr;
  if (d->ll_cell != 0) {
    CK_REQUIRE(d->ll != nullptr, "ck_tail_params_fwd: ll_cell=%d without ll (the pair's home while the cell is NULL)", d->ll_cell);
    ll_slot = ck::program_input_slot(d->ll_cell);
    CK_REQUIRE(d->ll_cell > 0 && ll_slot != nullptr, "ck_tail_params_fwd: ll_cell=%d names a program input, but no program is being "
               "recorded on this thread (or the index is outside 1..%d)", d->ll_cell, ck::kProgramInputs - 1);
  }
  return ck::dispatch(
      [=](hipStream_t s) {
        TailParamsArgs a = a0;
        if (ll_slot != nullptr && *ll_slot != nullptr) a.walk.ll = static_cast<double*>(const_cast<void*>(*ll_slot));
        if (lds > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tail_params_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
          if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(tail_params_kernel, dim3(blocks), dim3(512), lds, s, a);
        return hipGetLastError();
      },
      stream);
}
