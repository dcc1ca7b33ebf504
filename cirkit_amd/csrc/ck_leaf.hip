// Persistent form of the fused leaf region (ck_fused.hip, linear-domain variant): ONE workgroup per CU that stays
// resident and walks a list of (root fold, range of 32-row batch tiles) segments.
//
//     linear Categorical table (dense layer already pushed through it)  ->  D levels of CP-T layers
//
// Same arithmetic, in the same order, as subtree_linear_kernel -- the outputs are bit-identical
// (tests/test_gpu_parity.py) -- but organised for the machine instead of for the launch grid:
//
//  * the 2^D - 1 weight matrices of a root's subtree (60 KB at D = 4) travel global -> LDS ONCE per segment
//    (global_load_lds_dwordx4, no staging registers) and are shared by all waves of the CU; the grid-per-tile
//    kernel staged them once per 128 rows (1568 times instead of 256 at the north-star config);
//  * the waves of the workgroup draw tiles from an LDS counter, so a CU's SIMDs finish within one tile of each
//    other whatever the number of tiles per CU (24.5 at batch 4096 on 256 CUs);
//  * the leaf rows are gathered global -> LDS by the DMA path with EIGHT lanes per 128-byte table row (one
//    wave instruction = 8 full rows = 8 cache lines; the register gather touched 32 lines per instruction,
//    32 bytes of each) into a two-slot ring per wave, XOR-swizzled so that the ds_read_b128 that brings a row
//    into the MFMA operand layout is bank-conflict free; no prefetch registers, 3 waves per SIMD fit.
//
// Reference semantics per step (unchanged): TorchCategoricalLayer input.py:399-412, TorchSumLayer
// inner.py:266-273, TorchCPTLayer optimized.py:171-178, LSESumSemiring.apply_reduce semiring.py:383-408.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "ck_internal.h"
#include "ck_softmax.h"
#include "ck_tile.h"
#include "ck_tile16.h"
#include "ck_tailwalk.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxDepthP = 4;

struct LeafArgs {
  const float* table;    // (F0, C+1, 32) rows in LINEAR space (kind-5 prologue job)
  const float* scale;    // (F0, C+1) log scale of each table row
  const int32_t* xt;     // (Dvars, B) staged batch
  const int64_t* scope;  // (F_in) variable of each input-layer fold
  const float* w[kMaxDepthP];  // w[l-1]: (F_l, 1024 dwords) CK_W_TILED_F32 weights of CP-T level l
  const int32_t* nodes;        // packed node tables (as ck_subtree_cat_cpt_fwd)
  int node_off[kMaxDepthP + 1];
  int leaf_off;
  float* out;           // (F_root, B, 32)
  const int32_t* work;  // (n_seg, 4): root fold, first tile, end tile, 0
  int n_seg, B, C;
  int preclamped;  // xt holds -1 .. C - 1 only
  int32_t* redo;   // SIGNED: (F_root, tiles) flags of the tiles to evaluate again in log space (leaf_signed_redo_kernel)
  int w_rowmajor;  // the level weights are row-major (F_l, 32, 32) matrices instead of CK_W_TILED_F32
  // XRAW: the batch as the caller holds it, (B, D) int64 row-major -- no staging launch in front of this one
  const int64_t* x64;
  int D;
  int32_t* bad_flag;  // XRAW: raised (atomicOr 1) when a row holds an illegal value; nullptr = rows are not checked
  int x_pairs;        // XRAW: leaves 2j, 2j + 1 of every root read adjacent, 16-byte aligned variables (one load per pair)
  const int32_t* root_tab;  // nullptr, or (F_root, 3 * 2^D): per root the variable and the table fold of every leaf and the folds
                            // of its 2^D - 1 nodes in step order -- what the start of a segment otherwise collects from `nodes`,
                            // `scope` and the level tables in three dependent rounds of loads
  // TAIL: the trailing few-fold levels (ck_tail16.hip's walk) inside this launch -- see leaf_tail_phase
  const TailFold* tail_folds;    // (tail_n_folds) in level order
  const int32_t* tail_level_begin;  // (tail_n_levels + 1)
  int tail_n_folds, tail_n_levels;
  int tail_write;                // the 32-unit fold outputs of the tail are layer outputs somebody reads: store them too
  int tail_w_rowmajor;           // layout of the 32-output tail weights (else CK_W_TILED_F32)
  const int32_t* tail_bad_input; // staged batch: ck_stage_categories' sticky flag -> NaN circuit outputs (nullptr: none)
  double* ll;                    // nullptr, or [sum_b log p, B]
  double* ll_partial;            // (ceil(B / 16))
  unsigned int* ll_ticket;
  unsigned long long* arrive;    // monotonic arrival counter of the launches of this binding
  unsigned int* tail_state;      // (ceil(B / 16)) epoch in which each 16-row tile was last claimed
  // PARAMS: the launch evaluates the parameters it reads -- see leaf_params_phase
  const float* cat_logits;       // (F_cat, 32, C) logits of the Categorical layer
  const int64_t* cat_idx;        // (F0) Categorical fold of each table (dense) fold, or nullptr: the identity
  const float* dense_logits;     // (F0, 32, 32) logits of the dense layer pushed through the table
  const float* wraw[kMaxDepthP]; // wraw[l-1]: (F_l, 32, 32) logits of the weights of CP-T level l
  const int32_t* groot_off;      // (9): roots whose tables the workgroups b with b % 8 == g build are groot[groot_off[g] .. groot_off[g + 1])
  const int32_t* groot;
  unsigned long long* parrive;   // 8 arrival counters, 16 words apart
  const ck_rows32_job* xjobs;    // other 32-wide softmaxes (weights of the layers behind this launch)
  int n_xjobs;
};

// ---- the tail of the circuit inside the leaf launch -------------------------------------------------------------------
// The trailing few-fold levels (24, 11, 6, 4, 2, 1 folds at the north-star configuration) are a chain of tiny dependent
// steps; as a launch of their own (ck_tail16.hip) they cost a launch boundary, the start-up of 256 new workgroups, and
// a round trip through HBM for the roots -- 22 us behind a 70 us leaf launch.  Here the resident workgroups of the leaf
// launch walk them after their segments:
//   * root tiles are stored write-through (tile_store_wt), so a workgroup's roots are at the memory side when its
//     stores have completed; it then arrives on a monotonic counter (one 8-byte agent-scope atomic per workgroup);
//   * a workgroup waits until every workgroup of the launch has arrived (the roots of a row come from every workgroup),
//     then claims 16-row tiles of the batch -- its own first (tile = workgroup index), by an epoch compare-and-swap -- and
//     walks each exactly as tail16_kernel does: fold outputs of the tail stay in LDS (the leaf walk's 156 KB are free),
//     children produced by the leaf walk are read past the caches (load4_wt), the log-likelihood sum is folded in;
//   * NOTHING depends on all workgroups being resident at once: a workgroup that has waited 200 us -- another launch holds
//     compute units this one needs -- leaves without claiming; whoever passes the wait later finds its tiles unclaimed in
//     the sweep that follows its own tiles and walks them.
// Same arithmetic per fold as tail16_kernel (the fold -> wave assignment differs, the values do not).
constexpr unsigned long long kTailTimeoutTicks = 20000;  // wall_clock64 runs at 100 MHz: 200 us

template <int WAVES>
__device__ __forceinline__ void leaf_tail_phase(const LeafArgs& a, float* slots, int n_main, float* wbuf) {
  const int n_tiles = (a.B + 15) >> 4;
  const TailTiles tiles{slots, wbuf + kTailCtlFloats, n_main};
  TailFold* s_fold = reinterpret_cast<TailFold*>(wbuf);
  int32_t* s_level = reinterpret_cast<int32_t*>(s_fold + a.tail_n_folds);
  unsigned int* s_ctl = reinterpret_cast<unsigned int*>(s_level + a.tail_n_levels + 1);
  // this workgroup's roots: once the write-through stores have completed they are at the memory side
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // (every wave has left the walk: its LDS is free)
  if (threadIdx.x == 0) {
    const unsigned long long n = gridDim.x;
    const unsigned long long old = __hip_atomic_fetch_add(a.arrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ctl[0] = static_cast<unsigned int>(old / n + 1);  // the epoch of this launch
    s_ctl[2] = static_cast<unsigned int>(n - 1 - old % n);  // arrivals still missing (as of this one)
  }
  {  // fold descriptors and level table -> LDS while the others arrive
    const int n16 = a.tail_n_folds * static_cast<int>(sizeof(TailFold) / 16);
    const int4* src = reinterpret_cast<const int4*>(a.tail_folds);
    int4* dst = reinterpret_cast<int4*>(s_fold);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i <= a.tail_n_levels; i += blockDim.x) s_level[i] = a.tail_level_begin[i];
  }
  __syncthreads();
  const unsigned int epoch = s_ctl[0];
  if (threadIdx.x == 0) {
    const unsigned long long target = static_cast<unsigned long long>(epoch) * gridDim.x;
    const unsigned long long t0 = wall_clock64();
    unsigned int ok = 1;
    if (s_ctl[2] != 0)
      while (__hip_atomic_load(a.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > kTailTimeoutTicks) {
          ok = 0;
          break;
        }
      }
    s_ctl[1] = ok;
  }
  __syncthreads();
  if (s_ctl[1] == 0) return;  // gave up waiting: the tiles of this workgroup are left to the sweep of those that pass
  const bool poison = a.tail_bad_input != nullptr && *a.tail_bad_input != 0;
  TailWalkArgs wa{};
  wa.B = a.B;
  wa.n_levels = a.tail_n_levels;
  wa.n_folds = a.tail_n_folds;
  wa.w_rowmajor = a.tail_w_rowmajor;
  wa.write = a.tail_write;
  wa.ll = a.ll;
  wa.ll_partial = a.ll_partial;
  wa.ll_ticket = a.ll_ticket;
  auto claim = [&](int tile) -> bool {  // exactly one workgroup of the launch walks a tile
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int expected = epoch - 1;
      s_ctl[3] = __hip_atomic_compare_exchange_strong(a.tail_state + tile, &expected, epoch, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    __syncthreads();
    return s_ctl[3] != 0;
  };
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    if (claim(tile)) tail_walk<WAVES, true>(wa, tile, tiles, s_fold, s_level, poison, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), WorkgroupBarrier{});
  // sweep: tiles whose workgroup left without claiming them (never, unless launches compete for compute units)
  for (int base = 0; base < n_tiles; base += static_cast<int>(blockDim.x)) {
    const int tl = base + static_cast<int>(threadIdx.x);
    const int open = tl < n_tiles && __hip_atomic_load(a.tail_state + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch - 1;
    if (__syncthreads_or(open)) {
      const int end = min(n_tiles, base + static_cast<int>(blockDim.x));
      for (int t2 = base; t2 < end; ++t2)
        if (claim(t2)) tail_walk<WAVES, true>(wa, t2, tiles, s_fold, s_level, poison, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), WorkgroupBarrier{});
    }
  }
}

// ---- the parameters of the launch, evaluated by the launch ------------------------------------------------------------
// The reference re-evaluates every parameter graph on every forward (parameters/parameter.py:180-188); as a launch of its
// own (ck_param_softmax_batch) that is 21 us in front of a 70 us leaf launch at the north-star configuration -- 784
// table jobs that all load, then all compute, then all store -- although it is ~6 us of work per compute unit.  PARAMS:
//   * the 8 waves of a workgroup run TWO table jobs at a time (ck_softmax.h: the Categorical log-table of a leaf pushed
//     through its dense fold, 4 waves each, tiles in the gather slots that the walk does not need yet); the workgroups
//     b with b % 8 == g (one XCD, as workgroups are placed today -- nothing depends on it) share out the tables of the
//     roots that any of them walks (groot), 3-4 jobs each; the table rows are stored write-through;
//   * they then meet on an arrival counter of their own (32 pollers on one line, not 256).  A workgroup that has waited
//     200 us builds every table of its class itself: the jobs are idempotent -- whoever runs them writes the same bits --
//     so no workgroup ever depends on another one being resident;
//   * the 2^D - 1 weight matrices of a segment's root are softmaxed from their logits straight into the LDS layout the
//     walk reads (no tiled copy in memory, no DMA), and the 32-wide softmaxes of the layers BEHIND this launch (xjobs:
//     the weights the tail launch reads) are dealt to the workgroups, one matrix each.
// Same functions, same arithmetic as the prologue launch: bit-identical tables, weights and outputs.
constexpr unsigned long long kParamsTimeoutTicks = 20000;  // wall_clock64 at 100 MHz: 200 us

template <int D, int WAVES>
__device__ __forceinline__ void leaf_params_phase(const LeafArgs& a, float* slots) {
  static_assert(WAVES == 8, "two table jobs of four waves each");
  constexpr int kLeaves = 1 << D;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wave >> 2, w4 = wave & 3, kh = lane >> 5;
  const int C = a.C;
  float* tile = slots + half * (32 * (C + 4) + 1024);
  float* const table_w = const_cast<float*>(a.table);
  float* const scale_w = const_cast<float*>(a.scale);
  const int grp = blockIdx.x & 7, rank = blockIdx.x >> 3;
  const int gsize = (static_cast<int>(gridDim.x) - grp + 7) >> 3;  // workgroups of this class
  const int r0 = a.groot_off[grp], n_jobs = (a.groot_off[grp + 1] - r0) * kLeaves;
  auto sync = [] { __syncthreads(); };
  auto run_jobs = [&](int first, int stride) {
    for (int j0 = first; j0 < n_jobs; j0 += stride) {
      const int j = j0 + half;
      const float *theta = nullptr, *theta_w = nullptr;
      int d = 0;
      if (j < n_jobs) {
        const int root = a.groot[r0 + j / kLeaves];
        d = a.nodes[a.node_off[0] + root * kLeaves + j % kLeaves];
        const int64_t f = a.cat_idx != nullptr ? a.cat_idx[d] : d;
        theta = a.cat_logits + f * 32 * C;
        theta_w = a.dense_logits + static_cast<int64_t>(d) * 1024;
      }
      const __amdgpu_buffer_rsrc_t rt = wt_buffer(table_w + static_cast<int64_t>(d) * (C + 1) * 32);
      const __amdgpu_buffer_rsrc_t rs = wt_buffer(scale_w + static_cast<int64_t>(d) * (C + 1));
      table_dense_rows<4, true>(theta, theta_w, C, tile, w4, lane, sync, [&](int c, const float (&v)[16], float m) {
        if (c <= C) {
          if (kh == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), rs, static_cast<uint32_t>(c) * 4u, 0, kAuxWriteThrough);
          const uint32_t off = static_cast<uint32_t>(c * 32 + 4 * kh) * 4u;
#pragma unroll
          for (int g = 0; g < 4; ++g) store4_wt(rt, off + 32 * g, v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        }
      });
      __syncthreads();  // (the tiles are free for the next pair of jobs)
    }
  };
  run_jobs(rank * 2, gsize * 2);
  // 32-wide softmaxes of other layers: one (rows <= 32, 32) block per workgroup turn, two rows per wave pass
  for (int x = blockIdx.x; x < a.n_xjobs; x += gridDim.x) {
    const ck_rows32_job xj = a.xjobs[x];
    softmax_rows32<2>(xj.in, xj.rows, wave, 8, lane, [&](int row, int l, float p) { xj.out[w32_index(row, l, xj.tiled != 0)] = p; });
  }
  // this workgroup's tables are at the memory side once its write-through stores have completed: arrive, wait for the class
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ unsigned int s_alone;
  if (threadIdx.x == 0) {
    unsigned long long* ctr = a.parrive + grp * 16;
    const unsigned long long n = static_cast<unsigned long long>(gsize);
    const unsigned long long old = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (old / n + 1) * n;
    const unsigned long long t0 = wall_clock64();
    unsigned int alone = 0;
    if (old + 1 < target)
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > kParamsTimeoutTicks) {
          alone = 1;
          break;
        }
      }
    s_alone = alone;
  }
  __syncthreads();
  if (s_alone != 0) {  // (never, unless launches compete for compute units) every table of the class, here
    run_jobs(0, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// XRAW: the categories are read from the caller's (B, D) int64 batch directly (a.x64), one tile ahead; the staging launch
// (25.7 MB read + 12.9 MB written + a launch boundary per forward at the north-star configuration) disappears.  A tile
// reads 4 x 32 bytes of each of its 32 batch rows when the root covers a 4 x 4 pixel block (QuadTree): whole sectors.
// TAIL: the trailing levels of the circuit are walked by this launch too (leaf_tail_phase above): root tiles are then
// stored write-through.
// PARAMS: tables and weights are evaluated by this launch from the raw parameters (leaf_params_phase above).
// XP: (XRAW) leaves 2j, 2j + 1 of every root read adjacent, 16-byte aligned variables (LeafArgs::x_pairs) -- a template
// parameter, not a branch: with two load sequences of different lengths behind a run-time test the compiler's wait-count
// pass gives up counting and waits for EVERYTHING in flight (vmcnt(0)) where the batch values are packed, in every tile.
template <int D, int WAVES, bool SIGNED, bool XRAW, bool TAIL = false, bool PARAMS = false, bool XP = false>
__global__ void __launch_bounds__(WAVES * 64) leaf_persistent_kernel(const LeafArgs a) {
  constexpr int kLeaves = 1 << D, kNodes = kLeaves - 1, kSlots = (WAVES == 8 && D >= 2) ? 3 : 2;
  // (two arrays, not one: with the gather slots at a constant offset inside a single array their addresses became
  // values in scalar registers -- 110 spilled instead of 36)
  __shared__ __attribute__((aligned(16))) float w_lds[kNodes * 1024];  // subtree weights, in step order
  // gathered leaf tiles, a ring of kSlots 4 KB slots per wave (a gather that misses the XCD's L2 -- six roots' tables,
  // 3.6 MB, are live per XCD -- takes ~1 us: with three slots a leaf is requested three leaves, ~1.5 contractions,
  // before it is read).  The slots are READ with inline-asm ds_read_b128: the compiler's
  // wait-count insertion cannot tell a read of one slot from the DMA in flight into the other and would put
  // s_waitcnt vmcnt(0) before every slot read -- i.e. wait for the gather that has just been issued.  The vmcnt /
  // lgkmcnt waits around the slot reads are therefore explicit.
  __shared__ __attribute__((aligned(16))) float g_lds[WAVES * kSlots * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // CK_LEAF_STAMPS (scripts/leaf_stamps.py builds a copy of the library with it; never defined in the product): shader-clock
  // stamps of one tile of every wave of one workgroup -- where the instruction stream of a wave is when.  Stamp 3i: leaf i
  // begins; 3i + 1: its rows are in registers; 3i + 2: the request of leaf i + 3 is out; 48: the root's chain is issued;
  // 49: the tile is stored.  The workgroup is a.n_xjobs, the tile of each wave its a.tail_write-th; buffer: a.redo.
#ifdef CK_LEAF_STAMPS
  constexpr bool kStamps = XP && !TAIL && !PARAMS && WAVES == 8;  // (the others have no LDS to spare)
  __shared__ long long s_stamps[kStamps ? WAVES : 1][kStamps ? 64 : 1];
  bool stamp_on = false;
#define CK_STAMP(id)                               \
  do {                                             \
    if (kStamps && stamp_on) {                     \
      const long long c_ = clock64();              \
      if (lane == 0) s_stamps[wave][(id)] = c_;    \
    }                                              \
  } while (0)
  // per-workgroup wall-clock stamps (100 MHz): 0 entry, 1 weights landed, 2 first segment walked, 3 exit -> buffer[512 + 4 wg + k]
#define CK_WG_STAMP(k)                                                                                                   \
  do {                                                                                                                   \
    if (kStamps && a.redo != nullptr && threadIdx.x == 0)                                                                \
      reinterpret_cast<long long*>(a.redo)[512 + 4 * blockIdx.x + (k)] = static_cast<long long>(wall_clock64());          \
  } while (0)
#else
#define CK_STAMP(id)
#define CK_WG_STAMP(k)
#endif
  CK_WG_STAMP(0);
  const int b_in = lane & 31, kh = lane >> 5;
  float* const my_slots = g_lds + wave * (kSlots * 1024);
  uint32_t rd_addr[4];  // LDS byte address of chunk 2g + kh of row b_in in slot 0 (slot s: + 4096 s)
#pragma unroll
  for (int g = 0; g < 4; ++g)
    rd_addr[g] = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(my_slots)) + (b_in * 8 + ((2 * g + kh) ^ ((b_in >> 1) & 7))) * 16;

  // table + byte offset of the chunk this lane fetches: the swizzle (r >> 1) & 7 of row r = 8 q + (lane >> 3) is
  // (4 q + (lane >> 4)) & 7, i.e. one value for even q and one for odd q
  uint32_t g_coff[2];  // (the table is smaller than 4 GB: checked on the host)
#pragma unroll
  for (int q = 0; q < 2; ++q) g_coff[q] = ((lane & 7) ^ ((4 * q + (lane >> 4)) & 7)) * 16;

  if constexpr (PARAMS) leaf_params_phase<D, WAVES>(a, g_lds);

  for (int seg = blockIdx.x; seg < a.n_seg; seg += gridDim.x) {
    const int t = a.work[4 * seg], tile_begin = a.work[4 * seg + 1], tile_end = a.work[4 * seg + 2];
    if (seg != static_cast<int>(blockIdx.x)) __syncthreads();  // every wave has left the previous segment
    // weights of the 2^D - 1 nodes, in the static order of the steps: 4 x 1 KiB wave-DMAs per node.  The nodes' fold indices
    // are ALL fetched before the first DMA is issued: read one by one in front of each node's DMAs they cost a memory round
    // trip per node -- vmcnt counts in order, so every wait for an index also waited for the DMAs issued before it: fifteen
    // round trips in a row at the start of every workgroup.
    // Per-root constants (scalar registers): variable row of xt / column of the raw batch and first table row of every leaf.
    // With `root_tab` they and the node folds are ONE round of loads from one 192-byte row; without it the variables take
    // two more (leaf id -> scope).  Everything is fetched before the first DMA: what starts a workgroup is the chain
    // constants -> batch values -> table rows, and the weights travel beside it.
    int node_fold[kNodes];
    const int32_t* leaf_ids = a.nodes + a.leaf_off + t * kLeaves;
    const int32_t* fold0 = a.nodes + a.node_off[0] + t * kLeaves;
    int64_t var_off[kLeaves];  // element offset of the leaf's variable: row of xt, or column of the raw batch
    int32_t row_base[kLeaves];
    if (a.root_tab != nullptr) {
      const int32_t* rt = a.root_tab + static_cast<int64_t>(t) * (3 * kLeaves);
#pragma unroll
      for (int i = 0; i < kLeaves; ++i) {
        var_off[i] = XRAW ? static_cast<int64_t>(rt[i]) : rt[i] * static_cast<int64_t>(a.B);
        row_base[i] = rt[kLeaves + i] * (a.C + 1);
      }
#pragma unroll
      for (int k = 0; k < kNodes; ++k) node_fold[k] = rt[2 * kLeaves + k];
    } else {
      static_for<0, kLeaves>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, steps_after(i)>([&](auto lc) {
          constexpr int l = decltype(lc)::value, k = steps_before(i) + l;
          node_fold[k] = a.nodes[a.node_off[l + 1] + t * (kLeaves >> (l + 1)) + (i >> (l + 1))];
        });
      });
#pragma unroll
      for (int i = 0; i < kLeaves; ++i) {
        var_off[i] = XRAW ? a.scope[leaf_ids[i]] : a.scope[leaf_ids[i]] * static_cast<int64_t>(a.B);
        row_base[i] = fold0[i] * (a.C + 1);
      }
    }
#pragma unroll
    for (int k = 0; k < kNodes; ++k) asm volatile("" : "+v"(node_fold[k]));  // (loaded here, not sunk to the uses)
    static_for<0, kLeaves>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<0, steps_after(i)>([&](auto lc) {
        constexpr int l = decltype(lc)::value, k = steps_before(i) + l;
        const int fold = __builtin_amdgcn_readfirstlane(node_fold[k]);
        if constexpr (PARAMS) {
          // softmax of the fold's (32, 32) logits straight into the walk's LDS layout: rows 2 * wave + half and + 16 of it
          softmax_rows32<2>(a.wraw[l] + static_cast<int64_t>(fold) * 1024, 32, wave, WAVES, lane,
                            [&](int row, int ll, float p) { w_lds[k * 1024 + w32_index(row, ll, true)] = p; });
          return;
        }
        // lane's 16 bytes of chunk q: tiled, dword 256 q + 4 lane; row-major, W[lane & 31][8 q + 4 (lane >> 5) ..] (ck_tile.h)
        const float* src = a.w[l] + static_cast<int64_t>(fold) * 1024 + (a.w_rowmajor ? (lane & 31) * 32 + 4 * (lane >> 5) : lane * 4);
        const int qstride = a.w_rowmajor ? 8 : 256;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((k * 4 + q) % WAVES == wave)
            __builtin_amdgcn_global_load_lds((ck::gptr_t)(src + q * qstride), (ck::lptr_t)(w_lds + k * 1024 + q * 256), 16, 0, 0);
      });
    });
    if (seg == static_cast<int>(blockIdx.x)) CK_WG_STAMP(1024 + 0);  // (the weights' DMAs are out: the root's row has arrived)
    // The tiles of a segment are dealt round-robin to the waves (wave w: tile_begin + w, + WAVES, ...), so a wave knows
    // its next tiles and fetches their inputs while it computes: the batch values of tile k + 2 and, from those of tile
    // k + 1, its table rows' scales and first two leaf rows are requested when the gathers of tile k are over (the
    // last 2^(D-1)... contractions of a tile have no memory waits).  A tile therefore starts without the chain
    // batch values -> scales -> leaf rows (three dependent memory round trips) in front of it.
    auto batch_row = [&](int tile) { return min(tile * 32 + b_in, a.B - 1); };
    // Batch values of one tile in registers.  Staged: v[i] = value of leaf i (both halves of the wave hold the same).
    // XRAW: lane (b, kh) holds the two dwords (v[2j], v[2j + 1]) = (low, high) of the int64 value of leaf 2j + kh -- one
    // 8-byte load per PAIR of leaves: half the cache lines of a dword load per leaf (the 32 rows of a tile are 32
    // different lines whatever is loaded from them, and for these launches line requests are what the gathers already
    // spend the L1's time on), and the high dwords needed to validate the value come with them.
    using RawT = int32_t;
    auto load_x = [&](int tile, RawT (&xv)[kLeaves]) {
      // (a uniform pointer + a 32-bit lane offset: no 64-bit lane arithmetic per load)
      if constexpr (XRAW) {
        // (what these loads cost is their 32 different cache lines per instruction: with every lane of a tile pointed at
        // one cached line instead, the launch takes 71.3 instead of 76.5 us -- the batch is 5 us of this launch)
        const uint32_t rowb = static_cast<uint32_t>(batch_row(tile)) * (static_cast<uint32_t>(a.D) * 8u);
        if constexpr (kLeaves >= 4) {
          if constexpr (XP) {
            // leaves 2j and 2j + 1 read ADJACENT variables of the batch (16-byte aligned: checked on the host) -- what a
            // region graph over an image gives -- so lane (b, kh) takes the two values of leaves 4m + 2kh, 4m + 2kh + 1
            // with ONE 16-byte load: xv[4m .. 4m + 3] = (low, high, low, high); half the line requests again
#pragma unroll
            for (int m = 0; m < kLeaves / 4; ++m) {
              uint32_t off = rowb + (kh ? static_cast<uint32_t>(var_off[4 * m + 2]) : static_cast<uint32_t>(var_off[4 * m])) * 8u;
              asm volatile("" : "+v"(off));
              // (plain loads: a line of the batch is used by up to four roots of the XCD; fetched non-temporal -- so as
              // not to push the roots' tables out of the L2 -- the launch took 83.7 instead of 77.6 us)
              const int4 t = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(a.x64) + off);
              xv[4 * m] = t.x;
              xv[4 * m + 1] = t.y;
              xv[4 * m + 2] = t.z;
              xv[4 * m + 3] = t.w;
            }
            return;
          }
        }
#pragma unroll
        for (int j = 0; j < kLeaves / 2; ++j) {
          uint32_t off = rowb + (kh ? static_cast<uint32_t>(var_off[2 * j + 1]) : static_cast<uint32_t>(var_off[2 * j])) * 8u;
          asm volatile("" : "+v"(off));
          const int2 t = *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(a.x64) + off);
          xv[2 * j] = t.x;
          xv[2 * j + 1] = t.y;
        }
      } else {
        uint32_t boff = static_cast<uint32_t>(batch_row(tile)) * 4u;
        asm volatile("" : "+v"(boff));
#pragma unroll
        for (int i = 0; i < kLeaves; ++i)
          xv[i] = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.xt + var_off[i]) + boff);
      }
    };
    // categories of a tile as table rows, two per register (C < 65535 is checked on the host): negative = marginalised
    // -> the integral row C of the table.  XRAW: returns whether the lane's batch row holds a value that is not a legal
    // input -- a category >= C (TorchCategoricalLayer's advanced indexing raises IndexError there, layers/input.py:399-412)
    // or anything that does not fit 32 bits; such a value selects the integral row (memory-safe) and the row's root tile is
    // written as NaN below (the launches that consume it carry the NaN to the circuit output of that row).
    auto pack_categories = [&](const RawT (&xv)[kLeaves], uint32_t (&cp)[kLeaves / 2]) -> uint32_t {
      const uint32_t uc = static_cast<uint32_t>(a.C);
      uint32_t bad = 0;
      if constexpr (XRAW && kLeaves >= 4) {
        if constexpr (XP) {  // (see load_x) lane (b, kh) holds leaves 4m + 2kh and 4m + 2kh + 1: the packed pair 2m + kh
#pragma unroll
          for (int m = 0; m < kLeaves / 4; ++m) {
            const int32_t lo0 = xv[4 * m], hi0 = xv[4 * m + 1], lo1 = xv[4 * m + 2], hi1 = xv[4 * m + 3];
            bad |= static_cast<uint32_t>((hi0 != (lo0 >> 31)) | (lo0 >= a.C) | (hi1 != (lo1 >> 31)) | (lo1 >= a.C));
            const uint32_t mine = min(static_cast<uint32_t>(lo0), uc) | (min(static_cast<uint32_t>(lo1), uc) << 16);
            const auto r = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
            cp[2 * m] = r[0];
            cp[2 * m + 1] = r[1];
          }
          const auto rb = __builtin_amdgcn_permlane32_swap(bad, bad, false, false);
          return a.bad_flag != nullptr ? (rb[0] | rb[1]) : 0u;
        }
      }
#pragma unroll
      for (int j = 0; j < kLeaves / 2; ++j) {
        if constexpr (XRAW) {
          const int32_t lo = xv[2 * j], hi = xv[2 * j + 1];
          bad |= static_cast<uint32_t>((hi != (lo >> 31)) | (lo >= a.C));
          // (v_permlane32_swap of a value with itself: r[0] = what the lanes (b, 0) hold, r[1] = what the lanes (b, 1) hold)
          const uint32_t mine = min(static_cast<uint32_t>(lo), uc);
          const auto r = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
          cp[j] = r[0] | (r[1] << 16);
        } else {
          auto row = [&](int32_t v) -> uint32_t {
            // preclamped: -1 .. C - 1 (ck_stage_categories): -1 = 0xffffffff -> the integral row C (uniform branch)
            if (a.preclamped) return min(static_cast<uint32_t>(v), uc);
            return v < 0 ? uc : static_cast<uint32_t>(min(v, a.C - 1));
          };
          cp[j] = row(xv[2 * j]) | (row(xv[2 * j + 1]) << 16);
        }
      }
      if constexpr (XRAW) {
        const auto r = __builtin_amdgcn_permlane32_swap(bad, bad, false, false);
        bad = a.bad_flag != nullptr ? (r[0] | r[1]) : 0u;  // (no flag: the caller asked for no validation)
      }
      return bad;
    };
    auto row_of = [&](const uint32_t (&cp)[kLeaves / 2], auto ic) -> int32_t {  // table row of leaf ic.value for batch row b_in
      constexpr int i = decltype(ic)::value;
      return row_base[i] + static_cast<int32_t>((i & 1) ? cp[i >> 1] >> 16 : cp[i >> 1] & 0xffffu);
    };
    // leaf -> slot: lane (r8 = lane >> 3, c8 = lane & 7) of DMA q fetches 16-byte chunk c8 ^ swz(r) of row r = 8q + r8.
    // The address is the (uniform) table pointer plus a 32-bit lane offset (row index << 7) + chunk offset: one VALU
    // instruction per DMA (global_load_lds with an SGPR base).
    auto dma = [&](int32_t rowv, int slot) {
      // the four row indices of this lane with ONE wait (the compiler pairs them: two LDS round trips per leaf)
      uint32_t ridx[4];
      const uint32_t bp = 4 * (lane >> 3);
      asm volatile(
          "ds_bpermute_b32 %0, %4, %5\n\tds_bpermute_b32 %1, %4, %5 offset:32\n\tds_bpermute_b32 %2, %4, %5 offset:64\n\t"
          "ds_bpermute_b32 %3, %4, %5 offset:96\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(ridx[0]), "=&v"(ridx[1]), "=&v"(ridx[2]), "=&v"(ridx[3])
          : "v"(bp), "v"(rowv)
          : "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const char* src = reinterpret_cast<const char*>(a.table) + static_cast<uint32_t>((ridx[q] << 7) + g_coff[q & 1]);
        __builtin_amdgcn_global_load_lds((ck::gptr_t)src, (ck::lptr_t)(my_slots + slot * 1024 + q * 256), 16, 0, 0);
      }
    };
    // one request = the leaf rows (4 wave DMAs) and the log scale (1 load) of a leaf: 5 vector-memory operations
    float sld[4];  // scales of the leaves in flight (ring; leaf i -> sld[i & 3])
    auto request = [&](const uint32_t (&cp)[kLeaves / 2], auto ic) {
      constexpr int i = decltype(ic)::value;
      const int32_t r = row_of(cp, ic);
      dma(r, i % kSlots);
      uint32_t soff = static_cast<uint32_t>(r) << 2;  // (uniform pointer + 32-bit lane offset, as the gathers)
      asm volatile("" : "+v"(soff));
      sld[i & 3] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.scale) + soff);
    };
    // PIPE (three slots, depth >= 3).  The stamps of scripts/leaf_stamps.py show what two waves on a SIMD do to each other: a
    // VALU instruction of one is not issued while the other is inside a chain of contractions (16, 32, ... dependent MFMAs:
    // 1024 cycles each) -- and a wave issues in order, so everything behind that instruction waits with it.  A request as
    // written above is row index (VALU) -> ds_bpermute -> address (VALU) -> DMA: the wave that has just finished its chains
    // stood at the first of these for the whole chain of its neighbour, 14.5 k of the 44 k cycles of a tile, with its
    // gathers and slot reads behind it.  Here the leaves are walked in PAIRS and the instructions of a pair are ordered
    //   [slot reads of both leaves, the DMAs of the two leaves three ahead, the ds_bpermutes of the two leaves five ahead]
    //   [VALU: their addresses, the row indices of the two leaves seven ahead, the products]   [the chains]
    // so that between a wave's chains and its next VALU block there is only LDS and vector-memory work, which proceeds
    // while the neighbour computes.  State between pairs: addresses (4 + 4 registers), scale offsets, two row indices.
    constexpr bool kPipe = kSlots == 3 && D >= 3;
    uint32_t p_addr[2][4], p_soff[2], p_ridx[2][4];
    int32_t p_row[2];
    const uint32_t bperm = 4 * (lane >> 3);
    auto pipe_perm = [&](int h) {  // row indices of the lanes whose rows this lane fetches (results: p_ridx[h], in flight)
#pragma unroll
      for (int q = 0; q < 4; ++q) p_ridx[h][q] = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(bperm + 32 * q), p_row[h]));
    };
    auto pipe_addr = [&](int h) {  // VALU: addresses of the four DMAs and of the scale load from the exchanged row indices
#pragma unroll
      for (int q = 0; q < 4; ++q) p_addr[h][q] = (p_ridx[h][q] << 7) + g_coff[q & 1];
      p_soff[h] = static_cast<uint32_t>(p_row[h]) << 2;
    };
    auto pipe_issue = [&](int h, auto ic) {  // the request of leaf i: 4 DMAs + the scale load, no other instruction
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const char* src = reinterpret_cast<const char*>(a.table) + p_addr[h][q];
        __builtin_amdgcn_global_load_lds((ck::gptr_t)src, (ck::lptr_t)(my_slots + (i % kSlots) * 1024 + q * 256), 16, 0, 0);
      }
      sld[i & 3] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.scale) + p_soff[h]);
    };
    // what a tile starts with: the requests of its first three leaves as above, and the pipeline state of pair 0
    auto start_tile = [&](const uint32_t (&cp)[kLeaves / 2]) {
      static_for<0, kSlots>([&](auto jc) { request(cp, jc); });
      if constexpr (kPipe) {
        p_row[0] = row_of(cp, std::integral_constant<int, 3>{});
        p_row[1] = row_of(cp, std::integral_constant<int, 4>{});
        pipe_perm(0);
        pipe_perm(1);
        pipe_addr(0);
        pipe_addr(1);
        p_row[0] = row_of(cp, std::integral_constant<int, 5>{});
        p_row[1] = row_of(cp, std::integral_constant<int, 6>{});
      }
    };
    // The tiles of a segment in chunks of 64 x WAVES (one chunk, normally): a wave notes the tiles whose products left the
    // linear range in a 64-bit mask (scalar registers) and evaluates them in log space AFTER its walk over the chunk.
    // Doing that inside the walk -- an out-of-line call with the whole walk state live -- cost the hot loop its
    // registers (256 + spills against 237; 79 -> 74.5 us at the north-star configuration).
    // Batch values are requested two tiles ahead (after the last gather request of a tile) and turned into packed table
    // rows half a tile later: `xraw` is live from the last contractions of a tile to the middle of the next one.
    RawT xraw[kLeaves];            // batch values of the tile AFTER the current one
    uint32_t cat[kLeaves / 2];     // packed table rows of the current tile
    uint32_t catnext[kLeaves / 2];  // ... of the next tile of this wave
    uint32_t bad_cur = 0, bad_next = 0;  // XRAW: the lane's row of the current / next tile holds an illegal value
    for (int chunk_begin = tile_begin; chunk_begin < tile_end; chunk_begin += 64 * WAVES) {
    const int chunk_end = min(tile_end, chunk_begin + 64 * WAVES);
    int tile = chunk_begin + wave;
    if (tile < chunk_end) {  // the first tile of the wave: the chain is paid once per chunk
      load_x(tile, xraw);
      if (chunk_begin == tile_begin && seg == static_cast<int>(blockIdx.x)) CK_WG_STAMP(1024 + 1);  // (batch values requested)
      bad_cur = pack_categories(xraw, cat);
      if (chunk_begin == tile_begin && seg == static_cast<int>(blockIdx.x)) CK_WG_STAMP(1024 + 2);  // (... arrived and packed)
      start_tile(cat);
      if (chunk_begin == tile_begin && seg == static_cast<int>(blockIdx.x)) CK_WG_STAMP(1024 + 3);  // (first rows requested)
    }
    if (chunk_begin == tile_begin) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the weights (and its first leaf rows) have landed
      __syncthreads();
      CK_WG_STAMP(1);
    }
    uint64_t bad_tiles = 0;
    int nth = 0;  // tile number `nth` of this wave in the chunk

    for (; tile < chunk_end; tile += WAVES, ++nth, bad_cur = bad_next) {
      const int b = tile * 32 + b_in;
      const bool live = b < a.B;
      float stack[D][16], sstack[D];
      float cur[16], cs = 0.f, sprev = 0.f;
      bool bad = false;
#ifdef CK_LEAF_STAMPS
      stamp_on = static_cast<int>(blockIdx.x) == a.n_xjobs && nth == a.tail_write;
#endif
      constexpr int kXLoads = XRAW ? (XP ? kLeaves / 4 : kLeaves / 2) : kLeaves;  // (exactly: XP is a template parameter)
      if constexpr (kPipe) {
        static_for<0, kLeaves / 2>([&](auto pc) {
          constexpr int e = 2 * decltype(pc)::value, o = e + 1;
          float s_e = 0.f, s_o = 0.f;
          WRegs wfirst;
          // ---- LDS and vector memory only
          static_for<0, 2>([&](auto hc) {
            constexpr int h = decltype(hc)::value, i = e + h;
            CK_STAMP(3 * i);
            // younger than the request of leaf i: those of the next two leaves and, between the requests of leaves 4 and 5
            // (the VALU block of leaf 1), the batch values of the wave's next tile
            constexpr int kYounger = 5 * (kLeaves - 1 - i < 2 ? kLeaves - 1 - i : 2) + (i >= 2 && i <= 4 ? kXLoads : 0);
            f32x4 r0, r1, r2, r3;
            if constexpr (h == 1) {  // (the weights of the pair's first contraction with the slot reads: one LDS round trip)
#pragma unroll
              for (int q = 0; q < 4; ++q) wfirst.q[q] = *reinterpret_cast<const float4*>(w_lds + steps_before(o) * 1024 + q * 256 + lane * 4);
            }
            asm volatile(
                "s_waitcnt vmcnt(%9)\n\tds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\t"
                "ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                : "v"(rd_addr[0]), "v"(rd_addr[1]), "v"(rd_addr[2]), "v"(rd_addr[3]), "n"((i % kSlots) * 4096), "n"(kYounger)
                : "memory");
            float(&dst)[16] = h == 0 ? stack[0] : cur;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              dst[k] = r0[k];
              dst[4 + k] = r1[k];
              dst[8 + k] = r2[k];
              dst[12 + k] = r3[k];
            }
            (h == 0 ? s_e : s_o) = sld[i & 3];
            CK_STAMP(3 * i + 1);
            if constexpr (i + kSlots < kLeaves) pipe_issue(h, std::integral_constant<int, i + kSlots>{});
            if constexpr (i + 5 < kLeaves) pipe_perm(h);
            CK_STAMP(3 * i + 2);
          });
          // ---- VALU
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (e == 0) load_x(min(tile + WAVES, chunk_end - 1), xraw);
          if constexpr (e == kLeaves / 2) bad_next = pack_categories(xraw, catnext);
          cs = s_o + s_e;  // log scale of the pair
          tile_mul(cur, stack[0]);  // first level: the bare product (ck_tile.h)
          static_for<0, 2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            if constexpr (e + h + 5 < kLeaves) pipe_addr(h);
            if constexpr (e + h + 7 < kLeaves) p_row[h] = row_of(cat, std::integral_constant<int, e + h + 7>{});
          });
          if constexpr (o + 1 == kLeaves) {
            // the gathers of this tile are over: request what the next tile starts with
            if (tile + WAVES < chunk_end) {
#pragma unroll
              for (int j = 0; j < kLeaves / 2; ++j) cat[j] = catnext[j];
              start_tile(cat);
            }
          }
          static_for<0, steps_after(o)>([&](auto lc) {
            constexpr int l = decltype(lc)::value, step = steps_before(o) + l;
            WRegs wcur;
            if constexpr (l == 0) {
              wcur = wfirst;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) wcur.q[q] = *reinterpret_cast<const float4*>(w_lds + step * 1024 + q * 256 + lane * 4);
              linear_product<true, SIGNED>(cur, stack[l], cs, sstack[l], bad);
            }
            __builtin_amdgcn_sched_barrier(0);
            contract_linear<CK_W_TILED_F32>(wcur, cur);
          });
          if constexpr (steps_after(o) < D) {  // left sibling at this level: wait for the right one
            constexpr int l = steps_after(o);
#pragma unroll
            for (int j = 0; j < 16; ++j) stack[l][j] = cur[j];
            sstack[l] = cs;
          }
        });
      } else
      static_for<0, kLeaves>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        CK_STAMP(3 * i);
        // the request of leaf i has landed: at most those of the next kSlots - 1 leaves, 5 operations each, are younger --
        // and, for leaves 1 .. kSlots - 1, the batch values of the wave's next tile, requested at leaf 0 and not needed before
        // leaf kLeaves / 2.  kXLoads is the FEWEST load instructions a tile's batch values take (a smaller count only waits
        // longer; the previous tile's output stores, younger than its successor's first requests, are not counted for the
        // same reason).  Read the slot into the operand layout; the reads have returned before the slot is refilled
        f32x4 r0, r1, r2, r3;
        constexpr int kYounger = 5 * (kLeaves - 1 - i < kSlots - 1 ? kLeaves - 1 - i : kSlots - 1) + (i >= 1 && i < kSlots ? kXLoads : 0);
        // (the weights of the leaf's first contraction are requested in front of the slot reads: one LDS round trip for both)
        WRegs wfirst;
        if constexpr (steps_after(i) > 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) wfirst.q[q] = *reinterpret_cast<const float4*>(w_lds + steps_before(i) * 1024 + q * 256 + lane * 4);
        }
        asm volatile(
            "s_waitcnt vmcnt(%9)\n\tds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\t"
            "ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "v"(rd_addr[0]), "v"(rd_addr[1]), "v"(rd_addr[2]), "v"(rd_addr[3]), "n"((i % kSlots) * 4096), "n"(kYounger)
            : "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cur[e] = r0[e];
          cur[4 + e] = r1[e];
          cur[8 + e] = r2[e];
          cur[12 + e] = r3[e];
        }
        const float s_i = sld[i & 3];
        CK_STAMP(3 * i + 1);
        // The batch values of this wave's NEXT tile: requested here, behind the first slot read, and packed half a tile
        // later IN THE SAME ITERATION -- requested at the end of the previous tile (one more tile of lead) the compiler's
        // wait-count pass loses count across the loop's back edge and puts s_waitcnt vmcnt(0) in front of the packing: every
        // tile then waited for the leaf rows requested a moment before.  (They sit between the requests of leaves
        // kSlots - 1 and kSlots: leaves 1 .. kSlots - 1 count them among the younger operations.)
        if constexpr (i == 0) load_x(min(tile + WAVES, chunk_end - 1), xraw);
        if constexpr (i == kLeaves / 2) bad_next = pack_categories(xraw, catnext);
        if constexpr (i + kSlots < kLeaves) request(cat, std::integral_constant<int, i + kSlots>{});
        if constexpr (i + 1 == kLeaves) {
          // the gathers of this tile are over: request what the next tile starts with (see above)
          if (tile + WAVES < chunk_end) {
#pragma unroll
            for (int j = 0; j < kLeaves / 2; ++j) cat[j] = catnext[j];
            static_for<0, kSlots>([&](auto jc) { request(cat, jc); });
          }
        }
        if constexpr ((i & 1) != 0) cs = s_i + sprev;  // log scale of the pair (i - 1, i)
        else sprev = s_i;
        CK_STAMP(3 * i + 2);
        static_for<0, steps_after(i)>([&](auto lc) {
          constexpr int l = decltype(lc)::value, step = steps_before(i) + l;
          WRegs wcur;
          if constexpr (l == 0) {
            wcur = wfirst;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) wcur.q[q] = *reinterpret_cast<const float4*>(w_lds + step * 1024 + q * 256 + lane * 4);
          }
          if constexpr (l == 0) {
            tile_mul(cur, stack[0]);  // first level: the bare product (ck_tile.h)
          } else {
            linear_product<true, SIGNED>(cur, stack[l], cs, sstack[l], bad);
          }
          // (the products stay in front of the MFMA chain: interleaved into it, the compiler's pre-emit pass splits every
          // v_pk_mul_f32 that follows an MFMA back into two multiplies)
          __builtin_amdgcn_sched_barrier(0);
          contract_linear<CK_W_TILED_F32>(wcur, cur);
        });
        if constexpr (steps_after(i) < D) {  // left sibling at this level: wait for the right one
          constexpr int l = steps_after(i);
#pragma unroll
          for (int j = 0; j < 16; ++j) stack[l][j] = cur[j];
          sstack[l] = cs;
        }
      });
      if constexpr (D == 1) bad |= !((SIGNED ? tile_row_max_abs(cur) : tile_row_max(cur)) > kLinearFloor);  // (deeper roots are renormalised steps)
      CK_STAMP(48);
      if (__builtin_expect(__any(bad), 0)) {
        // rare: evaluated again in log space -- below, after the walk; SIGNED: by leaf_signed_redo_kernel (the signed
        // log-space walk as a callee costs this kernel 67 spilled registers: 84 -> 88 us at config 5)
        if constexpr (SIGNED) {
          if (lane == 0) a.redo[static_cast<int64_t>(t) * ((a.B + 31) >> 5) + tile] = 1;
        } else {
          bad_tiles |= uint64_t{1} << nth;
        }
      } else if (live) {
        if constexpr (XRAW) {
          if (__builtin_expect(__any(bad_cur != 0), 0)) {  // an illegal input in some row of this tile: that row is NaN
            if (a.bad_flag != nullptr && lane == 0) atomicOr(a.bad_flag, 1);
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[j] = bad_cur != 0 ? __builtin_nanf("") : cur[j];
          }
        }
        if constexpr (SIGNED) {  // the complex logarithm of a real number: (log|v|, 0 or pi)
          uint32_t sg = 0;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            sg |= (cur[j] < 0.f ? 1u : 0u) << j;
            cur[j] = fmaf(__builtin_amdgcn_logf(__builtin_fabsf(cur[j])), kLN2, cs);
          }
          tile_store_clog(a.out + ((static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh) * 2, cur, sg);
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = fmaf(__builtin_amdgcn_logf(cur[j]), kLN2, cs);
          if constexpr (TAIL) tile_store_wt(a.out + static_cast<int64_t>(t) * a.B * kK, static_cast<uint32_t>(b * kK + 4 * kh) * 4u, cur);
          else tile_store(a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh, cur);
        }
      }
      CK_STAMP(49);
    }
    // the noted tiles: a row of products fell out of the fp32 range -> the whole tile in log space (semiring.py:383-408)
    while (bad_tiles != 0) {
      const int k = __builtin_ctzll(bad_tiles);
      bad_tiles &= bad_tiles - 1;
      const int btile = chunk_begin + wave + k * WAVES;
      const int b = btile * 32 + b_in;
      SubtreeSource src{};
      src.table = a.table;
      src.scale = a.scale;
      src.xt = a.xt;
      src.x64 = a.x64;
      src.D = a.D;
      src.scope = a.scope;
      src.leaf_ids = leaf_ids;
      src.fold0 = fold0;
      src.w_steps = w_lds;
      src.t = t;
      src.B = a.B;
      src.C = a.C;
      src.bl = min(b, a.B - 1);
      float fb[16];
      if constexpr (!SIGNED) {
        subtree_tile_logspace<D, CK_W_TILED_F32>(src, lane, fb);
        if constexpr (XRAW) {  // (rows with an illegal value are NaN here too)
          RawT xb[kLeaves];
          uint32_t cb[kLeaves / 2];
          load_x(btile, xb);
          if (pack_categories(xb, cb) != 0) {
            if (a.bad_flag != nullptr) atomicOr(a.bad_flag, 1);
#pragma unroll
            for (int j = 0; j < 16; ++j) fb[j] = __builtin_nanf("");
          }
        }
        if (b < a.B) {
          if constexpr (TAIL) tile_store_wt(a.out + static_cast<int64_t>(t) * a.B * kK, static_cast<uint32_t>(b * kK + 4 * kh) * 4u, fb);
          else tile_store(a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh, fb);
        }
      }
    }
    }  // chunk
    CK_WG_STAMP(2);
  }
  if constexpr (TAIL) {
    static_assert(!SIGNED, "the in-launch tail walks unsigned values");
    leaf_tail_phase<WAVES>(a, g_lds, WAVES * kSlots * 2, w_lds);
  }
#ifdef CK_LEAF_STAMPS
  __syncthreads();
  CK_WG_STAMP(3);
  if (kStamps && static_cast<int>(blockIdx.x) == a.n_xjobs && a.redo != nullptr)
    for (int i = threadIdx.x; i < WAVES * 64; i += blockDim.x) reinterpret_cast<long long*>(a.redo)[i] = s_stamps[i >> 6][i & 63];
#endif
#undef CK_STAMP
#undef CK_WG_STAMP
}

// The tiles a SIGNED launch marked (a row of products below the linear-space floor), in log space with signs
// (tile_walk_logspace_signed): one wave per (root, tile); unmarked tiles exit at once.
template <int D>
__global__ void __launch_bounds__(64) leaf_signed_redo_kernel(const LeafArgs a) {
  const int tile = blockIdx.x, t = blockIdx.y;
  int32_t* flag = a.redo + static_cast<int64_t>(t) * gridDim.x + tile;
  if (*flag == 0) return;
  const int lane = threadIdx.x, b = tile * 32 + (lane & 31), kh = lane >> 5;
  SubtreeSource src{};
  src.table = a.table;
  src.scale = a.scale;
  src.xt = a.xt;
  src.x64 = a.x64;
  src.D = a.D;
  src.scope = a.scope;
  src.leaf_ids = a.nodes + a.leaf_off + t * (1 << D);
  src.fold0 = a.nodes + a.node_off[0] + t * (1 << D);
  src.w_steps = nullptr;
  for (int l = 0; l < D; ++l) src.w[l] = a.w[l];
  src.nodes = a.nodes;
  for (int l = 0; l <= D; ++l) src.node_off[l] = a.node_off[l];
  src.t = t;
  src.B = a.B;
  src.C = a.C;
  src.bl = min(b, a.B - 1);
  float v[16];
  uint32_t sg = 0;
  if (a.w_rowmajor) subtree_tile_logspace<D, CK_W_ROWMAJOR, true>(src, lane, v, &sg);
  else subtree_tile_logspace<D, CK_W_TILED_F32, true>(src, lane, v, &sg);
  if (b < a.B) tile_store_clog(a.out + ((static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh) * 2, v, sg);
  if (lane == 0) *flag = 0;  // ready for the next replay
}

template <int D, bool XRAW>
hipError_t launch_waves(const LeafArgs& a, int waves, bool is_signed, int n_roots, dim3 grid, hipStream_t s) {
  if (a.cat_logits != nullptr) {  // (checked by the caller: unsigned, 8 waves, no in-launch tail)
    hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, XRAW, false, true>), grid, dim3(512), 0, s, a);
    return hipGetLastError();
  }
  if (a.tail_folds != nullptr) {  // (checked by the caller: unsigned, 8 waves)
    hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, XRAW, true>), grid, dim3(512), 0, s, a);
    return hipGetLastError();
  }
  if (is_signed) {
    hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, true, XRAW>), grid, dim3(512), 0, s, a);
    if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
    hipLaunchKernelGGL((leaf_signed_redo_kernel<D>), dim3((a.B + 31) / 32, n_roots), dim3(64), 0, s, a);
  } else if (waves == 12) {
    if constexpr (XRAW) return hipErrorInvalidValue;  // (checked by the caller: 12 waves read the staged batch only)
    else hipLaunchKernelGGL((leaf_persistent_kernel<D, 12, false, false>), grid, dim3(768), 0, s, a);
  } else {
    if constexpr (XRAW && D >= 2) {
      if (a.x_pairs) {
        hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, false, false, true>), grid, dim3(512), 0, s, a);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, XRAW>), grid, dim3(512), 0, s, a);
  }
  return hipGetLastError();
}

template <bool XRAW>
hipError_t launch_depth(const LeafArgs& a, int depth, int waves, bool is_signed, int n_roots, dim3 grid, hipStream_t s) {
  switch (depth) {
    case 1:
      return launch_waves<1, XRAW>(a, waves, is_signed, n_roots, grid, s);
    case 2:
      return launch_waves<2, XRAW>(a, waves, is_signed, n_roots, grid, s);
    case 3:
      return launch_waves<3, XRAW>(a, waves, is_signed, n_roots, grid, s);
    default:
      return launch_waves<4, XRAW>(a, waves, is_signed, n_roots, grid, s);
  }
}

}  // namespace

extern "C" {

int ck_leaf_walk_fwd(const ck_leaf_launch* d, void* stream) {
  CK_REQUIRE(d != nullptr, "ck_leaf_walk_fwd: null descriptor");
  const bool raw = d->x_rows != nullptr || d->x_input >= 0;
  CK_REQUIRE(d->table && d->table_scale && d->scope && d->w_levels && d->nodes && d->node_off && d->out && d->work,
             "ck_leaf_walk_fwd: null pointer");
  CK_REQUIRE(raw != (d->xt != nullptr), "ck_leaf_walk_fwd: give either the staged batch xt or the raw batch (x_rows / x_input)");
  CK_REQUIRE(d->depth >= 1 && d->depth <= kMaxDepthP, "ck_leaf_walk_fwd: depth %d outside [1, %d]", d->depth, kMaxDepthP);
  CK_REQUIRE(d->n_seg > 0 && d->n_wg > 0 && d->B > 0 && d->C > 0, "ck_leaf_walk_fwd: non-positive size");
  CK_REQUIRE(d->waves == 8 || d->waves == 12, "ck_leaf_walk_fwd: waves must be 8 or 12 (got %d)", d->waves);
  if (d->K != kK) return ck::fail(CK_ERR_UNSUPPORTED, "ck_leaf_walk_fwd: K=%d (only K=32 is fused)", d->K);
  CK_REQUIRE(ck::aligned16(d->table) && ck::aligned16(d->out), "ck_leaf_walk_fwd: buffers must be 16-byte aligned");
  LeafArgs a{};
  a.table = d->table;
  a.scale = d->table_scale;
  a.xt = d->xt;
  a.scope = d->scope;
  for (int l = 0; l < d->depth; ++l) {
    CK_REQUIRE(d->cat_logits != nullptr || (d->w_levels[l] != nullptr && ck::aligned16(d->w_levels[l])), "ck_leaf_walk_fwd: bad weights of level %d", l + 1);
    a.w[l] = d->w_levels[l];
  }
  a.nodes = d->nodes;
  a.root_tab = d->root_tab;
  for (int l = 0; l <= d->depth; ++l) a.node_off[l] = d->node_off[l];
  a.leaf_off = d->leaf_off;
  a.out = d->out;
  a.work = d->work;
  a.n_seg = d->n_seg;
  a.B = d->B;
  a.C = d->C;
  a.preclamped = d->preclamped;
  CK_REQUIRE(d->w_layout == CK_W_TILED_F32 || d->w_layout == CK_W_ROWMAJOR, "ck_leaf_walk_fwd: weights must be CK_W_TILED_F32 or row-major");
  CK_REQUIRE(d->signed_redo == nullptr || (d->n_roots > 0 && d->n_roots <= 65535), "ck_leaf_walk_fwd: signed launch needs 0 < n_roots <= 65535");
  a.redo = d->signed_redo;
  a.w_rowmajor = d->w_layout == CK_W_ROWMAJOR ? 1 : 0;
  const void* const* slot = nullptr;
  if (raw) {
    CK_REQUIRE(d->waves == 8, "ck_leaf_walk_fwd: the raw batch is read by 8-wave workgroups only");
    CK_REQUIRE(d->D > 0 && static_cast<int64_t>(d->B) * d->D * 8 < (int64_t{1} << 32),
               "ck_leaf_walk_fwd: raw batch of B=%d x D=%d int64 values exceeds 32-bit byte offsets (stage it instead)", d->B, d->D);
    CK_REQUIRE(d->C < 65535, "ck_leaf_walk_fwd: C=%d categories do not fit the packed rows", d->C);
    a.D = d->D;
    a.bad_flag = d->bad_input;
    a.x_pairs = d->x_pairs != 0 && (d->D & 1) == 0;
    a.x64 = d->x_rows;
    if (d->x_input >= 0) {
      slot = ck::program_input_slot(d->x_input);
      CK_REQUIRE(slot != nullptr, "ck_leaf_walk_fwd: x_input=%d names a program input, but no program is being recorded on this "
                                  "thread (or the index is out of range)", d->x_input);
    }
  }
  if (d->tail_folds != nullptr) {
    CK_REQUIRE(d->waves == 8 && d->signed_redo == nullptr, "ck_leaf_walk_fwd: the in-launch tail needs 8 waves and unsigned values");
    CK_REQUIRE(d->tail_level_begin && d->tail_arrive && d->tail_state, "ck_leaf_walk_fwd: tail needs level_begin, arrive and state");
    CK_REQUIRE(d->tail_n_folds > 0 && d->tail_n_levels > 0 && d->tail_n_levels <= 15, "ck_leaf_walk_fwd: bad tail sizes");
    CK_REQUIRE(static_cast<int64_t>(d->B) * kK * 4 < (int64_t{1} << 31), "ck_leaf_walk_fwd: B=%d rows exceed the 32-bit offsets of the in-launch tail", d->B);
    CK_REQUIRE(ck::aligned16(d->tail_folds), "ck_leaf_walk_fwd: tail_folds not 16-byte aligned");
    CK_REQUIRE(d->ll == nullptr || (d->ll_partial != nullptr && d->ll_ticket != nullptr), "ck_leaf_walk_fwd: ll needs ll_partial and ll_ticket");
    CK_REQUIRE(d->tail_w_layout == CK_W_TILED_F32 || d->tail_w_layout == CK_W_ROWMAJOR, "ck_leaf_walk_fwd: tail weights must be CK_W_TILED_F32 or row-major");
    // LDS of the tail phase (leaf_tail_phase): fold tiles in the gather slots and behind the control block of the weight array
    CK_REQUIRE(d->depth >= 2, "ck_leaf_walk_fwd: the in-launch tail needs a fused depth of at least 2 (its control block lives in the weight array)");
    const size_t ctl = static_cast<size_t>(d->tail_n_folds) * sizeof(TailFold) + (d->tail_n_levels + 1) * sizeof(int32_t) + 16;
    const int cap = 8 * 3 * 2 + (((1 << d->depth) - 1) * 4096 - kTailCtlFloats * 4) / 2048;
    if (ctl > kTailCtlFloats * sizeof(float) || d->tail_n_folds > cap)
      return ck::fail(CK_ERR_UNSUPPORTED, "ck_leaf_walk_fwd: %d tail folds do not fit the launch's LDS (at most %d)", d->tail_n_folds, cap);
    a.tail_folds = reinterpret_cast<const TailFold*>(d->tail_folds);
    a.tail_level_begin = d->tail_level_begin;
    a.tail_n_folds = d->tail_n_folds;
    a.tail_n_levels = d->tail_n_levels;
    a.tail_write = d->tail_write;
    a.tail_w_rowmajor = d->tail_w_layout == CK_W_ROWMAJOR ? 1 : 0;
    a.tail_bad_input = d->tail_bad_input;
    a.ll = d->ll;
    a.ll_partial = d->ll_partial;
    a.ll_ticket = d->ll_ticket;
    a.arrive = reinterpret_cast<unsigned long long*>(d->tail_arrive);
    a.tail_state = d->tail_state;
  }
  if (d->cat_logits != nullptr) {
    CK_REQUIRE(d->waves == 8 && d->signed_redo == nullptr && d->tail_folds == nullptr,
               "ck_leaf_walk_fwd: parameters are evaluated in-launch by 8-wave, unsigned launches without a tail");
    CK_REQUIRE(d->dense_logits && d->w_logits && d->groot_off && d->groot && d->params_arrive, "ck_leaf_walk_fwd: in-launch parameters need "
               "dense_logits, w_logits, groot_off, groot and params_arrive");
    CK_REQUIRE(d->w_layout == CK_W_TILED_F32, "ck_leaf_walk_fwd: in-launch weights are written in the tiled layout");
    CK_REQUIRE(d->C <= 256 && (d->C & 3) == 0 && d->depth >= 2, "ck_leaf_walk_fwd: in-launch tables need C <= 256, C %% 4 == 0 and depth >= 2 "
               "(two 32 x (C + 4) tiles in the gather slots)");
    CK_REQUIRE(d->n_xjobs == 0 || d->xjobs != nullptr, "ck_leaf_walk_fwd: xjobs is null");
    a.cat_logits = d->cat_logits;
    a.cat_idx = d->cat_idx;
    a.dense_logits = d->dense_logits;
    for (int l = 0; l < d->depth; ++l) {
      CK_REQUIRE(d->w_logits[l] != nullptr, "ck_leaf_walk_fwd: no logits for the weights of level %d", l + 1);
      a.wraw[l] = d->w_logits[l];
    }
    a.groot_off = d->groot_off;
    a.groot = d->groot;
    a.parrive = reinterpret_cast<unsigned long long*>(d->params_arrive);
    a.xjobs = d->xjobs;
    a.n_xjobs = d->n_xjobs;
  }
#ifdef CK_LEAF_STAMPS
  if (d->cat_logits == nullptr && d->signed_redo == nullptr && getenv("CK_STAMP_PTR") != nullptr) {
    a.redo = reinterpret_cast<int32_t*>(strtoull(getenv("CK_STAMP_PTR"), nullptr, 0));
    a.n_xjobs = getenv("CK_STAMP_WG") ? atoi(getenv("CK_STAMP_WG")) : 0;
    a.tail_write = getenv("CK_STAMP_TILE") ? atoi(getenv("CK_STAMP_TILE")) : 1;
  }
#endif
  const int depth = d->depth, waves = d->waves, n_roots = d->n_roots;
  const bool is_signed = d->signed_redo != nullptr;
  dim3 grid(static_cast<unsigned>(std::min(d->n_wg, d->n_seg)));
  return ck::dispatch(
      [=](hipStream_t s) {
        if (!raw) return launch_depth<false>(a, depth, waves, is_signed, n_roots, grid, s);
        LeafArgs b = a;
        if (slot != nullptr) b.x64 = static_cast<const int64_t*>(*slot);  // the batch of THIS replay (ck_program_set_input)
        if (b.x64 == nullptr) return hipErrorInvalidValue;
        return launch_depth<true>(b, depth, waves, is_signed, n_roots, grid, s);
      },
      stream);
}

int ck_leaf_persistent_fwd(const float* table, const float* table_scale, const int32_t* xt, const int64_t* scope,
                           const float* const* w_levels, const int32_t* nodes, const int32_t* node_off, int leaf_off,
                           float* out, const int32_t* work, int n_seg, int n_wg, int waves, int depth, int B, int K,
                           int C, int preclamped, int w_layout, int32_t* signed_redo, int n_roots, void* stream) {
  ck_leaf_launch d{};
  d.table = table;
  d.table_scale = table_scale;
  d.xt = xt;
  d.scope = scope;
  d.w_levels = w_levels;
  d.nodes = nodes;
  d.node_off = node_off;
  d.leaf_off = leaf_off;
  d.out = out;
  d.work = work;
  d.n_seg = n_seg;
  d.n_wg = n_wg;
  d.waves = waves;
  d.depth = depth;
  d.B = B;
  d.K = K;
  d.C = C;
  d.preclamped = preclamped;
  d.w_layout = w_layout;
  d.signed_redo = signed_redo;
  d.n_roots = n_roots;
  d.x_rows = nullptr;
  d.x_input = -1;
  return ck_leaf_walk_fwd(&d, stream);
}

}  // extern "C"
