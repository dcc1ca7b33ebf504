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
#include "ck_tile.h"

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
  int32_t* redo;   // SIGNED: (F_root, tiles) flags of the tiles to evaluate again in log space (leaf_signed_redo_kernel); KEEP: see below
  int contraction;  // 0 exact fp32; 3 / 6: the bf16x3 / bf16x6 variants (depth 4, raw batch)
  int w_rowmajor;  // the level weights are row-major (F_l, 32, 32) matrices instead of CK_W_TILED_F32
  // XRAW: the batch as the caller holds it, (B, D) int64 row-major -- no staging launch in front of this one
  const int64_t* x64;
  int D;
  int32_t* bad_flag;  // XRAW: raised (atomicOr 1) when a row holds an illegal value; nullptr = rows are not checked
  int x_pairs;        // XRAW: leaves 2j, 2j + 1 of every root read adjacent, 16-byte aligned variables (one load per pair)
  // KEEP (training forward): the linear tile of the nodes of every SECOND level (l = 2, 4: kept_level) is stored as well -- what
  // the backward walk reads as the P of its two-level units; the levels in between it recomputes -- keep[l - 1]: (F_l, tiles, 1024) in tile-native
  // order (ck_tile.h tile_store_native; rows beyond B of the last tile hold whatever their lanes computed), the value the NEXT
  // level multiplies (the backward, ck_leaf_bwd.hip, needs a level's tiles to be consistent with each other, not their log
  // scales); tiles whose walk left the linear range are marked in `redo` (their kept tiles mean nothing)
  float* keep[kMaxDepthP];
#ifdef CK_LEAF_STAMPS
  int stamp_wg, stamp_tile;
#endif
  uint32_t* sout;  // SIGNED: nullptr (out is the (F_root, B, 32) complex64 block of (log|v|, 0 or pi)), or the sign words (F_root, Bp) of
                   // SIGNED-LOG blocks (ck_signed.hip): out is then (F_root, Bp, 32) fp32 log|v| in tile-native order, Bp = 32 ceil(B / 32)
  const int32_t* root_tab;  // nullptr, or (F_root, 3 * 2^D): per root the variable and the table fold of every leaf and the folds
                            // of its 2^D - 1 nodes in step order -- what the start of a segment otherwise collects from `nodes`,
                            // `scope` and the level tables in three dependent rounds of loads
};

// XRAW: the categories are read from the caller's (B, D) int64 batch directly (a.x64), one tile ahead; the staging launch
// (25.7 MB read + 12.9 MB written + a launch boundary per forward at the north-star configuration) disappears.  A tile
// reads 4 x 32 bytes of each of its 32 batch rows when the root covers a 4 x 4 pixel block (QuadTree): whole sectors.
// XP: (XRAW) leaves 2j, 2j + 1 of every root read adjacent, 16-byte aligned variables (LeafArgs::x_pairs) -- a template
// parameter, not a branch: with two load sequences of different lengths behind a run-time test the compiler's wait-count
// pass gives up counting and waits for EVERYTHING in flight (vmcnt(0)) where the batch values are packed, in every tile.
// KEEP: the training forward (LeafArgs::keep).  Its stores sit between the gathers of the walk: vector-memory operations of
// a wave complete in the order they were issued (loads and stores share vmcnt on gfx9: the compiler's own wait counts rely
// on it), so the explicit vmcnt(N) in front of a slot read counts the stores issued since that slot's request as well.
__host__ __device__ constexpr bool kept_level(int l) { return (l & 1) == 1; }  // step l = CP-T level l + 1: levels 2 and 4 are kept
__host__ __device__ constexpr int kept_steps(int n_steps) { return n_steps >> 1; }  // of the steps 0 .. n_steps - 1
__host__ __device__ constexpr int keep_stores_pipe(int i) {  // pair walk: between the request of leaf i >= 3 and its slot read
  int n = 0;
  for (int q = (i - 3) >> 1; q < (i >> 1); ++q) n += kept_steps(steps_after(2 * q + 1));
  return 4 * n;
}
__host__ __device__ constexpr int keep_stores_plain(int i, int slots) {  // leaf-by-leaf walk: leaf i >= slots
  int n = 0;
  for (int m = i - slots; m < i; ++m) n += kept_steps(steps_after(m));
  return 4 * n;
}
// A SIGNED tile as a signed-log block of ck_signed.hip: log|v| in tile-native order and ONE sign word per row (bit k: unit k is
// negative; a lane holds the bits of units 8 (j >> 2) + 4 kh + (j & 3), the two halves of a row are or-ed)
__device__ __forceinline__ void store_signed_log(const LeafArgs& a, int t, int tile, int b, int lane, const float (&v)[16], uint32_t sg) {
  const int kh = lane >> 5;
  const int64_t Bp = static_cast<int64_t>((a.B + 31) >> 5) * 32;
  tile_store_native(a.out + (static_cast<int64_t>(t) * Bp + static_cast<int64_t>(tile) * 32) * kK, lane, v);
  uint32_t so = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) so |= ((sg >> j) & 1u) << (8 * (j >> 2) + 4 * kh + (j & 3));
  so |= __shfl_xor(so, 32, 64);
  if (kh == 0 && b < a.B) a.sout[static_cast<int64_t>(t) * Bp + b] = so;
}

// CT: 0 = the contraction in exact fp32 (the product); 3 / 6 = the labelled bf16-split VARIANTS (ck_tile.h contract_bf16:
// "bf16x3" with two pieces per operand, "bf16x6" with three): the subtree weights are cut into pieces while they are staged
// (registers, not DMA), a node takes 4 / 6 KB of LDS -- with three pieces the gather ring has two slots per wave (the walk goes
// leaf by leaf) so that everything still fits 160 KB.
template <int D, int WAVES, bool SIGNED, bool XRAW, bool XP = false, bool KEEP = false, int CT = 0>
__global__ void __launch_bounds__(WAVES * 64) leaf_persistent_kernel(const LeafArgs a) {
  static_assert(!KEEP || WAVES == 8, "the training forward walks with 8 waves");
  static_assert(CT == 0 || (CT == 3 && !SIGNED && !KEEP) || (CT == 6 && !SIGNED && !KEEP), "bf16 variants: unsigned inference forward");
  constexpr int kLeaves = 1 << D, kNodes = kLeaves - 1, kSlots = (WAVES == 8 && D >= 2 && CT != 6) ? 3 : 2;
  constexpr int kWNode = CT == 6 ? 1536 : 1024;  // dwords of LDS per node
  // (two arrays, not one: with the gather slots at a constant offset inside a single array their addresses became
  // values in scalar registers -- 110 spilled instead of 36)
  __shared__ __attribute__((aligned(16))) float w_lds[kNodes * kWNode];  // subtree weights, in step order
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
  // 49: the tile is stored.  The workgroup is a.stamp_wg, the tile of each wave its a.stamp_tile-th; buffer: a.redo.
#ifdef CK_LEAF_STAMPS
  constexpr bool kStamps = XP && WAVES == 8;  // (the others have no LDS to spare)
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
    float* keep_base[KEEP ? kNodes : 1];  // KEEP: block of each node of this root in its level's (F_l, B, 32) array
    static_for<0, kLeaves>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<0, steps_after(i)>([&](auto lc) {
        constexpr int l = decltype(lc)::value, k = steps_before(i) + l;
        const int fold = __builtin_amdgcn_readfirstlane(node_fold[k]);
        if constexpr (KEEP && kept_level(l)) keep_base[k] = a.keep[l] + static_cast<int64_t>(fold) * ((a.B + 31) >> 5) * 1024;  // (F_l, tiles, 1024)
        if constexpr (CT != 0) {  // cut into bf16 pieces on the way (ck_tile.h bf16_piece_index)
          const float* wsrc = a.w[l] + static_cast<int64_t>(fold) * 1024;
          uint16_t* dst = reinterpret_cast<uint16_t*>(w_lds + k * kWNode);
          for (int idx = threadIdx.x; idx < 1024; idx += WAVES * 64) {
            int o, i;
            if (a.w_rowmajor) {
              o = idx >> 5;
              i = idx & 31;
            } else {  // CK_W_TILED_F32: dword (q, lane, t) = W[lane & 31][8q + 4 (lane >> 5) + t]
              const int ln = (idx >> 2) & 63;
              o = ln & 31;
              i = 8 * (idx >> 8) + 4 * (ln >> 5) + (idx & 3);
            }
            float rw = wsrc[idx];
#pragma unroll
            for (int p = 0; p < CT / 3 + 1; ++p) {
              const uint32_t bits = __float_as_uint(rw);
              dst[bf16_piece_index(p, o, i)] = static_cast<uint16_t>(bits >> 16);
              rw -= __uint_as_float(bits & 0xffff0000u);
            }
          }
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
      const int koff = tile * 1024;  // (KEEP) this tile's block in a node's (tiles, 1024) array
      float stack[D][16], sstack[D];
      float cur[16], cs = 0.f, sprev = 0.f;
      bool bad = false;
#ifdef CK_LEAF_STAMPS
      stamp_on = static_cast<int>(blockIdx.x) == a.stamp_wg && nth == a.stamp_tile;
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
            constexpr int kYounger = 5 * (kLeaves - 1 - i < 2 ? kLeaves - 1 - i : 2) + (i >= 2 && i <= 4 ? kXLoads : 0) +
                                     (KEEP && i >= 3 ? keep_stores_pipe(i) : 0);
            f32x4 r0, r1, r2, r3;
            if constexpr (h == 1 && CT == 0) {  // (the weights of the pair's first contraction with the slot reads: one LDS round trip)
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
              if constexpr (CT == 0) wcur = wfirst;
            } else {
              if constexpr (CT == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) wcur.q[q] = *reinterpret_cast<const float4*>(w_lds + step * 1024 + q * 256 + lane * 4);
              }
              linear_product<true, SIGNED>(cur, stack[l], cs, sstack[l], bad);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (CT == 0) contract_linear<CK_W_TILED_F32>(wcur, cur);
            else contract_bf16<CT / 3 + 1>(w_lds + step * kWNode, lane, cur);
            if constexpr (KEEP && kept_level(l)) tile_store_native(keep_base[step] + koff, lane, cur);
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
        constexpr int kYounger = 5 * (kLeaves - 1 - i < kSlots - 1 ? kLeaves - 1 - i : kSlots - 1) + (i >= 1 && i < kSlots ? kXLoads : 0) +
                                 (KEEP && i >= kSlots ? keep_stores_plain(i, kSlots) : 0);
        // (the weights of the leaf's first contraction are requested in front of the slot reads: one LDS round trip for both)
        WRegs wfirst;
        if constexpr (steps_after(i) > 0 && CT == 0) {
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
          if constexpr (CT == 0) {
            if constexpr (l == 0) {
              wcur = wfirst;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) wcur.q[q] = *reinterpret_cast<const float4*>(w_lds + step * 1024 + q * 256 + lane * 4);
            }
          }
          if constexpr (l == 0) {
            tile_mul(cur, stack[0]);  // first level: the bare product (ck_tile.h)
          } else {
            linear_product<true, SIGNED>(cur, stack[l], cs, sstack[l], bad);
          }
          // (the products stay in front of the MFMA chain: interleaved into it, the compiler's pre-emit pass splits every
          // v_pk_mul_f32 that follows an MFMA back into two multiplies)
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (CT == 0) contract_linear<CK_W_TILED_F32>(wcur, cur);
          else contract_bf16<CT / 3 + 1>(w_lds + step * kWNode, lane, cur);
          if constexpr (KEEP && kept_level(l)) tile_store_native(keep_base[step] + koff, lane, cur);
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
          if constexpr (KEEP) {
            if (lane == 0) a.redo[static_cast<int64_t>(t) * ((a.B + 31) >> 5) + tile] = 1;
          }
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
          if (a.sout != nullptr) store_signed_log(a, t, tile, b, lane, cur, sg);
          else tile_store_clog(a.out + ((static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh) * 2, cur, sg);
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = fmaf(__builtin_amdgcn_logf(cur[j]), kLN2, cs);
          tile_store(a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh, cur);
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
      src.w_steps = CT == 0 ? w_lds : nullptr;  // (the variants hold bf16 pieces in LDS: the log-space walk reads the fp32 weights)
      if constexpr (CT != 0) {
        for (int l = 0; l < D; ++l) src.w[l] = a.w[l];
        src.nodes = a.nodes;
        for (int l = 0; l <= D; ++l) src.node_off[l] = a.node_off[l];
      }
      src.t = t;
      src.B = a.B;
      src.C = a.C;
      src.bl = min(b, a.B - 1);
      float fb[16];
      if constexpr (!SIGNED) {
        if (CT != 0 && a.w_rowmajor) subtree_tile_logspace<D, CK_W_ROWMAJOR>(src, lane, fb);
        else subtree_tile_logspace<D, CK_W_TILED_F32>(src, lane, fb);
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
          tile_store(a.out + (static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh, fb);
        }
      }
    }
    }  // chunk
    CK_WG_STAMP(2);
  }
#ifdef CK_LEAF_STAMPS
  __syncthreads();
  CK_WG_STAMP(3);
  if (kStamps && static_cast<int>(blockIdx.x) == a.stamp_wg && a.redo != nullptr)
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
  if (a.sout != nullptr) store_signed_log(a, t, tile, b, lane, v, sg);  // (all 32 rows of the block; sign words of the live ones)
  else if (b < a.B) tile_store_clog(a.out + ((static_cast<int64_t>(t) * a.B + b) * kK + 4 * kh) * 2, v, sg);
  if (lane == 0 && a.keep[1] == nullptr) *flag = 0;  // ready for the next replay (a training forward: the backward clears it)
}

template <int D, bool XRAW>
hipError_t launch_waves(const LeafArgs& a, int waves, bool is_signed, int n_roots, dim3 grid, hipStream_t s) {
  if (D >= 2 && a.keep[1] != nullptr) {  // the training forward (checked by the caller: raw input, 8 waves, depth 2 | 4)
    if constexpr (XRAW) {
      if constexpr (D == 2 || D == 4) {
        if (is_signed) {  // signed values (a squared circuit's c(x) with real parameters): marked tiles by the signed log-space walk
          if (a.x_pairs) hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, true, true, true, true>), grid, dim3(512), 0, s, a);
          else hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, true, true, false, true>), grid, dim3(512), 0, s, a);
          if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
          hipLaunchKernelGGL((leaf_signed_redo_kernel<D>), dim3((a.B + 31) / 32, n_roots), dim3(64), 0, s, a);
          return hipGetLastError();
        }
      }
      if (is_signed) return hipErrorInvalidValue;
      if constexpr (D >= 2) {
        if (a.x_pairs) {
          hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, true, true>), grid, dim3(512), 0, s, a);
          return hipGetLastError();
        }
      }
      hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, false, true>), grid, dim3(512), 0, s, a);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (a.contraction != 0) {  // (checked by the caller: depth 4, raw input, unsigned, no kept tiles)
    if constexpr (XRAW && D == 4) {
      if (a.contraction == 3) {
        if (a.x_pairs) hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, true, false, 3>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, false, false, 3>), grid, dim3(512), 0, s, a);
      } else {
        if (a.x_pairs) hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, true, false, 6>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, false, false, 6>), grid, dim3(512), 0, s, a);
      }
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (is_signed) {
    bool pairs = false;
    if constexpr (XRAW && D >= 2) {  // (adjacent variables under the leaves 2j, 2j + 1: one 16-byte load per pair, as the unsigned launch)
      if (a.x_pairs) {
        pairs = true;
        hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, true, true, true>), grid, dim3(512), 0, s, a);
      }
    }
    if (!pairs) hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, true, XRAW>), grid, dim3(512), 0, s, a);
    if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
    hipLaunchKernelGGL((leaf_signed_redo_kernel<D>), dim3((a.B + 31) / 32, n_roots), dim3(64), 0, s, a);
  } else {
    if constexpr (XRAW && D >= 2) {
      if (a.x_pairs) {
        hipLaunchKernelGGL((leaf_persistent_kernel<D, 8, false, true, true>), grid, dim3(512), 0, s, a);
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
  CK_REQUIRE(d->waves == 8, "ck_leaf_walk_fwd: waves must be 8 (got %d)", d->waves);
  if (d->K != kK) return ck::fail(CK_ERR_UNSUPPORTED, "ck_leaf_walk_fwd: K=%d (only K=32 is fused)", d->K);
  CK_REQUIRE(ck::aligned16(d->table) && ck::aligned16(d->out), "ck_leaf_walk_fwd: buffers must be 16-byte aligned");
  LeafArgs a{};
  a.table = d->table;
  a.scale = d->table_scale;
  a.xt = d->xt;
  a.scope = d->scope;
  for (int l = 0; l < d->depth; ++l) {
    CK_REQUIRE(d->w_levels[l] != nullptr && ck::aligned16(d->w_levels[l]), "ck_leaf_walk_fwd: bad weights of level %d", l + 1);
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
  CK_REQUIRE(d->signs_out == nullptr || d->signed_redo != nullptr, "ck_leaf_walk_fwd: signs_out belongs to a signed launch");
  a.sout = d->signs_out;
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
  CK_REQUIRE(d->contraction == 0 || d->contraction == 3 || d->contraction == 6, "ck_leaf_walk_fwd: contraction %d (0, 3 or 6)", d->contraction);
  if (d->contraction != 0) {
    if (!(raw && d->depth == 4 && d->signed_redo == nullptr && d->keep_levels == nullptr))
      return ck::fail(CK_ERR_UNSUPPORTED, "ck_leaf_walk_fwd: the bf16-split contraction variants exist for depth-4 unsigned launches over the raw batch");
    a.contraction = d->contraction;
  }
  if (d->keep_levels != nullptr) {
    CK_REQUIRE(raw && d->waves == 8, "ck_leaf_walk_fwd: the training forward (keep_levels) reads the raw batch with 8-wave workgroups");
    CK_REQUIRE(d->signed_redo == nullptr || d->signed_redo == d->keep_redo, "ck_leaf_walk_fwd: a signed training forward marks its tiles in ONE "
               "flag array (signed_redo == keep_redo)");
    CK_REQUIRE(d->keep_redo != nullptr, "ck_leaf_walk_fwd: keep_levels needs keep_redo");
    CK_REQUIRE(static_cast<int64_t>(d->B) * kK < (int64_t{1} << 31), "ck_leaf_walk_fwd: B=%d rows exceed the 32-bit offsets of the kept tiles", d->B);
    CK_REQUIRE(d->depth == 2 || d->depth == 4, "ck_leaf_walk_fwd: keep_levels needs a region of 2 or 4 levels (got %d)", d->depth);
    for (int l = 1; l < d->depth; l += 2) {  // (levels 2 and 4; the entries of levels 1 and 3 are not read)
      CK_REQUIRE(d->keep_levels[l] != nullptr && ck::aligned16(d->keep_levels[l]), "ck_leaf_walk_fwd: bad keep_levels[%d]", l);
      a.keep[l] = d->keep_levels[l];
    }
    a.redo = d->keep_redo;
  }
#ifdef CK_LEAF_STAMPS
  if (d->signed_redo == nullptr && getenv("CK_STAMP_PTR") != nullptr) {
    a.redo = reinterpret_cast<int32_t*>(strtoull(getenv("CK_STAMP_PTR"), nullptr, 0));
    a.stamp_wg = getenv("CK_STAMP_WG") ? atoi(getenv("CK_STAMP_WG")) : 0;
    a.stamp_tile = getenv("CK_STAMP_TILE") ? atoi(getenv("CK_STAMP_TILE")) : 1;
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
        // (the 16-byte loads of the pair form need the batch of THIS replay 16-byte aligned: a view with an odd storage
        // offset takes the 8-byte loads)
        if ((reinterpret_cast<uintptr_t>(b.x64) & 15u) != 0) b.x_pairs = 0;
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
