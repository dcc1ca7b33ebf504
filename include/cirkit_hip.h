/* cirkit_hip.h -- C ABI of the MI355X (gfx950) layer-evaluation backend for cirkit's folded
 * log-space sum-product forward.
 *
 * cirkit has no FFI of its own: its "plugin/operator API" is Python (TorchLayer.forward and the
 * parameter nodes).  Each entry point below therefore names the reference Python function it
 * replaces (paths relative to the reference checkout); the Python shims in cirkit_amd/ bind them
 * with ctypes (see INTEGRATION.md for the stub a cirkit maintainer would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned storage (e.g. torch.Tensor.data_ptr());
 *    the library never frees or retains it past the call (ck_program_* excepted: a program keeps
 *    the pointers it was recorded with until ck_program_destroy);
 *  - every call only ENQUEUES work on the given hipStream_t (passed as void*); no implicit sync;
 *  - returns CK_OK (0) or a negative ck_status; ck_last_error() returns a thread-local message;
 *  - no C++ exception crosses the ABI; no global mutable state besides a device-property cache;
 *  - activations are fp32 (or interleaved complex64 for the *_c entry points), laid out
 *    (F, B, K) row-major with K fastest -- the reference's layout;
 *  - inner layers read their children from one activation ARENA through element offsets:
 *    child h of fold f is the (B, Ki) block at arena + row_off[f*H + h]  (offsets in elements of
 *    the activation type).  This replaces the materialising gather of LayerAddressBook.lookup,
 *    cirkit/backend/torch/circuits.py:42-47.
 */
#ifndef CIRKIT_HIP_H
#define CIRKIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ck_status {
  CK_OK = 0,
  CK_ERR_INVALID = -1,     /* bad shape / null pointer / misaligned buffer */
  CK_ERR_UNSUPPORTED = -2, /* valid request outside what the kernels cover */
  CK_ERR_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString) */
  CK_ERR_STATE = -4        /* program API misuse */
} ck_status;

/* modes of ck_sum_lse_fwd */
#define CK_SUM_CAT 0  /* TorchSumLayer: children concatenated -> N = H*Ki inputs        */
#define CK_SUM_PROD 1 /* TorchCPTLayer: children multiplied (log-space add) -> N = Ki   */
#define CK_SUM_KRON 2 /* TorchTuckerLayer (arity 2): Kronecker of the children -> N = Ki*Ki,
                         one maximum per child (optimized.py:89-103)                     */

/* weight layouts of the Ki = Ko = 32 sum kernels (cirkit_amd/csrc/ck_tile.h).  The tiled layout
 * is written by ck_param_softmax_batch (job kind 2).  (Layout 2 was a split-fp16 form: removed.) */
#define CK_W_ROWMAJOR 0    /* (F, Ko, N) fp32, the reference's layout                                 */
#define CK_W_TILED_F32 1   /* per fold: dword (q, lane, t) = W[lane&31][8q + 4(lane>>5) + t], fp32     */

/* unary parameter ops of ck_param_unary */
#define CK_UNARY_SIGMOID 0
#define CK_UNARY_SCALED_SIGMOID 1 /* sigmoid(x)*(b-a)+a ; nodes.py:698-699 */
#define CK_UNARY_EXP 2
#define CK_UNARY_LOG 3
#define CK_UNARY_SQUARE 4
#define CK_UNARY_CLAMP 5    /* min(max(x, a), b), a / b = -inf / +inf when absent ; TorchClampParameter nodes.py:702-728 */
#define CK_UNARY_SOFTPLUS 6 /* log(1 + exp(x)), x above 20 as it is ; TorchSoftplusParameter nodes.py:731-739 */

const char* ck_last_error(void);
/* ABI version; bumped on any signature change. */
int ck_abi_version(void);
/* Device facts used by the host for launch heuristics: out[0]=CU count, out[1]=LDS bytes/CU,
 * out[2]=wavefront size, out[3]=gcnArch is gfx950 (1/0). */
int ck_device_info(int device, int64_t out[4]);

/* ---------------------------------------------------------------- input layers ------------- */

/* (B, D) int64 -> (D, B) int32.  Stages the batch for the categorical/embedding gathers so that
 * they read x contiguously along B.  Replaces the `in_graph[..., scope_idx].permute(1,0,2)` copy of
 * circuits.py:66. */
int ck_transpose_i64_to_i32(const int64_t* x, int32_t* xt, int B, int D, void* stream);
/* The same staging copy WITH input validation: a value x[b, d] >= num_states[d] (num_states[d] > 0; 0 = variable not
 * checked) sets *flag (a sticky DEVICE int32 owned by the caller) -- TorchCategoricalLayer / TorchEmbeddingLayer raise
 * IndexError on such an index (layers/input.py:258-266, 399-412).  The kernels clamp the category for memory safety; the
 * caller turns the flag into NaN outputs (ck_tail16_lse_fwd's bad_input, or ck_poison_outputs) and into an IndexError on
 * the host when it next looks (HipCircuit.check_inputs).  Negative values stay what they are for this library: the
 * "marginalised variable" sentinel of the integral row.  clamp != 0: checked variables are stored range-mapped --
 * min(x, n - 1), or -1 for any negative x, n = num_states[d] -- which every consumer's own mapping leaves unchanged and
 * lets the persistent leaf launch (preclamped) turn a value into its table row with one unsigned minimum.  x_copy: NULL, or
 * (B, D) int64: the batch as it is, at an address recorded launches keep reading (ck_leaf_walk_fwd / _bwd x_rows). */
int ck_stage_categories(const int64_t* x, int32_t* xt, int B, int D, const int32_t* num_states, int32_t* flag, int clamp,
                        int64_t* x_copy, void* stream);
/* out[0..n) = NaN if *flag != 0 (one small launch; circuits whose last launch is ck_tail16_lse_fwd do not need it). */
int ck_poison_outputs(float* out, int64_t n, const int32_t* flag, void* stream);
/* p[0..n) = 0 if *flag != 0: the gradients of a batch that held an illegal category are dropped before the optimizer
 * sees them (the reference raises IndexError before any update, layers/input.py:399-412; a device-side flag cannot raise,
 * but it must not let NaN gradients into the parameters either). */
int ck_zero_if_flag(float* p, int64_t n, const int32_t* flag, void* stream);
/* (B, D) fp32 -> (D, B) fp32, same purpose for continuous inputs. */
int ck_transpose_f32(const float* x, float* xt, int B, int D, void* stream);

/* TorchCategoricalLayer.log_unnormalized_likelihood, layers/input.py:399-412.
 * table: (F, C+1, K) log-probabilities / logits ALREADY transposed so a (f, c) row is contiguous
 *        (built by ck_param_softmax_batch kind 1 or ck_param_transpose_last2 with out_rows = C+1);
 *        row C is the fold's INTEGRAL row (log-partition of each unit; ck_param_table_integral_row):
 *        a NEGATIVE category selects it = "this variable is marginalised" -- the per-row,
 *        per-variable mask of IntegrateQuery (cirkit/backend/torch/queries.py:103-150,
 *        TorchInputLayer.integrate input.py:280-282, 414-421);
 * xt   : (D, B) int32 (see above); scope[f] = variable of fold f;  out: (F, B, K). */
int ck_categorical_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out,
                       int F, int B, int K, int C, int D, void* stream);

/* TorchGaussianLayer.log_unnormalized_likelihood, layers/input.py:661-670:
 * out = -(x-mean)^2 / (2 stddev^2) - log(stddev) - log(sqrt(2 pi)) (+ log_partition).
 * xt: (D, B) fp32 (ck_transpose_f32); mean/stddev/log_partition: (F, K); log_partition may be
 * NULL.  A NaN input = marginalised variable: the output is log_partition (or 0), input.py:672-679. */
int ck_gaussian_fwd(const float* mean, const float* stddev, const float* log_partition,
                    const float* xt, const int64_t* scope, float* out, int F, int B, int K, int D,
                    void* stream);
/* A fully factorised multivariate input region (templates/region_graph/graph.py:531-540: one
 * Gaussian layer per variable multiplied by a TorchHadamardLayer, inner.py:126-127) in one pass:
 * out[f,b,:] = sum_j ck_gaussian_fwd(fold gfold[f,j])[b,:].  mean/stddev/log_partition/scope are
 * those of the Gaussian layer (G folds), gfold: (F, H) int32 fold ids into it, out: (F, B, K). */
int ck_gaussian_prod_fwd(const float* mean, const float* stddev, const float* log_partition, const float* xt,
                         const int64_t* scope, const int32_t* gfold, float* out, int F, int H, int B, int K,
                         void* stream);

/* TorchEmbeddingLayer.forward under complex-lse-sum, layers/input.py:258-266 + semiring.py:507-509:
 * out[f,b,k] = clog(weight[f,k,x[b,scope[f]]] + 0j).  table: (F, C+1, K) fp32 (transposed weight; row C = the
 * integral row sum_c weight[f,k,c], selected by a negative state as in ck_categorical_fwd). */
int ck_embedding_clog_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out_c,
                          int F, int B, int K, int C, int D, void* stream);
/* The same with COMPLEX weights (the reference compiles DataType.COMPLEX tensors, rules/parameters.py:75-86, and
 * TorchEmbeddingLayer.forward maps them with torch.log of a complex number, input.py:258-266, utils.py:32-35): table_c
 * (F, C+1, K) complex64 as float pairs (ck_param_transpose_last2_c + ck_param_table_integral_row mode 3 on 2 K floats);
 * out_c[f, b, k] = (log|w|, arg w) of w = table_c[f, x[b, scope[f]], k]. */
int ck_embedding_clog_c_fwd(const float* table_c, const int32_t* xt, const int64_t* scope, float* out_c,
                            int F, int B, int K, int C, int D, void* stream);
/* same under lse-sum (semiring.py:495-497): out = log(weight[...]) */
int ck_embedding_log_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out,
                         int F, int B, int K, int C, int D, void* stream);

/* TorchCategoricalLayer.forward under complex-lse-sum: the real log-likelihood gather of ck_categorical_fwd mapped into the
 * complex semiring (layers/input.py:276-278 -> semiring.py:512-514): out_c[f,b,k] = (table[f, x, k], 0). */
int ck_categorical_clog_fwd(const float* table, const int32_t* xt, const int64_t* scope, float* out_c,
                            int F, int B, int K, int C, int D, void* stream);
/* ComplexLSESumSemiring.map_from(x, LSESumSemiring) (semiring.py:512-514) on n fp32 values: out_c[i] = (in[i], 0).
 * Used for input layers whose log-density kernel is real (Gaussian) under complex-lse-sum. */
int ck_lse_to_clse(const float* in, float* out_c, int64_t n, void* stream);

/* TorchConstantValueLayer.forward, layers/input.py:739-743: broadcast value (F, K) over B.
 * complex_out: activations are complex64; value_is_complex: `value` is complex64;
 * log_space 0: apply the map from sum-product (log / clog), 1: value already in log space. */
int ck_constant_fwd(const float* value, float* out, int F, int B, int K, int log_space,
                    int value_is_complex, int complex_out, void* stream);

/* ---------------------------------------------------------------- inner layers ------------- */

/* TorchSumLayer.forward (inner.py:266-273, mode CK_SUM_CAT) and TorchCPTLayer.forward
 * (optimized.py:171-178, mode CK_SUM_PROD) fused with LSESumSemiring.apply_reduce
 * (semiring.py:383-408): v = cat_h/sum_h children; m = clamp(max v); out = log(W . exp(v - m)) + m.
 * w: (F, Ko, N) linear-space weights, N = H*Ki (cat) or Ki (prod), in layout `w_layout`
 * (CK_W_ROWMAJOR for any shape; the tiled layouts need Ki = Ko = 32, H = 1 or mode PROD).
 * out: (F, B, Ko). */
int ck_sum_lse_fwd(const float* arena, const int64_t* row_off, const float* w, float* out, int F,
                   int H, int B, int Ki, int Ko, int mode, int w_layout, void* stream);
/* ck_sum_lse_fwd with the contraction named: 0 = exact fp32 (what ck_sum_lse_fwd does); 3 / 6 = the labelled "bf16x3" / "bf16x6"
 * VARIANTS (operands cut into 2 / 3 bf16 pieces with exact residuals, 3 / 6 products per 16 inputs on v_mfma_f32_32x32x16_bf16,
 * fp32 accumulation) of the launches that have one: dense layers over concatenated children with 32 / 64 units, dense / CP-T
 * layers with 96..512 contracted inputs, Tucker layers (stream-K launch).  Every other shape runs in exact fp32. */
int ck_sum_lse_fwd_v(const float* arena, const int64_t* row_off, const float* w, float* out, int F,
                     int H, int B, int Ki, int Ko, int mode, int w_layout, int contraction, void* stream);
/* TorchTuckerLayer.forward (optimized.py:89-103) of arity 2 with 32 / 64 input units whose weight is softmax(theta) over its
 * last axis (parameters/nodes.py:764-772), WITHOUT the normalised weights in memory: theta (F, Ko, Ki^2) raw logits, which
 * the launch reads once and normalises online (running row maximum and sum beside the accumulators).  This is the stream-K
 * launch of ck_sum_lse_fwd in CK_SUM_KRON mode: it needs the workspace of ck_set_workspace and few tiles per resident
 * workgroup, otherwise CK_ERR_UNSUPPORTED (with one workgroup per tile the exponentials would be applied once per 128
 * rows: evaluate softmax(theta) with ck_param_softmax and call ck_sum_lse_fwd instead). */
int ck_tucker_logits_fwd(const float* arena, const int64_t* row_off, const float* theta, float* out, int F, int B, int Ki,
                         int Ko, void* stream);
/* The stream-K Tucker launch in full: w (F, Ko, Ki^2) holds logits (w_is_logits = 1, as ck_tucker_logits_fwd) or the weights
 * themselves (0, as ck_sum_lse_fwd in CK_SUM_KRON mode).  contraction: 0 = exact fp32 (v_mfma_f32_32x32x2_f32), the product;
 * 3 / 6 = the labelled "bf16x3" / "bf16x6" VARIANTS: the (exponentiated) weights and e_r cut by truncation into 2 / 3 bf16
 * pieces with exact residuals, 3 / 6 products per 16 right indices on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * (dropped terms <= 2^-15 / <= 2^-23 of a product).  CK_ERR_UNSUPPORTED where the stream-K launch does not apply (no
 * workspace, or more than 8 tiles per resident workgroup). */
int ck_tucker_fwd(const float* arena, const int64_t* row_off, const float* w, float* out, int F, int B, int Ki, int Ko,
                  int w_is_logits, int contraction, void* stream);
/* complex-lse-sum variant (semiring.py:441-476); w real (w_is_complex=0) or complex64. */
int ck_sum_lse_fwd_c(const float* arena_c, const int64_t* row_off, const float* w, float* out_c,
                     int F, int H, int B, int Ki, int Ko, int mode, int w_is_complex, void* stream);

/* complex-lse-sum CP-T / dense layer (32 -> 32 units, real weights) whose children are folds of an Embedding
 * layer: their rows are gathered from the layer's REAL weight table (F0, C+1, 32) (the table of
 * ck_embedding_clog_fwd) by the batch values and mapped to (log|w|, pi if w < 0) on the fly, instead of
 * reading a materialised (F0, B, 32) complex output.
 * child_fold / child_var: (F, H) int32 -- table fold and variable of every child; xt: (D, B) staged batch. */
int ck_sum_clse_gather_fwd(const float* table, const int32_t* xt, const int32_t* child_fold, const int32_t* child_var,
                           const float* w, float* out_c, int F, int H, int B, int C, void* stream);

/* Mixing layer = TorchSumLayer whose weight is TorchMixingWeightParameter (nodes.py:847-862):
 * out[f,b,k] = log(sum_h mw[f,k,h] * exp(x[f,h,b,k] - m)) + m,  m = max over all (h,k) of the row.
 * The (K, H*K) block-diagonal weight is never materialised.  mw: (F, K, H). */
int ck_mixing_lse_fwd(const float* arena, const int64_t* row_off, const float* mw, float* out,
                      int F, int H, int B, int K, void* stream);

/* CP sum-product block (what RegionGraph.build_circuit emits for sum_product='cp',
 * templates/region_graph/graph.py:424-456: one dense TorchSumLayer per child region, inner.py:266-273,
 * feeding a TorchHadamardLayer, inner.py:126-127) evaluated without materialising the dense outputs:
 *   out[f,b,:] = sum_s G_s,  G_s = log(W_{f,s} . exp(v_s - m_s)) + m_s  or  G_s = v_s (plain slot),
 *   v_s = sum_h arena[row_off[f,s,h] + b*K + :],  m_s = clamp(max v_s).
 * row_off: (F, S, H) element offsets.  w_addr: (F, S) DEVICE ADDRESSES (as int64) of row-major (K, K)
 * fp32 linear-space weight matrices, 0 for a plain slot; the matrices stay caller-owned.  K in {32, 64}.
 * w_post: NULL, or (F) device addresses of one more (K, K) matrix per fold applied to the product,
 *   out = log(W_post . exp(P - max P)) + max P with P = sum_s G_s -- the consumer is then a TorchCPTLayer
 *   (optimized.py:171-178; Hadamard -> Sum fused by the reference) instead of a bare Hadamard layer.
 * out_off: (F) element offsets of each fold's (B, K) output block inside `out`, or NULL for f*B*K
 * (lets one launch evaluate a subset of the folds of a layer in place).
 * g_var / g_addr (both NULL, or (F, S, H) like row_off): slots with g_var >= 0 do not read the arena but row
 *   x[b, g_var] (row C for a negative value) of the (C+1, K) table at device address g_addr -- the output of
 *   a dense layer over a Categorical layer, precomputed per category (ck_param_softmax_batch kind 4); such
 *   slots carry no weights.  xt: (D, B) staged batch. */
int ck_cp_lse_fwd(const float* arena, const int64_t* row_off, const int64_t* w_addr, const int64_t* w_post,
                  const int64_t* out_off, float* out, const int64_t* g_addr, const int32_t* g_var,
                  const int32_t* xt, int C, int F, int S, int H, int B, int K, void* stream);
/* ck_cp_lse_fwd with the contraction named: 0 = exact fp32 (what ck_cp_lse_fwd does); 3 / 6 = the labelled "bf16x3" / "bf16x6"
 * VARIANTS of the DMA-staged launch (blocks of one child per slot with contiguous output): a weight unit is cut in LDS, the
 * exponentiated tile in registers, into 2 / 3 bf16 pieces (truncation, exact residuals) and contracted with 3 / 6 products per
 * 16 inputs on v_mfma_f32_32x32x16_bf16, fp32 accumulation.  Blocks that launch does not take run in exact fp32. */
int ck_cp_lse_fwd_v(const float* arena, const int64_t* row_off, const int64_t* w_addr, const int64_t* w_post,
                    const int64_t* out_off, float* out, const int64_t* g_addr, const int32_t* g_var,
                    const int32_t* xt, int C, int F, int S, int H, int B, int K, int contraction, void* stream);

/* A region with H partitionings in one launch: the H CP blocks (as in ck_cp_lse_fwd, S slots each)
 * and the mixing layer that combines them (templates/region_graph/graph.py:556-583: `mix_ins` and the
 * arity-H SumLayer with TorchMixingWeightParameter weights, nodes.py:847-862):
 *   P_h = sum_s G_{h,s};  out[f,b,k] = log(sum_h mw[f,k,h] * exp(P_h[k] - M)) + M,  M = max_{h,k} P_h[k].
 * row_off, w_addr: (F, H, S) as in ck_cp_lse_fwd; mw: (F, K, H) mixing coefficients; out: (F, B, K).
 * The sum over h is accumulated online (running maximum), see ck_cp.hip.  g_addr / g_var / xt / C: table
 * slots as in ck_cp_lse_fwd, shaped (F, H, S).
 * redo: NULL, or a workspace of at least F * ceil(B / 128) int32 that is ZERO on entry (and zero again when the call's
 * launches have run).  With it, regions without table slots are evaluated with the products P_h and the mixing sum in
 * LINEAR space (row scales; one exponential per input element and two per row and partitioning instead of an exp, a
 * log and another exp per element), and every workgroup whose rows left the safe fp32 range (largest product of a row
 * <= 2^-80) marks itself and is evaluated again, in log space, by a second launch in which all other workgroups exit
 * at once.  Without it everything is evaluated in log space. */
int ck_region_lse_fwd(const float* arena, const int64_t* row_off, const int64_t* w_addr, const float* mw,
                      float* out, const int64_t* g_addr, const int32_t* g_var, const int32_t* xt, int C,
                      int32_t* redo, int F, int H, int S, int B, int K, void* stream);
/* ck_region_lse_fwd with the contraction named (0 exact fp32; 3 / 6 the bf16-split variants, as ck_cp_lse_fwd_v): regions
 * without table slots.  The tiles a linear-space launch marks in `redo` are evaluated again in exact fp32 log space. */
int ck_region_lse_fwd_v(const float* arena, const int64_t* row_off, const int64_t* w_addr, const float* mw,
                        float* out, const int64_t* g_addr, const int32_t* g_var, const int32_t* xt, int C,
                        int32_t* redo, int F, int H, int S, int B, int K, int contraction, void* stream);

/* TorchHadamardLayer.forward, inner.py:126-127 (lse: sum over the arity axis). esize = 1 (fp32)
 * or 2 (complex64: K counts complex elements). */
int ck_hadamard_fwd(const float* arena, const int64_t* row_off, float* out, int F, int H, int B,
                    int K, int esize, void* stream);

/* TorchKroneckerLayer.forward, inner.py:178-187, any arity H >= 2: out[f, b, r] = sum_h x_h[f, b, digit_h(r)] with
 * r = sum_h digit_h K^(H-1-h) (arity 2: out[f,b,i*K+j] = x0[f,b,i] + x1[f,b,j]); row_off (F, H). */
int ck_kronecker_fwd(const float* arena, const int64_t* row_off, float* out, int F, int H, int B, int K,
                     int esize, void* stream);

/* TorchTensorDotLayer.forward, optimized.py:287-300: x (F,B,Kj*Kq) viewed (Kj,Kq);
 * out[f,b,q*Kk+k] = log(sum_j w[f,k,j] * exp(x[f,b,j,q] - m[f,b,q])) + m[f,b,q]. w: (F, Kk, Kj). */
int ck_tensordot_lse_fwd(const float* arena, const int64_t* row_off, const float* w, float* out,
                         int F, int B, int Kj, int Kq, int Kk, void* stream);
int ck_tensordot_lse_fwd_c(const float* arena_c, const int64_t* row_off, const float* w,
                           float* out_c, int F, int B, int Kj, int Kq, int Kk, int w_is_complex,
                           void* stream);

/* ---------------------------------------------------------------- cross-layer fusion ------- */
/* Categorical input layer -> [dense TorchSumLayer] -> `depth` levels of arity-2 TorchCPTLayer,
 * evaluated depth-first per (root fold, 32-row batch tile) in registers; only the root level's
 * (F_root, B, K) output is written.  Same arithmetic per step as ck_categorical_fwd +
 * ck_sum_lse_fwd (reference: input.py:399-412, inner.py:266-273, optimized.py:171-178,
 * semiring.py:383-408); K must be 32.
 *   table (F0,C+1,K), xt (D,B) int32, scope (F0): as ck_categorical_fwd;
 *   w_dense: (F_dense,K,K) linear weights, or NULL when the leaves feed the CP-T layers directly --
 *            the table is then indexed by the LEVEL-0 fold (node_off[0] table), which lets a caller
 *            apply a dense layer to the (F,C,K) table itself first (one ck_sum_lse_fwd with B = C:
 *            a categorical input has only C distinct values per fold) and fuse only the CP-T levels;
 *   w_levels: HOST array of `depth` device pointers, w_levels[l-1] = (F_l,K,K) weights of level l;
 *   nodes: DEVICE int32 tables; node_off: HOST array of depth+1 offsets into `nodes`:
 *          nodes+node_off[l] is (F_root, 2^(depth-l)) = fold (in level l's layer) of every node of
 *          the subtree of each root fold, left to right (level 0 = dense layer, or input layer);
 *   leaf_off: offset of the (F_root, 2^depth) table of input-layer folds of the leaves;
 *   table_scale: NULL, or (F0, C+1) log scales when `table` holds its rows in LINEAR space (value =
 *          log(row) + scale; written by the kind-5 job of ck_param_softmax_batch).  The levels are then
 *          chained in linear space -- p = y_l y_r, e = p / max p, y = W e, one log per row for the
 *          scale -- instead of log followed by exp; mathematically the same sums (needs depth >= 1 and
 *          w_dense = NULL). */
int ck_subtree_cat_cpt_fwd(const float* table, const float* table_scale, const int32_t* xt, const int64_t* scope,
                           const float* w_dense, const float* const* w_levels,
                           const int32_t* nodes, const int32_t* node_off, int leaf_off, float* out,
                           int depth, int F_root, int B, int K, int C, int w_layout, void* stream);

/* The same fused leaf region (linear table, CK_W_TILED_F32 weights, no in-kernel dense layer, depth >= 1) as a
 * PERSISTENT launch: `n_wg` workgroups of `waves` (8 or 12) wavefronts, one per CU, walk the segment list
 * `work` = DEVICE (n_seg, 4) int32 rows {root fold, first 32-row tile, end tile, 0}: workgroup g takes segments
 * g, g + n_wg, ...; inside a segment its wavefronts draw tiles from an LDS counter.  A segment's 2^depth - 1
 * weight matrices are staged in LDS once; leaf rows are gathered global -> LDS (global_load_lds_dwordx4, 8 lanes
 * per 128-byte row).  Bit-identical outputs to ck_subtree_cat_cpt_fwd with table_scale; the caller chooses the
 * segment list (cirkit_amd/circuit.py: whole roots split evenly over the CUs of one XCD).  preclamped != 0: xt holds
 * -1 .. C - 1 only (ck_stage_categories with clamp != 0): the table row is then min_u32(x, C) -- the integral row C for
 * the marginalisation sentinel -- instead of a compare, a select and a minimum.  Reference semantics as
 * ck_subtree_cat_cpt_fwd.
 * Tiles whose products leave the linear-space range (largest product of a row <= 2^-80) are noted by the wave and
 * evaluated again, in log space, after its walk (reference arithmetic, semiring.py:383-408).
 * w_layout: CK_W_TILED_F32, or CK_W_ROWMAJOR (plain (F_l, 32, 32) parameter tensors).
 * signed_redo != NULL: the table rows are SIGNED linear values (an Embedding layer's weights, layers/input.py:258-266)
 * and the weights may be signed: a real-valued circuit under complex-lse-sum (semiring.py:441-476), whose activations
 * the reference carries as complex logarithms (log|v|, 0 or pi).  The same walk on signed tiles (renormalised by the
 * largest magnitude of a row); `out` is then (F_root, B, 32) complex64: out = (log|v|, pi if v < 0 else 0).  8 waves.
 * signed_redo is a workspace of n_roots * ceil(B / 32) int32, ZERO on entry and zero again afterwards: here the tiles that
 * leave the linear range are marked in it and evaluated, in log space with signs, by a second launch in which every other
 * wave exits at once (as a callee of the main kernel that walk costs it 67 spilled registers).  n_roots = folds of the root
 * layer (only read for signed launches). */
int ck_leaf_persistent_fwd(const float* table, const float* table_scale, const int32_t* xt, const int64_t* scope,
                           const float* const* w_levels, const int32_t* nodes, const int32_t* node_off, int leaf_off,
                           float* out, const int32_t* work, int n_seg, int n_wg, int waves, int depth, int B, int K,
                           int C, int preclamped, int w_layout, int32_t* signed_redo, int n_roots, void* stream);

/* ck_leaf_persistent_fwd with its arguments in a descriptor (every field as the parameter of the same name there; plain
 * pointers and sizes, the two marked arrays are HOST arrays read during the call), plus what the positional form cannot
 * express:
 *
 *  Raw input (x_rows != NULL or x_input >= 0; xt must then be NULL; 8 waves): the launch reads the categories from the
 *  caller's (B, D) int64 batch itself -- what TorchCategoricalLayer.log_unnormalized_likelihood indexes with
 *  (layers/input.py:399-412: `x.long()`) -- no staged copy, no ck_stage_categories launch in front.  A value v selects
 *  table row v for 0 <= v < C; a negative value down to -2^31 selects the integral row C (this library's marginalisation
 *  sentinel, cirkit/backend/torch/queries.py:19-184).  Anything else -- v >= C, an IndexError in the reference, or a value
 *  that does not fit 32 bits -- is an illegal input: with bad_input != NULL the root outputs of THAT BATCH ROW are written
 *  as NaN (every launch downstream carries the NaN to the circuit outputs of the row; other rows are unaffected) and
 *  *bad_input (a DEVICE int32 owned by the caller, as ck_stage_categories' flag) is raised; with bad_input == NULL such a
 *  value is evaluated as the integral row (memory-safe, not meaningful).  B * D * 8 must be below 2^32.
 *  Training forward (keep_levels != NULL; raw input, unsigned values, 8 waves): beside the root outputs the launch stores the
 *  LINEAR tile of the nodes of every SECOND level -- keep_levels[l - 1] is (F_l, ceil(B / 32), 1024) for the CP-T levels l = 2, 4
 *  (the entries of levels 1 and 3 are not read and may be NULL: ck_leaf_walk_bwd recomputes those tiles from their children) in
 *  tile-native order (see ck_leaf_walk_bwd), the value the next level multiplies (per row it differs from exp(layer output) by the power-of-two scale the walk carries; the backward,
 *  ck_leaf_walk_bwd, only needs a level's tiles to be consistent with each other) -- i.e. what the reference's autograd keeps
 *  alive as the outputs of those layers (graph/modules.py:303-335).  keep_redo: (n_roots, ceil(B / 32)) int32, zero on
 *  entry: tiles whose products left the linear range (evaluated again in log space for `out`) are marked 1 there; their
 *  kept tiles are not meaningful and ck_leaf_walk_bwd recomputes them in log space.
 *  x_input >= 0: the call is being RECORDED into a ck_program and the batch pointer is read, at every replay, from that
 *  program's input cell x_input (ck_program_set_input) -- a recorded forward then follows the caller's batch without a
 *  copy.  Not usable with use_graph != 0 launches (a hipGraph keeps the pointer of its capture). */
struct ck_tail16_fold; /* (defined with ck_tail16_lse_fwd below) */
/* out = softmax over the last axis of a (rows <= 32, 32) block of logits; tiled != 0: written in CK_W_TILED_F32 order
 * (rows must then be 32) instead of row-major -- one 32-wide parameter graph tensor -> softmax (nodes.py) */
typedef struct ck_rows32_job {
  const float* in;
  float* out;
  int32_t rows;
  int32_t tiled;
} ck_rows32_job;
typedef struct ck_leaf_launch {
  const float* table;
  const float* table_scale;
  const int32_t* xt;             /* staged (Dvars, B) int32 batch, or NULL (raw input) */
  const int64_t* scope;
  const float* const* w_levels;  /* HOST array of `depth` device pointers */
  const int32_t* nodes;
  const int32_t* node_off;       /* HOST array of depth + 1 offsets */
  int32_t leaf_off;
  int32_t n_seg;
  float* out;
  const int32_t* work;
  int32_t n_wg, waves, depth, B, K, C, preclamped, w_layout;
  int32_t* signed_redo;
  int32_t n_roots;
  int32_t x_input;               /* -1, or the program input cell holding the raw batch pointer */
  const int64_t* x_rows;         /* raw (B, D) int64 batch, or NULL */
  int32_t* bad_input;            /* raw input: validation flag (rows with illegal values become NaN), or NULL */
  int32_t D;                     /* variables per row of the raw batch */
  int32_t contraction;           /* 0: exact fp32 (v_mfma_f32_32x32x2_f32).  3 / 6: labelled VARIANTS -- every fp32 operand of a contraction cut
                                    into 2 / 3 bf16 pieces, 3 / 6 products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation ("bf16x3": ~2^-15 per
                                    product; "bf16x6": fp32-like); depth 4, raw input, unsigned, no keep_levels */
  int32_t reserved2;
  float* const* keep_levels;     /* NULL, or HOST array of `depth` DEVICE pointers: (F_l, B, 32) linear tiles of level l's nodes */
  int32_t* keep_redo;            /* with keep_levels: DEVICE (n_roots, ceil(B / 32)) flags of the tiles evaluated in log space */
  int32_t x_pairs;                          /* raw input: for every root, leaves 2j and 2j + 1 read variables v and v + 1 with v even
                                               (and D even): the launch then fetches both values with one 16-byte load */
  const int32_t* root_tab;                  /* DEVICE (n_roots of the region, 3 * 2^depth), or NULL: row t = for each leaf i of root t its
                                               variable scope[nodes[leaf_off + t 2^depth + i]] and its table fold
                                               nodes[node_off[0] + t 2^depth + i], then the folds of the root's 2^depth - 1 nodes in the
                                               order of the walk's steps (rest of the row unused): one load round at the start of a
                                               segment instead of three dependent ones */
  uint32_t* signs_out;                      /* signed launches: NULL, or DEVICE (n_roots, Bp) sign words, Bp = 32 ceil(B / 32): `out` is
                                               then a SIGNED-LOG block (ck_slse_fwd: (n_roots, Bp, 32) fp32 log|v| in tile-native order,
                                               bit k of a row's word = unit k is negative) instead of (n_roots, B, 32) complex64.  With
                                               keep_levels (signed_redo == keep_redo): the forward of a squared circuit's training step */
} ck_leaf_launch;
int ck_leaf_walk_fwd(const ck_leaf_launch* desc, void* stream);

/* The trailing few-fold levels of forward k (ck_tail16_lse_fwd's walk, 8 waves per 16-row tile, optional fused
 * `circuit(batch).sum()`) and the parameter graphs of forward k + 1 (ck_param_softmax_batch's kind-5 table jobs and 32-wide
 * softmaxes) in ONE launch: both are latency-bound on their own and independent of each other.  The reference
 * re-evaluates its parameter graphs once per forward (parameters/parameter.py:180-188); this launch does that at the END of
 * a forward, for the next one -- the caller evaluates them on their own (ck_param_softmax_batch) before a forward whose
 * parameters have changed since.  folds: as ck_tail16_lse_fwd, with `slot` = the LDS slot of each fold's tile (a slot may
 * be reused by a later fold once every reader of the earlier one has run) and child_src[h] = the child's SLOT; n_slots
 * slots of 2 KB + 8 KB of descriptors must stay below 80 KB.  Table job d (0 <= d < n_tables): log-table of Categorical
 * fold cat_idx[d] pushed through dense fold d -> table[d] (C + 1, 32) linear rows, table_scale[d] (C + 1) log scales
 * (layers/input.py:399-412, layers/inner.py:266-273).  rows: 32-wide softmaxes (nodes.py softmax over the last axis). */
typedef struct ck_tail_params_launch {
  const struct ck_tail16_fold* folds;
  const int32_t* level_begin;
  int32_t n_folds, n_levels, n_slots, B, w_layout, C;
  double* ll;
  double* ll_partial;
  uint32_t* ll_ticket;
  const int32_t* bad_input;
  const float* cat_logits;
  const int64_t* cat_idx;
  const float* dense_logits;
  float* table;
  float* table_scale;
  const struct ck_rows32_job* rows;
  int32_t n_tables, n_rows;
  int32_t ll_cell;                          /* 0: the pair is written to `ll`.  k in 1..3 while RECORDING: at every eager replay the pair
                                               goes to the pointer in program input cell k (ck_program_set_input), to `ll` while that
                                               cell is NULL -- a caller collecting the pairs of many steps in one buffer (one
                                               collective for all of them) hands each step its row instead of copying it there */
} ck_tail_params_launch;
int ck_tail_params_fwd(const ck_tail_params_launch* desc, void* stream);

/* The last levels of a circuit (few folds each) in one launch, on 16-row tiles: layer i is a TorchCPTLayer / dense TorchSumLayer
 * step over the product of its children (CK_SUM_PROD semantics), Ki = 32, Ko = 32 or < 32 for the terminal layer (e.g. the
 * scalar root).  (A 32-row form of this walk, ck_tail_lse_fwd, existed until round 5: nothing reached it any more.) */
/* The walk on 16-row tiles (v_mfma_f32_16x16x4_f32): one workgroup of 16 wavefronts per 16 batch rows; the fold
 * descriptors are staged in LDS once, every fold output is kept in LDS for the levels above it (and written to its
 * `out` block as before), the weights of a wave's next fold are requested before the level barrier.
 * folds: DEVICE array of n_folds descriptors in level order (16-byte aligned); level_begin: DEVICE (n_levels + 1)
 * first fold of each level (folds of one level never read each other).  A fold is a TorchCPTLayer / dense TorchSumLayer
 * step over the product of its H <= 4 children (CK_SUM_PROD semantics), Ki = 32, Ko = 32 or < 32 (e.g. the scalar root;
 * row-major weights; never a child inside the tail).  w_layout (Ko = 32 folds): CK_W_ROWMAJOR or CK_W_TILED_F32.
 * ll != NULL folds ck_ll_sum in (the last fold must be the scalar root): ll[0] = sum_b out[b] in fp64 -- per-workgroup
 * sums in row order, then workgroup order: deterministic -- ll[1] = B; ll_partial: ceil(B / 16) doubles of scratch,
 * ll_ticket: one zero-initialised uint32 (left zero).  bad_input: NULL, or the flag of ck_stage_categories -- when it is
 * nonzero the few-output folds (the circuit's root) write NaN: an invalid batch never yields a plausible likelihood. */
typedef struct ck_tail16_fold {
  const float* w;          /* (Ko, 32) linear weights of this fold                                             */
  float* out;              /* (B, Ko) output block of this fold                                                */
  const float* child[4];   /* (B, 32) block of child h when child_src[h] < 0                                   */
  int32_t child_src[4];    /* index (in `folds`) of child h when it is a 32-unit fold of the tail itself, else -1 */
  int32_t H, Ko;
  int32_t skip_store;      /* != 0 (32-unit folds only): the output is a child inside the tail and nobody else reads it --
                              it stays in LDS and `out` is not written (the reference keeps every layer output alive,
                              graph/modules.py:303-335; a forward that returns only the circuit outputs need not)           */
  int32_t slot;            /* LDS slot of the fold's tile for the launches that assign slots (ck_leaf_walk_fwd's tail:
                              the fold index; ck_tail_params_fwd: any reuse-respecting assignment); ck_tail16_lse_fwd
                              ignores it */
} ck_tail16_fold;
/* signed_values != 0: a real-valued circuit under complex-lse-sum (semiring.py:441-476): every block in memory (children,
 * outputs) is (B, Ko) complex64 holding (log|v|, 0 or pi), the weights may be signed; `ll` must then be NULL. */
int ck_tail16_lse_fwd(const ck_tail16_fold* folds, int n_folds, const int32_t* level_begin, int n_levels, int B, int K,
                      int w_layout, double* ll, double* ll_partial, uint32_t* ll_ticket, const int32_t* bad_input,
                      int signed_values, void* stream);


/* ---------------------------------------------------------------- parameter graphs --------- */
/* The reference re-evaluates each layer's parameter DAG on every forward
 * (parameters/parameter.py:180-188); these kernels do the same on the fold-stacked blocks. */

/* TorchSoftmaxParameter / TorchLogSoftmaxParameter (nodes.py:764-783) over the middle axis of an
 * (outer, len, inner) view. */
int ck_param_softmax(const float* in, float* out, int64_t outer, int len, int64_t inner,
                     int log_space, void* stream);
/* All `tensor -> softmax(last axis)` parameters of a circuit in one launch.  `jobs` is a HOST array
 * (copied into the launch).  kind 0: out[r, :] = softmax(in[r, :]) for `rows` rows of `len`.
 * kind 1 (Categorical probs, input.py:405-408): in (rows=F, k=K, len=C) logits ->
 * out (F, C+1, K) = log(softmax over C), transposed for the gather kernels, row C = 0 (integral).
 * kind 2: as kind 0 for (F*32, 32) weights, written in CK_W_TILED_F32 layout (rows must be a multiple of 32,
 * len = 32).  (kind 3 wrote the split-fp16 layout of a contraction that no longer exists: rejected.)
 * kind 4: kind 1 followed by a dense sum layer applied to the table (k = 32 or 64): for each of the `rows`
 * dense folds d, out[d] (C+1, 32) = log(softmax(in2[d]) . exp(T - m)) + m row by row, with T the kind-1
 * table of categorical fold idx[d] (a Categorical layer followed fold by fold by a dense layer only
 * takes C distinct values per fold, so the dense layer is evaluated on the table instead of on the batch).
 * kind 5: as kind 4 but every row is left in LINEAR space, out[d, c, :] = softmax(in2[d]) . exp(T[c] - m_c),
 * with its log scale m_c in out2[d, c] (the representation ck_subtree_cat_cpt_fwd takes with table_scale).
 * block_begin is ignored on input. */
typedef struct ck_softmax_job {
  const float* in;
  float* out;
  int64_t rows;
  int32_t len;
  int32_t k;
  int32_t kind;
  int32_t block_begin;
  const float* in2;   /* kind 4: (rows, 32, 32) logits of the dense layer */
  const int64_t* idx; /* kind 4/5: categorical fold of each dense fold, or NULL for the identity */
  float* out2;        /* kind 5: (rows, C+1) log scale of each table row */
} ck_softmax_job;
int ck_param_softmax_batch(const ck_softmax_job* jobs, int njobs, void* stream);
/* TorchBinomialLayer.log_unnormalized_likelihood (input.py:530-541) as a table: table (F, total_count + 2, K) --
 * row c = log-pmf of the value c for every unit, the last row the layer's integral (0); p (F, K) probabilities or logits.
 * ck_categorical_fwd then gathers row x[b] exactly as for a Categorical layer with total_count + 1 categories. */
int ck_param_binomial_table(const float* p, int is_logits, float* table, int64_t F, int K, int total_count, void* stream);
/* TorchGaussianProductLogPartition.forward (nodes.py:975-988): out[f, i*K2 + j] = log of the integral of the product
 * of Gaussian unit i of (mean1, stddev1) and unit j of (mean2, stddev2); all inputs (F, K). */
int ck_param_gaussian_product_logz(const float* mean1, const float* stddev1, const float* mean2, const float* stddev2,
                                   float* out, int64_t F, int K1, int K2, void* stream);
/* TorchGaussianProductMean / TorchGaussianProductStddev.forward (nodes.py:865-938) for every unit pair, out (F, K1 K2):
 * op 0 the mean (m1 s2^2 + m2 s1^2) / (s1^2 + s2^2), op 1 the standard deviation sqrt(s1^2 s2^2 / (s1^2 + s2^2)) (the means are
 * not read and may be NULL); _bwd: dout -> the operands' gradients, WRITTEN (op 1: dmean1 / dmean2 untouched, may be NULL). */
int ck_param_gaussian_product_ms(int op, const float* mean1, const float* stddev1, const float* mean2, const float* stddev2, float* out,
                                 int F, int K1, int K2, void* stream);
int ck_param_gaussian_product_ms_bwd(int op, const float* mean1, const float* stddev1, const float* mean2, const float* stddev2,
                                     const float* dout, float* dmean1, float* dstddev1, float* dmean2, float* dstddev2, int F, int K1, int K2,
                                     void* stream);
/* Its backward: dout (F, K1 K2) -> the gradients of the four operands, WRITTEN (four distinct buffers, (F, K1) / (F, K2)). */
int ck_param_gaussian_product_logz_bwd(const float* mean1, const float* stddev1, const float* mean2, const float* stddev2,
                                       const float* dout, float* dmean1, float* dstddev1, float* dmean2, float* dstddev2, int64_t F,
                                       int K1, int K2, void* stream);
/* TorchReduceProductParameter / TorchReduceLSEParameter (nodes.py:754-761): op 0 = product, 1 = log-sum-exp along the middle axis
 * of x viewed (outer, len, inner) -> y (outer, inner); _bwd: dx from x, y and dy.  TorchOuterSumParameter (nodes.py:615-653):
 * out[o, i1 n2 + i2, r] = a[o, i1, r] + b[o, i2, r]; _bwd: the gradient of operand `which` (0: a, 1: b) from dout. */
int ck_param_reduce(int op, const float* x, float* y, int64_t outer, int len, int64_t inner, void* stream);
int ck_param_reduce_bwd(int op, const float* x, const float* y, const float* dy, float* dx, int64_t outer, int len, int64_t inner, void* stream);
int ck_param_outer_sum(const float* a, const float* b, float* out, int64_t outer, int n1, int n2, int64_t inner, void* stream);
int ck_param_outer_sum_bwd(const float* dout, float* dx, int64_t outer, int n1, int n2, int64_t inner, int which, void* stream);
/* entrywise ops (nodes.py:656-739); a, b only used by CK_UNARY_SCALED_SIGMOID and CK_UNARY_CLAMP (vmin, vmax). */
int ck_param_unary(int op, const float* in, float* out, int64_t n, float a, float b, void* stream);
/* out[f] = in[idx[f]] over blocks of `per_fold` 4-byte words (pointer fold_idx nodes.py:277-279,
 * parameter address-book gathers parameter.py:41-47). */
int ck_param_gather_folds(const float* in, const int64_t* idx, float* out, int64_t F_out,
                          int64_t per_fold, void* stream);
/* TorchConjugateParameter (nodes.py:745-746) on complex64 data (n complex elements). */
int ck_param_conj(const float* in_c, float* out_c, int64_t n, void* stream);
/* TorchMixingWeightParameter (nodes.py:857-862): (F,K,H) -> dense (F,K,H*K). Only needed when the
 * mixing weight feeds another parameter op (e.g. MatMul after SumCollapse). */
int ck_param_mixing_weight(const float* in, float* out, int F, int K, int H, void* stream);
/* Batched matmul out[f] = op(a[f]) . op(b[f]): TorchMatMulParameter (nodes.py:802-805) and the
 * two-operand TorchEinsumParameter patterns (optimized.py:282-284).  a: (F,M,Kd) or (F,Kd,M) if
 * trans_a; b: (F,Kd,N) or (F,N,Kd) if trans_b; out: (F,M,N).  Extents that are multiples of 32: one wavefront per (32, 32)
 * output tile on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate); otherwise (32, 64) multiply-add tiles. */
/* `accumulate` != 0: out[f] += op(a[f]) . op(b[f]) -- the gradient of an einsum operand added straight into the
 * gradient of the stored tensor behind it (autograd's accumulation through TorchPointerParameter / TorchConjugateParameter /
 * TorchFlattenParameter, nodes.py:277-279, 745-746, 843-844) instead of a product buffer and an axpy.
 * trans_a == 2 (M == Kd, extents multiples of 32): op(a) = a + a^T -- both operand gradients of a Gram product y = x x^T
 * (autograd sends d y x and d y^T x to the same tensor) as ONE product. */
int ck_param_bmm(const float* a, const float* b, float* out, int F, int M, int N, int Kd, int trans_a, int trans_b, int accumulate,
                     void* stream);
/* (R, A, Bd) -> (R, out_rows >= Bd, A) transpose of the last two axes (rows beyond Bd untouched),
 * optionally taking log first (categorical: log(probs) -> table (F, C+1, K), input.py:405-408). */
/* TorchEinsumParameter (parameters/optimized.py:282-284) for any pattern: out[f, o...] = sum over the contracted indices of
 * prod_k x[k][f, ...].  Indices are numbered so that 0 .. n_out - 1 are the output's (in its order) and n_out .. n_idx - 1 the
 * contracted ones; stride[k][i] is the ELEMENT stride of operand k along index i inside a fold (0: the operand does not carry
 * it; a repeated index: the sum of its strides), fold_stride[k] its elements per fold.  Complex operands hold (re, im) pairs
 * (strides count complex elements); the output (F, extents of the output indices) is complex iff out_complex. */
#define CK_EINSUM_MAX_OPERANDS 4
#define CK_EINSUM_MAX_INDICES 8
typedef struct ck_einsum_desc {
  const float* x[CK_EINSUM_MAX_OPERANDS];
  float* out;
  int32_t n_ops, n_idx, n_out, F, out_complex;
  int32_t is_complex[CK_EINSUM_MAX_OPERANDS];
  int32_t extent[CK_EINSUM_MAX_INDICES];
  int64_t stride[CK_EINSUM_MAX_OPERANDS][CK_EINSUM_MAX_INDICES];
  int64_t fold_stride[CK_EINSUM_MAX_OPERANDS];
} ck_einsum_desc;
int ck_param_einsum(const ck_einsum_desc* d, void* stream);
int ck_param_transpose_last2(const float* in, float* out, int64_t R, int A, int Bd, int take_log,
                             int out_rows, void* stream);
/* The same transposition of COMPLEX elements (float pairs), no logarithm. */
int ck_param_transpose_last2_c(const float* in_c, float* out_c, int64_t R, int A, int Bd, int out_rows, void* stream);
/* Integral row (row C) of a gather table (F, C+1, K): mode 0 zeros (normalised probabilities),
 * 1 logsumexp over the C category rows (unnormalised logits, input.py:414-421), 2 ones (embedding),
 * 3 complex ones: K counts floats, the row becomes K / 2 pairs (1, 0) (embedding with complex weights). */
int ck_param_table_integral_row(float* table, int F, int C, int K, int mode, void* stream);

/* ---------------------------------------------------------------- backward (training) ------ */
/* Gradients of a scalar loss through the lse-sum forward -- what the reference gets from autograd
 * through LSESumSemiring.apply_reduce (semiring.py:383-408) and the layer forwards.  `garena` mirrors
 * the activation arena (same offsets).  accumulate: 0 store, 1 add (ordered launches), 2 atomic add
 * (a producer fold read several times by this layer). */
int ck_fill_f32(float* p, int64_t n, float value, void* stream);
/* garena[coff[c] + i] += sum_{j in [cptr[c], cptr[c+1])} tmp[clist[j] * block_elems + i]: the gradients that the folds
 * of one layer contribute to a child they share (autograd's accumulation into a tensor indexed twice by
 * LayerAddressBook.lookup, circuits.py:39-48), without float atomics and in list order. */
int ck_segment_add_rows(const float* tmp, const int32_t* cptr, const int32_t* clist, const int64_t* coff, float* garena,
                        int n_child, int64_t block_elems, void* stream);
/* TorchSumLayer / TorchCPTLayer / TorchTuckerLayer backward (modes CK_SUM_CAT / CK_SUM_PROD / CK_SUM_KRON:
 * optimized.py:89-103, one maximum per child, N = Ki^H contracted inputs), row-major linear
 * weights w (F,Ko,N): children gradients into garena at grad_row_off (NULL: at row_off, the mirror of the arena),
 * dW (F,Ko,N) accumulated with atomics
 * (zero it first). out/gout: (F,B,Ko) forward output and its gradient. */
int ck_sum_lse_bwd(const float* arena, float* garena, const int64_t* row_off, const int64_t* grad_row_off,
                   const float* w, const float* out, const float* gout, float* dw, int F, int H, int B, int Ki, int Ko,
                   int mode, int accumulate, void* stream);
/* The same layers under complex-lse-sum (ComplexLSESumSemiring.apply_reduce, semiring.py:441-476; ComplexSafeLog,
 * utils.py:22-50): arena / garena / out / gout hold complex64 (re, im) pairs, row_off counts complex elements, the
 * children's gradients are STORED at row_off of garena (torch's convention: conj(dy/dx) * gout), dw -- (F, Ko, N) floats, or
 * complex pairs when w_is_complex -- is accumulated with atomics (zero it first; real weights receive the real part). */
int ck_sum_lse_bwd_c(const float* arena_c, float* garena_c, const int64_t* row_off, const float* w, const float* out_c,
                     const float* gout_c, float* dw, int F, int H, int B, int Ki, int Ko, int mode, int w_is_complex,
                     void* stream);
/* TorchKroneckerLayer backward (inner.py:178-187): gout (F, B, K^H); the gradient of child h's unit i is the sum of gout
 * over the outputs whose digit h (child 0 most significant) is i. */
int ck_kronecker_bwd(float* garena, const int64_t* row_off, const float* gout, int F, int H, int B, int K,
                     int accumulate, void* stream);
/* TorchHadamardLayer backward: every child receives gout (F,B,K). */
int ck_hadamard_bwd(float* garena, const int64_t* row_off, const float* gout, int F, int H, int B, int K,
                    int accumulate, void* stream);
/* Gaussian input layer backward (input.py:661-670): dmean[f,k] = sum_b gout (x-mu)/sd^2,
 * dstddev[f,k] = sum_b gout ((x-mu)^2/sd^3 - 1/sd); NaN inputs (marginalised) contribute nothing. */
int ck_gaussian_bwd(const float* gout, const float* xt, const int64_t* scope, const float* mean, const float* stddev,
                    float* dmean, float* dstddev, int F, int B, int K, void* stream);
/* Mixing layer backward (forward: ck_mixing_lse_fwd): input gradients into garena at the children's
 * offsets (accumulate 0 store / 1 add / 2 atomic), dmw (F, K, H) += batch sums (zero it first). */
int ck_mixing_lse_bwd(const float* arena, float* garena, const int64_t* row_off, const int64_t* grad_row_off,
                      const float* mw, const float* gout, float* dmw, int F, int H, int B, int K, int accumulate,
                      void* stream);
/* Parameter-graph backward pieces: scaled sigmoid (nodes.py:698-699) from its OUTPUT y, the mixing-weight
 * expansion (nodes.py:857-862), y += a x. */
int ck_param_scaled_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, float vmin, float vmax,
                                int accumulate, void* stream);
/* softmax / log-softmax backward along any axis: tensors viewed as (outer, len, inner), y = the node's output
 * (TorchSoftmaxParameter / TorchLogSoftmaxParameter for any `dim`, nodes.py:764-783); entrywise nodes from input x and output y
 * (CK_UNARY_SIGMOID / EXP / LOG / SQUARE / CLAMP / SOFTPLUS, nodes.py:656-739; clamp: the gradient passes where y == x, as
 * torch.clamp's backward does inside [vmin, vmax]; softplus: sigmoid(x)). */
int ck_param_softmax_bwd_strided(const float* y, const float* dy, float* dx, int64_t outer, int len, int64_t inner, int log_space,
                                 int accumulate, void* stream);
int ck_param_unary_bwd(int op, const float* x, const float* y, const float* dy, float* dx, int64_t n, int accumulate, void* stream);
int ck_param_mixing_weight_bwd(const float* dy, float* dx, int F, int K, int H, int accumulate, void* stream);
int ck_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream);
/* Backward of ck_param_gather_folds: ddst[idx[i]] += dsrc[i] over blocks of `per_fold` fp32 words. */
int ck_param_scatter_add_folds(const float* dsrc, const int64_t* idx, float* ddst, int64_t n, int64_t per_fold,
                               void* stream);
/* TorchCategoricalLayer backward (the scatter-add that autograd performs for the advanced indexing of
 * layers/input.py:399-412): dtable[f,c,:] (+)= sum_{b: x[b,scope f]=c} gout[g(f),b,:]  (dtable (F,C+1,K), same transposed
 * layout as the forward table; accumulate 0 overwrites it -- every row of it -- 1 adds).  gfold: NULL (g(f) = f), or DEVICE (F) int32: the (B, K) block of `gout` that holds fold f's
 * gradient -- in a fused backward the two leaves of a product share ONE gradient tile (ck_leaf_walk_bwd).  fold_order: NULL,
 * or DEVICE (F) int32 permutation: workgroup b evaluates fold fold_order[b] (workgroups b and b + 8 run on the same XCD at about
 * the same time: folds that read the same gout block placed 8 apart fetch it from memory once); needs K % 32 == 0, B >= 256. */
int ck_categorical_bwd(const float* gout, const int32_t* gfold, const int32_t* xt, const int64_t* scope, float* dtable, int F,
                       int B, int K, int C, int accumulate, const int32_t* fold_order, void* stream);

/* Backward of the fused leaf region, two CP-T levels per launch (cirkit_amd/csrc/ck_leaf_bwd.hip) -- what autograd does for
 * TorchCPTLayer.forward (layers/optimized.py:171-178) under LSESumSemiring.apply_reduce (semiring.py:383-408), on the tiles
 * ck_leaf_walk_fwd kept (keep_levels) instead of materialised layer outputs and layer gradients.
 * A unit is a node P of level L, its children Q0, Q1 (level L - 1) and their four children c0..c3 (level L - 2, or -- leaf
 * != 0, L = 2 -- the Categorical table rows of four leaves), for one 32-row batch tile.  unit_tab: DEVICE (n_units, 16)
 * int32 rows [fold of P's gradient tile in `gin`, fold of P, of Q0, of Q1, of c0..c3 (leaf: table folds), variables of the four
 * leaves (leaf only), root fold of the region (for `redo`), 0, 0, 0]; work: DEVICE (n_seg, 4) [row of unit_tab, first tile,
 * end tile, 0] dealt to n_wg resident workgroups.  gin: gradient w.r.t. the LOG-space output of P (for the top launch the
 * gradient of the root layer's output, (F, B, 32) row-major: gin_rowmajor; below, the tile the previous launch left in ITS
 * `gout` for P's parent).  Tiles that only these launches exchange are TILE-NATIVE: a (F_l, ceil(B / 32), 1024) array whose
 * 4 KB block (fold, tile) holds, at dword (g, lane, t), unit 8g + 4 (lane >> 5) + t of row 32 tile + (lane & 31) -- the MFMA
 * register layout, one contiguous KiB per wave instruction.  y_p / y_c: the kept linear tiles of P's level and of the level two below (Q's tiles are recomputed)
 * (tile-native, ck_leaf_walk_fwd keep_levels); w_p / w_q: (F_l, 32, 32) row-major linear weights; dw_p / dw_q: their
 * gradients, accumulated (+=, atomically per segment); gout, written: the log-space gradient node Q leaves for BOTH its
 * children (the two children of a product receive the same one) -- tile-native (F_q, tiles, 1024), or, leaf != 0,
 * (F_q, B, 32) row-major (ck_categorical_bwd reads it row by row).
 * redo: NULL or ck_leaf_walk_fwd's keep_redo flags: flagged (root, tile) units are skipped (their kept tiles are not
 * meaningful; the caller evaluates them with the layer-wise kernels). */
typedef struct ck_leaf_bwd_launch {
  const int32_t* unit_tab;
  const int32_t* work;
  int32_t n_seg, n_wg, B, C, D, leaf;
  int32_t waves, gin_rowmajor;  /* gin_rowmajor != 0: gin is (F, B, 32) row-major (a layer-wise launch wrote it); wavefronts per workgroup: 8 (two per SIMD, a unit's loads issued at its start) or 4 (one per SIMD, up to 512
                               registers each: the next unit's tiles travel while the current one computes) */
  const float* gin;
  const float* y_p;
  const float* y_c;
  const float* table;
  const int64_t* x_rows;
  const float* w_p;
  const float* w_q;
  float* dw_p;
  float* dw_q;
  float* gout;
  const int32_t* redo;
  int32_t is_signed;  /* != 0: the tiles of ck_leaf_walk_fwd's SIGNED training forward (signs_out + keep_levels): linear values of either
                         sign, gradients w.r.t. log|.|; waves must be 8 */
  int32_t reserved;
} ck_leaf_bwd_launch;
int ck_leaf_walk_bwd(const ck_leaf_bwd_launch* desc, void* stream);
/* Backward of a circuit's trailing few-fold sum layers in ONE launch (cirkit_amd/csrc/ck_tail_bwd.hip): what autograd does
 * for TorchCPTLayer.forward (optimized.py:171-178) / dense TorchSumLayer.forward (inner.py:266-273) under
 * LSESumSemiring.apply_reduce (semiring.py:383-408), for all of these layers, one workgroup per 32-row batch tile walking them
 * top down.  folds: DEVICE array, the folds of the top layer first; level_begin: DEVICE (n_levels + 1) first fold of each
 * layer.  Per fold: w (Ko, 32) row-major LINEAR weights (Ko = 32, or 1 for a scalar root); gout (B, Ko) the gradient w.r.t.
 * its log-space output (a Ko = 1 fold must be the only fold of the first level); child[h] / gchild[h], h < H <= 2: the (B, 32)
 * log-space outputs of its children (multiplied: CP-T, or H = 1 -- then child[1] / gchild[1] must name the first child again)
 * and the blocks their gradient is WRITTEN to (each child must have one consumer); dw_part: the fold's (Ko, 32) slot
 * of tile 0 in a buffer of per-tile weight-gradient contributions, slots of consecutive tiles part_stride floats apart --
 * written, not added: ck_param_softmax_bwd_batch sums the ceil(B / 32) slots (n_part / part_stride of its jobs). */
typedef struct ck_tail_bwd_fold {
  const float* w;
  const float* gout;
  float* dw_part;
  const float* child[4];
  float* gchild[4];
  int32_t H, Ko;
} ck_tail_bwd_fold;
int ck_tail_bwd(const ck_tail_bwd_fold* folds, int n_folds, const int32_t* level_begin, int n_levels, int B, int64_t part_stride,
                void* stream);
/* Backward of `dense_on_table` (ck_param_softmax_batch's kind-5 job: T' = dense(log-table), layers/input.py:399-412 +
 * layers/inner.py:266-273 with both parameters softmaxes, nodes.py:764-772): dtable (F, C + 1, 32) is the gradient w.r.t. the
 * LOG-space table T' (ck_categorical_bwd's output); g_cat (F_cat, 32, C) and g_dense (F, 32, 32) receive (=) the gradients of
 * the raw parameters cat_logits / dense_logits (cat_idx: NULL or the Categorical fold of each dense fold; folds must not
 * repeat).  One workgroup per fold rebuilds T and W from the raw parameters and runs the (C + 1)-row backward in LDS. */
struct ck_opt_state; /* (defined with the job form of the training step below) */
/* opt != NULL (one rank, C % 4 == 0, cat_idx NULL): the optimizer in the launch's epilogue -- the workgroup that holds the
 * gradients of a fold's Categorical and dense logits updates both tensors and their moments in place (`optimizer.step()` of
 * the reference's loop for these two tensors: torch.optim.Adam / SGD, constants and clock in the DEVICE ck_opt_state that
 * ck_opt_tick advances; a dropped step changes nothing) and writes the fold's table of the NEXT forward, table (F, C + 1, 32)
 * linear rows and table_scale (F, C + 1) (what ck_param_softmax_batch's table job would make of the new logits, bit for bit). */
typedef struct ck_table_opt {
  const struct ck_opt_state* state;
  float* m1_cat;
  float* m2_cat;
  float* m1_dense;
  float* m2_dense;
  float* table;
  float* table_scale;
} ck_table_opt;
int ck_table_dense_bwd(const float* cat_logits, const int64_t* cat_idx, const float* dense_logits, const float* dtable, float* g_cat,
                       float* g_dense, int F, int C, const ck_table_opt* opt, void* stream);
/* The (root, tile) units ck_leaf_walk_fwd marked in keep_redo (products that left the linear range), for the whole region of
 * `depth` (2 or 4) levels at once: one wave per marked unit walks the subtree in LOG space -- the reference's arithmetic,
 * semiring.py:383-408, forward values recomputed -- adds the weight gradients of every level (dw_levels[l - 1], row-major,
 * float atomics), writes the gradient tiles of the level-1 nodes into gout1 (where the leaf launch of ck_leaf_walk_bwd
 * leaves them) and clears the mark.  Unmarked units exit at once.  table / table_scale / nodes / node_off / leaf_off /
 * scope as ck_leaf_walk_fwd; w_levels, dw_levels: HOST arrays of `depth` DEVICE pointers; gin: (F_root, B, 32) -- or,
 * gin_fold != NULL, a tile-native array of (ceil(B / 32), 1024) blocks of which block gin_fold[t] (DEVICE, n_roots) is root t's.
 * is_signed != 0: the tiles of a SIGNED training forward (values (log|v|, sign) as semiring.py:441-476 carries a real number). */
int ck_leaf_walk_bwd_redo(const float* table, const float* table_scale, const int64_t* x_rows, int B, int C, int D,
                          const int32_t* nodes, const int32_t* node_off, int leaf_off, const int64_t* scope, int depth,
                          const float* const* w_levels, float* const* dw_levels, const float* gin, float* gout1, int32_t* redo,
                          int n_roots, const int32_t* gin_fold, int is_signed, void* stream);
/* softmax parameter backward over the last axis: dtheta = W * (dW - sum(W*dW)). */
int ck_param_softmax_bwd(const float* w, const float* dw, float* dtheta, int64_t rows, int len,
                         int accumulate, void* stream);
/* Categorical probs backward: table (F,C+1,K) = transposed log softmax_C(theta (F,K,C));
 * dtheta[f,k,c] = dT[f,c,k] - exp(T[f,c,k]) sum_c' dT[f,c',k]. */
/* ck_param_softmax_bwd (accumulate 0) for a list of tensors in ONE launch.  jobs: DEVICE array, first_block ascending
 * from 0 with job k owning ceil(rows_k / 4) blocks; n_blocks = their total. */
typedef struct ck_softmax_bwd_job {
  const float* w;
  const float* dw;
  float* dtheta;
  int64_t rows;
  int32_t len;
  int32_t first_block;
  int64_t part_stride;  /* n_part > 1: dW is the sum of n_part slots, part_stride floats apart (ck_tail_bwd's dw_part) */
  int32_t n_part;       /* 0 or 1: dw is the gradient itself */
  int32_t reserved;
  float* theta;         /* with `opt` (len == 32 only): the logits, their moments and the (rows, 32) row-major buffer the NEXT forward */
  float* m1;            /* reads softmax(theta') from -- the row's wave updates them in place (all four NULL: gradients only) */
  float* m2;
  float* w_out;
} ck_softmax_bwd_job;
/* opt: NULL, or the DEVICE optimizer state (ck_opt_tick): jobs with `theta` take the optimizer's step on their rows in the launch's
 * epilogue (`optimizer.step()` for these tensors + TorchSoftmaxParameter.forward of the next step, nodes.py:764-772). */
int ck_param_softmax_bwd_batch(const ck_softmax_bwd_job* jobs, int n_jobs, int n_blocks, const struct ck_opt_state* opt, void* stream);
int ck_param_log_table_bwd(const float* table, const float* dtable, float* dtheta, int F, int K, int C,
                           int accumulate, void* stream);
/* Optimiser steps on one flat tensor; grad_scale multiplies the gradient first (e.g. 1/world). Adam
 * follows torch.optim.Adam's defaults semantics (bias-corrected, no weight decay, no amsgrad; the reference's training
 * loop, notebooks/learning-a-circuit.ipynb cell 18).  skip_flag: NULL, or a DEVICE int32: when it is nonzero at launch time
 * -- the step's batch held an illegal value, ck_leaf_walk_fwd / ck_stage_categories -- the launch changes NOTHING
 * (parameters, moments), and Adam adds 1 to *skipped (DEVICE int32, zero-initialised by the caller, or NULL): the bias
 * corrections use step - *skipped, so a skipped step does not count. */
int ck_adam_step(float* p, const float* g, float* m1, float* m2, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, const int32_t* skip_flag, int32_t* skipped,
                 void* stream);
int ck_sgd_step(float* p, const float* g, int64_t n, float lr, float grad_scale, const int32_t* skip_flag, void* stream);
/* TorchTensorDotLayer (optimized.py:289-296) for the partition function of a squared circuit -- a chain of such layers on ONE
 * row, every launch a fixed cost (cirkit_amd/csrc/ck_backward_c.hip): REAL weights; complex_values != 0: complex64 values (offsets
 * in complex numbers, the conventions of ck_sum_lse_bwd_c), 0: fp32 (lse-sum).
 * x = the SUM of the H blocks arena + row_off[f, h] -- the TorchHadamardLayer beneath (inner.py:126-127 in log space) read as a
 * list, never launched -- viewed (B, Kj, Kq); w (F, Kk, Kj); out (F, B, Kq Kk).
 * ck_tensordot2_*: the PAIR of layers a squared sum layer becomes (M' = W M W^T: one over W, one over conj W) in one launch:
 * stage 1 writes mid (F, B, Kq Kk1), stage 2 reads it as (Kq, Kk1) -> out (F, B, Kk1 Kk2) with w2 (F, Kk2, Kq).
 * Backward: the gradient of x is WRITTEN to garena + row_off[f, h] for every h (each factor of the product receives it), in x's
 * layout; gmid <- the gradient of mid; dw (F, Kk, Kj) += (float atomics). */
int ck_tensordot_lse_fwd_h(const float* arena, const int64_t* row_off, int H, const float* w, float* out, int F, int B, int Kj,
                           int Kq, int Kk, int complex_values, void* stream);
int ck_tensordot2_lse_fwd(const float* arena, const int64_t* row_off, int H, const float* w1, float* mid, const float* w2, float* out,
                          int F, int B, int Kj, int Kq, int Kk1, int Kk2, int complex_values, void* stream);
int ck_tensordot_lse_bwd(const float* arena, float* garena, const int64_t* row_off, int H, const float* w, const float* out,
                         const float* gout, float* dw, int F, int B, int Kj, int Kq, int Kk, int complex_values, void* stream);
int ck_tensordot2_lse_bwd(const float* arena, float* garena, const int64_t* row_off, int H, const float* w1, const float* mid,
                          float* gmid, const float* w2, const float* out, const float* gout, float* dw1, float* dw2, int F, int B,
                          int Kj, int Kq, int Kk1, int Kk2, int complex_values, void* stream);
/* Pieces of the squared-circuit loss -mean(2 Re c(x) - Re Z) (symbolic/functional.py:161,259,594 + the reference's training
 * loop) that are neither a layer nor a parameter node, so that a step needs no tensor-library arithmetic:
 * dst[i dst_stride] = src[i src_stride] (real parts of complex values); p[i stride] = value (the seed Re = v, Im = 0);
 * d w (F, K, C) = dtable / table on the gather-table layout (F, C + 1, K) (TorchEmbeddingLayer under the complex log,
 * input.py:258-266, utils.py:32-50; an entry nobody selected gets 0); out = [2 sum_b yc[b stride] - B z[0], B] in fp64. */
int ck_copy_strided_f32(const float* src, float* dst, int64_t n, int64_t src_stride, int64_t dst_stride, void* stream);
int ck_fill_strided_f32(float* p, int64_t n, int64_t stride, float value, void* stream);
int ck_embedding_weight_bwd(const float* table, const float* dtable, float* dw, int F, int C, int K, void* stream);
int ck_squared_ll(const float* yc, int64_t B, int64_t stride, const float* z, double* out, void* stream);
/* The whole backward of an Embedding layer under a logarithm in one launch (K % 32 == 0, (C + 1)(K + 1) + 2 C + 4099 words of
 * LDS; CK_ERR_UNSUPPORTED otherwise): d w[f, k, c] = (sum over the rows b with x[b, scope[f]] = c of gout[f, b, k]) / table[f, c, k],
 * 0 where nobody selected c.  gout_stride: floats between consecutive entries of gout (2: the real parts of a complex64 block);
 * gfold: NULL, or the (B, K) block of gout each fold reads (folds that were multiplied share their gradient: ck_slse_bwd);
 * fold_order: NULL, or the fold each workgroup takes (a permutation: folds that share a block placed 8 workgroups apart run
 * on one XCD at the same time and the second read of the block comes from its L2). */
int ck_embedding_bwd(const float* gout, int gout_stride, const int32_t* gfold, const int32_t* fold_order, const int32_t* xt,
                     const int64_t* scope, const float* table, float* dw, int F, int B, int K, int C, void* stream);
/* Signed-log sum layers (cirkit_amd/csrc/ck_signed.hip): TorchCPTLayer / TorchSumLayer (arity 1) under complex-lse-sum
 * (optimized.py:171-178, inner.py:266-273, semiring.py:441-476) for circuits whose parameters are all REAL -- every value is
 * real, the reference's (log|v|, 0 or pi) is stored as fp32 log|v| plus ONE sign word per row (bit k: unit k negative).
 * A fold's block of 32-unit rows is TILE-NATIVE: Bp = 32 ceil(B / 32) rows, tile t = rows 32 t .. 32 t + 31 in 1024 floats,
 * (row r, unit u) at dword 1024 (r / 32) + 256 (u / 8) + 4 ((r % 32) + 32 ((u / 4) % 2)) + u % 4 -- the MFMA register order,
 * one contiguous KiB per wave instruction; rows past B are padding (written, never meaningful).  arena / signs: the blocks and
 * their sign words; row_off (F, H): float offsets of the children's blocks (multiples of 32; the sign words of a block start
 * at offset / 32, one per row); 32 input units; w (F, Ko, 32) real, Ko = 32 or 1 .. 4; out: (F, Bp, 32) tile-native with
 * sout (F, Bp) (bit o: output o negative), or Ko <= 4: (F, B, Ko) row-major with sout (F, B).  log_table != NULL (Ko = 32): the children are folds of an Embedding
 * layer (layers/input.py:258-266), read as rows of the signed-log form of its weight table -- ck_slse_table: (rows, 32) real
 * weights -> log|w| and a sign word per row, rows = F0 (C + 1) of the gather table -- by the batch values (child_fold /
 * child_var (F, H): fold and variable of each child, xt (D, B) the staged batch); arena / signs / row_off are not read then.
 * Backward, for a loss that reads log|out|: gout + gout_off[f] (gout_off NULL: + f Bp 32, or + f B Ko for Ko <= 4) the real
 * gradient of fold f's output, in the layout of out; gx (F, Bp, 32) tile-native <- the gradient w.r.t. log|product of the
 * children| -- the SAME for each of the H children, who read it there (their gout_off) --, or, for a gathering layer,
 * (F, B, 32) ROW-MAJOR (ck_embedding_bwd's gfold scatters rows); dw (F, Ko, 32) += (float atomics: zero it first). */
int ck_slse_table(const float* table, float* log_table, uint32_t* table_signs, int64_t rows, void* stream);
/* ... or all three tables of an Embedding layer of 32 units from its weight (F, 32, C) (layers/input.py:258-266) in one launch:
 * table (F, C + 1, 32) (the transposed weight, row C the integral row of ones), its signed-log form and sign words (both NULL:
 * the linear table alone -- what ck_leaf_walk_fwd's signed launch reads). */
int ck_slse_tables(const float* weight, float* table, float* log_table, uint32_t* table_signs, int F, int C, void* stream);
int ck_slse_fwd(const float* arena, const uint32_t* signs, const int64_t* row_off, const float* w, float* out, uint32_t* sout,
                int F, int H, int B, int Ko, const float* log_table, const uint32_t* table_signs, const int32_t* child_fold,
                const int32_t* child_var, const int32_t* xt, int C, void* stream);
int ck_slse_bwd(const float* arena, const uint32_t* signs, const int64_t* row_off, const float* w, const float* out,
                const uint32_t* sout, const float* gout, const int64_t* gout_off, float* gx, float* dw, int F, int H, int B, int Ko,
                const float* log_table, const uint32_t* table_signs, const int32_t* child_fold, const int32_t* child_var,
                const int32_t* xt, int C, void* stream);
/* *dst |= *src; *src = 0 (DEVICE int32 flags): turns a per-step validation flag into a sticky one. */
int ck_latch_flag(int32_t* src, int32_t* dst, void* stream);
/* ck_fill_f32 that also hands a validation flag on: *step_flag = *src; if it is nonzero, *sticky |= *src and *src = 0
 * (DEVICE int32 words): the flag a forward raised (ck_leaf_walk_fwd's bad_input) becomes this step's flag -- what the
 * optimizer launch skips on -- and the sticky one in the launch that zeroes the gradient buffers anyway.  opt: NULL, or the
 * DEVICE optimizer state: the launch is this step's ck_opt_tick as well (the flag decides whether the step counts). */
int ck_fill_latch(float* p, int64_t n, float value, int32_t* src, int32_t* step_flag, int32_t* sticky, struct ck_opt_state* opt, void* stream);

/* ---------------------------------------------------------------- training step as job lists (64-unit CP circuits) ---- */
/* The reference trains with autograd through its layer-by-layer forward (notebooks/learning-a-circuit.ipynb cell 18:
 * `loss = -torch.mean(circuit(batch)); loss.backward(); optimizer.step()` over layers/inner.py:126-127, 266-273,
 * semiring.py:383-408, utils.py:10-30, parameters/nodes.py:764-772, 847-862).  For circuits made of 64-unit dense / CP-T /
 * mixing / Hadamard layers (the notebook's QuadGraph CP circuit, BASELINE config 4) a step is a short list of LEVEL launches
 * over JOBS (cirkit_amd/csrc/ck_jobs.hip, built by cirkit_amd/train_jobs.py).  All tables are DEVICE arrays; `pool` is a
 * DEVICE array of block pointers, blocks are (rows, 64) fp32 row-major.  A job's rows may be cut over n_split workgroups
 * (units of the launch); the units of one job share `part` (n_split slots) and `ticket` (zero, zero again afterwards). */
typedef struct ck_opt_state {  /* DEVICE: the optimizer's constants and clock, advanced by ck_opt_tick once per step */
  float lr, b1, b2, eps;
  float bc1, bc2;      /* Adam's bias corrections of THIS step */
  int32_t step;        /* steps taken (skipped ones do not count) */
  int32_t skipped;     /* steps dropped because their batch held an illegal category */
  int32_t skip_now;    /* this step is dropped: the epilogues change nothing */
  int32_t kind;        /* 0 SGD, 1 Adam */
  double b1d, b2d;     /* the betas in double: ck_opt_tick forms bc1 / bc2 = 1 - b^step in double, as torch.optim.Adam does */
} ck_opt_state;
/* One fold of a TorchSumLayer (arity 1) / TorchCPTLayer with 64 inputs and 64 outputs (layers/inner.py:266-273,
 * optimized.py:171-178):  v = sum of the n_in blocks pool[in_off ..] (the Hadamard product of the children in log space),
 * out = log(W exp(v - max v)) + max v.  Backward: G = sum of the n_g blocks pool[g_off ..]; gx = the gradient w.r.t. v;
 * mode 0: dtheta <- dW (64, 64);  mode 1: dtheta <- W (dW - <W, dW>), the gradient of the logits of W = softmax(theta)
 * (nodes.py:764-772);  mode 2: the optimizer's update of theta / m1 / m2 in place and w_out <- softmax(theta'). */
typedef struct ck_sum_job {
  const float* w;
  float* out;
  float* gx;
  float* dtheta;
  float* theta;
  float* m1;
  float* m2;
  float* w_out;
  float* part;
  uint32_t* ticket;
  int32_t in_off, n_in;
  int32_t g_off, n_g;
  int32_t row0, row1;
  int32_t split, n_split;
  int32_t mode;
  int32_t C;           /* with xrow: the number of categories */
  const int32_t* xrow; /* NULL, or the (B) int32 column of the staged batch: the job's ONE input block is a Categorical fold's (C + 1, 64)
                          table and batch row b reads the table row of its category (the input layer's output is never written) */
  /* backward through a MIXING fold without a launch of its own (mix_out != NULL): this job's output is one factor of slot
   * mix_h of a mixing fold with output block mix_out (B, 64) and coefficients mix_w (64, mix_H); the job's gradient list is the
   * MIXING fold's, and G = (sum of the list) * w[:, h] * exp(x_h - mix_out) with x_h = this job's output (recomputed) + the
   * n_partner other factors pool[partner_off ..] of the slot.  mix_dw: NULL, or (64, mix_H) floats whose column mix_h receives
   * d w[:, h] = sum_b G / w[:, h] (one job per slot writes it; ck_jobs_mix_params differentiates the softmax behind w). */
  const float* mix_out;
  const float* mix_w;
  float* mix_dw;
  int32_t partner_off, n_partner;
  int32_t mix_h, mix_H;
  int64_t reserved[3];
} ck_sum_job;
int ck_jobs_sum64_fwd(const ck_sum_job* jobs, int n_units, const float* const* pool, void* stream);
/* waves: 4 or 8 wavefronts per workgroup (a wave takes every waves-th 32-row tile of the unit's rows; 8: one workgroup per CU) */
int ck_jobs_sum64_bwd(const ck_sum_job* jobs, int n_units, const float* const* pool, const ck_opt_state* opt, int waves, void* stream);
/* One fold of a mixing layer (a TorchSumLayer whose weight ends in TorchMixingWeightParameter, nodes.py:847-862): H slots,
 * slot h = the sum of the S blocks pool[in_off + h S ..]; w (64, H) coefficients.  Backward: gx + h gx_stride <- the gradient
 * of slot h; dtheta (64, H) as for ck_sum_job (the softmax runs over h); part slots are 64 x (2 | 4 | 8 | 16) floats (the
 * smallest that holds h_max). */
typedef struct ck_mix_job {
  const float* w;
  float* out;
  float* gx;
  float* dtheta;
  float* theta;
  float* m1;
  float* m2;
  float* w_out;
  float* part;
  uint32_t* ticket;
  int32_t in_off, H;
  int32_t g_off, n_g;
  int32_t row0, row1;
  int32_t split, n_split;
  int32_t mode, S;
  int64_t reserved1;
} ck_mix_job;
/* The parameter step of mixing folds whose backward ran inside their factors' sum jobs (ck_sum_job.mix_out): one workgroup of 64
 * threads per job; `part` (64, H) holds d w (written by those sum jobs), mode as for ck_sum_job. */
int ck_jobs_mix_params(const ck_mix_job* jobs, int n_jobs, const ck_opt_state* opt, void* stream);
int ck_jobs_mix_fwd(const ck_mix_job* jobs, int n_units, const float* const* pool, int h_max, void* stream);
int ck_jobs_mix_bwd(const ck_mix_job* jobs, int n_units, const float* const* pool, int h_max, int64_t gx_stride,
                    const ck_opt_state* opt, void* stream);
/* out <- the sum of n_in blocks of `elems` floats (a Hadamard layer that is kept: layers/inner.py:126-127 in log space; or a
 * gradient that several jobs read). */
typedef struct ck_nsum_job {
  float* out;
  int32_t in_off, n_in;
} ck_nsum_job;
int ck_jobs_nsum(const ck_nsum_job* jobs, int n_jobs, const float* const* pool, int64_t elems, void* stream);
/* The top of the circuit in one launch: R <= 16 scalar sum folds (64 inputs, weight rows w[r]) under the final mixing layer
 * (coefficients c (R), or NULL with R = 1), `out` (B) log-likelihoods, ll <- [sum, B] (fp64; what ck_ll_sum gives), and with
 * gx (R, B, 64) their backward for loss = sum_b seed_b out_b (seed NULL: seed_const, the -1 / batch of the mean NLL).  mode as
 * for ck_sum_job, per weight row (dtheta_w[r] (64)) and for the coefficients (dtheta_c (R)); *bad_flag != 0 -> NaN outputs
 * (ck_poison_outputs).  part: n_wg x 1042 floats, ticket zero. */
typedef struct ck_root_launch {
  const float* const* pool;
  const int32_t* in_off;
  const int32_t* n_in;
  const float* const* w;
  const float* c;
  float* out;
  float* gx;
  const float* seed;
  double* ll;
  float* part;
  uint32_t* ticket;
  float* const* dtheta_w;
  float* dtheta_c;
  float* const* theta_w;
  float* const* m1_w;
  float* const* m2_w;
  float* const* w_out;
  float* theta_c;
  float* m1_c;
  float* m2_c;
  float* c_out;
  const ck_opt_state* opt;
  const int32_t* bad_flag;
  float seed_const;
  int32_t R, B, mode, n_wg;
  int32_t S; /* blocks per scalar fold: every fold's list pool[in_off[r] .. + S) has this length (pad with a block of zeros) */
} ck_root_launch;
int ck_jobs_root(const ck_root_launch* a, void* stream);
/* One fold of a TorchCategoricalLayer with 64 units and probs = softmax(theta) (layers/input.py:399-412, nodes.py:764-783), backward:
 * G = the sum of the n_g blocks pool[g_off ..], x the (B) int32 column of the staged batch (-1 = marginalised), theta (64, C) the
 * logits, table (C + 1, 64) the log-probabilities the forward gathered from.  mode 1: dtheta (64, C) <- the gradient of the
 * logits; mode 2: the optimizer's update of theta_out / m1 / m2 in place and table_out <- the next step's table (rows < C). */
typedef struct ck_cat_job {
  const int32_t* x;
  const float* theta;
  const float* table;
  float* dtheta;
  float* theta_out;
  float* m1;
  float* m2;
  float* table_out;
  int32_t g_off, n_g;
  int32_t mode, reserved;
} ck_cat_job;
int ck_jobs_cat_bwd(const ck_cat_job* jobs, int n_jobs, const float* const* pool, int B, int C, const ck_opt_state* opt, void* stream);
/* One fold of a TorchGaussianLayer with 64 units (layers/input.py:661-670), backward: G = the sum of the n_g blocks pool[g_off ..],
 * x the (B) column of the staged batch (NaN = marginalised); mode 1: dmean / dsd <- the gradients of the tensors behind mean and
 * stddev (has_ss: stddev = vmin + (vmax - vmin) sigmoid(theta), nodes.py:698-699); mode 2: the optimizer's update in place,
 * mean_out (or NULL: mean IS its tensor) / sd_out <- the values of the next step. */
typedef struct ck_gauss_job {
  const float* mean;
  const float* stddev;
  const float* x;
  float* dmean;
  float* dsd;
  float* th_mean;
  float* m1_mean;
  float* m2_mean;
  float* th_sd;
  float* m1_sd;
  float* m2_sd;
  float* mean_out;
  float* sd_out;
  int32_t g_off, n_g;
  float vmin, vmax;
  int32_t has_ss, mode;
} ck_gauss_job;
int ck_jobs_gauss_bwd(const ck_gauss_job* jobs, int n_jobs, const float* const* pool, int B, const ck_opt_state* opt, void* stream);
/* The optimizer step p <- p - ... on one flat range with the constants and the clock of a DEVICE ck_opt_state (m1 / m2 may be
 * NULL for SGD): what ck_adam_step / ck_sgd_step do, recordable (no step count in the launch) and skipped with the state. */
/* g2 (may be NULL): a second gradient buffer, the step runs on g + g2 -- a squared circuit's step adds the gradient of Z,
 * accumulated in a buffer of its own beside c's launches, where the optimizer reads it instead of in an axpy launch before. */
int ck_opt_step_range(float* p, const float* g, const float* g2, float* m1, float* m2, int64_t n, const ck_opt_state* opt, void* stream);
/* Once per step, before the backward launches: *flag != 0 (the forward's validation flag) -> this step is dropped (skip_now = 1,
 * skipped += 1, *sticky |= *flag, *flag = 0); else step += 1 and the bias corrections of this step.  flag / sticky may be NULL. */
int ck_opt_tick(ck_opt_state* state, int32_t* flag, int32_t* sticky, void* stream);

/* ---------------------------------------------------------------- reductions --------------- */
/* Sum of B log-likelihoods (stride in floats between consecutive rows) into out_dev[0] (fp64) and
 * the row count into out_dev[1]; the pair feeds the one RCCL all-reduce of the data-parallel NLL
 * (notebooks/learning-a-circuit.ipynb cell 20: `circuit(batch).sum()`).  out_dev is overwritten. */
int ck_ll_sum(const float* ll, int64_t B, int64_t stride, double* out_dev, void* stream);

/* ---------------------------------------------------------------- program (launch list) ---- */
/* A program records the exact sequence of the calls above for one (plan, batch size) and replays
 * it with ONE host call -- the native replacement for the per-batch Python interpreter loop
 * TorchDiAcyclicGraph.evaluate (graph/modules.py:303-335).  While recording, every ck_* call made
 * on the recording thread is appended instead of launched.  With use_graph != 0 the first launch
 * captures the sequence into a hipGraph and later launches replay the instantiated graph. */
typedef struct ck_program ck_program;
int ck_program_begin(ck_program** out);
int ck_program_end(ck_program* prog);
int ck_program_num_ops(const ck_program* prog);
int ck_program_launch(ck_program* prog, int use_graph, void* stream);
/* Per-launch inputs: a program has 4 input cells; a call recorded with a program-input index (ck_leaf_walk_fwd's x_input,
 * ck_tail_params_fwd's ll_cell) reads the pointer stored in that cell at every eager replay.  Set them before ck_program_launch; the memory must stay
 * valid until the replayed launches have run (stream order).  This is what lets the recorded forward -- the replacement
 * of TorchCircuit.forward(x), circuits.py:242-278 -- take a different batch per call without staging it. */
int ck_program_set_input(ck_program* prog, int index, const void* ptr);

/* ---- complex-lse-sum on linear (re, im) tiles (cirkit_amd/csrc/ck_clin.hip) --------------------------------------------
 * TorchEmbeddingLayer.forward (layers/input.py:258-266) -> TorchCPTLayer / TorchSumLayer (arity 1) of 32 units
 * (layers/optimized.py:171-178, inner.py:266-273) under ComplexLSESumSemiring.apply_reduce (semiring.py:441-476) for circuits
 * whose VALUES are complex (complex Embedding weights and / or complex sum weights): between launches a value is
 * (re + i im) 2^e -- tile-native blocks (fold, tile) -> 1024 dwords re + 1024 dwords im in the MFMA register order, one int32
 * exponent per (fold, row) -- a product of children is a complex multiply + a power-of-two renormalisation, a sum two (real
 * weights) or four (complex weights) fp32 MFMA chains; the reference's (log|v|, arg v) pairs are written only by the layer a
 * circuit outputs (`out_log`).  Results equal the reference's to fp32 rounding; phases modulo 2 pi.
 * ck_clin_table: w (F, 32, C) fp32 or complex64 -> table (F, C + 1, 32 | 64) = [re 32 | im 32 if complex], each row divided
 *   by 2^table_e[f, c] (largest |re|, |im| in [0.5, 1)); row C = the sum over the categories (what a negative
 *   category selects; the reference's Embedding layer has no integral and the host API refuses marginalisation through it).
 * ck_clin_leaf_fwd: `depth` (1..4) CP-T levels over the table in one launch.  xt (D, B) int32 staged categories (negative:
 *   row C) -- or xt NULL and the caller's (B, n_vars) int64 batch itself, as ck_leaf_walk_fwd takes it: x_rows, or x_input >= 0
 *   = the program input cell that holds its pointer at replay (ck_program_set_input); the launch then validates it: a category
 *   >= C makes its ROW NaN and raises *bad_flag (bad_flag NULL: clamped silently), i.e. x.long() + the index check of
 *   layers/input.py:258-266 without a staged copy.  leaf_fold / leaf_var (R, 2^depth) the Embedding fold and variable of every leaf in walk order, wnode
 *   (R, 2^depth - 1) DEVICE array of weight-matrix addresses (32, 32) fp32 or complex64 row-major in the order the depth-first
 *   walk contracts them (after leaf i: levels 1 .. number of trailing one bits of i); out (R, tiles, 2048), out_e (R, tiles 32).
 * ck_clin_layer_fwd: one layer of F folds, H children each at float offset child_off[f, h] (tile 0 of the child fold's
 *   block) in `lin` and child_eoff[f, h] in `lin_e`; w (F) DEVICE array of weight-matrix addresses (Ko, 32), Ko <= 32;
 *   out / out_e (tile blocks, may be NULL) and / or out_log (F, B, Ko) complex64. */
/* ck_clin_tail_fwd: the few-fold top of a circuit in one launch -- a workgroup of eight waves per 32-row tile walks the
 * n_levels layers in order (folds level_off[l] .. level_off[l + 1] of `folds`), a wave per fold, blocks through `lin` as between
 * launches. */
typedef struct ck_clin_tail_fold {
  int64_t co[2], ce[2];  /* children (H <= 2): float offset of the child fold's tile 0 in lin, int32 offset in lin_e */
  const float* w;        /* (Ko, 32) fp32 or complex64 */
  int64_t out, oute;     /* this fold's block: offsets of its tile 0, -1: not kept */
  float* out_log;        /* this fold's (B, Ko) complex64 rows (log|v|, arg v), or NULL */
  int32_t H, Ko;
} ck_clin_tail_fold;
int ck_clin_tail_fwd(float* lin, int32_t* lin_e, const void* folds, const int32_t* level_off, int n_levels, int w_is_complex, int B,
                     void* stream);
int ck_clin_table(const float* w, int w_is_complex, float* table, int32_t* table_e, int F, int C, void* stream);
int ck_clin_leaf_fwd(const float* table, const int32_t* table_e, const int32_t* xt, const int64_t* x_rows, int x_input, int n_vars,
                     int32_t* bad_flag, const int32_t* leaf_fold, const int32_t* leaf_var, const float* const* wnode, int w_is_complex,
                     int table_is_complex, float* out, int32_t* out_e, int R, int depth, int B, int C, void* stream);
int ck_clin_layer_fwd(const float* lin, const int32_t* lin_e, const int64_t* child_off, const int64_t* child_eoff, const float* const* w,
                      int w_is_complex, float* out, int32_t* out_e, float* out_log, int F, int H, int Ko, int B, void* stream);

/* ---- the exchange step -------------------------------------------------------------------------
 * The path shards on the batch axis: a rank evaluates its rows and ONE SUM all-reduce carries the [sum log p, count] pair
 * of a forward (or the flat gradient buffer of a training step) over RCCL / xGMI.  The reference has no distributed code
 * (SURVEY.md section 5): these replace the torch.distributed.all_reduce a user would write around TorchCircuit.forward
 * (circuits.py:242-278).  The collective is enqueued on the caller's stream by the library -- no host work between the
 * launch that writes the pair and the exchange -- and, inside ck_program_begin / _end, becomes a step of the recorded list.
 * RCCL is bound at run time (the copy already mapped into the process, else `librccl_path`, else the system's); without it
 * every ck_comm_* call returns CK_ERR_UNSUPPORTED.  Bootstrap: rank 0 calls ck_comm_unique_id, hands the 128 bytes to the
 * other ranks by any means (torch.distributed's store, a file, MPI), every rank calls ck_comm_init (collective: returns
 * when all `world` ranks have called it).  One communicator per process and device. */
typedef struct ck_comm ck_comm;
int ck_comm_load(const char* librccl_path);
int ck_comm_unique_id(void* id128);
int ck_comm_init(const void* id128, int rank, int world, int device, ck_comm** out);
/* in place, SUM over the ranks, ordered on `stream` like any launch of this library */
int ck_comm_all_reduce_f64(ck_comm* comm, double* buf, int64_t n, void* stream);
int ck_comm_all_reduce_f32(ck_comm* comm, float* buf, int64_t n, void* stream);
/* The same on the communicator's OWN stream, ordered behind what `after_stream` holds so far: nothing on `after_stream` waits
 * for it -- the [sum, count] pair of an evaluation step is exchanged beside the next step's launches.  ck_comm_wait: `stream`
 * waits (on the device, not the host) for every collective issued this way so far; an event that was never recorded counts
 * as complete. */
int ck_comm_all_reduce_async_f64(ck_comm* comm, double* buf, int64_t n, void* after_stream);
int ck_comm_wait(ck_comm* comm, void* stream);
/* out = [rank, world, device]; origin: which librccl was bound */
int ck_comm_info(const ck_comm* comm, int32_t out[3], char* origin, int origin_len);
int ck_comm_destroy(ck_comm* comm);

/* Lend a device scratch buffer to the launches this THREAD issues or records from now on (NULL, 0: take it back).  It
 * must be ZERO when lent; the part that has to stay zero (ticket counters behind the first CUs x 3 x (32 KiB + 512 B)) is zero
 * again after every launch that used it; launches that share it must be ordered (one stream, or one recorded program).  Used
 * by the stream-K Tucker launch of ck_sum_lse_fwd (CK_SUM_KRON, arity 2, 32 / 64 units): partial accumulators of tiles that
 * straddle workgroups (and, on logits, their row maxima and sums); without a workspace of CUs x 3 x (32 KiB + 512 B) +
 * 4 bytes per (fold, 32 outputs, 128 rows) tile those layers take one workgroup per tile. */
int ck_set_workspace(void* ptr, int64_t bytes);
int ck_program_destroy(ck_program* prog);

#ifdef __cplusplus
}
#endif
#endif /* CIRKIT_HIP_H */
