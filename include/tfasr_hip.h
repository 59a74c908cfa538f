/*
 * tfasr_hip.h — C ABI of libtfasr_hip.so: the MI355X (gfx950) hot path of TensorFlowASR's
 * Conformer-Transducer training / greedy inference.
 *
 * Conventions (mirroring the only native plug-in boundary the reference has, warp-transducer's
 * `compute_rnnt_loss(acts, grads, labels, label_lengths, input_lengths, alphabet_size, minibatch,
 * costs, workspace, options)`, reached from tensorflow_asr/losses/impl/rnnt.py:8,55 and built by
 * scripts/install_rnnt_loss.sh:12-49):
 *   - plain `extern "C"` functions, raw DEVICE pointers + explicit sizes, no torch / HIP C++ types;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); nothing synchronises;
 *   - no allocation inside: every scratch buffer is caller-owned, sized by a `*_workspace_size` query;
 *   - every entry returns a tfasr_status_t; `tfasr_status_string` decodes it;
 *   - re-entrant per stream, no assumption about the process-wide current device beyond "the pointers and the stream belong
 *     to the device that is current on this thread".  State the library keeps: (i) the three switches of the table below that it reads itself
 *     (once, function-local statics); (ii) the block executor's internal second stream + events per device (tfasr_block_io.wgrad_slot), the
 *     persistent-LSTM policy (tfasr_lstm_set_persist), and one flag set around a grouped launch that shares the chip with another stream;
 *     (iii) two measurement aids: a host-side launch counter (tfasr_launch_count) and the optional event record of tfasr_block_wgrad_probe -
 *     all of them assume what the rest of the design assumes anyway: ONE host thread queues the launches of a device.  Results never
 *     depend on any of it.  (The Python package also sets GPU_MAX_HW_QUEUES=8 at import unless the user chose a value - the HIP
 *     runtime's own switch.)
 *
 *     Environment switches (all of them; A/B aids, never needed for a result).  Read by the library:
 *       TFASR_FFN_FUSED=0        FFModule forward as LayerNorm + two products instead of tfasr_ffn_fused_fwd
 *       TFASR_LSTM_PERSIST=0|1   force the per-step / the persistent LSTM kernels (default: the caller's tfasr_lstm_set_persist policy)
 *       TFASR_DENSE_LN=0         (block executor) Dense data gradient and LayerNorm backward as two launches instead of tfasr_dense_ln_bwd
 *     Read by the Python host (tensorflowasr_amd/, bench.py):
 *       TFASR_LIB=<path>         load another build of this library (same-box A/B of two builds; probe builds)
 *       TFASR_HEAD_PAD=0, TFASR_FILTER_PAD=0   logical head size / subsampling filters instead of the 64-multiples
 *       TFASR_NATIVE_BLOCK=0     per-kernel Python path instead of the native block executor
 *       TFASR_ATTN_UNFUSED=1     unfused attention (score matrices in HBM; what f32 models always take)
 *       TFASR_CONV2_IM2COL=1     im2col route of the subsampling's second convolution
 *       TFASR_NO_PRED_STREAM=1, TFASR_PRED_SLICES=<n>   prediction network in line / in n slices between the encoder blocks
 *       TFASR_WGRAD_STREAM=0|1, TFASR_BLOCK_HOIST=0|1, TFASR_DEFER_SIDE=0, TFASR_FRONT_EARLY=0   stream placement of the weight gradients,
 *                                the hoisted per-block launches, the deferred small gradients, the front end (DESIGN.md section 2 "Streams")
 *       TFASR_JOINT_RECOMPUTE=1  joint + loss without materialised lattice logits (bench.py reports both)
 *       TFASR_DECODE_PRECISION=f32|bf16   encoder of the greedy search (token-exact f32 twin / training kernels)
 *       TFASR_DP_GRAD_WIRE=f32|bf16, TFASR_DP_FORCE_SPLIT=1   data-parallel gradient wire type; the world > 1 block route at one rank
 *       TFASR_BENCH_HOST=1, TFASR_BENCH_SECTIONS=1, TFASR_BENCH_STUB=1   bench.py: host enqueue time, per-section timers, launch-logic test
 *     Compile-time probe defines (tools/build_probe_lib.sh, tools/hwprobe/): TFASR_ATTN_TIMING, TFASR_GEMM_TIMING, TFASR_FFN_TIMING,
 *     TFASR_DECODE_TIMING, TFASR_DLN_TIMING (shader-clock stamps), TFASR_GLDS_BUILTIN, TFASR_QT_NT=0 (builtin LDS-DMA / plain dS stores).
 *   - `dtype`: storage type of activation tensors, TFASR_F32 or TFASR_BF16 (raw 16-bit bfloat16).
 *     All arithmetic accumulates in f32. Parameters, gradients and optimizer state are always f32.
 */
#ifndef TFASR_HIP_H_
#define TFASR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFASR_ABI_VERSION 43

typedef enum {
  TFASR_STATUS_SUCCESS = 0,
  TFASR_STATUS_INVALID_VALUE = 1,
  TFASR_STATUS_EXECUTION_FAILED = 2,
  TFASR_STATUS_UNSUPPORTED = 3
} tfasr_status_t;

typedef enum { TFASR_F32 = 0, TFASR_BF16 = 1 } tfasr_dtype_t;

const char* tfasr_status_string(int status);
/* ABI version of this header (bumped on any signature change). */
int tfasr_abi_version(void);
/* Kernel launches this library has queued since it was loaded (host-side count; every kernel goes through one launch macro). */
size_t tfasr_launch_count(void);

/* ------------------------------------------------------------------------------------------------
 * RNN-T loss  (replaces warprnnt_tensorflow.rnnt_loss / rnnt_loss_tf,
 *              tensorflow_asr/losses/impl/rnnt.py:41-58 and :181-331)
 *
 * logits  [B, T, U1, V] (dtype), raw activations (log-softmax is done inside, as the GPU build of
 *         warp-transducer does: impl/rnnt.py:53-54)
 * grads   [B, T, U1, V] (dtype) or NULL (loss only). May alias `logits` (in-place).
 *         grads = grad_scale[b] * dLoss_b/dlogits  (impl/rnnt.py:233-275,318-321); zero outside
 *         the valid lattice t < logit_len[b], u <= label_len[b].
 * labels  [B, U1-1] int32; label_len, logit_len [B] int32 (caller applies the BaseLoss clamp
 *         logit_len = max(logit_len, label_len): losses/base_loss.py:36)
 * grad_scale [B] f32 or NULL (= 1): the upstream dL/dloss_b (e.g. 1/B for the Keras
 *         sum_over_batch_size reduction, rnnt_loss.py:34)
 * costs   [B] f32: loss_b = -log p(labels_b | x_b) = -beta[b,0,0]  (impl/rnnt.py:277)
 * blank must be 0 (losses/base_loss.py:24).
 * ---------------------------------------------------------------------------------------------- */
int tfasr_rnnt_loss_workspace_size(int B, int T, int U1, int V, size_t* bytes);
int tfasr_rnnt_loss(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                    const int32_t* logit_len, const float* grad_scale, int B, int T, int U1, int V,
                    int blank, int dtype, float* costs, void* workspace, size_t workspace_bytes,
                    void* stream);
/* PACKED lattice: only the valid nodes exist.  Utterance b owns rows [cell_off[b], cell_off[b+1]) of logits/grads
 * [total_cells, V], laid out row-major (t, u) with (label_len[b]+1) columns and logit_len[b] rows; cell_off [B+1] int64
 * on the device.  Same loss/gradient as the dense entry (padded nodes carry zero gradient there, impl/rnnt.py:218-224),
 * at sum_b T_b*U1_b instead of B*T*U1 rows.  Workspace: tfasr_rnnt_loss_workspace_size(1, total_cells, 1, V). */
/* The vocabulary id each lattice row's label transition emits (labels[b, u] for u < label_len[b], else -1), for the GEMM
   epilogue's `row_label`; rows follow the packed (cell_off != NULL) or dense [B,T,U1] lattice order. */
/* tfasr_rnnt_loss_packed_stats without logits: costs and, instead of the gradient tensor, the per-row coefficients from which a
 * re-computed logit tile becomes the gradient (tfasr_gemm_args.rgrad_coef): coef [4 * total_cells] f32. */
int tfasr_rnnt_loss_packed_coef(const int32_t* labels, const int32_t* label_len, const int32_t* logit_len, const float* grad_scale,
                                const long* cell_off, long total_cells, const float* lse_part, int lse_parts, const float* pick, int B, int T,
                                int U1, int V, int blank, float* costs, float* coef, void* workspace, size_t workspace_bytes, void* stream);
int tfasr_rnnt_row_labels(const int32_t* labels, const int32_t* label_len, const int32_t* logit_len, const long* cell_off,
                          long total_cells, int B, int T, int U1, int V, int32_t* row_label, void* stream);
/* tfasr_rnnt_loss_packed whose log-softmax statistics were already produced by the projection GEMM's epilogue (tfasr_gemm_args
   lse_part / pick): the first pass over the logits (max / sum-exp / blank and label gathers) is skipped. */
int tfasr_rnnt_loss_packed_stats(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                                 const int32_t* logit_len, const float* grad_scale, const long* cell_off, long total_cells,
                                 const float* lse_part, int lse_parts, const float* pick, int B, int T, int U1, int V, int blank,
                                 int dtype, float* costs, void* workspace, size_t workspace_bytes, void* stream);
int tfasr_rnnt_loss_packed(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                           const int32_t* logit_len, const float* grad_scale, const long* cell_off, long total_cells, int B,
                           int T, int U1, int V, int blank, int dtype, float* costs, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CTC loss (CtcLoss.call -> tf.nn.ctc_loss(logits_time_major=False, blank_index=0), losses/ctc_loss.py:47-66) and
 * CTC greedy decoding (tf.nn.ctc_greedy_decoder(merge_repeated=True), models/ctc/base_ctc.py:102-124).
 * logits [B,T,V] raw activations, labels [B,U] dense, costs [B] = -log p(labels|x); grads (may alias logits) =
 * grad_scale[b] * dcost_b/dlogits, zero for t >= logit_len[b].  2U+1 <= 1024.
 * greedy: tokens [B,T] (blank padded), tokens_len [B]; workspace_argmax [B*T] int32.
 * ---------------------------------------------------------------------------------------------- */
int tfasr_ctc_loss_workspace_size(int B, int T, int U, int V, size_t* bytes);
int tfasr_ctc_loss(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                   const int32_t* logit_len, const float* grad_scale, int B, int T, int U, int V, int blank, int dtype,
                   float* costs, void* workspace, size_t workspace_bytes, void* stream);
int tfasr_ctc_greedy_decode(const void* logits, const int32_t* logit_len, int32_t* workspace_argmax, int32_t* tokens,
                            int32_t* tokens_len, int B, int T, int V, int blank, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM family (Dense / EinsumDense / pointwise Conv1D / attention products / joint vocab projection:
 * keras Dense sites conformer.py:72-87, multihead_attention.py:628-637,654, base_transducer.py:238-293)
 *
 *   D[b] = epilogue( alpha * op(A[b]) x op(B[b]) )          b = b1*nb2 + b2 (two-level batch)
 *   op(A) is [M,K]: trans_a==0 -> A stored [M,K] (lda), trans_a==1 -> A stored [K,M] (lda)
 *   op(B) is [K,N]: trans_b==0 -> B stored [K,N] (ldb), trans_b==1 -> B stored [N,K] (ldb)
 *   epilogue: v = alpha*acc + bias[n]; if (prez) prez = v; v = act(v);
 *             if (dact_z) v *= dact'(dact_z[m,n]);
 *             if (drop_p > 0) v = keep(drop_seed, m*ldd+n) ? v/(1-drop_p) : 0      (keras Dropout, conformer.py:80-87)
 *             if (res) v = res[m,n] + beta*v;
 *             D = v   (out_f32 ? f32 : dtype);  accumulate!=0 -> atomicAdd into f32 D (split-K legal)
 * A, B, res, dact_z, prez are `dtype`; bias is f32.
 * ---------------------------------------------------------------------------------------------- */
/* TANH_OUT is a `dact` only: dact_z holds tanh's OUTPUT h (not its argument), the factor is 1 - h^2 (the joint network keeps h) */
typedef enum { TFASR_ACT_NONE = 0, TFASR_ACT_SWISH = 1, TFASR_ACT_TANH = 2, TFASR_ACT_SIGMOID = 3, TFASR_ACT_TANH_OUT = 4,
               TFASR_ACT_FACTOR = 5 /* dact only: dact_z holds the derivative factor itself (tfasr_ffn_fused_fwd2) */ } tfasr_act_t;

typedef struct {
  const void* A; const void* B; void* D;
  const float* bias;       /* [N] or NULL */
  const void* res;         /* [M,N] (ldd) or NULL */
  const void* dact_z;      /* [M,N] (ldd) or NULL: multiply by act'(z) (backward of a fused activation) */
  void* prez;              /* [M,N] (ldd) or NULL: also store the pre-activation */
  int M, N, K;
  int lda, ldb, ldd;
  int trans_a, trans_b;
  int nb1, nb2;            /* batch counts (>=1) */
  long sA1, sA2, sB1, sB2, sD1, sD2; /* element strides per batch level */
  float alpha, beta;
  int act;                 /* tfasr_act_t applied to v */
  int dact;                /* tfasr_act_t whose derivative multiplies v when dact_z != NULL */
  int dtype;               /* tfasr_dtype_t of A,B,res,dact_z,prez and of D unless out_f32 */
  int out_f32;             /* D is f32 */
  int accumulate;          /* D += (atomic, requires out_f32) */
  int split_k;             /* >=1; >1 requires accumulate */
  float drop_p;            /* dropout rate applied to v (0 = off); the mask is a pure function of (drop_seed, element index) */
  long drop_seed;
  float* ws;               /* optional split-K workspace (f32, >= split_k*M*N elements), NULL = reduce with atomics.  With it the
                            * k-slices store their partial tiles with plain vector stores and a second small kernel sums them
                            * into D: the f32 atomics of a [256,1024] weight gradient (4 M of them) otherwise cost more than
                            * its MFMAs.  Used only when accumulate != 0, split_k > 1 and nb1*nb2 == 1. */
  long ws_elems;
  float* colsum;           /* optional [N] f32, only with accumulate != 0, trans_a == 1, trans_b == 0, nb1*nb2 == 1 (a Dense layer's
                            * weight gradient x^T dy): colsum[n] += alpha * sum_k B[k, n], i.e. the BIAS gradient, produced by the
                            * same launch (one extra all-ones MFMA row in the first row of tiles) instead of a second pass over dy */
  /* Optional log-softmax statistics of the OUTPUT rows, produced by the epilogue from the f32 accumulators (the joint's vocabulary
     projection feeding the RNN-T loss: base_transducer.py:291 -> impl/rnnt.py:211): per row m and per 64-column slice c of the
     row, lse_part[(m*lse_parts + c)*2 + {0,1}] = (max, sum exp(x - max)) over that slice; pick[2m] = x[m,0] (blank logit),
     pick[2m+1] = x[m, row_label[m]] (label logit; untouched when row_label[m] < 0).  lse_parts must be ceil(N/128)*2.
     Only the bf16 fast path (plain NN product + bias, 128-wide tiles) implements it: otherwise tfasr_gemm returns UNSUPPORTED. */
  float* lse_part; int lse_parts; const int32_t* row_label; float* pick;
  /* Optional K-segments (bf16 fast path, plain or bias epilogue, no split-K, K % 64 == 0, seg_k % 64 == 0, K / 64 <= 64): the k
     range [s*seg_k, (s+1)*seg_k) reads op(A) from A + seg_a_off[s] and (when seg_b_off != NULL) op(B) from B + seg_b_off[s]
     (element offsets, DEVICE arrays of K / seg_k entries) instead of a contiguous K: a convolution tap = the same rows shifted. */
  const long* seg_a_off; const long* seg_b_off; int seg_k;
  /* Joint network without materialised lattice logits (SURVEY section 7 step 8 / 8(d) "recompute" variant; base_transducer.py:280-293 +
     losses/impl/rnnt.py:211-278).  (1) With lse_part set, D may be NULL: the projection's epilogue emits only the row statistics.
     (2) rgrad_coef != NULL: the product is the RE-computed logit tile and the epilogue turns it into the loss gradient before the
     store, D[m, v] = exp2(x * log2(e) - c.x) * c.y + [v == 0] c.z + [v == row_label[m]] c.w with c = rgrad_coef[4m .. 4m+3]
     (tfasr_rnnt_loss_packed_coef: c.x = lse * log2(e), c.y = -(g_blank + g_label) * scale, c.z = g_blank * scale, c.w = g_label * scale).
     Only the 256-row bf16 kernel (plain NN product + bias) implements both: otherwise UNSUPPORTED. */
  const float* rgrad_coef;
  /* Optional BatchNorm backward statistics of the OUTPUT (bf16 fast path, plain NT product D = alpha A B^T, 64-column tiles, N % 8 == 0;
     anything else: UNSUPPORTED and nothing is launched): the product is the gradient w.r.t. swish(BatchNorm(x)) (ConvModule,
     conformer.py:305-333) and the epilogue adds what tfasr_bn_bwd_stats would take of it in a second pass - per channel c
     sum dz and sum dz * xhat with dz = D * swish'(x * fin[2N + c] + fin[3N + c]), xhat = (x - fin[c]) * fin[N + c] - into
     bns_out[copy][2][N] (workgroup i adds into copy i % bns_copies; the consumer adds the copies up).  bns_x = the BatchNorm input
     [M, N] (compute dtype, row stride ldd). */
  const void* bns_x; const float* bns_fin; float* bns_out; int bns_copies;
  int bns_c;  /* > 0: the N columns are (position, channel) pairs, channel = column % bns_c (bns_fin [4][bns_c], bns_out [copies][2][bns_c]); 0: N channels */
} tfasr_gemm_args;

int tfasr_gemm(const tfasr_gemm_args* args, void* stream);
/* n independent products queued as ONE launch when all of them are bf16 weight gradients (trans_a, !trans_b, accumulate into
   f32, no batch, no epilogue terms; split_k is chosen for the group): the Dense-layer gradients of one Conformer block
   (models/encoders/conformer.py:101-109,209-239,366-377 under keras autodiff).  Anything else is launched one by one, in order. */
int tfasr_gemm_group(const tfasr_gemm_args* args, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (keras LayerNormalization eps 1e-3: conformer.py:59-64, base_transducer.py:88-93) and
 * BatchNorm in training mode (keras BatchNormalization(synchronized=True), momentum .99, eps 1e-3:
 * conformer.py:327-333, subsampling.py:197-203).  x,y,dy,dx are [rows, C] `dtype`; parameters,
 * statistics and gradients f32.  dgamma/dbeta/stats are ACCUMULATED (atomicAdd): zero them first.
 *   bn: stats[2C] = (sum x, sum x^2) -> (all-reduce across ranks for sync-BN) -> finalize ->
 *       fin[4C] = (mean, rstd, scale, shift); y = act(x*scale+shift).  Backward: bstats[2C] =
 *       (sum dz, sum dz*xhat) -> (all-reduce) -> dx; dgamma = bstats[C:2C], dbeta = bstats[0:C].
 * ---------------------------------------------------------------------------------------------- */
int tfasr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        long rows, int C, float eps, int dtype, void* stream);
/* FFModule.call forward in ONE launch (tensorflow_asr/models/encoders/conformer.py:101-109 + models/layers/residual.py:58-62):
 *     y = x + res_factor * dropout2( dropout1( swish( LN(x) W1 + b1 ) ) W2 + b2 )
 * x, y, ln [rows, d]; W1 [d, F], W2 [F, d] (row-major, compute dtype); gamma / beta / b1 / b2 f32.  Saved for the backward exactly as
 * the three-launch route (tfasr_layernorm_fwd + two tfasr_gemm) writes them: ln = LN(x), mean / rstd [rows], z = LN(x) W1 + b1 (may be
 * NULL: not saved), h = dropout1(swish(z)) (may be NULL), same dropout hash and element indices (seed1 on [rows, F], seed2 on [rows, d]).
 * The F-wide hidden activation is never a GEMM operand in HBM.  TFASR_STATUS_UNSUPPORTED outside bf16 / d = 256 / F % 64 == 0 /
 * 16-byte aligned pointers (the caller then takes the three-launch route). */
int tfasr_ffn_fused_fwd(const void* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                        const float* b2, void* y, void* ln, float* mean, float* rstd, void* z, void* h, long rows, int d, int F,
                        float ln_eps, float res_factor, float drop_p, long drop_seed1, long drop_seed2, int dtype, void* stream);
/* z_factor != 0: `z` receives the backward's factor g = swish'(LN(x) W1 + b1) * mask1 / (1 - p) (compute dtype) instead of the
 * pre-activation: the data gradient of the second Dense layer is then tfasr_gemm with dact_z = g, dact = TFASR_ACT_FACTOR and NO dropout
 * term (one multiply per element in its epilogue instead of swish' and the dropout hash). */
int tfasr_ffn_fused_fwd2(const void* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                         const float* b2, void* y, void* ln, float* mean, float* rstd, void* z, int z_factor, void* h, long rows, int d, int F,
                         float ln_eps, float res_factor, float drop_p, long drop_seed1, long drop_seed2, int dtype, void* stream);
/* LayerNormalization + Dense in ONE launch (the head of MHSAModule / ConvModule, conformer.py:59-64 + the fused q/k/v projection
 * multihead_attention.py:628-637 / the first pointwise conv convolution.py:159-228): out [rows, N] = LN(x) W + b, ln / mean / rstd
 * stored as tfasr_layernorm_fwd writes them (row sums in another order: single-ulp differences); out bitwise tfasr_gemm on that ln.
 * UNSUPPORTED outside bf16 /
 * d = 256 / N % 64 == 0 / 128 <= N <= 1024 / 16-byte aligned pointers. */
int tfasr_ln_dense_fwd(const void* x, const float* gamma, const float* beta, const void* W, const float* b, void* out, void* ln, float* mean,
                       float* rstd, long rows, int d, int N, float ln_eps, int dtype, void* stream);
int tfasr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        const void* add, void* dx, float* dgamma, float* dbeta, long rows, int C, int dtype,
                        void* stream);
/* tfasr_layernorm_bwd with a second output: dx_dropped = tfasr_dropout(dx, drop_p, drop_seed) (the gradient entering the NEXT
   module's dropped branch in backward order), written by the same kernel instead of a separate pass over dx.
   dx_dropped == NULL: identical to tfasr_layernorm_bwd. */
int tfasr_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                             const void* add, void* dx, float* dgamma, float* dbeta, void* dx_dropped, float drop_p,
                             long drop_seed, long rows, int C, int dtype, void* stream);
/* The same with the gamma / beta gradients left as PER-BLOCK partial sums part[nblk][2C] (plain stores; nblk =
   tfasr_layernorm_bwd_part_blocks(rows, C, dtype), 0 = no such kernel for the shape): the backward's tail is otherwise a chain of
   ~190 same-address atomics per column (~3 us).  tfasr_layernorm_bwd_fold adds the partial sums of up to 8 LayerNorms (part =
   [nsets][nblk][2C], same nblk) to their dgamma[i] / dbeta[i] in one launch, one writer per column. */
int tfasr_layernorm_bwd_part_blocks(long rows, int C, int dtype);
int tfasr_layernorm_bwd_part(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* add,
                             void* dx, float* part, void* dx_dropped, float drop_p, long drop_seed, long rows, int C, int dtype,
                             void* stream);
/* ..._part with an explicit number of partial-sum slots (>= 1; a buffer shared with other producers of LayerNorm partial sums whose slot
   count differs): exactly nblk blocks run, the ones without rows store zeros. */
int tfasr_layernorm_bwd_part_n(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* add,
                               void* dx, float* part, int nblk, void* dx_dropped, float drop_p, long drop_seed, long rows, int C, int dtype,
                               void* stream);
/* Dense data gradient + the LayerNorm backward in front of it in ONE launch (bf16, d = 256, K % 64 == 0; else UNSUPPORTED and the caller
 * keeps tfasr_gemm + tfasr_layernorm_bwd_part_n): the backward of `y = Dense(LayerNorm(x))` heads (FFModule encoders/conformer.py:66-109,
 * the fused q/k/v projection multihead_attention.py:628-637, ConvModule's first pointwise conv convolution.py:159-228):
 *     dln = alpha * dy @ W^T  (never stored);  dx = add + LayerNorm'(dln | x, gamma, mean, rstd);  part[tile] = (sum dln*xhat, sum dln)
 * dy [rows, K], W [d, K] row-major (the Dense kernel as stored, [din = d, dout = K]); x / add / dx / dx_dropped [rows, d]; part =
 * [nblk][2d] partial sums exactly as tfasr_layernorm_bwd_part_n leaves them (slots without rows are zeroed); dx_dropped (optional) =
 * tfasr_dropout(dx, drop_p, drop_seed).  UNSUPPORTED also when rows need more than nblk tiles of 96 rows. */
int tfasr_dense_ln_bwd(const void* dy, const void* W, int K, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* add, void* dx, float* part, int nblk, void* dx_dropped, float drop_p, long drop_seed, long rows,
                       int d, float alpha, int dtype, void* stream);
int tfasr_layernorm_bwd_fold(const float* part, int nsets, int nblk, int C, float* const* dgamma, float* const* dbeta, void* stream);
/* The same for up to 128 sets that lie in different buffers (part[i] = [nblk][2C] of set i; host arrays of device pointers): the
   LayerNorms of every Conformer block of a step in one launch. */
int tfasr_layernorm_bwd_fold_sets(const float* const* part, int nsets, int nblk, int C, float* const* dgamma, float* const* dbeta,
                                  void* stream);
int tfasr_bn_stats(const void* x, float* stats, long rows, int C, int dtype, void* stream);
int tfasr_bn_finalize(const float* stats, float count, const float* gamma, const float* beta, float* fin,
                      float* moving_mean, float* moving_var, float momentum, float eps, int C, int training,
                      void* stream);
/* tfasr_bn_finalize + tfasr_bn_apply_fwd in one launch (row kernel shapes: C % 8 == 0, 64 <= C <= 2048, 256 % (C/8) == 0, else
 * UNSUPPORTED): `fin` and the moving statistics are written as tfasr_bn_finalize would. */
int tfasr_bn_finalize_apply_fwd(const void* x, const float* stats, float count, const float* gamma, const float* beta, float* fin,
                                float* moving_mean, float* moving_var, float momentum, float eps, void* y, long rows, int C, int act,
                                int training, int dtype, void* stream);
/* the same with the statistics spread over `copies` copies [copies][2][C] (summed on the fly) */
int tfasr_bn_finalize_apply_fwd_copies(const void* x, const float* stats, int copies, float count, const float* gamma, const float* beta, float* fin,
                                       float* moving_mean, float* moving_var, float momentum, float eps, void* y, long rows, int C, int act,
                                       int training, int dtype, void* stream);
int tfasr_bn_apply_fwd(const void* x, const float* fin, void* y, long rows, int C, int act, int dtype, void* stream);
int tfasr_bn_bwd_stats(const void* x, const void* dy, const float* fin, float* bstats, long rows, int C, int act,
                       int dtype, void* stream);
/* + the BatchNorm parameter gradients in the same launch: dbeta += grad_scale * bstats[0:C], dgamma += grad_scale * bstats[C:2C]
 * (either may be NULL) */
int tfasr_bn_apply_bwd_grads(const void* x, const void* dy, const float* fin, const float* bstats, float count, void* dx, long rows,
                            int C, int act, float* dgamma, float* dbeta, float grad_scale, int dtype, void* stream);
/* the same with the sums spread over `copies` copies [copies][2][C] (added up on the fly; row-kernel channel counts only, else UNSUPPORTED) */
int tfasr_bn_apply_bwd_grads_copies(const void* x, const void* dy, const float* fin, const float* bstats, int copies, float count, void* dx,
                                    long rows, int C, int act, float* dgamma, float* dbeta, float grad_scale, int dtype, void* stream);
int tfasr_bn_apply_bwd(const void* x, const void* dy, const float* fin, const float* bstats, float count, void* dx,
                       long rows, int C, int act, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pointwise / small-reduction stages (see csrc/elementwise.hip for the reference sites).
 * ---------------------------------------------------------------------------------------------- */
int tfasr_cast(const void* src, void* dst, long n, int src_dtype, int dst_dtype, void* stream);
/* y[i] = keep(seed, i) ? x[i]/(1-p) : 0 — the same counter-based mask as the GEMM epilogue's (index = row*ld+col) */
int tfasr_dropout(const void* x, void* y, long n, float p, long seed, int dtype, void* stream);
/* out[c] += scale * sum_r x[r*ld + c]   (bias gradients) */
int tfasr_colsum(const void* x, long ld, float* out, long rows, int C, float scale, int dtype, void* stream);
/* nmat <= 64 f32 matrices x + b * stride [rows, C] -> bf16 copies y + b * stride (stride in elements, a multiple of 4; C % 4 == 0) and
   colsum[b][c] += sum_r x_b[r, c] from the f32 values (host array of device pointers, NULL entries skipped), one launch: the operand and
   the bias gradient of the positional-projection weight gradients of every Conformer block (tfasr_block_io.defer_pos_grad). */
int tfasr_cast_colsum_many(const float* x, void* y, long stride, int nmat, int rows, int C, float* const* colsum, void* stream);
/* GLU over the last axis: x [rows, 2C] -> y [rows, C] = x[:, :C] * sigmoid(x[:, C:])  (activations/glu.py:25-28) */
int tfasr_glu_fwd(const void* x, void* y, long rows, int C, int dtype, void* stream);
int tfasr_glu_bwd(const void* x, const void* dy, void* dx, long rows, int C, int dtype, void* stream);
/* causal depthwise Conv1D, channel-last, w [K, C] f32 (keras depthwise kernel [K,C,1]), K <= 32
 * (convolution.py:159-228, conformer.py:305-313) */
int tfasr_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int B, int T, int C, int K, int dtype,
                     void* stream);
/* forward + the statistics tfasr_bn_stats would take of its output, in one launch: stats [ncopy][2][C] f32 is ACCUMULATED into (sum and sum
 * of squares of the bf16-rounded outputs per channel; workgroup i adds into copy i % ncopy so that same-address atomics do not queue; the
 * consumer - tfasr_bn_finalize_apply_fwd_copies - adds the copies up).  UNSUPPORTED (f32, C % 8 != 0, K > 32, unaligned): the two calls. */
int tfasr_dwconv_fwd_stats(const void* x, const float* w, const float* bias, void* y, float* stats, int ncopy, int B, int T, int C, int K,
                           int dtype, void* stream);
/* ConvModule backward between the two pointwise convs in ONE launch (conformer.py:300-333): the BatchNorm backward's apply pass
 * (tfasr_bn_apply_bwd_grads_copies with act = swish: dcv = gradient w.r.t. the depthwise conv's output, WRITTEN - the depthwise weight
 * gradient's operand - and dgamma / dbeta accumulated), the depthwise data gradient and the GLU backward (tfasr_dwconv_bwd_data_glu).
 * bn_x = the BatchNorm input, dsw = the gradient w.r.t. its swish output, both [B*T, C].  UNSUPPORTED: the two calls. */
int tfasr_bn_dwconv_bwd_data_glu(const void* bn_x, const void* dsw, const float* fin, const float* bstats, int copies, float count, float* dgamma,
                                 float* dbeta, float grad_scale, void* dcv, const float* w, const void* glu_x, void* dglu, int B, int T, int C,
                                 int K, int dtype, void* stream);
/* ... with the GLU in front of the conv (ConvModule, conformer.py:300-313) in the same launch: glu_x [B*T, 2C] -> g = a * sigmoid(b)
 * [B*T, C] (written: the depthwise weight gradient's operand; bitwise tfasr_glu_fwd) -> y, stats.  UNSUPPORTED (f32, C % 8, K <= 8 or
 * > 32, unaligned): tfasr_glu_fwd + tfasr_dwconv_fwd_stats. */
int tfasr_glu_dwconv_fwd_stats(const void* glu_x, void* g, const float* w, const float* bias, void* y, float* stats, int ncopy, int B, int T,
                               int C, int K, int dtype, void* stream);
int tfasr_dwconv_bwd_data(const void* dy, const float* w, void* dx, int B, int T, int C, int K, int dtype, void* stream);
/* data gradient followed by the backward of the GLU in front of the conv (glu_x = the GLU's input [B*T, 2C], dglu = its gradient) in
 * one launch; TFASR_STATUS_UNSUPPORTED (f32, C % 8 != 0, K > 32): call tfasr_dwconv_bwd_data + tfasr_glu_bwd instead */
int tfasr_dwconv_bwd_data_glu(const void* dy, const float* w, const void* glu_x, void* dglu, int B, int T, int C, int K, int dtype,
                              void* stream);
int tfasr_dwconv_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int T, int C, int K,
                            int dtype, void* stream);
/* same result through a caller-owned workspace of per-block partial sums + a reduce kernel (no atomics; 2.5x faster at the
 * Conformer-M shape).  Falls back to the entry above when the workspace is NULL / too small or the shape is not covered. */
int tfasr_dwconv_bwd_weight_workspace_size(int B, int T, int C, int K, size_t* bytes);
int tfasr_dwconv_bwd_weight_ws(const void* x, const void* dy, float* dw, float* dbias, int B, int T, int C, int K, int dtype,
                               void* workspace, size_t workspace_bytes, void* stream);
/* n <= 32 depthwise weight gradients of ONE shape in one tile launch + one reduce launch (host arrays of device pointers; dbias may be NULL
   or hold NULL entries; workspace >= n * tfasr_dwconv_bwd_weight_workspace_size(B, T, C, K) bytes, else / f32: one launch pair or launch each). */
int tfasr_dwconv_bwd_weight_many(const void* const* x, const void* const* dy, float* const* dw, float* const* dbias, int n, int B, int T, int C,
                                 int K, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* y1 = x + u, y2 = x + v (content / positional attention biases, multihead_attention.py:554-558) and its backward */
int tfasr_bias2_fwd(const void* x, long ldx, const float* u, const float* v, void* y1, void* y2, long rows, int C,
                    int dtype, void* stream);
int tfasr_bias2_bwd(const void* d1, const void* d2, void* dx, long lddx, float* du, float* dv, long rows, int C,
                    int dtype, void* stream);
/* Embedding gather (table f32 [V,E]) and scatter-add gradient (embedding.py:41-48) */
int tfasr_embedding_fwd(const int32_t* idx, const float* table, void* out, long rows, int E, int V, int dtype,
                        void* stream);
int tfasr_embedding_bwd(const int32_t* idx, const void* dout, float* dtable, long rows, int E, int V, int dtype,
                        void* stream);
/* TransducerJointMerge + tanh: h[b,t,u,:] = tanh(enc[b,t,:] + pred[b,u,:])  (base_transducer.py:199-207,291);
 * backward reduces dh*(1-h^2) over u (-> denc [B,T,J]) and over t (-> dpred [B,U1,J]) */
int tfasr_joint_fwd(const void* enc, const void* pred, void* h, int B, int T, int U1, int J, int dtype, void* stream);
int tfasr_joint_bwd(const void* h, const void* dh, void* denc, void* dpred, int B, int T, int U1, int J, int dtype,
                    void* stream);
/* packed-lattice variants (see tfasr_rnnt_loss_packed): h/dh [total_cells, J]; denc [B,T,J] / dpred [B,U1,J] are dense
 * (zero outside the valid ranges) */
int tfasr_joint_fwd_packed(const void* enc, const void* pred, void* h, const long* cell_off, const int32_t* label_len,
                           long total_cells, int B, int T, int U1, int J, int dtype, void* stream);
/* h == NULL: dh already carries the tanh' factor (the producing GEMM's dact = TFASR_ACT_TANH_OUT epilogue): plain segment sums */
int tfasr_joint_bwd_packed(const void* h, const void* dh, void* denc, void* dpred, const long* cell_off,
                           const int32_t* label_len, const int32_t* logit_len, int B, int T, int U1, int J, int dtype,
                           void* stream);
/* keras Adam step over one flat f32 buffer: p -= lr*wd*p; g' = grad_scale*g (+ 2*l2*p for i < n_reg);
 * m,v update; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)   (small.yml.j2:73-87; L2: :67-69) */
int tfasr_adam(float* p, const float* g, float* m, float* v, long n, long n_reg, float lr, float beta1, float beta2,
               float eps, float weight_decay, float l2, float grad_scale, long step, void* stream);
/* the same update; shadow_bf16 != NULL: the bf16 copy of the parameters (what the bf16 kernels read as weights) is written in the same
   pass, bitwise what tfasr_cast of the updated buffer gives. */
int tfasr_adam_shadow(float* p, const float* g, float* m, float* v, long n, long n_reg, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float l2, float grad_scale, long step, void* shadow_bf16, void* stream);
int tfasr_sumsq(const float* p, long n, float* out, void* stream);
/* x[i] += stddev * N(0,1) over an f32 vector, the deviate a pure function of (seed, i).  Replaces tf.random.normal in variational
 * weight noise (utils/layer_util.py:42-52 add_gwn, models/transducer/base_transducer.py:382-425) and gradient noise
 * (utils/math_util.py add_gauss_noise, models/base_model.py:185-191). */
int tfasr_gauss_noise(float* x, long n, float stddev, long seed, void* stream);
/* y += alpha * x over f32 vectors (sync-BN gamma/beta gradient hand-off, gradient accumulation: accumulation.py:54-70) */
int tfasr_axpy(float* y, const float* x, float alpha, long n, void* stream);
/* SpecAugment mask application, in place: fmask [B,nf,2] = (f0, width), tmask [B,nt,2] = (t0, width)
 * (augmentations/methods/specaugment.py:58-87,108-137; random draws are made by the caller) */
int tfasr_specaugment(void* x, const int32_t* fmask, const int32_t* tmask, int nf, int nt, int B, int T, int F,
                      float mask_value, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Relative-position attention softmax (multihead_attention.py:543-582, 27-77; positional_encoding.py:152-172)
 *   content [B,H,T,ldc>=T], pos [B,H,T,ldp>=2T] (column 2T-1 = bias column), probs like content (may alias it),
 *   lengths [B] or NULL; use_mask: padded query rows -> uniform.  Backward: dcontent may alias dprobs.
 *   ldc / ldp are row strides in elements (pad them to a multiple of 8 so the MFMA GEMMs can DMA the rows).
 * ---------------------------------------------------------------------------------------------- */
int tfasr_relattn_softmax_fwd(const void* content, const void* pos, const int32_t* lengths, void* probs, int B, int H,
                              int T, int ldc, int ldp, int use_mask, int dtype, void* stream);
int tfasr_relattn_softmax_bwd(const void* probs, const void* dprobs, const int32_t* lengths, void* dcontent, void* dpos,
                              int B, int H, int T, int ldc, int ldp, int use_mask, int dtype, void* stream);
/* Same forward with the streaming attention mask of compute_streaming_mask (multihead_attention.py:104-143,331-345) ANDed into
 * the auto mask: query i sees key j iff max(0, c*chunk - hist) <= j < min(T, c*chunk + chunk), c = i / chunk (history_size < 0
 * = unlimited history); chunk_size <= 0 = no streaming mask.  Masked scores are -1e9 in the reference, i.e. exactly zero
 * probability, so the backward (tfasr_relattn_softmax_bwd, which works from the probabilities) is unchanged. */
int tfasr_relattn_softmax_fwd_streaming(const void* content, const void* pos, const int32_t* lengths, void* probs, int B, int H,
                                        int T, int ldc, int ldp, int use_mask, int chunk_size, int history_size, int dtype,
                                        void* stream);

/* Fused (flash-style) forward of the same attention: nothing of size T x T is written.  qkv [B*T, 3*H*dh] (fused
 * projection output, q|k|v column blocks), ubias/vbias [H*dh] f32 (content / positional biases), pext [2T, H*dh]
 * (projected relative table + bias row), out [B*T, H*dh], lse [B,H,T] f32.  bf16, dh == 64 only (else UNSUPPORTED; narrower heads are
 * stored zero-padded to 64 by the caller, scale = 1 / sqrt(reference head size)).
 * chunk / hist: streaming attention mask (compute_streaming_mask, multihead_attention.py:104-143,331-345): query i sees keys
 * [max(0, c - hist), min(T, c + chunk)) with c = floor(i / chunk) * chunk; chunk <= 0: off; hist < 0: unlimited history.  Key blocks
 * outside every window of a query block are skipped.  The same pair of arguments on the backward entry points below. */
int tfasr_relattn_fused_fwd(const void* qkv, const float* ubias, const float* vbias, const void* pext,
                            const int32_t* lengths, void* out, float* lse, int B, int H, int T, int dh, float scale,
                            int use_mask, int chunk, int hist, int dtype, void* stream);

/* Fused backward, query side (transposed orientation, csrc/attn_fused.hip relattn_fused_bwd_qT_kernel): recomputes the probabilities
 * from lse; o / dout [B*T, H*dh].  The query gradient leaves the kernel complete: dq (row stride lddq, a multiple of 4, e.g. the q
 * columns of the fused [B*T, 3*H*dh] gradient) = dqu + dqv with dqu = d/d(q+u), dqv = d/d(q+v) (formed against the window rows);
 * du [H*dh] += column sums of dqu, dv += column sums of dqv.  The UNSKEWED score gradient ds [B,H,T,lds] (lds >= T, multiple of 8) is
 * stored for tfasr_relattn_dpext, dvec [2,B,H,T] = rowsum(dout * o) and the bias-row score (q_i + v) . pext[2T-1] of every query for the key side, and the bias row's share is added into
 * dpext [2T, H*dh] f32 (zeroed by the caller).  With use_mask, the ds rows (and dvec) of a 64-row query block that lies entirely in the
 * padding (i0 >= lengths[b]) are NOT written: their gradient is zero and tfasr_relattn_dpext skips those tiles.
 * qu / qv (both or neither, [B*T, H*dh], 16-byte aligned): also written = q + u / q + v for tfasr_relattn_fused_bwd_k and
 * tfasr_relattn_dpext (what tfasr_bias2_fwd writes), except the rows of wholly padded 64-query blocks, which those two never read.
 * (Entry points _bwd_q / _bwd_q2 of ABI <= 34 - the row-oriented kernels - are gone: ABI 35.) */
int tfasr_relattn_fused_bwd_q3(const void* qkv, const float* ubias, const float* vbias, const void* pext, const int32_t* lengths,
                               const void* o, const void* dout, const float* lse, void* dq, long lddq, float* du, float* dv, void* ds,
                               float* dvec, float* dpext, void* qu, void* qv, int B, int H, int T, int dh, int lds, float scale, int use_mask,
                               int chunk, int hist, int dtype, void* stream);
int tfasr_relattn_dpext(const void* ds, const void* qv, const int32_t* lengths, float* dpext, int B, int H, int T, int dh, int lds,
                        int use_mask, int dtype, void* stream);
/* Fused backward, key side (run after _bwd_q3, which also emits dvec [2,B,H,T] = rowsum(dout*o) | bias-row scores): writes the k and v column
 * blocks of dqkv [B*T, 3*H*dh]; qu/qv [B*T, H*dh] = q+u / q+v (tfasr_bias2_fwd). */
int tfasr_relattn_fused_bwd_k(const void* qkv, const void* qu, const void* qv, const void* pext, const int32_t* lengths,
                              const void* dout, const float* lse, const float* dvec, void* dqkv, int B, int H, int T, int dh,
                              float scale, int use_mask, int chunk, int hist, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LSTM cell pointwise stages (keras LSTM, gates i,f,c,o; base_transducer.py:71-85,123-159)
 * ---------------------------------------------------------------------------------------------- */
int tfasr_lstm_step_fwd(const void* xg, long xg_stride_b, const float* hr, const void* h_prev, long hprev_stride_b,
                        const float* c_prev, long cprev_stride_b, const int32_t* lengths, int t, void* gates,
                        long gates_stride_b, float* c_out, long c_stride_b, void* h_out, long h_stride_b, void* y_out,
                        long y_stride_b, int B, int P, int dtype, void* stream);
int tfasr_lstm_step_bwd(const void* dy, long dy_stride_b, const float* dhr, float* dh_carry, float* dc_carry,
                        const void* gates, long gates_stride_b, const float* c_t, long c_stride_b, const float* c_prev,
                        long cprev_stride_b, const int32_t* lengths, int t, void* dz, long dz_stride_b, int B, int P,
                        int dtype, void* stream);

/* The whole recurrence of the prediction network queued by one call (per step: recurrent GEMM h_{t-1} @ R into `hr` + the cell stage).
 * xg / gates [B,U1,4P], cseq (f32) / hseq / yseq [B,U1,P] row-major; rk = recurrent kernel [P,4P]; h0 / c0 may be NULL (zero state);
 * hr [B,4P], dhr [B,P] f32 scratch; dh_carry / dc_carry [B,P] f32, zero on entry of the backward. */
/* Which recurrence tfasr_lstm_seq_fwd / _bwd run: mode 1 = the persistent one-launch kernels where the shape allows, 0 = the per-step
   kernels, -1 = TFASR_LSTM_PERSIST from the environment (default: persistent).  Returns the previous override.  The persistent kernels
   finish a direction in 0.74 / 1.18 ms at U1 = 111, P = 640 (per-step path: 1.6 / 2.4 ms) but hold P / 16 CUs for that long: beside an
   encoder running on another stream the per-step path makes the whole train step 0.2 ms FASTER (measured, M and S), so the Python
   model switches per call (conformer.py: persistent when the prediction network has the device to itself). */
int tfasr_lstm_set_persist(int mode);
int tfasr_lstm_seq_fwd(const void* xg, const void* rk, const void* h0, long h0_stride_b, const float* c0, long c0_stride_b,
                       const int32_t* lengths, void* gates, float* cseq, void* hseq, void* yseq, float* hr, int B, int U1, int P,
                       int dtype, void* stream);
int tfasr_lstm_seq_bwd(const void* dy, const void* rk, const void* gates, const float* cseq, const int32_t* lengths, void* dz,
                       float* dh_carry, float* dc_carry, float* dhr, int B, int U1, int P, int dtype, void* stream);
/* Steps [t0, t1) of the same recurrences with the per-step kernels only (never the persistent launch), so that a caller can queue the
   chain in slices between other work: forward slices in ascending order, backward slices in DESCENDING order ([t, U1) first); the
   buffers - including the carries - are those of the whole-sequence calls and live across the slices. */
int tfasr_lstm_seq_fwd_range(const void* xg, const void* rk, const void* h0, long h0_stride_b, const float* c0, long c0_stride_b,
                             const int32_t* lengths, void* gates, float* cseq, void* hseq, void* yseq, float* hr, int B, int U1, int P,
                             int dtype, int t0, int t1, void* stream);
int tfasr_lstm_seq_bwd_range(const void* dy, const void* rk, const void* gates, const float* cseq, const int32_t* lengths, void* dz,
                             float* dh_carry, float* dc_carry, float* dhr, int B, int U1, int P, int dtype, int t0, int t1, void* stream);
/* The whole recurrence of one direction as ONE persistent launch (csrc/lstm_persist.hip): workgroup j keeps the recurrent
 * weights of 16 hidden units in registers for the whole sequence, the per-step exchange of h_t (forward) / dz_t (backward) between
 * the workgroups goes through the sequence buffers themselves with write-through stores, one device-scope arrival counter and one
 * agent-scope acquire per step.  Same buffers and semantics as tfasr_lstm_seq_fwd / _bwd (which take this path by themselves when it
 * applies; TFASR_LSTM_PERSIST=0 forces the step kernels).  bf16, B <= 64, P a multiple of 32, P <= 1024, else UNSUPPORTED.
 * `sync`: tfasr_lstm_persist_sync_bytes() bytes of device memory, zeroed by the call; every spin is bounded (1 s): after a stream
 * synchronisation the second 32-bit word != 0 means a wait timed out and the results are invalid. */
size_t tfasr_lstm_persist_sync_bytes(void);
int tfasr_lstm_persist_fwd(const void* xg, const void* rk, const void* h0, long h0_stride_b, const float* c0, long c0_stride_b,
                           const int32_t* lengths, void* gates, float* cseq, void* hseq, void* yseq, int B, int U1, int P, int dtype,
                           void* sync, void* stream);
int tfasr_lstm_persist_bwd(const void* dy, const void* rk, const void* gates, const float* cseq, const int32_t* lengths, void* dz,
                           float* dh_carry, float* dc_carry, int B, int U1, int P, int dtype, void* sync, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy transducer search control (Transducer.recognize_batch / recognize_single, base_transducer.py:496-712).
 * mode 0 = batch variant, 1 = single (B == 1, `per_frame` [nframes] zero-initialised, tok_idx starts at -1).
 * `active` [1]: the while_loop condition evaluated on device by `prepare`; `update` is a no-op when it is 0.
 * encj [B,T,J] = joint encoder projection of every frame; ecur [B,J] receives the current frames.
 * ---------------------------------------------------------------------------------------------- */
int tfasr_decode_prepare(const void* encj, const int32_t* nframes, const int32_t* frame_idx, const int32_t* tok_idx,
                         int32_t* active, void* ecur, int B, int T, int J, int max_tokens, int mode, int dtype,
                         void* stream);
/* One search step up to the logits in three launches (embedding + LSTM cell incl. the loop condition, LayerNorm + prediction
 * projection + frame gather + tanh, vocabulary projection), f32 on the f32 master weights: emb [V,E], lstm_k [E,4P], lstm_rk
 * [P,4P], lstm_b [4P], ln_g/ln_b [P] (NULL: no prediction LayerNorm), joint_pred_w [P,J], vocab_w [J,V]; encj [B,T,J] f32.
 * Writes active[0], h_new / c_new [B,P], z [B,J], logits [B,V]; follow with tfasr_decode_update (dtype f32).  B <= 64;
 * P, J % 4 == 0 and V % 8 == 0, else UNSUPPORTED (use tfasr_decode_prepare + the per-op entry points). */
int tfasr_decode_step(const float* emb, const float* lstm_k, const float* lstm_rk, const float* lstm_b, const float* ln_g,
                      const float* ln_b, const float* joint_pred_w, const float* joint_pred_b, const float* vocab_w,
                      const float* vocab_b, const float* packed, const float* encj, const int32_t* nframes, const int32_t* frame_idx,
                      const int32_t* tok_idx, const int32_t* prev_tok, const float* h, const float* c, int32_t* active,
                      float* h_new, float* c_new, float* z, float* logits, int B, int T, int E, int P, int J, int V,
                      int max_tokens, int mode, float ln_eps, void* stream);
/* `iters` iterations of (tfasr_decode_step + tfasr_decode_update) queued by one host call (f32 states h, c [B,P]). */
int tfasr_decode_steps(const float* emb, const float* lstm_k, const float* lstm_rk, const float* lstm_b, const float* ln_g,
                       const float* ln_b, const float* joint_pred_w, const float* joint_pred_b, const float* vocab_w,
                       const float* vocab_b, const float* packed, const float* encj, const int32_t* nframes, int32_t* frame_idx,
                       int32_t* tok_idx, int32_t* prev_tok, float* h, float* c, int32_t* active, float* h_new, float* c_new, float* z,
                       float* logits, int32_t* tokens, int32_t* per_frame, int B, int T, int E, int P, int J, int V, int max_tokens,
                       int blank, int mode, int max_tokens_per_frame, float ln_eps, int iters, void* stream);
/* `packed` of the two entry points above (NULL: the vector-ALU kernels on the row-major masters): the weights of a search step re-laid
 * once per recognize call for the exact-f32 MFMA kernels (csrc/decode_step.hip): the recurrent kernel, the joint's prediction projection
 * and the vocabulary projection with the 16 output columns of a workgroup contiguous in fragment order, and G = emb @ lstm_k [V, 4P]
 * (keras LSTM: x @ kernel is a product of its own), so that a step gathers the input half of the pre-activation instead of multiplying
 * the embedding row again.  tfasr_decode_pack_floats = floats `packed` must hold, 0 when the shapes have no MFMA route (P, J % 16,
 * P <= 1024, J <= 1280): pass NULL then.  With `packed`, `z` also carries the gathered encoder frames between the step's launches. */
size_t tfasr_decode_pack_floats(int E, int P, int J, int V);
int tfasr_decode_pack(const float* emb, const float* lstm_k, const float* lstm_rk, const float* joint_pred_w, const float* vocab_w,
                      float* packed, int E, int P, int J, int V, void* stream);
int tfasr_decode_update(const void* logits, const int32_t* active, const int32_t* nframes, int32_t* frame_idx,
                        int32_t* prev_tok, int32_t* tok_idx, int32_t* tokens, int32_t* per_frame, const void* h_new,
                        const float* c_new, void* h, float* c, int B, int V, int P, int max_tokens, int blank, int mode,
                        int max_tokens_per_frame, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Conv2dSubsampling pieces (subsampling.py:163-254; causal 3x3 stride 2, convolution.py:25-37,132-144)
 *   conv1: x [B,T0,F0] (Cin=1) -> y [B,ceil(T0/2),ceil(F0/2),C]; w [3,3,1,C] f32
 *   im2col/col2im for the second conv: x [B,T1,F1,C] <-> col [B*T2*F2, 9*C]
 * ---------------------------------------------------------------------------------------------- */
int tfasr_conv1_fwd(const void* x, const float* w, const float* bias, void* y, int B, int T0, int F0, int C, int dtype,
                    void* stream);
int tfasr_conv1_bwd_weight(const void* x, const void* dy, float* dw, float* db, int B, int T0, int F0, int C,
                           int dtype, void* stream);
/* Haloed space-to-depth ("S") layout of a channel-last [B, T1, F1, C] activation: [B, T2+1, F2+1, 2, 2, C] (T2 = ceil(T1/2),
 * F2 = ceil(F1/2)); element (b, t, f, :) lives in row (b, t/2+1, f/2+1), parity block (t%2, f%2); row 0 / column 0 of every sample
 * and the slots past an odd T1 / F1 are zero.  There the input of tap (kh, kw) of the causal 3x3 stride-2 Conv2D
 * (subsampling.py:218-230, convolution.py:25-37,132-144) for output row (b, tt, ff) is the same row index shifted by a constant,
 * so conv2 forward / data gradient are tfasr_gemm calls with K-segments (tfasr_gemm_args.seg_*) and its weight gradient is 9
 * shifted-pointer products (tfasr_gemm_group): no patch matrix exists.  _s2d variants of conv1 write / read that layout directly
 * (only the valid slots); tfasr_halo_zero clears the halo rows of a [B, T2+1, F2+1, W] tensor, tfasr_s2d_edge_zero the slots past
 * an odd edge. */
int tfasr_conv1_fwd_s2d(const void* x, const float* w, const float* bias, void* y, int B, int T0, int F0, int C, int dtype,
                        void* stream);
int tfasr_conv1_bwd_weight_s2d(const void* x, const void* dy, float* dw, float* db, int B, int T0, int F0, int C, int dtype,
                               void* stream);
/* conv1 + BatchNorm(+swish) of Conv2dSubsampling's first block WITHOUT materialising conv1's output (it is recomputed from the
   feature map: 9 FMAs per element instead of ~10 GB of HBM passes per step): statistics [2C] (sum, sum of squares; -> the caller's
   all-reduce -> tfasr_bn_finalize), apply into the S layout, backward statistics [2C] (sum dz, sum dz*xhat), backward apply fused
   with conv1's weight / bias gradients (`count` = rows x world of the statistics).  Act = swish (subsampling.py:163-230). */
int tfasr_conv1_stats(const void* x, const float* w, const float* bias, float* stats, int B, int T0, int F0, int C, int dtype,
                      void* stream);
int tfasr_conv1_bn_apply_s2d(const void* x, const float* w, const float* bias, const float* fin, void* y, int B, int T0, int F0, int C,
                             int dtype, void* stream);
int tfasr_conv1_bn_bwd_stats_s2d(const void* x, const float* w, const float* bias, const float* fin, const void* dy, float* bstats, int B,
                                 int T0, int F0, int C, int dtype, void* stream);
int tfasr_conv1_bn_bwd_apply_s2d(const void* x, const float* w, const float* bias, const float* fin, const float* bstats, float count,
                                 const void* dy, float* dw, float* db, int B, int T0, int F0, int C, int dtype, void* stream);
/* The same through the Gram matrix of conv1's 3x3 patches (conv1 has one input channel: every sum over positions is a function of
   gram = { G[9][9] = sum p p^T, s[9] = sum p, N } (a buffer of TFASR_CONV1_GRAM_DOUBLES doubles: 8 partial copies of the 91 sums, zeroed and
   filled by tfasr_conv1_gram from the feature map; the consumers add the copies up) and of sums
   against the incoming gradient): forward statistics [2C] without a pass over C channels x 9 taps per position; backward in ONE pass
   over the gradient (bstats [2C] as tfasr_conv1_bn_bwd_stats_s2d + pbuf [10][C]: P[k][c] = sum p_k dz_c, row 9 = this rank's sum dz),
   then tfasr_conv1_bn_bwd_finalize (after the caller's all-reduce of bstats; `count` = positions x world) adds conv1's weight / bias
   gradients.  Same results as the two-pass route up to summation order (formulas: csrc/conv2d.hip). */
#define TFASR_CONV1_GRAM_DOUBLES 768
int tfasr_conv1_gram(const void* x, double* gram, int B, int T0, int F0, int dtype, void* stream);
int tfasr_conv1_stats_from_gram(const double* gram, const float* w, const float* bias, float* stats, int C, void* stream);
int tfasr_conv1_bn_bwd_onepass_s2d(const void* x, const float* w, const float* bias, const float* fin, const void* dy, float* bstats,
                                   float* pbuf, int B, int T0, int F0, int C, int dtype, void* stream);
int tfasr_conv1_bn_bwd_finalize(const double* gram, const float* w, const float* bias, const float* fin, const float* bstats, float count,
                                const float* pbuf, float* dw, float* db, int C, void* stream);
int tfasr_halo_zero(void* x, int B, int T2, int F2, int W, int dtype, void* stream);
int tfasr_s2d_edge_zero(void* x, int B, int T1, int F1, int C, int dtype, void* stream);
int tfasr_im2col_3x3s2(const void* x, void* col, int B, int T1, int F1, int C, int dtype, void* stream);
int tfasr_col2im_3x3s2(const void* dcol, void* dx, int B, int T1, int F1, int C, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Log-mel frontend (feature_extraction.py:170-231,255-303): signal [B,N] f32 -> out [B,T0,F] (dtype),
 * T0 = ceil(N/frame_step).  window [frame_len], melw [nfft/2+1, F], band [F,2] = first/last non-zero row of melw.
 * ---------------------------------------------------------------------------------------------- */
int tfasr_logmel(const float* signal, int B, int N, float preemph, const float* window, int frame_len, int frame_step,
                 int nfft, const float* melw, const int32_t* band, int F, float eps, void* out, int T0, int dtype,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * ContextNet encoder pieces (models/encoders/contextnet.py:40-298) beyond the shared conv / BN / GEMM kernels.
 * x, y, dy, dx: [B, T, C] channel-last `dtype`, C % 8 == 0; pool / scale / dscale / dpool: [B, C] f32.
 *   rows_subsample: y[b, t'] = x[b, t' * stride], T' = ceil(T / stride) (a strided CAUSAL conv = the stride-1 one sampled);
 *                   _bwd scatters dy back and zero-fills the skipped rows
 *   se_pool: masked mean over t < lengths[b] (GlobalAveragePooling1D with the sequence mask, :159-163)
 *   se_scale_fwd: y = x * scale[b, c] (:168-169); se_scale_bwd_reduce: dscale[b,c] = sum_t dy*x;
 *   se_bwd_apply: dx = dy * scale[b,c] + (t < len_b) * dpool[b,c] / len_b
 *   add_act: y = act(a + b) (b may be NULL), _bwd: d = dy * act'(a + b)      (ConvBlock residual + activation, :283-292)
 * ---------------------------------------------------------------------------------------------- */
int tfasr_rows_subsample_fwd(const void* x, void* y, int B, int T, int C, int stride, int dtype, void* stream);
int tfasr_rows_subsample_bwd(const void* dy, void* dx, int B, int T, int C, int stride, int dtype, void* stream);
int tfasr_se_pool(const void* x, const int32_t* lengths, float* pool, int B, int T, int C, int dtype, void* stream);
int tfasr_se_scale_fwd(const void* x, const float* scale, void* y, int B, int T, int C, int dtype, void* stream);
int tfasr_se_scale_bwd_reduce(const void* x, const void* dy, float* dscale, int B, int T, int C, int dtype, void* stream);
int tfasr_se_bwd_apply(const void* dy, const float* scale, const float* dpool, const int32_t* lengths, void* dx, int B, int T,
                       int C, int dtype, void* stream);
int tfasr_add_act_fwd(const void* a, const void* b, void* y, long n, int act, int dtype, void* stream);
int tfasr_add_act_bwd(const void* a, const void* b, const void* dy, void* d, long n, int act, int dtype, void* stream);

/* CTC prefix beam search (CtcModel.recognize_beam -> tf.nn.ctc_beam_search_decoder(beam_width), base_ctc.py:127-149).  A HOST
 * routine, like the reference's op: logits [B,T,V] and logit_len [B] are HOST pointers; tokens [B,T] (0-padded, the dense
 * form of the top path), tokens_len [B], log_prob [B] (optional) are HOST outputs.  blank_index: the reference's call treats
 * class V-1 as blank (TF's decoder convention) although the model's blank is 0 - pass V-1 to reproduce it. */
int tfasr_ctc_beam_search_host(const float* logits, const int32_t* logit_len, int B, int T, int V, int beam_width,
                               int blank_index, int32_t* tokens, int32_t* tokens_len, float* log_prob);

/* ------------------------------------------------------------------------------------------------
 * Native executor of one Conformer block (ConformerBlock.call, encoders/conformer.py:430-520, and its backward):
 * queues every kernel of FFModule -> MHSAModule -> ConvModule -> FFModule -> LayerNorm on `stream` with one host call.
 * All parameters live in three flat buffers (f32 master, compute-dtype shadow, f32 gradient) addressed by the element
 * offsets off[TFASR_BP_*]; intermediates are carved from the caller's arenas: `stash` keeps what the backward needs
 * (sizes: tfasr_block_workspace_sizes), `scratch` is free again after the call (between phase A and B of one backward it
 * must stay untouched).  ctx = tfasr_block_ctx_bytes() bytes of host memory, written by _fwd and read by _bwd.
 * Phases (TFASR_PHASE_A|TFASR_PHASE_B = everything): A ends after the conv module's BatchNorm statistics are in
 * io->bn_stats [2d+1] (forward) / io->bn_bstats [2d] (backward) so that a data-parallel caller can all-reduce them
 * before phase B (keras BatchNormalization(synchronized=True), conformer.py:327-333).
 * ---------------------------------------------------------------------------------------------- */
typedef enum {
  TFASR_BP_FF1_LN_G = 0, TFASR_BP_FF1_LN_B, TFASR_BP_FF1_D1_W, TFASR_BP_FF1_D1_B, TFASR_BP_FF1_D2_W, TFASR_BP_FF1_D2_B,
  TFASR_BP_FF2_LN_G, TFASR_BP_FF2_LN_B, TFASR_BP_FF2_D1_W, TFASR_BP_FF2_D1_B, TFASR_BP_FF2_D2_W, TFASR_BP_FF2_D2_B,
  TFASR_BP_AT_LN_G, TFASR_BP_AT_LN_B, TFASR_BP_AT_QKV_W, TFASR_BP_AT_QKV_B, TFASR_BP_AT_POS_W, TFASR_BP_AT_POS_B,
  TFASR_BP_AT_O_W, TFASR_BP_AT_O_B, TFASR_BP_AT_U, TFASR_BP_AT_V,
  TFASR_BP_CV_LN_G, TFASR_BP_CV_LN_B, TFASR_BP_CV_PW1_W, TFASR_BP_CV_PW1_B, TFASR_BP_CV_DW_W, TFASR_BP_CV_DW_B,
  TFASR_BP_CV_BN_G, TFASR_BP_CV_BN_B, TFASR_BP_CV_PW2_W, TFASR_BP_CV_PW2_B,
  TFASR_BP_LN_G, TFASR_BP_LN_B, TFASR_BP_COUNT
} tfasr_block_param_t;
#define TFASR_PHASE_A 1
#define TFASR_PHASE_B 2

typedef struct {
  int B, T, d, H, dh, dff, ksize;   /* batch, frames, model dim, heads, head size, FFN dim, depthwise kernel size */
  int dtype;                        /* tfasr_dtype_t of activations / shadow weights */
  int training;                     /* dropout + batch statistics on */
  int save;                         /* keep what the backward needs (pre-activations) */
  int use_mask;                     /* attention auto mask (query rows >= length) */
  int force_unfused;                /* 1: never use the fused attention kernels */
  int world;                        /* data-parallel world size (BatchNorm count / gradient scaling) */
  int site0;                        /* first dropout site id of this block */
  long drop_epoch;                  /* bumped once per forward pass: seed = drop_epoch*8192 + site */
  float drop_p, ffm_res, mhsa_res, conv_res, ln_eps, bn_eps, bn_momentum;
  int chunk_size, history_size;     /* streaming attention mask (chunk_size <= 0: off; history_size < 0: unlimited) */
  int dw_norm_layer;                /* 1: LayerNormalization after the depthwise conv (encoder_convm_dw_norm_type "layer",
                                       encoders/conformer.py:334-340) in the CV_BN_G/B parameter slots; no batch statistics */
  int dh_logical;                   /* the reference's head size (softmax scale 1 / sqrt(dh_logical), multihead_attention.py:554-558) when
                                       `dh` is a zero-padded physical head dimension (e.g. 36 stored as 64); <= 0: = dh */
} tfasr_block_cfg;

typedef struct {
  const float* flat; const void* shadow; float* grad;  /* flat parameter buffers */
  float* bn_mm; float* bn_mv;                          /* conv-module BatchNorm moving mean / variance [d] */
  const void* pe;                                      /* relative sinusoid table [2T, d], compute dtype (row 2T-1 zero) */
  long off[TFASR_BP_COUNT];
} tfasr_block_params;

typedef struct {
  const void* x_in; void* x_out;        /* forward: block input / output [B*T, d] */
  const void* dy; void* dx;             /* backward: gradient w.r.t. output / input */
  const int32_t* lengths;               /* [B] valid frames */
  float* bn_stats; float* bn_bstats;    /* [2d+1] / [2d] f32 */
  void* stash; size_t stash_bytes;
  void* scratch; size_t scratch_bytes;
  /* Optional: accumulators the CALLER has already zeroed (e.g. slices of one buffer cleared by a single memset for all the
     blocks of a step, instead of three small in-stream memsets per block).  prezeroed & 1: bn_stats is zero on entry of
     forward phase A; & 2: bn_bstats is zero on entry of backward phase A; dpext_zero != NULL: a zeroed f32 [2T * H*dh]
     buffer for the positional-projection gradient (otherwise carved out of scratch and cleared in-stream). */
  int prezeroed;
  float* dpext_zero;
  /* Backward: wgrad_slot = 1 or 2 lets the block queue its grouped weight-gradient launch on an internal second stream, so that it
     runs beside the NEXT block's backward (whose chain of small dependent kernels leaves CUs idle) instead of in line.  The caller
     then (i) gives the two slots disjoint `scratch` arenas and alternates them block by block - a call with slot k first makes
     `stream` wait for the launches queued by the earlier calls with slot k, whose operands live in that arena - (ii) keeps the
     forward stash of a block alive until tfasr_block_wgrad_join, and (iii) calls tfasr_block_wgrad_join before anything reads the
     gradients.  0 = in line on `stream` (default). */
  int wgrad_slot;
  /* Work that is the same for every block of a step, or independent of a block's dependent chain, taken OUT of the chain (a kernel
     boundary costs 2.65 us on this chip whatever the kernel does: tools/hwprobe/anyorder_test; 16 blocks x 4 such launches per step):
     pext_pre != NULL (forward): the projected relative-position table [2T, H*dh] of THIS block (pe @ Wpos + bpos, compute dtype),
       computed by the caller ahead of the chain (e.g. on another stream while the subsampling runs); the block does not launch the
       projection.  Must stay alive until the block's backward.
     defer_pos_grad != 0 (backward; needs dpext_zero): the block only ACCUMULATES the table's gradient into dpext_zero (f32); cast,
       projection weight gradient (gWpos += pe^T dpext) and bias gradient (column sums) are left to the caller, who runs them for all
       blocks at once after the last block's backward.
     ln_part_ext != NULL (backward): caller-owned partial-sum buffer of the block's LayerNorm gamma / beta gradients (at least
       8 * tfasr_layernorm_bwd_part_blocks(rows, d, dtype) * 2d floats, alive until tfasr_block_ln_fold_all): the block does not launch
       its own fold; the caller folds every block of the step with ONE launch (tfasr_block_ln_fold_all over the blocks' ctx). */
  const void* pext_pre;
  int defer_pos_grad;
  float* ln_part_ext;
  size_t ln_part_ext_floats;
  /* dcv_keep != NULL (backward, bf16, BatchNorm variant): caller-owned [B*T, d] buffer that receives the gradient of the depthwise conv's
     output and stays alive until tfasr_block_dwconv_wgrad_all; the block does not launch its depthwise WEIGHT gradient (nothing on the
     chain waits for it): the caller runs the weight gradients of every block of the step as one launch pair.  The forward stash of the
     block (its GLU output is the other operand) must stay alive until then as well. */
  void* dcv_keep;
  /* ds_keep / qv_keep != NULL (backward, with defer_pos_grad, fused attention): caller-owned buffers for the unskewed score gradient
     dS [B, H, T, ceil8(T)] and q + v [B*T, H*dh] (compute dtype); the block does not launch tfasr_relattn_dpext - the caller accumulates
     the table gradient into dpext_zero itself (e.g. on another stream beside the next block's backward: nothing on the chain waits for it). */
  void* ds_keep;
  void* qv_keep;
  /* > 1: bn_stats is [bn_stats_copies][2d] (+ 1 float) - the depthwise conv accumulates the BatchNorm statistics itself, its workgroups
     spreading their atomics over the copies (tfasr_dwconv_fwd_stats), and phase B adds the copies up; a data-parallel caller all-reduces
     all of them.  Likewise bn_bstats [bn_stats_copies][2d]: filled by the epilogue of the pointwise conv's data gradient
     (tfasr_gemm_args.bns_out).  0 / 1: the single [2d+1] / [2d] buffers filled by tfasr_bn_stats / tfasr_bn_bwd_stats. */
  int bn_stats_copies;
} tfasr_block_io;

size_t tfasr_block_ctx_bytes(void);
int tfasr_block_workspace_sizes(const tfasr_block_cfg* cfg, size_t* stash_bytes, size_t* fwd_scratch_bytes,
                                size_t* bwd_scratch_bytes);
int tfasr_block_fwd(const tfasr_block_cfg* cfg, const tfasr_block_params* params, const tfasr_block_io* io, void* ctx,
                    int phase, void* stream);
/* `stream` waits for the weight-gradient launches queued under the slots in slot_mask (bit 0 = slot 1, bit 1 = slot 2); no-op if none */
int tfasr_block_wgrad_join(int slot_mask, void* stream);
int tfasr_block_bwd(const tfasr_block_cfg* cfg, const tfasr_block_params* params, const tfasr_block_io* io, void* ctx,
                    int phase, void* stream);
/* What the last tfasr_block_bwd call with this ctx LEFT TO THE CALLER - the executor, not the caller, decides whether an optional
   tfasr_block_io request is honoured (it depends on the attention route, the storage type and the executor's own switches), so a caller
   that asked for a deferral must look here before it runs the deferred launch itself:
   bit 0: the table gradient from dS (ds_keep / qv_keep are filled; tfasr_relattn_dpext was NOT launched),
   bit 1: the depthwise-conv weight gradient (dcv_keep is filled), bit 2: the positional-projection gradients (defer_pos_grad),
   bit 3: the LayerNorm gamma / beta fold (ln_part_ext).  A bit that is clear means the block did that work in line. */
int tfasr_block_bwd_left(const void* ctx);
/* The executor's second stream of the current device (lowest priority, non-blocking; created on first use): where a block's grouped
   weight gradients run with wgrad_slot != 0.  A caller with off-chain work of its own (positional tables ahead of the chain, table
   gradients beside the next block) can queue it THERE instead of on one more stream of its own: every additional HIP stream is one more
   hardware queue, and on this chip the step time depends on how many are live (38 instead of 23 ms with eight). */
int tfasr_block_side_stream(void** stream);
/* Measurement probe: while enabled, tfasr_block_bwd brackets every grouped weight-gradient launch with HIP events on the stream that
   launch runs on (its own second stream with wgrad_slot != 0).  _read waits for the recorded launches, returns their summed duration and
   count, and clears the record.  Off by default (two event records per block). */
int tfasr_block_wgrad_probe(int enable);
int tfasr_block_wgrad_probe_read(float* total_ms, int* launches);
/* dgamma / dbeta of every LayerNorm of `n` blocks whose backward ran with io->ln_part_ext: one launch instead of one per block.
   ctx[i] = the ctx of block i's tfasr_block_bwd call (host memory); blocks without pending partial sums are skipped. */
int tfasr_block_ln_fold_all(void* const* ctx, int n, int d, void* stream);
/* depthwise-conv weight / bias gradients of n <= 32 blocks whose backward ran with io->dcv_keep (dcv[i] = that buffer, ctx[i] / params[i]
   = the block's; workspace >= n * tfasr_dwconv_bwd_weight_workspace_size(B, T, d, ksize) bytes, else one launch pair per block). */
int tfasr_block_dwconv_wgrad_all(const tfasr_block_cfg* cfg, const tfasr_block_params* const* params, void* const* ctx, const void* const* dcv,
                                 int n, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFASR_HIP_H_ */
