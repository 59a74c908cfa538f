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
 *   - re-entrant per stream, no global state, no assumption about the process-wide current device
 *     beyond "the pointers and the stream belong to the device that is current on this thread".
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

#define TFASR_ABI_VERSION 1

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

/* ------------------------------------------------------------------------------------------------
 * GEMM family (Dense / EinsumDense / pointwise Conv1D / attention products / joint vocab projection:
 * keras Dense sites conformer.py:72-87, multihead_attention.py:628-637,654, base_transducer.py:238-293)
 *
 *   D[b] = epilogue( alpha * op(A[b]) x op(B[b]) )          b = b1*nb2 + b2 (two-level batch)
 *   op(A) is [M,K]: trans_a==0 -> A stored [M,K] (lda), trans_a==1 -> A stored [K,M] (lda)
 *   op(B) is [K,N]: trans_b==0 -> B stored [K,N] (ldb), trans_b==1 -> B stored [N,K] (ldb)
 *   epilogue: v = alpha*acc + bias[n]; if (prez) prez = v; v = act(v);
 *             if (dact_z) v *= dact'(dact_z[m,n]);  if (res) v = res[m,n] + beta*v;
 *             D = v   (out_f32 ? f32 : dtype);  accumulate!=0 -> atomicAdd into f32 D (split-K legal)
 * A, B, res, dact_z, prez are `dtype`; bias is f32.
 * ---------------------------------------------------------------------------------------------- */
typedef enum { TFASR_ACT_NONE = 0, TFASR_ACT_SWISH = 1, TFASR_ACT_TANH = 2, TFASR_ACT_SIGMOID = 3 } tfasr_act_t;

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
} tfasr_gemm_args;

int tfasr_gemm(const tfasr_gemm_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFASR_HIP_H_ */
