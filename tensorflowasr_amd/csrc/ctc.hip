// CTC loss + gradient and CTC greedy decoding for gfx950.
//
// Reference: CtcLoss.call -> tf.nn.ctc_loss(labels dense [B,U], logits [B,T,V], label_length, logit_length,
// logits_time_major=False, blank_index=0) (losses/ctc_loss.py:47-66) and CtcModel.recognize ->
// tf.nn.ctc_greedy_decoder(merge_repeated=True, blank_index) (models/ctc/base_ctc.py:102-124).
// Standard CTC (Graves 2006) over the extended label sequence l' = (b, l1, b, l2, ..., b), S = 2U+1 states:
//   alpha_t(s) = lse(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if l'_s != b and l'_s != l'_{s-2}]) + lp_t(l'_s)
//   loss = -lse(alpha_{T-1}(S-1), alpha_{T-1}(S-2));  dL/dlogit_t(v) = softmax_t(v) - sum_{s: l'_s = v} exp(alpha_t(s)+beta_t(s)-lp_t(v)-logP)
// Kernels: per-frame log-sum-exp (one wave per frame, HBM-bound single pass over V), the state scan (one workgroup per
// (utterance, {alpha|beta}), one thread per state, T sequential steps through a double-buffered LDS row), the gradient
// (one workgroup per frame; per-label occupancies combined with LDS atomics, then one pass over V).
#include "common.h"
#include <algorithm>

namespace {

template <typename T>
__global__ __launch_bounds__(256) void ctc_lse_kernel(const T* __restrict__ logits, float* __restrict__ lse,
                                                      int32_t* __restrict__ amax, long rows, int V) {
  const int lane = threadIdx.x & 63;
  const long w0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long r = w0; r < rows; r += nw) {
    const T* row = logits + r * V;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) { const float x = Num<T>::ld(row + v); if (x > m) { m = x; mi = v; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(m, o, 64);
      const int oi = __shfl_xor(mi, o, 64);
      if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(Num<T>::ld(row + v) - m);
    s = wave_sum(s);
    if (lane == 0) { if (lse) lse[r] = m + logf(s); if (amax) amax[r] = mi; }
  }
}

__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// grid (B, 2); blockDim >= S = 2*U+1
template <typename T>
__global__ void ctc_scan_kernel(const T* __restrict__ logits, const float* __restrict__ lse, const int32_t* __restrict__ labels,
                                const int32_t* __restrict__ label_len, const int32_t* __restrict__ logit_len, int Tm, int U,
                                int V, int blank, float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ costs) {
  extern __shared__ float sh[];
  const int b = blockIdx.x, s = threadIdx.x, nthr = blockDim.x;
  const int Tl = min(logit_len[b], Tm), Ul = min(label_len[b], U);
  const int S = 2 * Ul + 1, Smax = 2 * U + 1;
  float* buf0 = sh;
  float* buf1 = sh + nthr;
  buf0[s] = -INFINITY;
  buf1[s] = -INFINITY;
  __syncthreads();
  if (Tl <= 0) { if (s == 0 && blockIdx.y == 0) costs[b] = (Ul == 0) ? 0.f : INFINITY; return; }
  const bool act = s < S;
  int lab = blank;
  bool skip = false;  // transition s-2 -> s (alpha) allowed
  bool skipb = false; // transition s -> s+2 (beta) allowed
  if (act && (s & 1)) {
    lab = min(max(labels[(long)b * U + (s >> 1)], 0), V - 1);
    // neighbours clamped like `lab` itself (an out-of-range label is folded into [0, V-1] everywhere, not only at its own state)
    if (s >= 2) skip = lab != min(max(labels[(long)b * U + (s >> 1) - 1], 0), V - 1);
    if (s + 2 < S) skipb = lab != min(max(labels[(long)b * U + (s >> 1) + 1], 0), V - 1);
  }
  const T* lg = logits + (long)b * Tm * V;
  const float* ls = lse + (long)b * Tm;
  const long abase = (long)b * Tm * Smax;
  if (blockIdx.y == 0) {
    for (int t = 0; t < Tl; ++t) {
      float* cur = (t & 1) ? buf1 : buf0;
      const float* prev = (t & 1) ? buf0 : buf1;
      if (act) {
        const float lp = Num<T>::ld(lg + (long)t * V + lab) - ls[t];
        float a;
        if (t == 0) a = (s <= 1) ? lp : -INFINITY;
        else a = lse3(prev[s], s >= 1 ? prev[s - 1] : -INFINITY, skip ? prev[s - 2] : -INFINITY) + lp;
        alpha[abase + (long)t * Smax + s] = a;
        cur[s] = a;
      }
      __syncthreads();
    }
    if (s == 0) {
      const float* last = ((Tl - 1) & 1) ? buf1 : buf0;
      const float a1 = last[S - 1], a2 = (S >= 2) ? last[S - 2] : -INFINITY;
      costs[b] = -lse3(a1, a2, -INFINITY);
    }
  } else {
    int it = 0;
    for (int t = Tl - 1; t >= 0; --t, ++it) {
      float* cur = (it & 1) ? buf1 : buf0;
      const float* prev = (it & 1) ? buf0 : buf1;
      if (act) {
        const float lp = Num<T>::ld(lg + (long)t * V + lab) - ls[t];
        float v;
        if (t == Tl - 1) v = (s >= S - 2) ? lp : -INFINITY;
        else v = lse3(prev[s], (s + 1 < S) ? prev[s + 1] : -INFINITY, skipb ? prev[s + 2] : -INFINITY) + lp;
        beta[abase + (long)t * Smax + s] = v;
        cur[s] = v;
      }
      __syncthreads();
    }
  }
}

// grid = B*T blocks
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const T* logits, T* grads, const float* __restrict__ lse,
                                                       const int32_t* __restrict__ labels, const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ logit_len, const float* __restrict__ grad_scale,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const float* __restrict__ costs, int Tm, int U, int V, int blank) {
  extern __shared__ float occ[];  // V floats
  const int b = blockIdx.x / Tm, t = blockIdx.x % Tm;
  const int Tl = min(logit_len[b], Tm), Ul = min(label_len[b], U);
  const T* row = logits + ((long)b * Tm + t) * V;
  T* out = grads + ((long)b * Tm + t) * V;
  if (t >= Tl) { for (int v = threadIdx.x; v < V; v += blockDim.x) Num<T>::st(out + v, 0.f); return; }
  for (int v = threadIdx.x; v < V; v += blockDim.x) occ[v] = 0.f;
  __syncthreads();
  const int S = 2 * Ul + 1, Smax = 2 * U + 1;
  const float logp = -costs[b];
  const float l = lse[(long)b * Tm + t];
  const long abase = ((long)b * Tm + t) * Smax;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const int lab = (s & 1) ? min(max(labels[(long)b * U + (s >> 1)], 0), V - 1) : blank;
    const float lp = Num<T>::ld(row + lab) - l;
    const float e = alpha[abase + s] + beta[abase + s] - lp - logp;
    if (e > -80.f) atomicAdd(&occ[lab], expf(e));
  }
  __syncthreads();
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  const bool finite = isfinite(logp);
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    const float p = expf(Num<T>::ld(row + v) - l);
    Num<T>::st(out + v, finite ? (p - occ[v]) * sc : 0.f);
  }
}

// merge repeated + drop blanks (tf.nn.ctc_greedy_decoder); one thread per utterance
__global__ void ctc_collapse_kernel(const int32_t* __restrict__ amax, const int32_t* __restrict__ logit_len, int32_t* __restrict__ out,
                                    int32_t* __restrict__ out_len, int B, int Tm, int blank) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int Tl = min(logit_len[b], Tm);
  int prev = -1, n = 0;
  for (int t = 0; t < Tl; ++t) {
    const int c = amax[(long)b * Tm + t];
    if (c != prev && c != blank) out[(long)b * Tm + n++] = c;
    prev = c;
  }
  for (int i = n; i < Tm; ++i) out[(long)b * Tm + i] = blank;
  out_len[b] = n;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int tfasr_ctc_loss_workspace_size(int B, int T, int U, int V, size_t* bytes) {
  if (!bytes || B <= 0 || T <= 0 || U < 0 || V <= 0) return TFASR_STATUS_INVALID_VALUE;
  const size_t S = 2 * (size_t)U + 1;
  *bytes = align256((size_t)B * T * 4) + 2 * align256((size_t)B * T * S * 4);
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_ctc_loss(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                              const int32_t* logit_len, const float* grad_scale, int B, int T, int U, int V, int blank,
                              int dtype, float* costs, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!logits || !labels || !label_len || !logit_len || !costs || !workspace) return TFASR_STATUS_INVALID_VALUE;
  if (B <= 0 || T <= 0 || U < 0 || V <= 1 || blank < 0 || blank >= V || 2 * U + 1 > 1024) return TFASR_STATUS_INVALID_VALUE;
  size_t need = 0;
  tfasr_ctc_loss_workspace_size(B, T, U, V, &need);
  if (workspace_bytes < need) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const size_t S = 2 * (size_t)U + 1;
  char* ws = (char*)workspace;
  float* lse = (float*)ws;
  float* alpha = (float*)(ws + align256((size_t)B * T * 4));
  float* beta = (float*)(ws + align256((size_t)B * T * 4) + align256((size_t)B * T * S * 4));
  const long rows = (long)B * T;
  const int grid = (int)std::max<long>(1, std::min<long>((rows + 3) / 4, 8192));
  const int nthr = (int)((S + 63) / 64) * 64;
  if (dtype == TFASR_F32) {
    TFASR_KLAUNCH(ctc_lse_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)logits, lse, (int32_t*)nullptr, rows, V);
    TFASR_KLAUNCH(ctc_scan_kernel<float>, dim3(B, 2), dim3(nthr), 2 * nthr * sizeof(float), s, (const float*)logits, lse, labels, label_len, logit_len, T, U, V, blank, alpha, beta, costs);
    if (grads) TFASR_KLAUNCH(ctc_grad_kernel<float>, dim3(B * T), dim3(256), V * sizeof(float), s, (const float*)logits, (float*)grads, lse, labels, label_len, logit_len, grad_scale, alpha, beta, costs, T, U, V, blank);
  } else if (dtype == TFASR_BF16) {
    TFASR_KLAUNCH(ctc_lse_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)logits, lse, (int32_t*)nullptr, rows, V);
    TFASR_KLAUNCH(ctc_scan_kernel<bf16_t>, dim3(B, 2), dim3(nthr), 2 * nthr * sizeof(float), s, (const bf16_t*)logits, lse, labels, label_len, logit_len, T, U, V, blank, alpha, beta, costs);
    if (grads) TFASR_KLAUNCH(ctc_grad_kernel<bf16_t>, dim3(B * T), dim3(256), V * sizeof(float), s, (const bf16_t*)logits, (bf16_t*)grads, lse, labels, label_len, logit_len, grad_scale, alpha, beta, costs, T, U, V, blank);
  } else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_ctc_greedy_decode(const void* logits, const int32_t* logit_len, int32_t* workspace_argmax, int32_t* tokens,
                                       int32_t* tokens_len, int B, int T, int V, int blank, int dtype, void* stream_) {
  if (!logits || !logit_len || !workspace_argmax || !tokens || !tokens_len || B <= 0 || T <= 0 || V <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const long rows = (long)B * T;
  const int grid = (int)std::max<long>(1, std::min<long>((rows + 3) / 4, 8192));
  if (dtype == TFASR_F32) TFASR_KLAUNCH(ctc_lse_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)logits, (float*)nullptr, workspace_argmax, rows, V);
  else if (dtype == TFASR_BF16) TFASR_KLAUNCH(ctc_lse_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)logits, (float*)nullptr, workspace_argmax, rows, V);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_KLAUNCH(ctc_collapse_kernel, dim3((B + 63) / 64), dim3(64), 0, s, workspace_argmax, logit_len, tokens, tokens_len, B, T, blank);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
