// Greedy transducer search bookkeeping for gfx950 (Transducer.recognize_batch / recognize_single,
// models/transducer/base_transducer.py:496-712).  The arithmetic of one decoding step (embedding, LSTM cell, LayerNorm,
// joint, vocabulary projection) runs on the same kernels as training; these two kernels hold the data-dependent control:
//   decode_prepare : evaluates the while_loop condition ON DEVICE (so the host can enqueue many iterations without a
//                    sync; iterations after termination are no-ops), gathers the current encoder frame of every sample
//   decode_update  : log_softmax + argmax (first maximal index, like tf.argmax) + the masked state updates, including
//                    the reference's quirks (tokens start at column 2, column 0 collects blanks)
#include "common.h"

namespace {

// mode 0 = recognize_batch, 1 = recognize_single (B == 1)
template <typename T>
__global__ void decode_prepare_kernel(const T* __restrict__ encj, const int32_t* __restrict__ nframes,
                                      const int32_t* __restrict__ frame_idx, const int32_t* __restrict__ tok_idx,
                                      int32_t* __restrict__ active, T* __restrict__ ecur, int B, int Tn, int J, int max_tokens,
                                      int mode) {
  __shared__ int all_frames, all_tokens;
  if (threadIdx.x == 0) { all_frames = 1; all_tokens = 1; }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (mode == 0) {
      if (!(frame_idx[b] >= nframes[b] - 1)) atomicAnd(&all_frames, 0);
      if (!(tok_idx[b] >= max_tokens - 1)) atomicAnd(&all_tokens, 0);
    } else {
      if (frame_idx[b] < nframes[b]) atomicAnd(&all_frames, 0);  // loop while frame < nframes
      atomicAnd(&all_tokens, 0);
    }
  }
  __syncthreads();
  const int act = !(all_frames || all_tokens);
  if (threadIdx.x == 0) active[0] = act;
  if (!act) return;
  for (int i = threadIdx.x; i < B * J; i += blockDim.x) {
    const int b = i / J, j = i % J;
    int f = min(frame_idx[b], nframes[b] - 1);
    f = max(min(f, Tn - 1), 0);
    ecur[i] = encj[((long)b * Tn + f) * J + j];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void decode_update_kernel(
    const T* __restrict__ logits, const int32_t* __restrict__ active, const int32_t* __restrict__ nframes,
    int32_t* __restrict__ frame_idx, int32_t* __restrict__ prev_tok, int32_t* __restrict__ tok_idx,
    int32_t* __restrict__ tokens, int32_t* __restrict__ per_frame, const T* __restrict__ h_new,
    const float* __restrict__ c_new, T* __restrict__ h, float* __restrict__ c, int B, int V, int P, int max_tokens,
    int blank, int mode, int max_tokens_per_frame) {
  if (!active[0]) return;
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ int s_keep;  // 1 -> keep previous decoder state (blank / finished)
  const int b = blockIdx.x;
  const T* row = logits + (long)b * V;
  // log_softmax in f32 (tf.nn.log_softmax, base_transducer.py:463) then first-max argmax
  float m = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) m = fmaxf(m, Num<T>::ld(row + v));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red_v[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_v[0], red_v[1]), fmaxf(red_v[2], red_v[3]));
  __syncthreads();
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += expf(Num<T>::ld(row + v) - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red_v[threadIdx.x >> 6] = s;
  __syncthreads();
  const float lse = logf(red_v[0] + red_v[1] + red_v[2] + red_v[3]);
  __syncthreads();
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    const float lp = (Num<T>::ld(row + v) - m) - lse;
    if (lp > best || (lp == best && v < bi)) { best = lp; bi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { red_v[threadIdx.x >> 6] = best; red_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (red_v[w] > best || (red_v[w] == best && red_i[w] < bi)) { best = red_v[w]; bi = red_i[w]; }
    const int cur = bi;
    if (mode == 0) {
      const int ti = tok_idx[b], fi = frame_idx[b];
      const bool eq_blank = (cur == blank) || (ti >= max_tokens) || (fi > nframes[b]);
      const int nxt = min(ti + 1, max_tokens - 1);
      tokens[(long)b * max_tokens + (eq_blank ? 0 : nxt)] = eq_blank ? blank : cur;
      if (!eq_blank) { tok_idx[b] = nxt; prev_tok[b] = cur; }
      else frame_idx[b] = fi + 1;
      s_keep = eq_blank ? 1 : 0;
    } else {
      const int fi = frame_idx[b];
      const bool is_blank = (cur == blank);
      int nf = per_frame[fi];
      if (!is_blank) { nf += 1; per_frame[fi] = nf; }
      if (is_blank || nf >= max_tokens_per_frame) frame_idx[b] = fi + 1;
      int ti = tok_idx[b];  // token_index, starts at -1
      if (!is_blank) { ti += 1; tok_idx[b] = ti; prev_tok[b] = cur; }
      if (ti >= 0) tokens[ti] = prev_tok[b];
      s_keep = is_blank ? 1 : 0;
    }
  }
  __syncthreads();
  if (!s_keep)
    for (int p = threadIdx.x; p < P; p += blockDim.x) { h[(long)b * P + p] = h_new[(long)b * P + p]; c[(long)b * P + p] = c_new[(long)b * P + p]; }
}

// The same bookkeeping with ONE batch of loads (f32 logits, V <= 256 NV, P <= 1024): the kernel above walks the logits row three times
// and reads the counters and the new state only after its reductions - six dependent round trips (~6.7 us per search iteration for a
// few hundred bytes of work).  Here a thread keeps its NV logits, the counters and its share of h_new / c_new in registers from the
// first instruction on; what remains is three LDS reductions.  Arithmetic and tie-breaks are those of decode_update_kernel.
template <int NV>
__global__ __launch_bounds__(256) void decode_update_regs_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ active, const int32_t* __restrict__ nframes,
    int32_t* __restrict__ frame_idx, int32_t* __restrict__ prev_tok, int32_t* __restrict__ tok_idx, int32_t* __restrict__ tokens,
    int32_t* __restrict__ per_frame, const float* __restrict__ h_new, const float* __restrict__ c_new, float* __restrict__ h,
    float* __restrict__ c, int B, int V, int P, int max_tokens, int blank, int mode, int max_tokens_per_frame) {
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ int s_keep;
  const int b = blockIdx.x;
  const float* row = logits + (long)b * V;
  float x[NV], hn[4], cn[4];
#pragma unroll
  for (int t = 0; t < NV; ++t) x[t] = (threadIdx.x + 256 * t < V) ? row[threadIdx.x + 256 * t] : -INFINITY;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = threadIdx.x + 256 * t;
    hn[t] = p < P ? h_new[(long)b * P + p] : 0.f;
    cn[t] = p < P ? c_new[(long)b * P + p] : 0.f;
  }
  const int ti = tok_idx[b], fi = frame_idx[b], nfr = nframes[b], ptok = prev_tok[b];
  const int act = active[0];
  int nf = 0;
  if (mode == 1) nf = per_frame[fi];  // (second round trip, single-utterance variant only)
  if (!act) return;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NV; ++t) m = fmaxf(m, x[t]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red_v[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_v[0], red_v[1]), fmaxf(red_v[2], red_v[3]));
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NV; ++t)
    if (threadIdx.x + 256 * t < V) s += expf(x[t] - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red_v[threadIdx.x >> 6] = s;
  __syncthreads();
  const float lse = logf(red_v[0] + red_v[1] + red_v[2] + red_v[3]);
  __syncthreads();
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int v = threadIdx.x + 256 * t;
    if (v < V) {
      const float lp = (x[t] - m) - lse;
      if (lp > best || (lp == best && v < bi)) { best = lp; bi = v; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { red_v[threadIdx.x >> 6] = best; red_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (red_v[w] > best || (red_v[w] == best && red_i[w] < bi)) { best = red_v[w]; bi = red_i[w]; }
    const int cur = bi;
    if (mode == 0) {
      const bool eq_blank = (cur == blank) || (ti >= max_tokens) || (fi > nfr);
      const int nxt = min(ti + 1, max_tokens - 1);
      tokens[(long)b * max_tokens + (eq_blank ? 0 : nxt)] = eq_blank ? blank : cur;
      if (!eq_blank) { tok_idx[b] = nxt; prev_tok[b] = cur; }
      else frame_idx[b] = fi + 1;
      s_keep = eq_blank ? 1 : 0;
    } else {
      const bool is_blank = (cur == blank);
      if (!is_blank) { nf += 1; per_frame[fi] = nf; }
      if (is_blank || nf >= max_tokens_per_frame) frame_idx[b] = fi + 1;
      int t2 = ti;  // token_index, starts at -1
      if (!is_blank) { t2 += 1; tok_idx[b] = t2; prev_tok[b] = cur; }
      if (t2 >= 0) tokens[t2] = is_blank ? ptok : cur;
      s_keep = is_blank ? 1 : 0;
    }
  }
  __syncthreads();
  if (!s_keep) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = threadIdx.x + 256 * t;
      if (p < P) { h[(long)b * P + p] = hn[t]; c[(long)b * P + p] = cn[t]; }
    }
  }
}

}  // namespace

extern "C" int tfasr_decode_prepare(const void* encj, const int32_t* nframes, const int32_t* frame_idx,
                                    const int32_t* tok_idx, int32_t* active, void* ecur, int B, int T, int J, int max_tokens,
                                    int mode, int dtype, void* stream_) {
  if (!encj || !nframes || !frame_idx || !tok_idx || !active || !ecur || B <= 0 || T <= 0 || J <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(decode_prepare_kernel<float>, dim3(1), dim3(256), 0, s, (const float*)encj, nframes, frame_idx, tok_idx, active, (float*)ecur, B, T, J, max_tokens, mode);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(decode_prepare_kernel<bf16_t>, dim3(1), dim3(256), 0, s, (const bf16_t*)encj, nframes, frame_idx, tok_idx, active, (bf16_t*)ecur, B, T, J, max_tokens, mode);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_decode_update(const void* logits, const int32_t* active, const int32_t* nframes, int32_t* frame_idx,
                                   int32_t* prev_tok, int32_t* tok_idx, int32_t* tokens, int32_t* per_frame,
                                   const void* h_new, const float* c_new, void* h, float* c, int B, int V, int P,
                                   int max_tokens, int blank, int mode, int max_tokens_per_frame, int dtype, void* stream_) {
  if (!logits || !active || !nframes || !frame_idx || !prev_tok || !tok_idx || !tokens || !h_new || !c_new || !h || !c)
    return TFASR_STATUS_INVALID_VALUE;
  if (B <= 0 || V <= 0 || P <= 0 || (mode == 1 && (B != 1 || !per_frame))) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_F32 && P <= 1024 && V <= 4096) {
#define TFASR_DU(NV) TFASR_KLAUNCH(decode_update_regs_kernel<NV>, dim3(B), dim3(256), 0, s, (const float*)logits, active, nframes, frame_idx, prev_tok, \
                                        tok_idx, tokens, per_frame, (const float*)h_new, c_new, (float*)h, c, B, V, P, max_tokens, blank, mode, max_tokens_per_frame)
    if (V <= 1024) TFASR_DU(4);
    else if (V <= 2048) TFASR_DU(8);
    else TFASR_DU(16);
#undef TFASR_DU
  } else if (dtype == TFASR_F32)
    TFASR_KLAUNCH(decode_update_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)logits, active, nframes, frame_idx, prev_tok, tok_idx, tokens, per_frame, (const float*)h_new, c_new, (float*)h, c, B, V, P, max_tokens, blank, mode, max_tokens_per_frame);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(decode_update_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)logits, active, nframes, frame_idx, prev_tok, tok_idx, tokens, per_frame, (const bf16_t*)h_new, c_new, (bf16_t*)h, c, B, V, P, max_tokens, blank, mode, max_tokens_per_frame);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
