// One greedy-search step of the transducer (Transducer.call_next, base_transducer.py:437-464, inside the loops of :496-712) in
// FOUR launches instead of ten: the step's products are [B <= 64, <= 1280] x [<= 1280, <= 2560] in f32 - far too skinny for the
// 128x128 MFMA tiles (20 workgroups walking 40 k-slabs each took ~43 us per product) - so they run as many-workgroup "skinny" dot
// products on the vector ALUs with the surrounding pointwise work fused in:
//   decode_lstm_kernel  : loop condition (decode_prepare's) + embedding lookup + x W + h R + b + LSTM cell     -> h_new, c_new
//   decode_joint_kernel : LayerNorm(h_new) + prediction projection + encoder frame gather + tanh              -> z [B, J]
//   decode_vocab_kernel : vocabulary projection                                                             -> logits [B, V] f32
//   tfasr_decode_update (decode.hip): log-softmax, arg-max, token / frame / state bookkeeping of the reference loops
// All arithmetic is f32 on the f32 master weights (the reference's CPU path), whatever the model's storage type.
// Thread mapping of the dot products: 256 threads = NB batch slots x KS k-slices; a thread accumulates NC output columns of its
// batch row over its k-slice, the slices meet in LDS.  The weight rows a workgroup reads are its own NC columns only, so the
// matrices stay resident in the L2 of the XCD that owns those columns across the iterations of the search.
#include "common.h"

namespace {

constexpr int MAXB = 64;
constexpr int NT = 1024;  // threads per workgroup: 16 k-slices at B = 64, 32 at B = 32 (short dependent load chains per thread)

// acc[c] += sum_{k in slice} x(k) * W[k * ldw + col0 + c]   for c < NC (NC % 4 == 0, col0 % 4 == 0 -> float4 loads)
template <int NC, typename XF>
__device__ __forceinline__ void slice_dot(float (&acc)[NC], XF x, const float* __restrict__ W, long ldw, int k0, int k1) {
#pragma unroll 8
  for (int k = k0; k < k1; ++k) {
    const float xv = x(k);
    const float* wr = W + (long)k * ldw;
#pragma unroll
    for (int c = 0; c < NC; c += 4) {
      const float4 wv = *reinterpret_cast<const float4*>(wr + c);
      acc[c] += xv * wv.x; acc[c + 1] += xv * wv.y; acc[c + 2] += xv * wv.z; acc[c + 3] += xv * wv.w;
    }
  }
}

struct Map { int nb, ks, b, s; };
__device__ __forceinline__ Map make_map(int B) {
  Map m;
  m.nb = B <= 8 ? 8 : (B <= 16 ? 16 : (B <= 32 ? 32 : 64));
  m.ks = NT / m.nb;
  m.b = threadIdx.x % m.nb;
  m.s = threadIdx.x / m.nb;
  return m;
}

// the while_loop condition of the reference loops, evaluated by every workgroup (B values); see decode_prepare_kernel
__device__ __forceinline__ bool loop_active(const int32_t* nframes, const int32_t* frame_idx, const int32_t* tok_idx, int B, int max_tokens,
                                            int mode) {
  __shared__ int all_frames, all_tokens;
  if (threadIdx.x == 0) { all_frames = 1; all_tokens = 1; }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (mode == 0) {
      if (!(frame_idx[b] >= nframes[b] - 1)) atomicAnd(&all_frames, 0);
      if (!(tok_idx[b] >= max_tokens - 1)) atomicAnd(&all_tokens, 0);
    } else {
      if (frame_idx[b] < nframes[b]) atomicAnd(&all_frames, 0);
      atomicAnd(&all_tokens, 0);
    }
  }
  __syncthreads();
  return !(all_frames || all_tokens);
}

// ---- 1. embedding + LSTM cell for UPW hidden units per workgroup ----------------------------------------------------------------
template <int UPW>
__global__ __launch_bounds__(NT) void decode_lstm_kernel(
    const float* __restrict__ emb, const float* __restrict__ Wk, const float* __restrict__ Wr, const float* __restrict__ bias,
    const int32_t* __restrict__ prev_tok, const float* __restrict__ h, const float* __restrict__ c, const int32_t* __restrict__ nframes,
    const int32_t* __restrict__ frame_idx, const int32_t* __restrict__ tok_idx, int32_t* __restrict__ active, float* __restrict__ h_new,
    float* __restrict__ c_new, int B, int E, int P, int V, int max_tokens, int mode) {
  constexpr int NC = 4 * UPW;
  __shared__ float red[NT][NC + 1];
  const bool act = loop_active(nframes, frame_idx, tok_idx, B, max_tokens, mode);
  if (blockIdx.x == 0 && threadIdx.x == 0) active[0] = act ? 1 : 0;
  if (!act) return;
  const Map m = make_map(B);
  const int u0 = blockIdx.x * UPW;  // first hidden unit of this workgroup (P % UPW == 0 checked by the host)
  const int K = E + P;
  float acc[4][UPW];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < UPW; ++u) acc[q][u] = 0.f;
  if (m.b < B) {
    const int per = (K + m.ks - 1) / m.ks, k0 = m.s * per, k1 = min(K, k0 + per);
    const int tok = min(max(prev_tok[m.b], 0), V - 1);
    const float* er = emb + (long)tok * E;
    const float* hr = h + (long)m.b * P;
    auto run = [&](const float* xr, const float* Wm, int ka, int kb) {  // k in [ka, kb) of one operand (embedding row x Wk, h x Wr)
#pragma unroll 8
      for (int k = ka; k < kb; ++k) {
        const float xv = xr[k];
        const float* wr = Wm + (long)k * 4 * P + u0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (UPW == 4) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + (long)q * P);
            acc[q][0] += xv * wv.x; acc[q][1] += xv * wv.y; acc[q][2] += xv * wv.z; acc[q][3] += xv * wv.w;
          } else {
#pragma unroll
            for (int u = 0; u < UPW; ++u) acc[q][u] += xv * wr[(long)q * P + u];
          }
        }
      }
    };
    run(er, Wk, min(k0, E), min(k1, E));
    run(hr, Wr, max(k0, E) - E, max(k1, E) - E);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < UPW; ++u) red[threadIdx.x][q * UPW + u] = acc[q][u];
  __syncthreads();
  // one thread per (b, unit): sum the k-slices, add the bias, run the cell (gate order i, f, c, o; lstm.hip)
  for (int i = threadIdx.x; i < B * UPW; i += blockDim.x) {
    const int b = i / UPW, u = i % UPW;
    if (u0 + u >= P) continue;
    float z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sum = bias[q * P + u0 + u];
      for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + b][q * UPW + u];
      z[q] = sum;
    }
    const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), gg = tanh_fast(z[2]), og = sigmoidf_(z[3]);
    const float cn = fg * c[(long)b * P + u0 + u] + ig * gg;
    c_new[(long)b * P + u0 + u] = cn;
    h_new[(long)b * P + u0 + u] = og * tanh_fast(cn);
  }
}

// ---- 2. LayerNorm + prediction projection + tanh(enc + pred) for NC joint columns per workgroup -------------------------------
template <int NC>
__global__ __launch_bounds__(NT) void decode_joint_kernel(
    const float* __restrict__ h_new, const float* __restrict__ ln_g, const float* __restrict__ ln_b, const float* __restrict__ Wjp,
    const float* __restrict__ bjp, const float* __restrict__ encj, const int32_t* __restrict__ nframes,
    const int32_t* __restrict__ frame_idx, const int32_t* __restrict__ active, float* __restrict__ z, int B, int T, int P, int J,
    float ln_eps) {
  if (!active[0]) return;
  __shared__ float red[NT][NC + 1];
  __shared__ float s_mean[MAXB], s_rstd[MAXB];
  const Map m = make_map(B);
  const int j0 = blockIdx.x * NC;
  const int per = (P + m.ks - 1) / m.ks, k0 = m.s * per, k1 = min(P, k0 + per);
  const float* hr = h_new + (long)min(m.b, B - 1) * P;
  if (ln_g) {  // keras LayerNormalization (eps 1e-3): two passes over the row for the moments
    float part = 0.f;
    if (m.b < B) for (int k = k0; k < k1; ++k) part += hr[k];
    red[threadIdx.x][0] = part;
    __syncthreads();
    if (threadIdx.x < B) {
      float sum = 0.f;
      for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + threadIdx.x][0];
      s_mean[threadIdx.x] = sum / P;
    }
    __syncthreads();
    const float mu = s_mean[min(m.b, B - 1)];
    part = 0.f;
    if (m.b < B) for (int k = k0; k < k1; ++k) { const float dlt = hr[k] - mu; part += dlt * dlt; }
    __syncthreads();
    red[threadIdx.x][0] = part;
    __syncthreads();
    if (threadIdx.x < B) {
      float sum = 0.f;
      for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + threadIdx.x][0];
      s_rstd[threadIdx.x] = rsqrtf(sum / P + ln_eps);
    }
    __syncthreads();
  }
  float acc[NC];
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) acc[cidx] = 0.f;
  if (m.b < B) {
    if (ln_g) {
      const float mu = s_mean[m.b], rs = s_rstd[m.b];
      slice_dot<NC>(acc, [&](int k) { return (hr[k] - mu) * rs * ln_g[k] + ln_b[k]; }, Wjp + j0, J, k0, k1);
    } else {
      slice_dot<NC>(acc, [&](int k) { return hr[k]; }, Wjp + j0, J, k0, k1);
    }
  }
  __syncthreads();
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) red[threadIdx.x][cidx] = acc[cidx];
  __syncthreads();
  for (int i = threadIdx.x; i < B * NC; i += blockDim.x) {
    const int b = i / NC, cidx = i % NC;
    if (j0 + cidx >= J) continue;
    float sum = bjp[j0 + cidx];
    for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + b][cidx];
    int f = min(frame_idx[b], nframes[b] - 1);
    f = max(min(f, T - 1), 0);
    z[(long)b * J + j0 + cidx] = tanhf(encj[((long)b * T + f) * J + j0 + cidx] + sum);  // TransducerJointMerge add + tanh (:199-207,291)
  }
}

// ---- 3. vocabulary projection for NC classes per workgroup -----------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(NT) void decode_vocab_kernel(const float* __restrict__ z, const float* __restrict__ Wv, const float* __restrict__ bv,
                                                           const int32_t* __restrict__ active, float* __restrict__ logits, int B, int J, int V) {
  if (!active[0]) return;
  __shared__ float red[NT][NC + 1];
  const Map m = make_map(B);
  const int v0 = blockIdx.x * NC;
  float acc[NC];
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) acc[cidx] = 0.f;
  if (m.b < B) {
    const int per = (J + m.ks - 1) / m.ks, k0 = m.s * per, k1 = min(J, k0 + per);
    const float* zr = z + (long)m.b * J;
    slice_dot<NC>(acc, [&](int k) { return zr[k]; }, Wv + v0, V, k0, k1);
  }
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) red[threadIdx.x][cidx] = acc[cidx];
  __syncthreads();
  for (int i = threadIdx.x; i < B * NC; i += blockDim.x) {
    const int b = i / NC, cidx = i % NC;
    if (v0 + cidx >= V) continue;
    float sum = bv[v0 + cidx];
    for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + b][cidx];
    logits[(long)b * V + v0 + cidx] = sum;
  }
}

}  // namespace

extern "C" int tfasr_decode_step(const float* emb, const float* lstm_k, const float* lstm_rk, const float* lstm_b, const float* ln_g,
                                 const float* ln_b, const float* joint_pred_w, const float* joint_pred_b, const float* vocab_w,
                                 const float* vocab_b, const float* encj, const int32_t* nframes, const int32_t* frame_idx,
                                 const int32_t* tok_idx, const int32_t* prev_tok, const float* h, const float* c, int32_t* active,
                                 float* h_new, float* c_new, float* z, float* logits, int B, int T, int E, int P, int J, int V,
                                 int max_tokens, int mode, float ln_eps, void* stream_) {
  if (!emb || !lstm_k || !lstm_rk || !lstm_b || !joint_pred_w || !joint_pred_b || !vocab_w || !vocab_b || !encj || !nframes || !frame_idx ||
      !tok_idx || !prev_tok || !h || !c || !active || !h_new || !c_new || !z || !logits)
    return TFASR_STATUS_INVALID_VALUE;
  if (B <= 0 || B > MAXB || T <= 0 || E <= 0 || P <= 0 || J <= 0 || V <= 1) return TFASR_STATUS_INVALID_VALUE;
  if ((P % 4) || (J % 4) || (V % 8) || (((uintptr_t)lstm_k | (uintptr_t)lstm_rk | (uintptr_t)joint_pred_w | (uintptr_t)vocab_w) & 15))
    return TFASR_STATUS_UNSUPPORTED;  // float4 weight loads
  hipStream_t s = (hipStream_t)stream_;
  hipLaunchKernelGGL(decode_lstm_kernel<4>, dim3(P / 4), dim3(NT), 0, s, emb, lstm_k, lstm_rk, lstm_b, prev_tok, h, c, nframes, frame_idx, tok_idx,
                     active, h_new, c_new, B, E, P, V, max_tokens, mode);
  TFASR_CHECK_LAUNCH();
  hipLaunchKernelGGL(decode_joint_kernel<4>, dim3((J + 3) / 4), dim3(NT), 0, s, h_new, ln_g, ln_b, joint_pred_w, joint_pred_b, encj, nframes,
                     frame_idx, active, z, B, T, P, J, ln_eps);
  TFASR_CHECK_LAUNCH();
  hipLaunchKernelGGL(decode_vocab_kernel<8>, dim3((V + 7) / 8), dim3(NT), 0, s, z, vocab_w, vocab_b, active, logits, B, J, V);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

// `iters` search iterations (fused step + bookkeeping each) queued by ONE host call: the Python host costs ~100 us per iteration
// in ctypes argument marshalling alone, more than the four launches themselves.  Iterations after the loop condition turned false
// are no-ops on the device (active[0] == 0), so the caller checks `active` only every `iters` iterations.
extern "C" int tfasr_decode_steps(const float* emb, const float* lstm_k, const float* lstm_rk, const float* lstm_b, const float* ln_g,
                                  const float* ln_b, const float* joint_pred_w, const float* joint_pred_b, const float* vocab_w,
                                  const float* vocab_b, const float* encj, const int32_t* nframes, int32_t* frame_idx, int32_t* tok_idx,
                                  int32_t* prev_tok, float* h, float* c, int32_t* active, float* h_new, float* c_new, float* z, float* logits,
                                  int32_t* tokens, int32_t* per_frame, int B, int T, int E, int P, int J, int V, int max_tokens, int blank,
                                  int mode, int max_tokens_per_frame, float ln_eps, int iters, void* stream_) {
  if (iters <= 0 || !tokens) return TFASR_STATUS_INVALID_VALUE;
  for (int i = 0; i < iters; ++i) {
    int st = tfasr_decode_step(emb, lstm_k, lstm_rk, lstm_b, ln_g, ln_b, joint_pred_w, joint_pred_b, vocab_w, vocab_b, encj, nframes, frame_idx,
                               tok_idx, prev_tok, h, c, active, h_new, c_new, z, logits, B, T, E, P, J, V, max_tokens, mode, ln_eps, stream_);
    if (st != TFASR_STATUS_SUCCESS) return st;
    st = tfasr_decode_update(logits, active, nframes, frame_idx, prev_tok, tok_idx, tokens, per_frame, h_new, c_new, h, c, B, V, P, max_tokens, blank,
                             mode, max_tokens_per_frame, TFASR_F32, stream_);
    if (st != TFASR_STATUS_SUCCESS) return st;
  }
  return TFASR_STATUS_SUCCESS;
}
