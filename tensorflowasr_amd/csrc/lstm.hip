// LSTM cell pointwise stages (forward / backward) of the transducer prediction network for gfx950.
//
// Reference: keras.layers.LSTM(units=P, return_sequences, zero_output_for_mask=True) in
// TransducerPrediction (base_transducer.py:71-85,123-132): gate order i,f,c,o; kernel [E,4P],
// recurrent kernel [P,4P], bias [4P]; sigmoid recurrent activation, tanh activation; on masked steps
// (t >= length) the state is carried and the emitted output is zero.
// The two GEMMs (x@W+b for all steps at once, h_{t-1}@R per step) run on the MFMA GEMM family; these
// kernels do the per-step gate math, one thread per (batch row, hidden unit).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

// z = xg[b,t] + hr[b]; i,f,g,o; c = f*c_prev + i*g; h = o*tanh(c)
template <typename T>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(
    const T* __restrict__ xg, long xg_stride_b,         // [B, 4P] view of step t
    const float* __restrict__ hr,                        // [B, 4P] or null
    const T* __restrict__ h_prev, long hprev_stride_b,   // [B, P] view or null (zeros)
    const float* __restrict__ c_prev, long cprev_stride_b,  // [B, P] view or null (zeros)
    const int32_t* __restrict__ lengths, int t,
    T* __restrict__ gates, long gates_stride_b,          // [B, 4P] view (activated gates) or null
    float* __restrict__ c_out, long c_stride_b,          // [B, P] view
    T* __restrict__ h_out, long h_stride_b,              // [B, P] view (carried state)
    T* __restrict__ y_out, long y_stride_b,              // [B, P] view (masked output) or null
    int B, int P) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * P) return;
  const int b = idx / P, p = idx % P;
  const float hp = h_prev ? Num<T>::ld(h_prev + b * hprev_stride_b + p) : 0.f;
  const float cp = c_prev ? c_prev[b * cprev_stride_b + p] : 0.f;
  const bool valid = lengths ? (t < lengths[b]) : true;
  if (!valid) {
    c_out[b * c_stride_b + p] = cp;
    Num<T>::st(h_out + b * h_stride_b + p, hp);
    if (y_out) Num<T>::st(y_out + b * y_stride_b + p, 0.f);
    if (gates) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Num<T>::st(gates + b * gates_stride_b + q * P + p, 0.f);
    }
    return;
  }
  float z[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    z[q] = Num<T>::ld(xg + b * xg_stride_b + q * P + p);
    if (hr) z[q] += hr[(long)b * 4 * P + q * P + p];
  }
  const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), gg = tanh_fast(z[2]), og = sigmoidf_(z[3]);
  const float c = fg * cp + ig * gg;
  const float h = og * tanh_fast(c);
  c_out[b * c_stride_b + p] = c;
  Num<T>::st(h_out + b * h_stride_b + p, h);
  if (y_out) Num<T>::st(y_out + b * y_stride_b + p, h);
  if (gates) {
    Num<T>::st(gates + b * gates_stride_b + 0 * P + p, ig);
    Num<T>::st(gates + b * gates_stride_b + 1 * P + p, fg);
    Num<T>::st(gates + b * gates_stride_b + 2 * P + p, gg);
    Num<T>::st(gates + b * gates_stride_b + 3 * P + p, og);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(
    const T* __restrict__ dy, long dy_stride_b,           // [B,P] view: dL/d(output_t)
    const float* __restrict__ dhr,                         // [B,P] or null: dz_{t+1} @ R^T
    float* __restrict__ dh_carry, float* __restrict__ dc_carry,  // [B,P] in/out
    const T* __restrict__ gates, long gates_stride_b,      // [B,4P] view
    const float* __restrict__ c_t, long c_stride_b,        // [B,P] view
    const float* __restrict__ c_prev, long cprev_stride_b, // [B,P] view or null
    const int32_t* __restrict__ lengths, int t,
    T* __restrict__ dz, long dz_stride_b,                  // [B,4P] view
    int B, int P) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * P) return;
  const int b = idx / P, p = idx % P;
  const long sidx = (long)b * P + p;
  float dh_total = dh_carry[sidx] + (dhr ? dhr[sidx] : 0.f);
  const bool valid = lengths ? (t < lengths[b]) : true;
  if (!valid) {
    dh_carry[sidx] = dh_total;  // state passes straight through a masked step
#pragma unroll
    for (int q = 0; q < 4; ++q) Num<T>::st(dz + b * dz_stride_b + q * P + p, 0.f);
    return;
  }
  const float dh = dh_total + Num<T>::ld(dy + b * dy_stride_b + p);
  const float ig = Num<T>::ld(gates + b * gates_stride_b + p);
  const float fg = Num<T>::ld(gates + b * gates_stride_b + P + p);
  const float gg = Num<T>::ld(gates + b * gates_stride_b + 2 * P + p);
  const float og = Num<T>::ld(gates + b * gates_stride_b + 3 * P + p);
  const float c = c_t[b * c_stride_b + p];
  const float cp = c_prev ? c_prev[b * cprev_stride_b + p] : 0.f;
  const float tc = tanhf(c);
  const float dc = dc_carry[sidx] + dh * og * (1.f - tc * tc);
  Num<T>::st(dz + b * dz_stride_b + 0 * P + p, dc * gg * ig * (1.f - ig));
  Num<T>::st(dz + b * dz_stride_b + 1 * P + p, dc * cp * fg * (1.f - fg));
  Num<T>::st(dz + b * dz_stride_b + 2 * P + p, dc * ig * (1.f - gg * gg));
  Num<T>::st(dz + b * dz_stride_b + 3 * P + p, dh * tc * og * (1.f - og));
  dc_carry[sidx] = dc * fg;
  dh_carry[sidx] = 0.f;
}

}  // namespace

extern "C" int tfasr_lstm_step_fwd(const void* xg, long xg_stride_b, const float* hr, const void* h_prev,
                                   long hprev_stride_b, const float* c_prev, long cprev_stride_b,
                                   const int32_t* lengths, int t, void* gates, long gates_stride_b, float* c_out,
                                   long c_stride_b, void* h_out, long h_stride_b, void* y_out, long y_stride_b, int B,
                                   int P, int dtype, void* stream_) {
  if (!xg || !c_out || !h_out || B <= 0 || P <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = (B * P + 255) / 256;
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(lstm_step_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)xg, xg_stride_b, hr,
                       (const float*)h_prev, hprev_stride_b, c_prev, cprev_stride_b, lengths, t, (float*)gates,
                       gates_stride_b, c_out, c_stride_b, (float*)h_out, h_stride_b, (float*)y_out, y_stride_b, B, P);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(lstm_step_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)xg, xg_stride_b, hr,
                       (const bf16_t*)h_prev, hprev_stride_b, c_prev, cprev_stride_b, lengths, t, (bf16_t*)gates,
                       gates_stride_b, c_out, c_stride_b, (bf16_t*)h_out, h_stride_b, (bf16_t*)y_out, y_stride_b, B, P);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_lstm_step_bwd(const void* dy, long dy_stride_b, const float* dhr, float* dh_carry, float* dc_carry,
                                   const void* gates, long gates_stride_b, const float* c_t, long c_stride_b,
                                   const float* c_prev, long cprev_stride_b, const int32_t* lengths, int t, void* dz,
                                   long dz_stride_b, int B, int P, int dtype, void* stream_) {
  if (!dy || !dh_carry || !dc_carry || !gates || !c_t || !dz || B <= 0 || P <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = (B * P + 255) / 256;
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(lstm_step_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, dy_stride_b, dhr,
                       dh_carry, dc_carry, (const float*)gates, gates_stride_b, c_t, c_stride_b, c_prev, cprev_stride_b,
                       lengths, t, (float*)dz, dz_stride_b, B, P);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(lstm_step_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, dy_stride_b, dhr,
                       dh_carry, dc_carry, (const bf16_t*)gates, gates_stride_b, c_t, c_stride_b, c_prev, cprev_stride_b,
                       lengths, t, (bf16_t*)dz, dz_stride_b, B, P);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The whole recurrence queued from C: per step one recurrent GEMM (h_{t-1} @ R, f32 out) + the cell kernel.  From Python the 2 x U1
// launches of a direction cost ~15 us of host time each (ctypes marshalling), 2.8 ms per direction at U1 = 86 - with the train step's
// host side at 25.8 of 28.0 ms the GPU idled between the steps of this chain.
// Layouts as the step kernels take them: xg / gates [B, U1, 4P], cseq (f32) / hseq / yseq [B, U1, P]; hr [B, 4P] f32 scratch.
// ---------------------------------------------------------------------------------------------------------------------------------
static int g_persist_override = -1;  // tfasr_lstm_set_persist: -1 = TFASR_LSTM_PERSIST from the environment (default on), 0 / 1 = forced
static bool persist_enabled() {
  if (g_persist_override >= 0) return g_persist_override == 1;
  static int v = -1;
  if (v < 0) { const char* e = getenv("TFASR_LSTM_PERSIST"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}
extern "C" int tfasr_lstm_set_persist(int mode) {
  const int prev = g_persist_override;
  g_persist_override = mode < 0 ? -1 : (mode ? 1 : 0);
  return prev;
}

// steps [t0, t1) of the forward recurrence with the per-step kernels (never the persistent launch): what a caller uses to queue the chain
// in SLICES between other work - the host blocks in hipLaunchKernel once a stream's launch queue holds ~1 ms of work, and while it is
// blocked on this chain's stream the other streams starve (conformer.py interleaves the slices with the encoder blocks)
extern "C" int tfasr_lstm_seq_fwd_range(const void* xg, const void* rk, const void* h0, long h0_stride_b, const float* c0, long c0_stride_b,
                                        const int32_t* lengths, void* gates, float* cseq, void* hseq, void* yseq, float* hr, int B, int U1, int P,
                                        int dtype, int t0, int t1, void* stream) {
  if (!xg || !rk || !gates || !cseq || !hseq || !hr || B <= 0 || U1 <= 0 || P <= 0 || t0 < 0 || t1 > U1 || t0 > t1) return TFASR_STATUS_INVALID_VALUE;
  const long esz = dtype == TFASR_F32 ? 4 : 2;
  for (int t = t0; t < t1; ++t) {
    const char* hprev = t > 0 ? (const char*)hseq + (long)(t - 1) * P * esz : (const char*)h0;
    const long hps = t > 0 ? (long)U1 * P : h0_stride_b;
    const float* cprev = t > 0 ? cseq + (long)(t - 1) * P : c0;
    const long cps = t > 0 ? (long)U1 * P : c0_stride_b;
    if (hprev) {
      tfasr_gemm_args a;
      memset(&a, 0, sizeof(a));
      a.A = hprev; a.B = rk; a.D = hr; a.M = B; a.N = 4 * P; a.K = P; a.lda = hps; a.ldb = 4 * P; a.ldd = 4 * P;
      a.nb1 = a.nb2 = 1; a.alpha = 1.f; a.beta = 1.f; a.dtype = dtype; a.out_f32 = 1; a.split_k = 1;
      const int st = tfasr_gemm(&a, stream);
      if (st != TFASR_STATUS_SUCCESS) return st;
    }
    const int st = tfasr_lstm_step_fwd((const char*)xg + (long)t * 4 * P * esz, (long)U1 * 4 * P, hprev ? hr : nullptr, hprev, hps, cprev, cps, lengths, t,
                                       (char*)gates + (long)t * 4 * P * esz, (long)U1 * 4 * P, cseq + (long)t * P, (long)U1 * P,
                                       (char*)hseq + (long)t * P * esz, (long)U1 * P, yseq ? (char*)yseq + (long)t * P * esz : nullptr, (long)U1 * P, B, P,
                                       dtype, stream);
    if (st != TFASR_STATUS_SUCCESS) return st;
  }
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_lstm_seq_fwd(const void* xg, const void* rk, const void* h0, long h0_stride_b, const float* c0, long c0_stride_b,
                                  const int32_t* lengths, void* gates, float* cseq, void* hseq, void* yseq, float* hr, int B, int U1, int P,
                                  int dtype, void* stream) {
  if (!xg || !rk || !gates || !cseq || !hseq || !hr || B <= 0 || U1 <= 0 || P <= 0) return TFASR_STATUS_INVALID_VALUE;
  // one persistent launch for the whole sequence where the shape allows it (lstm_persist.hip); its 64-byte synchronisation record
  // lives at the front of the `hr` scratch (B x 4P floats, unused by that path)
  if (persist_enabled() && (size_t)B * 4 * P * sizeof(float) >= tfasr_lstm_persist_sync_bytes()) {
    const int st = tfasr_lstm_persist_fwd(xg, rk, h0, h0_stride_b, c0, c0_stride_b, lengths, gates, cseq, hseq, yseq, B, U1, P, dtype, hr, stream);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  return tfasr_lstm_seq_fwd_range(xg, rk, h0, h0_stride_b, c0, c0_stride_b, lengths, gates, cseq, hseq, yseq, hr, B, U1, P, dtype, 0, U1, stream);
}

// steps t1-1 down to t0 of the backward recurrence with the per-step kernels (see tfasr_lstm_seq_fwd_range); slices must be queued in
// descending order: [t, U1), then [t', t), ... - the carries dh_carry / dc_carry / dhr live across the calls
extern "C" int tfasr_lstm_seq_bwd_range(const void* dy, const void* rk, const void* gates, const float* cseq, const int32_t* lengths, void* dz,
                                        float* dh_carry, float* dc_carry, float* dhr, int B, int U1, int P, int dtype, int t0, int t1, void* stream) {
  if (!dy || !rk || !gates || !cseq || !dz || !dh_carry || !dc_carry || !dhr || B <= 0 || U1 <= 0 || P <= 0 || t0 < 0 || t1 > U1 || t0 > t1)
    return TFASR_STATUS_INVALID_VALUE;
  const long esz = dtype == TFASR_F32 ? 4 : 2;
  for (int t = t1 - 1; t >= t0; --t) {
    int st = tfasr_lstm_step_bwd((const char*)dy + (long)t * P * esz, (long)U1 * P, t < U1 - 1 ? dhr : nullptr, dh_carry, dc_carry,
                                 (const char*)gates + (long)t * 4 * P * esz, (long)U1 * 4 * P, cseq + (long)t * P, (long)U1 * P,
                                 t > 0 ? cseq + (long)(t - 1) * P : nullptr, (long)U1 * P, lengths, t, (char*)dz + (long)t * 4 * P * esz, (long)U1 * 4 * P, B, P,
                                 dtype, stream);
    if (st != TFASR_STATUS_SUCCESS) return st;
    if (t > 0) {  // dhr = dz_t @ R^T
      tfasr_gemm_args a;
      memset(&a, 0, sizeof(a));
      a.A = (const char*)dz + (long)t * 4 * P * esz; a.B = rk; a.D = dhr; a.M = B; a.N = P; a.K = 4 * P; a.lda = (long)U1 * 4 * P; a.ldb = 4 * P; a.ldd = P;
      a.trans_b = 1; a.nb1 = a.nb2 = 1; a.alpha = 1.f; a.beta = 1.f; a.dtype = dtype; a.out_f32 = 1; a.split_k = 1;
      st = tfasr_gemm(&a, stream);
      if (st != TFASR_STATUS_SUCCESS) return st;
    }
  }
  return TFASR_STATUS_SUCCESS;
}

// backward through time: dz [B, U1, 4P] out; dh_carry / dc_carry [B, P] f32 zeroed by the caller; dhr [B, P] f32 scratch
extern "C" int tfasr_lstm_seq_bwd(const void* dy, const void* rk, const void* gates, const float* cseq, const int32_t* lengths, void* dz,
                                  float* dh_carry, float* dc_carry, float* dhr, int B, int U1, int P, int dtype, void* stream) {
  if (!dy || !rk || !gates || !cseq || !dz || !dh_carry || !dc_carry || !dhr || B <= 0 || U1 <= 0 || P <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (persist_enabled() && (size_t)B * P * sizeof(float) >= tfasr_lstm_persist_sync_bytes()) {  // (synchronisation record at the front of `dhr`)
    const int st = tfasr_lstm_persist_bwd(dy, rk, gates, cseq, lengths, dz, dh_carry, dc_carry, B, U1, P, dtype, dhr, stream);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  return tfasr_lstm_seq_bwd_range(dy, rk, gates, cseq, lengths, dz, dh_carry, dc_carry, dhr, B, U1, P, dtype, 0, U1, stream);
}
