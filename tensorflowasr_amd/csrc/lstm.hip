// LSTM cell pointwise stages (forward / backward) of the transducer prediction network for gfx950.
//
// Reference: keras.layers.LSTM(units=P, return_sequences, zero_output_for_mask=True) in
// TransducerPrediction (base_transducer.py:71-85,123-132): gate order i,f,c,o; kernel [E,4P],
// recurrent kernel [P,4P], bias [4P]; sigmoid recurrent activation, tanh activation; on masked steps
// (t >= length) the state is carried and the emitted output is zero.
// The two GEMMs (x@W+b for all steps at once, h_{t-1}@R per step) run on the MFMA GEMM family; these
// kernels do the per-step gate math, one thread per (batch row, hidden unit).
#include "common.h"

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
    hipLaunchKernelGGL(lstm_step_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)xg, xg_stride_b, hr,
                       (const float*)h_prev, hprev_stride_b, c_prev, cprev_stride_b, lengths, t, (float*)gates,
                       gates_stride_b, c_out, c_stride_b, (float*)h_out, h_stride_b, (float*)y_out, y_stride_b, B, P);
  else if (dtype == TFASR_BF16)
    hipLaunchKernelGGL(lstm_step_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)xg, xg_stride_b, hr,
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
    hipLaunchKernelGGL(lstm_step_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, dy_stride_b, dhr,
                       dh_carry, dc_carry, (const float*)gates, gates_stride_b, c_t, c_stride_b, c_prev, cprev_stride_b,
                       lengths, t, (float*)dz, dz_stride_b, B, P);
  else if (dtype == TFASR_BF16)
    hipLaunchKernelGGL(lstm_step_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, dy_stride_b, dhr,
                       dh_carry, dc_carry, (const bf16_t*)gates, gates_stride_b, c_t, c_stride_b, c_prev, cprev_stride_b,
                       lengths, t, (bf16_t*)dz, dz_stride_b, B, P);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
