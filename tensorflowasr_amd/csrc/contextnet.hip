// ContextNet encoder pieces that the Conformer path does not already provide (models/encoders/contextnet.py:40-298):
//   * row subsampling for the strided causal SeparableConv1D (a causal stride-s convolution IS the stride-1 causal convolution
//     sampled at t = s*t': left pad K-1, VALID, stride s - convolution.py:159-228 / keras SeparableConv1D);
//   * squeeze-and-excite (SEModule.call, contextnet.py:159-170): masked global average pool over the valid frames
//     (keras GlobalAveragePooling1D with the propagated sequence mask), broadcast scale, and their gradients;
//   * residual add + activation of ConvBlock.call (contextnet.py:283-292).
// All HBM-bound pointwise / row-reduction kernels: [B, T, C] channel-last, 16-byte lanes, C % 8 == 0.
#include "common.h"
#include <algorithm>

namespace {

inline int flat_grid(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 256L * 32)); }

template <typename T>
__global__ __launch_bounds__(256) void subsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Tn, int T2, int C, int stride) {
  const int c8 = C >> 3;
  const long n = (long)B * T2 * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const long row = i / c8;
    const int t2 = (int)(row % T2), b = (int)(row / T2);
    float v[8];
    ld8(x + ((long)b * Tn + (long)t2 * stride) * C + c, v);
    st8(y + i * 8, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void subsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Tn, int T2, int C, int stride) {
  const int c8 = C >> 3;
  const long n = (long)B * Tn * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const long row = i / c8;
    const int t = (int)(row % Tn), b = (int)(row / Tn);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t % stride == 0 && t / stride < T2) ld8(dy + ((long)b * T2 + t / stride) * C + c, v);
    st8(dx + i * 8, v);
  }
}

// pool[b, c] = mean_{t < len_b} x[b, t, c]   (MODE 0)      ds[b, c] = sum_t dy[b,t,c] * x[b,t,c]   (MODE 1, all t)
// grid (C/256 slabs, B): 32 lanes x 8 channels per row, 8 rows per block iteration, LDS reduction over the 8 row lanes
template <typename T, int MODE>
__global__ __launch_bounds__(256) void se_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const int32_t* __restrict__ lengths,
                                                        float* __restrict__ out, int Tn, int C) {
  __shared__ float red[8][256];
  const int b = blockIdx.y, li = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + li * 8;
  const int len = MODE == 0 ? min(max(lengths ? lengths[b] : Tn, 0), Tn) : Tn;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C)
    for (int t = rl; t < len; t += 8) {
      float v[8];
      ld8(x + ((long)b * Tn + t) * C + c0, v);
      if (MODE == 1) {
        float d[8];
        ld8(dy + ((long)b * Tn + t) * C + c0, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k] * d[k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      }
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl][li * 8 + k] = acc[k];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][threadIdx.x];
    out[(long)b * C + c] = MODE == 0 ? s / (float)max(len, 1) : s;
  }
}

// y = x * s[b, c]
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y, int B, int Tn, int C) {
  const int c8 = C >> 3;
  const long n = (long)B * Tn * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const int b = (int)(i / ((long)Tn * c8));
    float v[8], sc[8];
    ld8(x + i * 8, v);
    ld8(s + (long)b * C + c, sc);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= sc[k];
    st8(y + i * 8, v);
  }
}
// dx = dy * s[b, c] + (t < len_b ? dpool[b, c] / len_b : 0)
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const T* __restrict__ dy, const float* __restrict__ s, const float* __restrict__ dpool,
                                                           const int32_t* __restrict__ lengths, T* __restrict__ dx, int B, int Tn, int C) {
  const int c8 = C >> 3;
  const long n = (long)B * Tn * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const long row = i / c8;
    const int t = (int)(row % Tn), b = (int)(row / Tn);
    const int len = min(max(lengths ? lengths[b] : Tn, 0), Tn);
    float v[8], sc[8], dp[8];
    ld8(dy + i * 8, v);
    ld8(s + (long)b * C + c, sc);
    ld8(dpool + (long)b * C + c, dp);
    const float w = t < len ? 1.f / (float)max(len, 1) : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + w * dp[k];
    st8(dx + i * 8, v);
  }
}

// y = act(a + b) ; backward: d = dy * act'(a + b)
template <typename T>
__global__ __launch_bounds__(256) void add_act_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n8, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float u[8], v[8];
    ld8(a + i * 8, u);
    if (b) { ld8(b + i * 8, v); } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = u[k] + v[k];
      u[k] = act == TFASR_ACT_SWISH ? swishf_(z) : (act == TFASR_ACT_SIGMOID ? sigmoidf_(z) : z);
    }
    st8(y + i * 8, u);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void add_act_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ dy, T* __restrict__ d,
                                                          long n8, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float u[8], v[8], g[8];
    ld8(a + i * 8, u);
    if (b) { ld8(b + i * 8, v); } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    ld8(dy + i * 8, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = u[k] + v[k];
      if (act == TFASR_ACT_SWISH) g[k] *= dswishf_(z);
      else if (act == TFASR_ACT_SIGMOID) { const float sg = sigmoidf_(z); g[k] *= sg * (1.f - sg); }
    }
    st8(d + i * 8, g);
  }
}

}  // namespace

#define CN_DISPATCH(dtype, F32, BF16) do { if ((dtype) == TFASR_F32) { F32; } else if ((dtype) == TFASR_BF16) { BF16; } else return TFASR_STATUS_INVALID_VALUE; } while (0)

extern "C" int tfasr_rows_subsample_fwd(const void* x, void* y, int B, int T, int C, int stride, int dtype, void* stream_) {
  if (!x || !y || B <= 0 || T <= 0 || C <= 0 || (C & 7) || stride <= 0) return TFASR_STATUS_INVALID_VALUE;
  const int T2 = (T + stride - 1) / stride;
  hipStream_t s = (hipStream_t)stream_;
  const int g = flat_grid((long)B * T2 * C / 8);
  CN_DISPATCH(dtype, TFASR_KLAUNCH(subsample_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, (float*)y, B, T, T2, C, stride),
              TFASR_KLAUNCH(subsample_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, T, T2, C, stride));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_rows_subsample_bwd(const void* dy, void* dx, int B, int T, int C, int stride, int dtype, void* stream_) {
  if (!dy || !dx || B <= 0 || T <= 0 || C <= 0 || (C & 7) || stride <= 0) return TFASR_STATUS_INVALID_VALUE;
  const int T2 = (T + stride - 1) / stride;
  hipStream_t s = (hipStream_t)stream_;
  const int g = flat_grid((long)B * T * C / 8);
  CN_DISPATCH(dtype, TFASR_KLAUNCH(subsample_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (float*)dx, B, T, T2, C, stride),
              TFASR_KLAUNCH(subsample_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, B, T, T2, C, stride));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_se_pool(const void* x, const int32_t* lengths, float* pool, int B, int T, int C, int dtype, void* stream_) {
  if (!x || !pool || B <= 0 || T <= 0 || C <= 0 || (C & 7)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  dim3 g((C + 255) / 256, B);
  CN_DISPATCH(dtype, TFASR_KLAUNCH((se_reduce_kernel<float, 0>), g, dim3(256), 0, s, (const float*)x, (const float*)nullptr, lengths, pool, T, C),
              TFASR_KLAUNCH((se_reduce_kernel<bf16_t, 0>), g, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, lengths, pool, T, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_se_scale_fwd(const void* x, const float* scale, void* y, int B, int T, int C, int dtype, void* stream_) {
  if (!x || !scale || !y || B <= 0 || T <= 0 || C <= 0 || (C & 7)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int g = flat_grid((long)B * T * C / 8);
  CN_DISPATCH(dtype, TFASR_KLAUNCH(se_scale_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, scale, (float*)y, B, T, C),
              TFASR_KLAUNCH(se_scale_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)x, scale, (bf16_t*)y, B, T, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_se_scale_bwd_reduce(const void* x, const void* dy, float* dscale, int B, int T, int C, int dtype, void* stream_) {
  if (!x || !dy || !dscale || B <= 0 || T <= 0 || C <= 0 || (C & 7)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  dim3 g((C + 255) / 256, B);
  CN_DISPATCH(dtype, TFASR_KLAUNCH((se_reduce_kernel<float, 1>), g, dim3(256), 0, s, (const float*)x, (const float*)dy, (const int32_t*)nullptr, dscale, T, C),
              TFASR_KLAUNCH((se_reduce_kernel<bf16_t, 1>), g, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (const int32_t*)nullptr, dscale, T, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_se_bwd_apply(const void* dy, const float* scale, const float* dpool, const int32_t* lengths, void* dx, int B, int T, int C,
                                  int dtype, void* stream_) {
  if (!dy || !scale || !dpool || !dx || B <= 0 || T <= 0 || C <= 0 || (C & 7)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int g = flat_grid((long)B * T * C / 8);
  CN_DISPATCH(dtype, TFASR_KLAUNCH(se_bwd_apply_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, scale, dpool, lengths, (float*)dx, B, T, C),
              TFASR_KLAUNCH(se_bwd_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, scale, dpool, lengths, (bf16_t*)dx, B, T, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_add_act_fwd(const void* a, const void* b, void* y, long n, int act, int dtype, void* stream_) {
  if (!a || !y || n <= 0 || (n & 7)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int g = flat_grid(n / 8);
  CN_DISPATCH(dtype, TFASR_KLAUNCH(add_act_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)y, n / 8, act),
              TFASR_KLAUNCH(add_act_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n / 8, act));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_add_act_bwd(const void* a, const void* b, const void* dy, void* d, long n, int act, int dtype, void* stream_) {
  if (!a || !dy || !d || n <= 0 || (n & 7)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int g = flat_grid(n / 8);
  CN_DISPATCH(dtype, TFASR_KLAUNCH(add_act_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)a, (const float*)b, (const float*)dy, (float*)d, n / 8, act),
              TFASR_KLAUNCH(add_act_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)dy, (bf16_t*)d, n / 8, act));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
