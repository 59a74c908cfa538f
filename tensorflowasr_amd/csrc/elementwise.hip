// HBM-bound pointwise / small-reduction kernels of the Conformer-Transducer hot path (gfx950).
// Every kernel moves 16 B per lane where the layout allows it (channel-last, C % 8 == 0) and keeps
// channel = fastest index so a wave reads/writes whole 128-B lines.
//   cast, colsum (bias gradients), GLU fwd/bwd (activations/glu.py:25-28), causal depthwise conv
//   fwd/bwd (convolution.py:159-228), shared attention biases u/v (conformer.py:647-663,
//   multihead_attention.py:554-558), embedding gather/scatter (embedding.py:41-48), joint
//   broadcast-add+tanh fwd/bwd (base_transducer.py:199-207,291), Adam (+L2, decoupled weight decay).
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

inline int flat_grid(long n, int per = 256) { return (int)std::max<long>(1, std::min<long>((n + per - 1) / per, 256L * 16)); }

// ---------------------------------------------------------------------------------------- cast
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long n) {
  const long n8 = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    ld8(src + i * 8, v);
    st8(dst + i * 8, v);
  }
  for (long i = n8 * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    Num<D>::st(dst + i, Num<S>::ld(src + i));
}

// ------------------------------------------------------------------------------------- dropout
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* x, T* y, long n, float p, uint64_t seed) {
  const long n8 = n / 8;
  const float inv = 1.f / (1.f - p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = drop_keep(seed, (uint64_t)(i * 8 + k), p) ? v[k] * inv : 0.f;
    st8(y + i * 8, v);
  }
  for (long i = n8 * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    Num<T>::st(y + i, drop_keep(seed, (uint64_t)i, p) ? Num<T>::ld(x + i) * inv : 0.f);
}

// -------------------------------------------------------------------------------------- colsum
// out[c] (+)= scale * sum_r x[r, c]; x has row stride ld.  grid = (row blocks, ceil(C/64))
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long ld, float* __restrict__ out,
                                                     long rows, int C, float scale) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  float acc = 0.f;
  if (c < C)
    for (long r = (long)blockIdx.x * 4 + w; r < rows; r += (long)gridDim.x * 4) acc += Num<T>::ld(x + r * ld + c);
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < C) atomicAdd(out + c, scale * (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]));
}

// 16-B variant: 32 lanes x 8 columns = 256 columns per block.y, 8 rows per block iteration, LDS reduction, 1 atomic/column/block
template <typename T>
__global__ __launch_bounds__(512) void colsum_vec_kernel(const T* __restrict__ x, long ld, float* __restrict__ out,
                                                          long rows, int C, float scale) {
  // fat blocks (see norm.hip): 32 lanes x 8 columns per row, 2 rows per wave, 4 row loads in flight per lane, LDS atomics
  __shared__ float red[16][256];  // up to 8 waves x 2 rows
  const int lane = threadIdx.x & 63, li = lane & 31, sub = lane >> 5, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c0 = blockIdx.y * 256 + li * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C) {
    const long S = (long)gridDim.x * nw * 2;
    for (long r = ((long)blockIdx.x * nw + w) * 2 + sub; r < rows; r += 4 * S) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + u * S < rows) ld8(x + (r + u * S) * ld + c0, v[u]);
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[u][k] = 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += (v[0][k] + v[1][k]) + (v[2][k] + v[3][k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[w * 2 + sub][li * 8 + k] = acc[k];
  __syncthreads();
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (threadIdx.x < 256 && c < C) {
    float s = 0.f;
    for (int q = 0; q < nw * 2; ++q) s += red[q][threadIdx.x];
    atomicAdd(out + c, scale * s);
  }
}

// f32 matrices x_b [rows, C] (b = blockIdx.z, x_b = x + b * stride) -> bf16 copies y_b (same stride, in elements) AND their f32 column sums
// added to out.p[b][C]: what the positional-projection gradients of all Conformer blocks need from the accumulated table gradients
// (the GEMM operand and the bias gradient, the latter from the f32 values) in ONE launch.  A workgroup owns 64 columns of one matrix:
// 16 lanes x 4 columns per row, 16 rows per iteration, 4 row loads in flight per lane; one writer per column (plain +=).
constexpr int CCS_MAX = 64;
struct CastColsumOut { float* p[CCS_MAX]; };
__global__ __launch_bounds__(256) void cast_colsum_many_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long stride, int rows, int C,
                                                              CastColsumOut out) {
  __shared__ float red[16][64];
  const int b = blockIdx.z, c0 = blockIdx.x * 64 + (threadIdx.x & 15) * 4, rl = threadIdx.x >> 4;
  const float* xb = x + (long)b * stride;
  bf16_t* yb = y + (long)b * stride;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    for (int r = rl; r < rows; r += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (r + 16 * u < rows) ? *reinterpret_cast<const float4*>(xb + (long)(r + 16 * u) * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + 16 * u < rows) {
          uint2 pk;
          pk.x = pack2_bf16(v[u].x, v[u].y);
          pk.y = pack2_bf16(v[u].z, v[u].w);
          *reinterpret_cast<uint2*>(yb + (long)(r + 16 * u) * C + c0) = pk;
        }
        acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[rl][(threadIdx.x & 15) * 4 + k] = acc[k];
  __syncthreads();
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && c < C && out.p[b]) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][threadIdx.x];
    out.p[b][c] += s;
  }
}

// ----------------------------------------------------------------------------------------- GLU
template <typename T>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int C) {
  const long n8 = rows * C / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 8, r = e / C;
    const int c = (int)(e % C);
    float a[8], b[8];
    ld8(x + r * 2 * C + c, a);
    ld8(x + r * 2 * C + C + c, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] *= sigmoidf_(b[k]);
    st8(y + e, a);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                      T* __restrict__ dx, long rows, int C) {
  const long n8 = rows * C / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 8, r = e / C;
    const int c = (int)(e % C);
    float a[8], b[8], d[8];
    ld8(x + r * 2 * C + c, a);
    ld8(x + r * 2 * C + C + c, b);
    ld8(dy + e, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float s = sigmoidf_(b[k]);
      b[k] = d[k] * a[k] * s * (1.f - s);
      a[k] = d[k] * s;
    }
    st8(dx + r * 2 * C + c, a);
    st8(dx + r * 2 * C + C + c, b);
  }
}

// ---------------------------------------------------------------------- causal depthwise conv
constexpr int DW_TT = 8;
constexpr int DW_MAXK = 32;
// y[b,t,c] = bias[c] + sum_k w[k,c] * x[b, t-(K-1)+k, c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ y, int Tn,
                                                         int C, int K) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int t0 = blockIdx.y * DW_TT;
  const long base = (long)blockIdx.z * Tn * C;
  float wk[DW_MAXK];
#pragma unroll
  for (int k = 0; k < DW_MAXK; ++k) wk[k] = (k < K) ? w[k * C + c] : 0.f;
  float acc[DW_TT];
  const float bv = bias ? bias[c] : 0.f;
#pragma unroll
  for (int i = 0; i < DW_TT; ++i) acc[i] = bv;
#pragma unroll
  for (int j = 0; j < DW_MAXK - 1 + DW_TT; ++j) {  // input time index ti = t0-(K-1)+j (fully unrolled: register indices static)
    if (j < K - 1 + DW_TT) {
      const int ti = t0 - (K - 1) + j;
      const float xv = (ti >= 0 && ti < Tn) ? Num<T>::ld(x + base + (long)ti * C + c) : 0.f;
#pragma unroll
      for (int i = 0; i < DW_TT; ++i) {
        const int k = j - i;  // ti = (t0+i)-(K-1)+k  (compile-time after unrolling)
        if (k >= 0 && k < DW_MAXK) acc[i] += wk[k] * xv;  // wk[k] == 0 for k >= K
      }
    }
  }
#pragma unroll
  for (int i = 0; i < DW_TT; ++i)
    if (t0 + i < Tn) Num<T>::st(y + base + (long)(t0 + i) * C + c, acc[i]);
}
// dx[b,t,c] = sum_k w[k,c] * dy[b, t+(K-1)-k, c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                              T* __restrict__ dx, int Tn, int C, int K) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int t0 = blockIdx.y * DW_TT;
  const long base = (long)blockIdx.z * Tn * C;
  float wr[DW_MAXK];  // reversed taps: wr[q] = w[K-1-q]
#pragma unroll
  for (int q = 0; q < DW_MAXK; ++q) wr[q] = (q < K) ? w[(K - 1 - q) * C + c] : 0.f;
  float acc[DW_TT];
#pragma unroll
  for (int i = 0; i < DW_TT; ++i) acc[i] = 0.f;
#pragma unroll
  for (int j = 0; j < DW_MAXK - 1 + DW_TT; ++j) {  // output-gradient time index to = t0 + j
    if (j < K - 1 + DW_TT) {
      const int to = t0 + j;
      const float dv = (to < Tn) ? Num<T>::ld(dy + base + (long)to * C + c) : 0.f;
#pragma unroll
      for (int i = 0; i < DW_TT; ++i) {
        const int q = j - i;  // to = (t0+i)+q, tap k = K-1-q
        if (q >= 0 && q < DW_MAXK) acc[i] += wr[q] * dv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < DW_TT; ++i)
    if (t0 + i < Tn) Num<T>::st(dx + base + (long)(t0 + i) * C + c, acc[i]);
}
// dw[k,c] += sum_{b,t} dy[b,t,c]*x[b,t-(K-1)+k,c]; dbias[c] += sum dy.  grid = (C/blk, t chunks of DW_WT, B)
constexpr int DW_WT = 64;
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ dw, float* __restrict__ dbias,
                                                                int Tn, int C, int K) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int t0 = blockIdx.y * DW_WT;
  const long base = (long)blockIdx.z * Tn * C;
  float acc[DW_MAXK];
#pragma unroll
  for (int k = 0; k < DW_MAXK; ++k) acc[k] = 0.f;
  float ab = 0.f;
  // sliding window of x over ti in [t-(K-1), t]
  float win[DW_MAXK];
#pragma unroll
  for (int k = 0; k < DW_MAXK; ++k) {
    const int ti = t0 - (K - 1) + k - 1;  // window before first shift: positions t0-1-(K-1)+k
    win[k] = (k < K && ti >= 0 && ti < Tn) ? Num<T>::ld(x + base + (long)ti * C + c) : 0.f;
  }
  const int tend = min(t0 + DW_WT, Tn);
  for (int t = t0; t < tend; ++t) {
#pragma unroll
    for (int k = 0; k < DW_MAXK - 1; ++k) win[k] = win[k + 1];
    // newest element sits at index K-1 (ti = t)
    const float xn = Num<T>::ld(x + base + (long)t * C + c);
#pragma unroll
    for (int k = 0; k < DW_MAXK; ++k) if (k == K - 1) win[k] = xn;
    const float d = Num<T>::ld(dy + base + (long)t * C + c);
    ab += d;
#pragma unroll
    for (int k = 0; k < DW_MAXK; ++k) acc[k] += d * win[k];
  }
#pragma unroll
  for (int k = 0; k < DW_MAXK; ++k) if (k < K) atomicAdd(dw + k * C + c, acc[k]);
  if (dbias) atomicAdd(dbias + c, ab);
}

// ---------------------------------------------------------------- shared attention biases u,v
// y1 = x + u, y2 = x + v  (x has row stride ldx; y contiguous [rows, C])
template <typename T>
__global__ __launch_bounds__(256) void bias2_fwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ u,
                                                        const float* __restrict__ v, T* __restrict__ y1,
                                                        T* __restrict__ y2, long rows, int C) {
  const long n = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i % C);
    const float xv = Num<T>::ld(x + r * ldx + c);
    Num<T>::st(y1 + i, xv + u[c]);
    Num<T>::st(y2 + i, xv + v[c]);
  }
}
// dx = d1 + d2 (row stride lddx); du += colsum(d1); dv += colsum(d2)
template <typename T>
__global__ __launch_bounds__(256) void bias2_bwd_kernel(const T* __restrict__ d1, const T* __restrict__ d2,
                                                        T* __restrict__ dx, long lddx, float* __restrict__ du,
                                                        float* __restrict__ dv, long rows, int C) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  float a1 = 0.f, a2 = 0.f;
  if (c < C)
    for (long r = (long)blockIdx.x * 4 + w; r < rows; r += (long)gridDim.x * 4) {
      const float x1 = Num<T>::ld(d1 + r * C + c), x2 = Num<T>::ld(d2 + r * C + c);
      Num<T>::st(dx + r * lddx + c, x1 + x2);
      a1 += x1;
      a2 += x2;
    }
  red[0][w][lane] = a1;
  red[1][w][lane] = a2;
  __syncthreads();
  if (w == 0 && c < C) {
    atomicAdd(du + c, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
    atomicAdd(dv + c, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
  }
}

// 16-B variants (bf16, C % 8 == 0, 16-B aligned rows): no per-element division, 8 channels per lane
__global__ __launch_bounds__(256) void bias2_fwd_vec_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ u,
                                                            const float* __restrict__ v, bf16_t* __restrict__ y1, bf16_t* __restrict__ y2,
                                                            long rows, int C) {
  const int c8 = C >> 3;
  const long n = rows * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8;
    const int c = (int)(i - r * c8) << 3;
    float xv[8], a[8], b[8];
    ld8(x + r * ldx + c, xv);
    const float4 u0 = *reinterpret_cast<const float4*>(u + c), u1 = *reinterpret_cast<const float4*>(u + c + 4);
    const float4 v0 = *reinterpret_cast<const float4*>(v + c), v1 = *reinterpret_cast<const float4*>(v + c + 4);
    a[0] = xv[0] + u0.x; a[1] = xv[1] + u0.y; a[2] = xv[2] + u0.z; a[3] = xv[3] + u0.w;
    a[4] = xv[4] + u1.x; a[5] = xv[5] + u1.y; a[6] = xv[6] + u1.z; a[7] = xv[7] + u1.w;
    b[0] = xv[0] + v0.x; b[1] = xv[1] + v0.y; b[2] = xv[2] + v0.z; b[3] = xv[3] + v0.w;
    b[4] = xv[4] + v1.x; b[5] = xv[5] + v1.y; b[6] = xv[6] + v1.z; b[7] = xv[7] + v1.w;
    st8(y1 + r * C + c, a);
    st8(y2 + r * C + c, b);
  }
}
// grid (row chunks, C/256): 32 lanes x 8 columns per row, fat 1024-thread blocks, LDS atomics, 1 global atomic/column/block
__global__ __launch_bounds__(512) void bias2_bwd_vec_kernel(const bf16_t* __restrict__ d1, const bf16_t* __restrict__ d2,
                                                             bf16_t* __restrict__ dx, long lddx, float* __restrict__ du,
                                                             float* __restrict__ dv, long rows, int C) {
  __shared__ float red[2][16][256];  // up to 8 waves x 2 rows
  const int lane = threadIdx.x & 63, li = lane & 31, sub = lane >> 5, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c0 = blockIdx.y * 256 + li * 8;
  float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C) {
    const long S = (long)gridDim.x * nw * 2;
    for (long r = ((long)blockIdx.x * nw + w) * 2 + sub; r < rows; r += 2 * S) {
      float p[2][8], q[2][8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (r + t * S < rows) { ld8(d1 + (r + t * S) * C + c0, p[t]); ld8(d2 + (r + t * S) * C + c0, q[t]); }
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) { p[t][k] = 0.f; q[t][k] = 0.f; }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (r + t * S < rows) {
          float o[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { o[k] = p[t][k] + q[t][k]; a1[k] += p[t][k]; a2[k] += q[t][k]; }
          st8(dx + (r + t * S) * lddx + c0, o);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][w * 2 + sub][li * 8 + k] = a1[k]; red[1][w * 2 + sub][li * 8 + k] = a2[k]; }
  __syncthreads();
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (threadIdx.x < 256 && c < C) {
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < nw * 2; ++q) { s1 += red[0][q][threadIdx.x]; s2 += red[1][q][threadIdx.x]; }
    atomicAdd(du + c, s1);
    atomicAdd(dv + c, s2);
  }
}

// ------------------------------------------------------------------------------------ embedding
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int32_t* __restrict__ idx,
                                                            const float* __restrict__ table, T* __restrict__ out,
                                                            long rows, int E, int V) {
  const long n = rows * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / E;
    const int e = (int)(i % E);
    const int id = min(max(idx[r], 0), V - 1);
    Num<T>::st(out + i, table[(long)id * E + e]);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int32_t* __restrict__ idx, const T* __restrict__ dout,
                                                            float* __restrict__ dtable, long rows, int E, int V) {
  const long n = rows * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / E;
    const int e = (int)(i % E);
    const int id = min(max(idx[r], 0), V - 1);
    atomicAdd(dtable + (long)id * E + e, Num<T>::ld(dout + i));
  }
}

// ---------------------------------------------------------------------------------------- joint
// h[b,t,u,:] = tanh(enc[b,t,:] + pred[b,u,:])
template <typename T>
__global__ __launch_bounds__(256) void joint_fwd_kernel(const T* __restrict__ enc, const T* __restrict__ pred,
                                                        T* __restrict__ h, int B, int Tn, int U1, int J) {
  const long n8 = (long)B * Tn * U1 * J / 8;
  const int j8 = J / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % j8) * 8;
    const long row = i / j8;  // (b,t,u)
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int b = (int)(bt / Tn);
    float a[8], p[8];
    ld8(enc + bt * J + j, a);
    ld8(pred + ((long)b * U1 + u) * J + j, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = tanh_fast(a[k] + p[k]);
    st8(h + i * 8, a);
  }
}
// MODE 0: denc[b,t,:] = sum_u dh*(1-h^2)   (grid.x = B*T)
// MODE 1: dpred[b,u,:] = sum_t dh*(1-h^2)  (grid.x = B*U1)
template <typename T, int MODE>
__global__ __launch_bounds__(128) void joint_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dh,
                                                        T* __restrict__ dout, int B, int Tn, int U1, int J) {
  const int j = (blockIdx.y * blockDim.x + threadIdx.x) * 8;
  if (j >= J) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long base, stride;
  int count;
  if (MODE == 0) { base = (long)blockIdx.x * U1 * J; stride = J; count = U1; }
  else {
    const int b = blockIdx.x / U1, u = blockIdx.x % U1;
    base = ((long)b * Tn * U1 + u) * J; stride = (long)U1 * J; count = Tn;
  }
  for (int k = 0; k < count; ++k) {
    float hv[8], dv[8];
    ld8(dh + base + k * stride + j, dv);
    if (h) {  // uniform
      ld8(h + base + k * stride + j, hv);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += dv[q] * (1.f - hv[q] * hv[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += dv[q];
    }
  }
  st8(dout + (long)blockIdx.x * J + j, acc);
}

// packed lattice variants: only the valid nodes (t < Tl_b, u <= Ul_b) exist, utterance b starts at row cell_off[b]
template <typename T>
__global__ __launch_bounds__(256) void joint_fwd_packed_kernel(const T* __restrict__ enc, const T* __restrict__ pred,
                                                               T* __restrict__ h, const long* __restrict__ cell_off,
                                                               const int32_t* __restrict__ label_len, long nrows, int B, int Tn,
                                                               int U1, int J) {
  const int j8 = J / 8;
  const long n8 = nrows * j8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % j8) * 8;
    const long row = i / j8;
    int lo = 0, hi = B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cell_off[mid] <= row) lo = mid; else hi = mid; }
    const int b = lo, u1b = min(label_len[b], U1 - 1) + 1;
    const long q = row - cell_off[b];
    const int t = (int)(q / u1b), u = (int)(q % u1b);
    float a[8], p[8];
    ld8(enc + ((long)b * Tn + t) * J + j, a);
    ld8(pred + ((long)b * U1 + u) * J + j, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = tanh_fast(a[k] + p[k]);
    st8(h + i * 8, a);
  }
}
// row-structured variant: block = (t, b); thread = (u sub-lane, 8-column chunk): no per-chunk binary search / 64-bit division,
// the encoder row is read once per thread and reused for every u
template <typename T>
__global__ __launch_bounds__(256) void joint_fwd_packed_rows_kernel(const T* __restrict__ enc, const T* __restrict__ pred,
                                                                    T* __restrict__ h, const long* __restrict__ cell_off,
                                                                    const int32_t* __restrict__ label_len, int Tn, int U1, int J) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int u1b = min(label_len[b], U1 - 1) + 1;
  const long r0 = cell_off[b];
  const int Tl = (int)((cell_off[b + 1] - r0) / u1b);
  if (t >= Tl) return;
  const int j8 = J / 8, nu = blockDim.x / j8;
  const int us = threadIdx.x / j8, c = (threadIdx.x - us * j8) * 8;
  if (us >= nu) return;
  float a[8];
  ld8(enc + ((long)b * Tn + t) * J + c, a);
  for (int u = us; u < u1b; u += nu) {
    float p[8];
    ld8(pred + ((long)b * U1 + u) * J + c, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = tanh_fast(a[k] + p[k]);
    st8(h + (r0 + (long)t * u1b + u) * J + c, p);
  }
}
// blockDim = (128, JB_RL): threadIdx.y walks the summed axis in strides of JB_RL (four independent load streams per output instead of
// one dependent loop: the t-strided MODE 1 ran at 1.4 TB/s), partial sums meet in LDS.
constexpr int JB_RL = 4;
template <typename T, int MODE>
__global__ __launch_bounds__(128 * JB_RL) void joint_bwd_packed_kernel(const T* __restrict__ h, const T* __restrict__ dh,
                                                                        T* __restrict__ dout, const long* __restrict__ cell_off,
                                                                        const int32_t* __restrict__ label_len,
                                                                        const int32_t* __restrict__ logit_len, int B, int Tn, int U1, int J) {
  __shared__ float red[JB_RL][128][8];
  const int j = (blockIdx.y * 128 + threadIdx.x) * 8;
  const bool live = j < J;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long base = 0, stride = 0;
  int count = 0;
  if (MODE == 0) {
    const int b = blockIdx.x / Tn, t = blockIdx.x % Tn;
    const int Tl = min(logit_len[b], Tn), u1b = min(label_len[b], U1 - 1) + 1;
    if (t < Tl) { base = (cell_off[b] + (long)t * u1b) * J; stride = J; count = u1b; }
  } else {
    const int b = blockIdx.x / U1, u = blockIdx.x % U1;
    const int Tl = min(logit_len[b], Tn), u1b = min(label_len[b], U1 - 1) + 1;
    if (u < u1b) { base = (cell_off[b] + u) * J; stride = (long)u1b * J; count = Tl; }
  }
  if (live) {
    for (int k = threadIdx.y; k < count; k += JB_RL) {
      float hv[8], dv[8];
      ld8(dh + base + k * stride + j, dv);
      if (h) {  // uniform
        ld8(h + base + k * stride + j, hv);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += dv[q] * (1.f - hv[q] * hv[q]);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += dv[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[threadIdx.y][threadIdx.x][q] = acc[q];
  __syncthreads();
  if (threadIdx.y == 0 && live) {
#pragma unroll
    for (int y = 1; y < JB_RL; ++y)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += red[y][threadIdx.x][q];
    st8(dout + (long)blockIdx.x * J + j, acc);
  }
}

// ----------------------------------------------------------------------------------------- Adam
// keras.optimizers.Adam semantics (bias-corrected, decoupled weight_decay applied first) + the L2
// kernel regulariser gradient 2*l2*p on the first n_reg elements (small.yml.j2:67-69,73-87).
// shadow != nullptr: the bf16 copy of the parameters the MFMA kernels read is written in the same pass (it was a separate cast over the
// whole buffer after the optimizer: one more read of 4 B / parameter and one more launch per step)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, long n_reg,
                                                   float lr, float b1, float b2, float eps, float wd, float l2,
                                                   float gscale, float bc1, float bc2, bf16_t* __restrict__ shadow) {
  const float alpha = lr * sqrtf(bc2) / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pv = p[i];
    float gv = g[i] * gscale;
    if (i < n_reg) gv += 2.f * l2 * pv;
    pv -= pv * wd * lr;
    const float mv = m[i] + (gv - m[i]) * (1.f - b1);
    const float vv = v[i] + (gv * gv - v[i]) * (1.f - b2);
    m[i] = mv;
    v[i] = vv;
    const float pn = pv - alpha * mv / (sqrtf(vv) + eps);
    p[i] = pn;
    if (shadow) shadow[i] = f32_to_bf16(pn);
  }
}

// y += alpha * x  (f32 vectors: sync-BN gradient hand-off, gradient accumulation)
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

// x[i] += stddev * N(0,1), the normal deviate a pure function of (seed, i): two hashed 32-bit uniforms through Box-Muller (one pair of
// uniforms serves elements 2k and 2k+1: cosine / sine branch).  Weight noise (layer_util.add_gwn, base_transducer.py:382-425) and
// gradient noise (math_util.add_gauss_noise, base_model.py:185-191).
__global__ __launch_bounds__(256) void gauss_noise_kernel(float* __restrict__ x, long n, float stddev, uint64_t seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const uint64_t pair = (uint64_t)i >> 1;
    const uint32_t h1 = drop_hash(seed, pair), h2 = drop_hash(seed ^ 0x5851F42D4C957F2DULL, pair);
    const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    x[i] += stddev * r * ((i & 1) ? sn : cs);
  }
}

// sum of squares (regularisation loss term); out[0] += sum p^2
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ p, long n, float* __restrict__ out) {
  __shared__ float red[16];
  float a = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a += p[i] * p[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) atomicAdd(out, a);
}

// ------------------------------------------------------------------------------- SpecAugment
// x[b,t,f] = mval where any (f0<=f<f0+fw) or (t0<=t<t0+tw)  (specaugment.py:78-86,128-136)
template <typename T>
__global__ __launch_bounds__(256) void specaug_kernel(T* __restrict__ x, const int32_t* __restrict__ fmask,
                                                      const int32_t* __restrict__ tmask, int nf, int nt, int B, int Tn,
                                                      int F, float mval) {
  const long n = (long)B * Tn * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int t = (int)((i / F) % Tn);
    const int b = (int)(i / ((long)F * Tn));
    bool hit = false;
    for (int k = 0; k < nf; ++k) {
      const int f0 = fmask[(b * nf + k) * 2], fw = fmask[(b * nf + k) * 2 + 1];
      hit |= (f >= f0 && f < f0 + fw);
    }
    for (int k = 0; k < nt; ++k) {
      const int t0 = tmask[(b * nt + k) * 2], tw = tmask[(b * nt + k) * 2 + 1];
      hit |= (t >= t0 && t < t0 + tw);
    }
    if (hit) Num<T>::st(x + i, mval);
  }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
  do { if ((dtype) == TFASR_F32) { CALL_F32; } else if ((dtype) == TFASR_BF16) { CALL_BF16; } else return TFASR_STATUS_INVALID_VALUE; } while (0)

extern "C" int tfasr_cast(const void* src, void* dst, long n, int src_dtype, int dst_dtype, void* stream_) {
  if (!src || !dst || n <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid(n / 8 + 1);
  if (src_dtype == TFASR_F32 && dst_dtype == TFASR_BF16)
    TFASR_KLAUNCH((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == TFASR_BF16 && dst_dtype == TFASR_F32)
    TFASR_KLAUNCH((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == TFASR_F32 && dst_dtype == TFASR_F32)
    TFASR_KLAUNCH((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else if (src_dtype == TFASR_BF16 && dst_dtype == TFASR_BF16)
    TFASR_KLAUNCH((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_dropout(const void* x, void* y, long n, float p, long seed, int dtype, void* stream_) {
  if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid(n / 8 + 1);
  DISPATCH_T(dtype, TFASR_KLAUNCH(dropout_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, n, p, (uint64_t)seed),
             TFASR_KLAUNCH(dropout_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n, p, (uint64_t)seed));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_colsum(const void* x, long ld, float* out, long rows, int C, float scale, int dtype, void* stream_) {
  if (!x || !out || rows <= 0 || C <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16 && (C % 8) == 0 && (ld % 8) == 0 && ((((uintptr_t)x) & 15) == 0)) {
    const int cb = (C + 255) / 256;
    static const int thr = 512;
    static const int cap = 192;
    dim3 gridv((int)std::max<long>(1, std::min<long>(rows / (thr / 8) + 1, std::max(32, cap / cb))), cb);
    TFASR_KLAUNCH(colsum_vec_kernel<bf16_t>, gridv, dim3(thr), 0, s, (const bf16_t*)x, ld, out, rows, C, scale);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  const int gx = (int)std::max<long>(1, std::min<long>(rows / 64 + 1, 256));
  dim3 grid(gx, (C + 63) / 64);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ld, out, rows, C, scale),
             TFASR_KLAUNCH(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ld, out, rows, C, scale));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_cast_colsum_many(const float* x, void* y, long stride, int nmat, int rows, int C, float* const* colsum, void* stream_) {
  if (!x || !y || !colsum || nmat <= 0 || nmat > CCS_MAX || rows <= 0 || C <= 0 || stride < (long)rows * C) return TFASR_STATUS_INVALID_VALUE;
  if ((C & 3) || (stride & 3) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 7)) return TFASR_STATUS_UNSUPPORTED;
  CastColsumOut o;
  for (int i = 0; i < CCS_MAX; ++i) o.p[i] = i < nmat ? colsum[i] : nullptr;
  TFASR_KLAUNCH(cast_colsum_many_kernel, dim3((C + 63) / 64, 1, nmat), dim3(256), 0, (hipStream_t)stream_, x, (bf16_t*)y, stride, rows, C, o);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_glu_fwd(const void* x, void* y, long rows, int C, int dtype, void* stream_) {
  if (!x || !y || rows <= 0 || C <= 0 || C % 8) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid(rows * C / 8);
  DISPATCH_T(dtype, TFASR_KLAUNCH(glu_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, rows, C),
             TFASR_KLAUNCH(glu_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, rows, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_glu_bwd(const void* x, const void* dy, void* dx, long rows, int C, int dtype, void* stream_) {
  if (!x || !dy || !dx || rows <= 0 || C <= 0 || C % 8) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid(rows * C / 8);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(glu_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)dx, rows, C),
             TFASR_KLAUNCH(glu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, rows, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

// dwconv.hip: bf16 channel-pair kernels (which: 0 fwd, 1 data gradient, 2 weight gradient); UNSUPPORTED -> scalar kernels below
int tfasr_dwconv_pair_try(int which, const void* x, const void* dy, const float* w, const float* bias, void* y, float* dw, float* dbias, int B,
                          int T, int C, int K, hipStream_t s);

extern "C" int tfasr_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int B, int T, int C, int K,
                                int dtype, void* stream_) {
  if (!x || !w || !y || B <= 0 || T <= 0 || C <= 0 || K <= 0 || K > DW_MAXK) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16) {
    const int st = tfasr_dwconv_pair_try(0, x, nullptr, w, bias, y, nullptr, nullptr, B, T, C, K, s);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  const int bx = C >= 256 ? 256 : ((C + 63) / 64) * 64;
  dim3 grid((C + bx - 1) / bx, (T + DW_TT - 1) / DW_TT, B);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(dwconv_fwd_kernel<float>, grid, dim3(bx), 0, s, (const float*)x, w, bias, (float*)y, T, C, K),
             TFASR_KLAUNCH(dwconv_fwd_kernel<bf16_t>, grid, dim3(bx), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, T, C, K));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_dwconv_bwd_data(const void* dy, const float* w, void* dx, int B, int T, int C, int K, int dtype,
                                     void* stream_) {
  if (!dy || !w || !dx || B <= 0 || T <= 0 || C <= 0 || K <= 0 || K > DW_MAXK) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16) {
    const int st = tfasr_dwconv_pair_try(1, nullptr, dy, w, nullptr, dx, nullptr, nullptr, B, T, C, K, s);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  const int bx = C >= 256 ? 256 : ((C + 63) / 64) * 64;
  dim3 grid((C + bx - 1) / bx, (T + DW_TT - 1) / DW_TT, B);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(dwconv_bwd_data_kernel<float>, grid, dim3(bx), 0, s, (const float*)dy, w, (float*)dx, T, C, K),
             TFASR_KLAUNCH(dwconv_bwd_data_kernel<bf16_t>, grid, dim3(bx), 0, s, (const bf16_t*)dy, w, (bf16_t*)dx, T, C, K));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
int tfasr_dwconv_wgrad_ws_try(const void* x, const void* dy, float* dw, float* dbias, int B, int T, int C, int K, float* ws, size_t ws_bytes,
                              hipStream_t s);  // dwconv.hip

extern "C" int tfasr_dwconv_bwd_weight_workspace_size(int B, int T, int C, int K, size_t* bytes) {
  if (!bytes || B <= 0 || T <= 0 || C <= 0 || K <= 0) return TFASR_STATUS_INVALID_VALUE;
  *bytes = (size_t)B * ((T + 63) / 64) * ((C + 255) / 256) * (size_t)(K + 1) * 256 * 4;
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_dwconv_bwd_weight_ws(const void* x, const void* dy, float* dw, float* dbias, int B, int T, int C, int K, int dtype,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
  if (!x || !dy || !dw || B <= 0 || T <= 0 || C <= 0 || K <= 0 || K > DW_MAXK) return TFASR_STATUS_INVALID_VALUE;
  if (dtype == TFASR_BF16 && workspace) {
    const int st = tfasr_dwconv_wgrad_ws_try(x, dy, dw, dbias, B, T, C, K, (float*)workspace, workspace_bytes, (hipStream_t)stream_);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  return tfasr_dwconv_bwd_weight(x, dy, dw, dbias, B, T, C, K, dtype, stream_);
}

int tfasr_dwconv_wgrad_many_try(const void* const* x, const void* const* dy, float* const* dw, float* const* dbias, int n, int B, int T, int C, int K,
                                float* ws, size_t ws_bytes, hipStream_t s);  // dwconv.hip

// n <= 32 weight gradients of ONE shape (e.g. the layers of a ContextNet stage, or every Conformer block's) as one tile launch + one reduce
// launch; fallback: one by one.  workspace >= n * tfasr_dwconv_bwd_weight_workspace_size(B, T, C, K) bytes for the batched route.
extern "C" int tfasr_dwconv_bwd_weight_many(const void* const* x, const void* const* dy, float* const* dw, float* const* dbias, int n, int B, int T, int C,
                                            int K, int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!x || !dy || !dw || n <= 0 || n > 32 || B <= 0 || T <= 0 || C <= 0 || K <= 0 || K > DW_MAXK) return TFASR_STATUS_INVALID_VALUE;
  for (int i = 0; i < n; ++i)
    if (!x[i] || !dy[i] || !dw[i]) return TFASR_STATUS_INVALID_VALUE;
  if (dtype == TFASR_BF16 && workspace) {
    const int st = tfasr_dwconv_wgrad_many_try(x, dy, dw, dbias, n, B, T, C, K, (float*)workspace, workspace_bytes, (hipStream_t)stream_);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  for (int i = 0; i < n; ++i) {
    const int st = tfasr_dwconv_bwd_weight(x[i], dy[i], dw[i], dbias ? dbias[i] : nullptr, B, T, C, K, dtype, stream_);
    if (st != TFASR_STATUS_SUCCESS) return st;
  }
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_dwconv_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int T, int C,
                                       int K, int dtype, void* stream_) {
  if (!x || !dy || !dw || B <= 0 || T <= 0 || C <= 0 || K <= 0 || K > DW_MAXK) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16) {
    const int st = tfasr_dwconv_pair_try(2, x, dy, nullptr, nullptr, nullptr, dw, dbias, B, T, C, K, s);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  const int bx = C >= 256 ? 256 : ((C + 63) / 64) * 64;
  dim3 grid((C + bx - 1) / bx, (T + DW_WT - 1) / DW_WT, B);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(dwconv_bwd_weight_kernel<float>, grid, dim3(bx), 0, s, (const float*)x, (const float*)dy, dw, dbias, T, C, K),
             TFASR_KLAUNCH(dwconv_bwd_weight_kernel<bf16_t>, grid, dim3(bx), 0, s, (const bf16_t*)x, (const bf16_t*)dy, dw, dbias, T, C, K));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_bias2_fwd(const void* x, long ldx, const float* u, const float* v, void* y1, void* y2, long rows,
                               int C, int dtype, void* stream_) {
  if (!x || !u || !v || !y1 || !y2 || rows <= 0 || C <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (dtype == TFASR_BF16 && (C % 8) == 0 && (ldx % 8) == 0 && al(x) && al(u) && al(v) && al(y1) && al(y2)) {
    TFASR_KLAUNCH(bias2_fwd_vec_kernel, dim3(flat_grid(rows * C / 8)), dim3(256), 0, s, (const bf16_t*)x, ldx, u, v, (bf16_t*)y1, (bf16_t*)y2, rows, C);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  const int grid = flat_grid(rows * C);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(bias2_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, u, v, (float*)y1, (float*)y2, rows, C),
             TFASR_KLAUNCH(bias2_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ldx, u, v, (bf16_t*)y1, (bf16_t*)y2, rows, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_bias2_bwd(const void* d1, const void* d2, void* dx, long lddx, float* du, float* dv, long rows,
                               int C, int dtype, void* stream_) {
  if (!d1 || !d2 || !dx || !du || !dv || rows <= 0 || C <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (dtype == TFASR_BF16 && (C % 8) == 0 && (lddx % 8) == 0 && al(d1) && al(d2) && al(dx)) {
    static const int thr = 512;
    static const int cap = 192;
    dim3 gridv((int)std::max<long>(1, std::min<long>(rows / (thr / 8) + 1, cap)), (C + 255) / 256);
    TFASR_KLAUNCH(bias2_bwd_vec_kernel, gridv, dim3(thr), 0, s, (const bf16_t*)d1, (const bf16_t*)d2, (bf16_t*)dx, lddx, du, dv, rows, C);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  const int gx = (int)std::max<long>(1, std::min<long>(rows / 64 + 1, 256));
  dim3 grid(gx, (C + 63) / 64);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(bias2_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)d1, (const float*)d2, (float*)dx, lddx, du, dv, rows, C),
             TFASR_KLAUNCH(bias2_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)d1, (const bf16_t*)d2, (bf16_t*)dx, lddx, du, dv, rows, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_embedding_fwd(const int32_t* idx, const float* table, void* out, long rows, int E, int V,
                                   int dtype, void* stream_) {
  if (!idx || !table || !out || rows <= 0 || E <= 0 || V <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid(rows * E);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(embedding_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, idx, table, (float*)out, rows, E, V),
             TFASR_KLAUNCH(embedding_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, idx, table, (bf16_t*)out, rows, E, V));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_embedding_bwd(const int32_t* idx, const void* dout, float* dtable, long rows, int E, int V,
                                   int dtype, void* stream_) {
  if (!idx || !dout || !dtable || rows <= 0 || E <= 0 || V <= 0) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid(rows * E);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(embedding_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, idx, (const float*)dout, dtable, rows, E, V),
             TFASR_KLAUNCH(embedding_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, idx, (const bf16_t*)dout, dtable, rows, E, V));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_joint_fwd(const void* enc, const void* pred, void* h, int B, int T, int U1, int J, int dtype,
                               void* stream_) {
  if (!enc || !pred || !h || B <= 0 || T <= 0 || U1 <= 0 || J <= 0 || J % 8) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid((long)B * T * U1 * J / 8);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(joint_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)enc, (const float*)pred, (float*)h, B, T, U1, J),
             TFASR_KLAUNCH(joint_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)enc, (const bf16_t*)pred, (bf16_t*)h, B, T, U1, J));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_joint_bwd(const void* h, const void* dh, void* denc, void* dpred, int B, int T, int U1, int J,
                               int dtype, void* stream_) {
  if (!h || !dh || !denc || !dpred || B <= 0 || T <= 0 || U1 <= 0 || J <= 0 || J % 8) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int gy = (J / 8 + 127) / 128;
  dim3 g0(B * T, gy), g1(B * U1, gy);
  DISPATCH_T(dtype,
             { TFASR_KLAUNCH((joint_bwd_kernel<float, 0>), g0, dim3(128), 0, s, (const float*)h, (const float*)dh, (float*)denc, B, T, U1, J);
               TFASR_KLAUNCH((joint_bwd_kernel<float, 1>), g1, dim3(128), 0, s, (const float*)h, (const float*)dh, (float*)dpred, B, T, U1, J); },
             { TFASR_KLAUNCH((joint_bwd_kernel<bf16_t, 0>), g0, dim3(128), 0, s, (const bf16_t*)h, (const bf16_t*)dh, (bf16_t*)denc, B, T, U1, J);
               TFASR_KLAUNCH((joint_bwd_kernel<bf16_t, 1>), g1, dim3(128), 0, s, (const bf16_t*)h, (const bf16_t*)dh, (bf16_t*)dpred, B, T, U1, J); });
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_joint_fwd_packed(const void* enc, const void* pred, void* h, const long* cell_off, const int32_t* label_len,
                                      long total_cells, int B, int T, int U1, int J, int dtype, void* stream_) {
  if (!enc || !pred || !h || !cell_off || !label_len || total_cells <= 0 || B <= 0 || T <= 0 || U1 <= 0 || J <= 0 || J % 8)
    return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (J / 8 <= 256) {
    dim3 g2(T, B);
    DISPATCH_T(dtype,
               TFASR_KLAUNCH(joint_fwd_packed_rows_kernel<float>, g2, dim3(256), 0, s, (const float*)enc, (const float*)pred, (float*)h, cell_off, label_len, T, U1, J),
               TFASR_KLAUNCH(joint_fwd_packed_rows_kernel<bf16_t>, g2, dim3(256), 0, s, (const bf16_t*)enc, (const bf16_t*)pred, (bf16_t*)h, cell_off, label_len, T, U1, J));
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  const int grid = flat_grid(total_cells * J / 8);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(joint_fwd_packed_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)enc, (const float*)pred, (float*)h, cell_off, label_len, total_cells, B, T, U1, J),
             TFASR_KLAUNCH(joint_fwd_packed_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)enc, (const bf16_t*)pred, (bf16_t*)h, cell_off, label_len, total_cells, B, T, U1, J));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_joint_bwd_packed(const void* h, const void* dh, void* denc, void* dpred, const long* cell_off,
                                      const int32_t* label_len, const int32_t* logit_len, int B, int T, int U1, int J, int dtype,
                                      void* stream_) {
  if (!dh || !denc || !dpred || !cell_off || !label_len || !logit_len || B <= 0 || T <= 0 || U1 <= 0 || J <= 0 || J % 8)
    return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int gy = (J / 8 + 127) / 128;
  dim3 g0(B * T, gy), g1(B * U1, gy);
  DISPATCH_T(dtype,
             { TFASR_KLAUNCH((joint_bwd_packed_kernel<float, 0>), g0, dim3(128, JB_RL), 0, s, (const float*)h, (const float*)dh, (float*)denc, cell_off, label_len, logit_len, B, T, U1, J);
               TFASR_KLAUNCH((joint_bwd_packed_kernel<float, 1>), g1, dim3(128, JB_RL), 0, s, (const float*)h, (const float*)dh, (float*)dpred, cell_off, label_len, logit_len, B, T, U1, J); },
             { TFASR_KLAUNCH((joint_bwd_packed_kernel<bf16_t, 0>), g0, dim3(128, JB_RL), 0, s, (const bf16_t*)h, (const bf16_t*)dh, (bf16_t*)denc, cell_off, label_len, logit_len, B, T, U1, J);
               TFASR_KLAUNCH((joint_bwd_packed_kernel<bf16_t, 1>), g1, dim3(128, JB_RL), 0, s, (const bf16_t*)h, (const bf16_t*)dh, (bf16_t*)dpred, cell_off, label_len, logit_len, B, T, U1, J); });
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_adam_shadow(float* p, const float* g, float* m, float* v, long n, long n_reg, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, float l2, float grad_scale, long step, void* shadow_bf16,
                                 void* stream_) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return TFASR_STATUS_INVALID_VALUE;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  TFASR_KLAUNCH(adam_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream_, p, g, m, v, n, n_reg, lr, beta1,
                     beta2, eps, weight_decay, l2, grad_scale, bc1, bc2, (bf16_t*)shadow_bf16);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_adam(float* p, const float* g, float* m, float* v, long n, long n_reg, float lr, float beta1,
                          float beta2, float eps, float weight_decay, float l2, float grad_scale, long step,
                          void* stream_) {
  return tfasr_adam_shadow(p, g, m, v, n, n_reg, lr, beta1, beta2, eps, weight_decay, l2, grad_scale, step, nullptr, stream_);
}

extern "C" int tfasr_gauss_noise(float* x, long n, float stddev, long seed, void* stream_) {
  if (!x || n < 0) return TFASR_STATUS_INVALID_VALUE;
  if (n == 0 || stddev == 0.f) return TFASR_STATUS_SUCCESS;
  TFASR_KLAUNCH(gauss_noise_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream_, x, n, stddev, (uint64_t)seed);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_axpy(float* y, const float* x, float alpha, long n, void* stream_) {
  if (!x || !y || n <= 0) return TFASR_STATUS_INVALID_VALUE;
  TFASR_KLAUNCH(axpy_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream_, y, x, alpha, n);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_sumsq(const float* p, long n, float* out, void* stream_) {
  if (!p || !out || n <= 0) return TFASR_STATUS_INVALID_VALUE;
  TFASR_KLAUNCH(sumsq_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream_, p, n, out);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_specaugment(void* x, const int32_t* fmask, const int32_t* tmask, int nf, int nt, int B, int T,
                                 int F, float mask_value, int dtype, void* stream_) {
  if (!x || B <= 0 || T <= 0 || F <= 0 || (nf > 0 && !fmask) || (nt > 0 && !tmask)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid((long)B * T * F);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(specaug_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)x, fmask, tmask, nf, nt, B, T, F, mask_value),
             TFASR_KLAUNCH(specaug_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t*)x, fmask, tmask, nf, nt, B, T, F, mask_value));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
