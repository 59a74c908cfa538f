// Conv2dSubsampling pieces for gfx950 (subsampling.py:163-254; causal 3x3 stride-2 Conv2D = left-pad 2 in
// time AND frequency then VALID, convolution.py:25-37,132-144; output length ceil(L/2), math_util.py:282-305).
//   conv1 (Cin = 1): direct kernel, 9 taps per output, channel-last write (HBM-bound on the output)
//   conv2 (Cin = C): im2col -> MFMA GEMM (K = 9*C) -> col2im for the data gradient
// Layout: activations [B, T, F, C] channel-last; kernels keras-style [kh, kw, cin, cout].
#include "common.h"
#include <type_traits>
#include <algorithm>

namespace {

inline int flat_grid(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 256L * 32)); }

// Haloed space-to-depth ("S") layout of a [B, T1, F1, C] activation: [B, T2+1, F2+1, 2, 2, C] with T2 = ceil(T1/2), F2 = ceil(F1/2);
// element (b, t, f, :) sits in row (b, t/2 + 1, f/2 + 1), parity block (t%2, f%2); row 0 / column 0 of every sample and the
// slots past an odd T1 / F1 are zero.  In this layout the input of tap (kh, kw) of the causal 3x3 stride-2 conv for output row
// (b, tt, ff) is the SAME row shifted by a constant, so conv2 is a plain GEMM over 9 K-segments (no patch matrix).
__device__ __forceinline__ long s2d_off(int b, int t, int f, int T2, int F2, int C) {
  return ((((long)b * (T2 + 1) + (t >> 1) + 1) * (F2 + 1) + (f >> 1) + 1) * 4 + ((t & 1) * 2 + (f & 1))) * C;
}

// y[b,t,f,c] = bias[c] + sum_{kh,kw} w[kh,kw,0,c] * x[b, 2t+kh-2, 2f+kw-2]
// The grid stride is a multiple of C/8, so every thread keeps ONE channel group: its 72 taps + 8 biases live in registers.
template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y, int B, int T0,
                                                        int F0, int T1, int F1, int C, int s2d) {
  const int c8n = C / 8;
  const long n8 = (long)B * T1 * F1 * c8n;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;  // host guarantees stride % c8n == 0
  const int c = (int)(i0 % c8n) * 8;
  float wr[9][8], br[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) br[k] = bias ? bias[c + k] : 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int k = 0; k < 8; ++k) wr[tap][k] = w[tap * C + c + k];
  for (long i = i0; i < n8; i += stride) {
    long pos = i / c8n;
    const int f = (int)(pos % F1); pos /= F1;
    const int t = (int)(pos % T1);
    const int b = (int)(pos / T1);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = br[k];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ti = 2 * t + kh - 2;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int fi = 2 * f + kw - 2;
        float xv = 0.f;
        if (ti >= 0 && ti < T0 && fi >= 0 && fi < F0) xv = Num<T>::ld(x + ((long)b * T0 + ti) * F0 + fi);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += wr[kh * 3 + kw][k] * xv;
      }
    }
    st8(s2d ? y + s2d_off(b, t, f, (T1 + 1) / 2, (F1 + 1) / 2, C) + c : y + i * 8, acc);
  }
}

// dw[kh,kw,c] += sum dy[b,t,f,c]*x[...]; db[c] += sum dy.  LPR lanes x 8 channels cover one position (16-B loads of dy);
// 64/LPR positions per wave iteration; per-block LDS reduction, one atomic per (tap, channel) per block.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void conv1_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               float* __restrict__ dw, float* __restrict__ db, int B,
                                                               int T0, int F0, int T1, int F1, int C, int s2d) {
  constexpr int PPW = 64 / LPR;
  __shared__ float red[4][10][LPR * 8 + 1];
  const int lane = threadIdx.x & 63, li = lane % LPR, sub = lane / LPR, w = threadIdx.x >> 6;
  const int c0 = li * 8;
  const bool act = c0 < C;
  const long npos = (long)B * T1 * F1;
  float acc[10][8];
#pragma unroll
  for (int q = 0; q < 10; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[q][k] = 0.f;
  const long p0 = ((long)blockIdx.x * 4 + w) * PPW + sub, pstep = (long)gridDim.x * 4 * PPW;
  if (act)
    for (long pos = p0; pos < npos; pos += pstep) {
      long pp = pos;
      const int f = (int)(pp % F1); pp /= F1;
      const int t = (int)(pp % T1);
      const int b = (int)(pp / T1);
      float d[8];
      ld8(s2d ? dy + s2d_off(b, t, f, (T1 + 1) / 2, (F1 + 1) / 2, C) + c0 : dy + pos * C + c0, d);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ti = 2 * t + kh - 2, fi = 2 * f + kw - 2;
          const float xv = (ti >= 0 && ti < T0 && fi >= 0 && fi < F0) ? Num<T>::ld(x + ((long)b * T0 + ti) * F0 + fi) : 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[kh * 3 + kw][k] += d[k] * xv;
        }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[9][k] += d[k];
    }
#pragma unroll
  for (int q = 0; q < 10; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = acc[q][k];
      if (PPW == 2) v += __shfl_xor(v, 32, 64);  // the two positions handled by one wave own the same channels
      if (sub == 0) red[w][q][c0 + k] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 10 * C; i += blockDim.x) {
    const int q = i / C, c = i % C;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) sum += red[r][q][c];
    if (q < 9) atomicAdd(dw + q * C + c, sum);
    else if (db) atomicAdd(db + c, sum);
  }
}

// Row-structured variants for C == 256: one (b, t) output row per block iteration, thread = (channel group, f sub-lane), so no
// thread ever divides a flat 64-bit index (three 64-bit div/mods per element made the flat kernels VALU-bound: 543 us to write
// 1 GB, 935 us to read it back).  The three input rows a block needs are staged in LDS once.
template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_rows_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ y, int B, int T0,
                                                             int F0, int T1, int F1, int s2d) {
  constexpr int C = 256;
  __shared__ float xs[3][260];
  const int c = (threadIdx.x & 31) * 8, fs = threadIdx.x >> 5;
  float wr[9][8], br[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) br[k] = bias ? bias[c + k] : 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int k = 0; k < 8; ++k) wr[tap][k] = w[tap * C + c + k];
  const int nrows = B * T1;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int b = row / T1, t = row - b * T1;
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * (F0 + 2); i += 256) {  // xs[kh][fi + 2] = x[b, 2t+kh-2, fi], zero outside
      const int kh = i / (F0 + 2), fi = i - kh * (F0 + 2) - 2, ti = 2 * t + kh - 2;
      xs[kh][fi + 2] = (ti >= 0 && ti < T0 && fi >= 0 && fi < F0) ? Num<T>::ld(x + ((long)b * T0 + ti) * F0 + fi) : 0.f;
    }
    __syncthreads();
    for (int f = fs; f < F1; f += 8) {
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = br[k];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float xv = xs[kh][2 * f + kw];
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] += wr[kh * 3 + kw][k] * xv;
        }
      st8(s2d ? y + s2d_off(b, t, f, (T1 + 1) / 2, (F1 + 1) / 2, C) + c : y + ((long)row * F1 + f) * C + c, acc);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv1_bwd_weight_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    float* __restrict__ dw, float* __restrict__ db, int B, int T0,
                                                                    int F0, int T1, int F1, int s2d) {
  constexpr int C = 256;
  __shared__ float xs[3][260];
  __shared__ float red[4][10][C + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (threadIdx.x & 31) * 8, fs = threadIdx.x >> 5;
  float acc[10][8];
#pragma unroll
  for (int q = 0; q < 10; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[q][k] = 0.f;
  const int nrows = B * T1;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int b = row / T1, t = row - b * T1;
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * (F0 + 2); i += 256) {
      const int kh = i / (F0 + 2), fi = i - kh * (F0 + 2) - 2, ti = 2 * t + kh - 2;
      xs[kh][fi + 2] = (ti >= 0 && ti < T0 && fi >= 0 && fi < F0) ? Num<T>::ld(x + ((long)b * T0 + ti) * F0 + fi) : 0.f;
    }
    __syncthreads();
    for (int f = fs; f < F1; f += 8) {
      float d[8];
      ld8(s2d ? dy + s2d_off(b, t, f, (T1 + 1) / 2, (F1 + 1) / 2, C) + c : dy + ((long)row * F1 + f) * C + c, d);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float xv = xs[kh][2 * f + kw];
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[kh * 3 + kw][k] += d[k] * xv;
        }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[9][k] += d[k];
    }
  }
  (void)lane;
#pragma unroll
  for (int q = 0; q < 10; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = acc[q][k];
      v += __shfl_xor(v, 32, 64);  // the two f sub-lanes of one wave own the same channels
      if ((threadIdx.x & 32) == 0) red[w][q][c + k] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 10 * C; i += blockDim.x) {
    const int q = i / C, cc = i % C;
    const float sum = red[0][q][cc] + red[1][q][cc] + red[2][q][cc] + red[3][q][cc];
    if (q < 9) atomicAdd(dw + q * C + cc, sum);
    else if (db) atomicAdd(db + cc, sum);
  }
}

// col[(b,t2,f2), (kh*3+kw)*C + c] = x[b, 2*t2+kh-2, 2*f2+kw-2, c] (0 outside)
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int T1, int F1,
                                                     int T2, int F2, int C) {
  const int c8n = C / 8;
  const long n8 = (long)B * T2 * F2 * 9 * c8n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8n) * 8;
    long r = i / c8n;
    const int tap = (int)(r % 9); r /= 9;
    const int f = (int)(r % F2); r /= F2;
    const int t = (int)(r % T2);
    const int b = (int)(r / T2);
    const int ti = 2 * t + tap / 3 - 2, fi = 2 * f + tap % 3 - 2;
    uint4 v = make_uint4(0, 0, 0, 0);
    float vf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ti >= 0 && ti < T1 && fi >= 0 && fi < F1) ld8(x + (((long)b * T1 + ti) * F1 + fi) * C + c, vf);
    (void)v;
    st8(col + i * 8, vf);
  }
}

// dx[b,t1,f1,c] = sum over taps with matching parity of dcol[(b,(t1+2-kh)/2,(f1+2-kw)/2), tap*C + c]
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int B, int T1,
                                                     int F1, int T2, int F2, int C) {
  const int c8n = C / 8;
  const long n8 = (long)B * T1 * F1 * c8n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8n) * 8;
    long r = i / c8n;
    const int f1 = (int)(r % F1); r /= F1;
    const int t1 = (int)(r % T1);
    const int b = (int)(r / T1);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int tt = t1 + 2 - kh;
      if (tt < 0 || (tt & 1)) continue;
      const int t2 = tt >> 1;
      if (t2 >= T2) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ff = f1 + 2 - kw;
        if (ff < 0 || (ff & 1)) continue;
        const int f2 = ff >> 1;
        if (f2 >= F2) continue;
        float v[8];
        ld8(dcol + ((((long)b * T2 + t2) * F2 + f2) * 9 + kh * 3 + kw) * C + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      }
    }
    st8(dx + i * 8, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv1 + BatchNorm + swish without ever storing conv1's output: conv1 (Cin = 1) costs 9 FMAs per element, its [B,T1,F1,C] output
// is ~1 GB at the bench shape, and every pass over it (write, statistics, apply, backward statistics, backward apply, weight
// gradient: ~10 GB per step) is pure HBM time.  Four kernels RECOMPUTE it from the 30 MB feature map instead:
//   conv1_stats      : sum / sum of squares per channel                                    (-> tfasr_bn_finalize as usual)
//   conv1_bn_apply   : y = swish(conv1(x) * scale + shift), written in the S layout
//   conv1_bn_bwd_stats : sum dz, sum dz * xhat with dz = dy * swish'(z)                   (dy read from the S layout)
//   conv1_bn_bwd_apply : d conv1 = scale (dz - mean dz - xhat mean(dz xhat)); its weight / bias gradients accumulated on the fly
// One (b, t) output row per block iteration, thread = (8-channel group, f sub-lane); taps + bias live in registers.
// MODE 0 stats, 1 apply, 2 backward stats, 3 backward apply + weight gradient.
// CPT channels per thread: 8 (MODE 0 / 1) or 4 (the backward passes: 80 accumulators + 72 taps at 8 channels per thread left one
// wave per SIMD - 257 registers - and the dependent activation chain had nothing to hide behind: 885 us for 0.3 ms of VALU work)
template <int N> __device__ __forceinline__ void ldn(const float* p, float (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; k += 4) { const float4 a = *reinterpret_cast<const float4*>(p + k); v[k] = a.x; v[k + 1] = a.y; v[k + 2] = a.z; v[k + 3] = a.w; }
}
template <int N> __device__ __forceinline__ void ldn(const bf16_t* p, float (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; k += 4) {
    const uint2 a = *reinterpret_cast<const uint2*>(p + k);
    v[k] = __uint_as_float(a.x << 16); v[k + 1] = __uint_as_float(a.x & 0xffff0000u); v[k + 2] = __uint_as_float(a.y << 16); v[k + 3] = __uint_as_float(a.y & 0xffff0000u);
  }
}
// Round 5: the arithmetic that is not a transcendental runs as PACKED f32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two channels per
// instruction - the kernels are VALU-bound: 9 conv FMAs + 9 weight-gradient FMAs + the BatchNorm affine per element against two
// quarter-rate transcendentals); same fused multiply-adds in the same order per channel, so results are bitwise unchanged.
typedef __attribute__((ext_vector_type(2))) float c1f2_t;
__device__ __forceinline__ c1f2_t c1fma(c1f2_t a, c1f2_t b, c1f2_t c) { return __builtin_elementwise_fma(a, b, c); }
template <typename T, int MODE, int CPT>
__global__ __launch_bounds__(256) void conv1_bn_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ fin, const float* __restrict__ bstats, float inv_count,
                                                       T* __restrict__ y, const T* __restrict__ dy, float* __restrict__ out0,
                                                       float* __restrict__ out1, int B, int T0, int F0, int T1, int F1, int C) {
  extern __shared__ float sm[];
  float* xs = sm;                       // [3][F0 + 2] input rows of the current output row (zero padded)
  float* red = sm + 3 * (F0 + 2);       // [NQ][C] block reduction
  constexpr int NQ = MODE == 3 ? 10 : (MODE == 4 ? 11 : 2);
  constexpr int CP2 = CPT / 2;          // channel pairs per thread
  static_assert(CPT % 2 == 0, "channel pairs");
  const int cgn = C / CPT, FS = 256 / cgn;
  const int cg = threadIdx.x % cgn, fs = threadIdx.x / cgn;
  const bool on = fs < FS;
  const int c = cg * CPT;
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  c1f2_t wr[9][CP2], br[CP2];
#pragma unroll
  for (int k = 0; k < CP2; ++k) br[k] = bias ? c1f2_t{bias[c + 2 * k], bias[c + 2 * k + 1]} : c1f2_t{0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int k = 0; k < CP2; ++k) wr[tap][k] = c1f2_t{w[tap * C + c + 2 * k], w[tap * C + c + 2 * k + 1]};
  c1f2_t mean[CP2], rstd[CP2], sc[CP2], sh[CP2], s0[CP2], s1[CP2];
  if (MODE >= 1) {
#pragma unroll
    for (int k = 0; k < CP2; ++k) {
      mean[k] = c1f2_t{fin[c + 2 * k], fin[c + 2 * k + 1]}; rstd[k] = c1f2_t{fin[C + c + 2 * k], fin[C + c + 2 * k + 1]};
      sc[k] = c1f2_t{fin[2 * C + c + 2 * k], fin[2 * C + c + 2 * k + 1]}; sh[k] = c1f2_t{fin[3 * C + c + 2 * k], fin[3 * C + c + 2 * k + 1]};
    }
  }
  if (MODE == 3) {
#pragma unroll
    for (int k = 0; k < CP2; ++k) {
      s0[k] = c1f2_t{bstats[c + 2 * k] * inv_count, bstats[c + 2 * k + 1] * inv_count};
      s1[k] = c1f2_t{bstats[C + c + 2 * k] * inv_count, bstats[C + c + 2 * k + 1] * inv_count};
    }
  }
  c1f2_t acc2[NQ][CP2];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int k = 0; k < CP2; ++k) acc2[q][k] = c1f2_t{0.f, 0.f};
  const int nrows = B * T1;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int b = row / T1, t = row - b * T1;
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * (F0 + 2); i += 256) {  // xs[kh][fi + 2] = x[b, 2t+kh-2, fi], zero outside
      const int kh = i / (F0 + 2), fi = i - kh * (F0 + 2) - 2, ti = 2 * t + kh - 2;
      xs[i] = (ti >= 0 && ti < T0 && fi >= 0 && fi < F0) ? Num<T>::ld(x + ((long)b * T0 + ti) * F0 + fi) : 0.f;
    }
    __syncthreads();
    if (!on) continue;
    // one output position: conv1 recomputed, then the mode's sums / stores; `dr` = this thread's channels of the incoming gradient
    auto body = [&](int f, const float (&dr)[CPT]) {
      c1f2_t a[CP2];
      float xv[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = xs[kh * (F0 + 2) + 2 * f + kw];
#pragma unroll
      for (int k = 0; k < CP2; ++k) a[k] = br[k];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const c1f2_t x2 = c1f2_t{xv[tap], xv[tap]};
#pragma unroll
        for (int k = 0; k < CP2; ++k) a[k] = c1fma(wr[tap][k], x2, a[k]);
      }
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < CP2; ++k) { acc2[0][k] += a[k]; acc2[1][k] = c1fma(a[k], a[k], acc2[1][k]); }
      } else if (MODE == 1) {
        float o[CPT];
#pragma unroll
        for (int k = 0; k < CP2; ++k) {
          const c1f2_t z = c1fma(a[k], sc[k], sh[k]);
          o[2 * k] = swishf_(z[0]); o[2 * k + 1] = swishf_(z[1]);
        }
        static_assert(MODE != 1 || CPT == 8, "apply pass: 8 channels per thread");
        if constexpr (CPT == 8) st8(y + s2d_off(b, t, f, T2, F2, C) + c, o);
      } else {
        c1f2_t d[CP2];
#pragma unroll
        for (int k = 0; k < CP2; ++k) {
          const c1f2_t z = c1fma(a[k], sc[k], sh[k]);
          const c1f2_t dz = c1f2_t{dr[2 * k] * dswishf_(z[0]), dr[2 * k + 1] * dswishf_(z[1])};
          const c1f2_t xh = (a[k] - mean[k]) * rstd[k];
          if (MODE == 2 || MODE == 4) { acc2[0][k] += dz; acc2[1][k] = c1fma(dz, xh, acc2[1][k]); }
          d[k] = (MODE == 3) ? sc[k] * (dz - s0[k] - xh * s1[k]) : dz;  // MODE 3: gradient w.r.t. conv1's output
        }
        if (MODE == 4) {  // P[tap][c] += patch[tap] * dz_c: with the patch Gram matrix this is all conv1's weight gradient needs (below)
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const c1f2_t x2 = c1f2_t{xv[tap], xv[tap]};
#pragma unroll
            for (int k = 0; k < CP2; ++k) acc2[2 + tap][k] = c1fma(d[k], x2, acc2[2 + tap][k]);
          }
        }
        if (MODE == 3) {
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const c1f2_t x2 = c1f2_t{xv[tap], xv[tap]};
#pragma unroll
            for (int k = 0; k < CP2; ++k) acc2[tap][k] = c1fma(d[k], x2, acc2[tap][k]);
          }
#pragma unroll
          for (int k = 0; k < CP2; ++k) acc2[9][k] += d[k];
        }
      }
    };
    // (measured and dropped in round 5: requesting the incoming gradient of all ten positions of the row up front - 218 instead of 142
    // registers = two waves per SIMD instead of three: 596 vs 520 us; the kernel is bound by its ~60 vector instructions per element,
    // not by the load chain - and packed-f32 FMAs, which are kept, did not move it either: 520 vs 517 us)
    for (int f = fs; f < F1; f += FS) {
      float dr[CPT];
      if (MODE >= 2) ldn<CPT>(dy + s2d_off(b, t, f, T2, F2, C) + c, dr);
      body(f, dr);
    }
  }
  if (MODE == 1) return;
  // block reduction over the f sub-lanes that share a channel group (LDS float atomics: once per block), then one global atomic
  // per (quantity, channel) per block
  __syncthreads();
  for (int i = threadIdx.x; i < NQ * C; i += 256) red[i] = 0.f;
  __syncthreads();
  if (on) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int k = 0; k < CP2; ++k) { atomicAdd(&red[q * C + c + 2 * k], acc2[q][k][0]); atomicAdd(&red[q * C + c + 2 * k + 1], acc2[q][k][1]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NQ * C; i += 256) {
    const int q = i / C, cc = i % C;
    if (MODE == 3) {
      if (q < 9) atomicAdd(out0 + q * C + cc, red[i]);
      else if (out1) atomicAdd(out1 + cc, red[i]);
    } else if (MODE == 4) {
      if (q < 2) atomicAdd(out0 + q * C + cc, red[i]);        // bstats (the caller all-reduces them across ranks)
      if (q >= 2) atomicAdd(out1 + (q - 2) * C + cc, red[i]);  // P[9][C]
      if (q == 0) atomicAdd(out1 + 9 * C + cc, red[i]);        // this rank's own sum of dz (the bias gradient's first term)
    } else {
      atomicAdd(out0 + q * C + cc, red[i]);  // stats[0:C] = first quantity, stats[C:2C] = second
    }
  }
}

// ---- conv1 + BatchNorm0 through the GRAM MATRIX of the 3x3 patches ---------------------------------------------------------------
// conv1 has ONE input channel: z_c(pos) = sum_k W[k,c] p_k(pos) + b_c with the 9-element patch p(pos).  Every sum over positions that
// the BatchNorm around it needs is therefore a function of  G = sum_pos p p^T (9x9),  s = sum_pos p (9),  N = #positions  - 91 numbers
// computed once per step from the 30 MB feature map - and of sums against the incoming gradient:
//   forward statistics   sum z_c   = W_c.s + N b_c                  sum z_c^2 = W_c^T G W_c + 2 b_c W_c.s + N b_c^2
//   backward             d_c(pos)  = sc_c (dz_c - s0_c - xhat_c s1_c)                    (s0, s1 = global means of dz, dz xhat)
//                        dW[k,c]   = sum_pos p_k d_c = sc_c (P[k,c] - s0_c s_k - s1_c X[k,c]),   P[k,c] = sum_pos p_k dz_c
//                        X[k,c]    = sum_pos p_k xhat_c = rstd_c ((G W_c)_k + (b_c - mean_c) s_k)
//                        db[c]     = sc_c (sum dz_c - s0_c N - s1_c rstd_c (W_c.s + N (b_c - mean_c)))      (all sums of THIS rank)
// so the backward is ONE pass over the 1 GB gradient (sum dz, sum dz xhat, P) instead of two (statistics, then apply + weight
// gradient), and the forward statistics pass over 256 channels x 9 taps per position is replaced by the 54 sums of the patch moments.
// Accumulated in double (the combinations cancel: log-mel features have mean^2 >> variance).
constexpr int GRAM_N = 81 + 9 + 1;
// the sums are spread over GRAM_COPIES copies (stride GRAM_STRIDE doubles) by block index: same-address atomics serialise (~25 ns each),
// 1024 blocks on one copy would queue for 25 us; the consumers add the copies up
constexpr int GRAM_COPIES = 8, GRAM_STRIDE = 96;
template <typename T>
__global__ __launch_bounds__(256) void conv1_gram_kernel(const T* __restrict__ x, double* __restrict__ out, int B, int T0, int F0, int T1, int F1) {
  extern __shared__ float sm[];  // [RB][3][F0 + 2]
  constexpr int RB = 6;
  const int W2 = F0 + 2;
  double g[45], sv[9];
#pragma unroll
  for (int i = 0; i < 45; ++i) g[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) sv[i] = 0.0;
  const int nrows = B * T1;
  for (int row0 = blockIdx.x * RB; row0 < nrows; row0 += gridDim.x * RB) {
    __syncthreads();
    for (int i = threadIdx.x; i < RB * 3 * W2; i += 256) {
      const int rr = i / (3 * W2), j = i - rr * 3 * W2, kh = j / W2, fi = j - kh * W2 - 2;
      const int row = row0 + rr;
      float v = 0.f;
      if (row < nrows) {
        const int b = row / T1, t = row - b * T1, ti = 2 * t + kh - 2;
        if (ti >= 0 && ti < T0 && fi >= 0 && fi < F0) v = Num<T>::ld(x + ((long)b * T0 + ti) * F0 + fi);
      }
      sm[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RB * F1; i += 256) {
      const int rr = i / F1, f = i - rr * F1;
      if (row0 + rr >= nrows) continue;
      double pv[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) pv[kh * 3 + kw] = (double)sm[(rr * 3 + kh) * W2 + 2 * f + kw];
      int q = 0;
#pragma unroll
      for (int a = 0; a < 9; ++a) {
        sv[a] += pv[a];
#pragma unroll
        for (int bb = a; bb < 9; ++bb) g[q++] += pv[a] * pv[bb];
      }
    }
  }
  // block reduction: wave shuffles, then one LDS slot per wave, then 54 global atomics per block
  __shared__ double wred[4][54];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 54; ++i) {
    double v = i < 45 ? g[i] : sv[i - 45];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) wred[w][i] = v;
  }
  __syncthreads();
  out += (blockIdx.x % GRAM_COPIES) * GRAM_STRIDE;
  if (threadIdx.x < 54) {
    const double v = wred[0][threadIdx.x] + wred[1][threadIdx.x] + wred[2][threadIdx.x] + wred[3][threadIdx.x];
    if (threadIdx.x < 45) {  // upper triangle entry -> both (a, b) and (b, a)
      int a = 0, rem = threadIdx.x;
      while (rem >= 9 - a) { rem -= 9 - a; ++a; }
      const int bb = a + rem;
      atomicAdd(out + a * 9 + bb, v);
      if (bb != a) atomicAdd(out + bb * 9 + a, v);
    } else {
      atomicAdd(out + 81 + (threadIdx.x - 45), v);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 64) out[90] = (double)nrows * (double)F1;
}
// statistics [2C] = (sum z, sum z^2) from the patch moments
__global__ __launch_bounds__(256) void conv1_stats_from_gram_kernel(const double* __restrict__ gram_copies, const float* __restrict__ w, const float* __restrict__ bias,
                                                                    float* __restrict__ stats, int C) {
  __shared__ double gram[GRAM_N];
  for (int i = threadIdx.x; i < GRAM_N; i += blockDim.x) {
    double v = 0.0;
    for (int q = 0; q < GRAM_COPIES; ++q) v += gram_copies[q * GRAM_STRIDE + i];
    gram[i] = v;
  }
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double wc[9], ws = 0.0, wgw = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { wc[k] = (double)w[k * C + c]; ws += wc[k] * gram[81 + k]; }
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    double t = 0.0;
#pragma unroll
    for (int bb = 0; bb < 9; ++bb) t += gram[a * 9 + bb] * wc[bb];
    wgw += wc[a] * t;
  }
  const double N = gram[90], b = bias ? (double)bias[c] : 0.0;
  stats[c] = (float)(ws + N * b);
  stats[C + c] = (float)(wgw + 2.0 * b * ws + N * b * b);
}
// conv1 weight / bias gradients from P (one-pass backward), the patch moments and the (all-reduced) BatchNorm statistics
__global__ __launch_bounds__(256) void conv1_bn_bwd_finalize_kernel(const double* __restrict__ gram_copies, const float* __restrict__ w, const float* __restrict__ bias,
                                                                    const float* __restrict__ fin, const float* __restrict__ bstats, float inv_count,
                                                                    const float* __restrict__ pbuf, float* __restrict__ dw, float* __restrict__ db, int C) {
  __shared__ double gram[GRAM_N];
  for (int i = threadIdx.x; i < GRAM_N; i += blockDim.x) {
    double v = 0.0;
    for (int q = 0; q < GRAM_COPIES; ++q) v += gram_copies[q * GRAM_STRIDE + i];
    gram[i] = v;
  }
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = fin[c], rstd = fin[C + c], sc = fin[2 * C + c];
  const double s0 = (double)bstats[c] * inv_count, s1 = (double)bstats[C + c] * inv_count;
  const double N = gram[90], b = bias ? (double)bias[c] : 0.0;
  double wc[9], ws = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { wc[k] = (double)w[k * C + c]; ws += wc[k] * gram[81 + k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    double gw = 0.0;
#pragma unroll
    for (int bb = 0; bb < 9; ++bb) gw += gram[k * 9 + bb] * wc[bb];
    const double sk = gram[81 + k];
    const double X = rstd * (gw + (b - mean) * sk);
    dw[k * C + c] += (float)(sc * ((double)pbuf[k * C + c] - s0 * sk - s1 * X));
  }
  if (db) db[c] += (float)(sc * ((double)pbuf[9 * C + c] - s0 * N - s1 * rstd * (ws + N * (b - mean))));
}

// Zero the halo rows (tt == 0 or ff == 0) of a [B, T2+1, F2+1, W] tensor (W = 4C for the S layout, C for the conv2 output side).
template <typename T>
__global__ __launch_bounds__(256) void halo_zero_kernel(T* __restrict__ x, int B, int T2, int F2, int W) {
  const int w8 = W / 8, per_b = (F2 + 1) + T2;
  const long n = (long)B * per_b * w8;
  const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % w8);
    long r = i / w8;
    const int k = (int)(r % per_b), b = (int)(r / per_b);
    const int tt = k <= F2 ? 0 : k - F2, ff = k <= F2 ? k : 0;
    st8(x + (((long)b * (T2 + 1) + tt) * (F2 + 1) + ff) * W + (long)c8 * 8, z);
  }
}
// Zero the slots of an S-layout tensor past an odd T1 (row tt = T2, parity pt = 1) / odd F1 (column ff = F2, parity pf = 1): only
// those slots are visited (the first version scanned every slot of the tensor: 135 us for a 1 GB activation).
template <typename T>
__global__ __launch_bounds__(256) void s2d_edge_zero_kernel(T* __restrict__ x, int B, int T1, int F1, int T2, int F2, int C) {
  const int c8n = C / 8;
  const int nt = (T1 & 1) ? F2 * 2 : 0;          // (ff in 1..F2) x (pf in 0..1) at tt = T2, pt = 1
  const int nf = (F1 & 1) ? T2 * 2 : 0;          // (tt in 1..T2) x (pt in 0..1) at ff = F2, pf = 1
  const long n = (long)B * (nt + nf) * c8n;
  const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % c8n);
    long r = i / c8n;
    const int k = (int)(r % (nt + nf)), b = (int)(r / (nt + nf));
    int tt, ff, blk;
    if (k < nt) { tt = T2; ff = 1 + (k >> 1); blk = 2 + (k & 1); }
    else { const int q = k - nt; tt = 1 + (q >> 1); ff = F2; blk = (q & 1) * 2 + 1; }
    st8(x + ((((long)b * (T2 + 1) + tt) * (F2 + 1) + ff) * 4 + blk) * C + (long)c8 * 8, z);
  }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
  do { if ((dtype) == TFASR_F32) { CALL_F32; } else if ((dtype) == TFASR_BF16) { CALL_BF16; } else return TFASR_STATUS_INVALID_VALUE; } while (0)

static int conv1_fwd_impl(const void* x, const float* w, const float* bias, void* y, int B, int T0, int F0, int C,
                          int dtype, void* stream_, int s2d) {
  if (!x || !w || !y || B <= 0 || T0 <= 0 || F0 <= 0 || C <= 0 || C % 8) return TFASR_STATUS_INVALID_VALUE;
  const int T1 = (T0 + 1) / 2, F1 = (F0 + 1) / 2;
  hipStream_t s = (hipStream_t)stream_;
  if (C == 256 && F0 + 2 <= 260) {
    const int g2 = std::min(B * T1, 8192);
    DISPATCH_T(dtype,
               TFASR_KLAUNCH(conv1_fwd_rows_kernel<float>, dim3(g2), dim3(256), 0, s, (const float*)x, w, bias, (float*)y, B, T0, F0, T1, F1, s2d),
               TFASR_KLAUNCH(conv1_fwd_rows_kernel<bf16_t>, dim3(g2), dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, B, T0, F0, T1, F1, s2d));
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  int grid = flat_grid((long)B * T1 * F1 * C / 8);
  if ((256 % (C / 8)) != 0) {  // keep (grid*256) a multiple of C/8 so each thread owns one channel group
    const int c8n = C / 8;
    grid = ((grid + c8n - 1) / c8n) * c8n;
  }
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(conv1_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, w, bias, (float*)y, B, T0, F0, T1, F1, C, s2d),
             TFASR_KLAUNCH(conv1_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, B, T0, F0, T1, F1, C, s2d));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_conv1_fwd(const void* x, const float* w, const float* bias, void* y, int B, int T0, int F0, int C,
                               int dtype, void* stream_) {
  return conv1_fwd_impl(x, w, bias, y, B, T0, F0, C, dtype, stream_, 0);
}
extern "C" int tfasr_conv1_fwd_s2d(const void* x, const float* w, const float* bias, void* y, int B, int T0, int F0, int C,
                                   int dtype, void* stream_) {
  return conv1_fwd_impl(x, w, bias, y, B, T0, F0, C, dtype, stream_, 1);
}

static int conv1_bwd_weight_impl(const void* x, const void* dy, float* dw, float* db, int B, int T0, int F0, int C,
                                 int dtype, void* stream_, int s2d) {
  if (!x || !dy || !dw || B <= 0 || T0 <= 0 || F0 <= 0 || C <= 0 || C > 256 || (C % 8)) return TFASR_STATUS_INVALID_VALUE;
  const int T1 = (T0 + 1) / 2, F1 = (F0 + 1) / 2;
  hipStream_t s = (hipStream_t)stream_;
  if (C == 256 && F0 + 2 <= 260) {
    const int g2 = std::min(B * T1, 1024);
    DISPATCH_T(dtype,
               TFASR_KLAUNCH(conv1_bwd_weight_rows_kernel<float>, dim3(g2), dim3(256), 0, s, (const float*)x, (const float*)dy, dw, db, B, T0, F0, T1, F1, s2d),
               TFASR_KLAUNCH(conv1_bwd_weight_rows_kernel<bf16_t>, dim3(g2), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, dw, db, B, T0, F0, T1, F1, s2d));
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  const long npos = (long)B * T1 * F1;
  const int grid = (int)std::max<long>(1, std::min<long>(npos / 64 + 1, 1024));
  DISPATCH_T(dtype,
             TFASR_KLAUNCH((conv1_bwd_weight_kernel<float, 32>), dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)dy, dw, db, B, T0, F0, T1, F1, C, s2d),
             TFASR_KLAUNCH((conv1_bwd_weight_kernel<bf16_t, 32>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, dw, db, B, T0, F0, T1, F1, C, s2d));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_conv1_bwd_weight(const void* x, const void* dy, float* dw, float* db, int B, int T0, int F0, int C,
                                      int dtype, void* stream_) {
  return conv1_bwd_weight_impl(x, dy, dw, db, B, T0, F0, C, dtype, stream_, 0);
}
extern "C" int tfasr_conv1_bwd_weight_s2d(const void* x, const void* dy, float* dw, float* db, int B, int T0, int F0, int C,
                                          int dtype, void* stream_) {
  return conv1_bwd_weight_impl(x, dy, dw, db, B, T0, F0, C, dtype, stream_, 1);
}

template <int MODE>
static int conv1_bn_launch(const void* x, const float* w, const float* bias, const float* fin, const float* bstats, float count, void* y,
                           const void* dy, float* out0, float* out1, int B, int T0, int F0, int C, int dtype, void* stream_) {
  if (!x || !w || B <= 0 || T0 <= 0 || F0 <= 0 || C <= 0 || C > 256 || (C % 8)) return TFASR_STATUS_INVALID_VALUE;
  constexpr int CPT = MODE >= 2 ? 4 : 8;
  constexpr int NQL = MODE == 3 ? 10 : (MODE == 4 ? 11 : 2);
  const int T1 = (T0 + 1) / 2, F1 = (F0 + 1) / 2;
  hipStream_t s = (hipStream_t)stream_;
  const size_t smem_q = (size_t)(3 * (F0 + 2) + NQL * C) * sizeof(float);
  // backward passes: every block walks ~B*T1/grid rows, so the grid is exactly ONE round of resident blocks (occupancy x CUs, asked once per
  // instantiation and LDS size): 1024 blocks on 768 slots were a full round plus a third of one.  TFASR_CONV1_GRID overrides (A/B).
  static const int env_grid = 0;
  static int res_blocks[2] = {0, 0};
  static size_t res_smem[2] = {0, 0};
  const int di = dtype == TFASR_F32 ? 0 : 1;
  if (MODE != 1 && !env_grid && (res_blocks[di] == 0 || res_smem[di] != smem_q)) {
    int dev = 0, cus = 0, per = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess)
      e = di == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, conv1_bn_kernel<float, MODE, CPT>, 256, smem_q)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, conv1_bn_kernel<bf16_t, MODE, CPT>, 256, smem_q);
    res_blocks[di] = (e == hipSuccess && per > 0 && cus > 0) ? per * cus : 768;
    res_smem[di] = smem_q;
  }
  const int grid = std::min(B * T1, MODE == 1 ? 8192 : std::max(64, env_grid ? env_grid : res_blocks[di]));
  const size_t smem = (size_t)(3 * (F0 + 2) + NQL * C) * sizeof(float);
  const float inv = count > 0.f ? 1.f / count : 0.f;
  DISPATCH_T(dtype,
             TFASR_KLAUNCH((conv1_bn_kernel<float, MODE, CPT>), dim3(grid), dim3(256), smem, s, (const float*)x, w, bias, fin, bstats, inv, (float*)y,
                                (const float*)dy, out0, out1, B, T0, F0, T1, F1, C),
             TFASR_KLAUNCH((conv1_bn_kernel<bf16_t, MODE, CPT>), dim3(grid), dim3(256), smem, s, (const bf16_t*)x, w, bias, fin, bstats, inv, (bf16_t*)y,
                                (const bf16_t*)dy, out0, out1, B, T0, F0, T1, F1, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_conv1_stats(const void* x, const float* w, const float* bias, float* stats, int B, int T0, int F0, int C, int dtype,
                                 void* stream_) {
  if (!stats) return TFASR_STATUS_INVALID_VALUE;
  return conv1_bn_launch<0>(x, w, bias, nullptr, nullptr, 0.f, nullptr, nullptr, stats, nullptr, B, T0, F0, C, dtype, stream_);
}
extern "C" int tfasr_conv1_bn_apply_s2d(const void* x, const float* w, const float* bias, const float* fin, void* y, int B, int T0, int F0,
                                        int C, int dtype, void* stream_) {
  if (!fin || !y) return TFASR_STATUS_INVALID_VALUE;
  return conv1_bn_launch<1>(x, w, bias, fin, nullptr, 0.f, y, nullptr, nullptr, nullptr, B, T0, F0, C, dtype, stream_);
}
extern "C" int tfasr_conv1_bn_bwd_stats_s2d(const void* x, const float* w, const float* bias, const float* fin, const void* dy, float* bstats,
                                            int B, int T0, int F0, int C, int dtype, void* stream_) {
  if (!fin || !dy || !bstats) return TFASR_STATUS_INVALID_VALUE;
  return conv1_bn_launch<2>(x, w, bias, fin, nullptr, 0.f, nullptr, dy, bstats, nullptr, B, T0, F0, C, dtype, stream_);
}
extern "C" int tfasr_conv1_bn_bwd_apply_s2d(const void* x, const float* w, const float* bias, const float* fin, const float* bstats,
                                            float count, const void* dy, float* dw, float* db, int B, int T0, int F0, int C, int dtype,
                                            void* stream_) {
  if (!fin || !bstats || !dy || !dw || count <= 0.f) return TFASR_STATUS_INVALID_VALUE;
  return conv1_bn_launch<3>(x, w, bias, fin, bstats, count, nullptr, dy, dw, db, B, T0, F0, C, dtype, stream_);
}

extern "C" int tfasr_conv1_gram(const void* x, double* gram, int B, int T0, int F0, int dtype, void* stream_) {
  if (!x || !gram || B <= 0 || T0 <= 0 || F0 <= 0) return TFASR_STATUS_INVALID_VALUE;
  const int T1 = (T0 + 1) / 2, F1 = (F0 + 1) / 2;
  hipStream_t s = (hipStream_t)stream_;
  if (hipMemsetAsync(gram, 0, GRAM_COPIES * GRAM_STRIDE * sizeof(double), s) != hipSuccess) return TFASR_STATUS_EXECUTION_FAILED;
  const int grid = std::max(1, std::min((B * T1 + 5) / 6, 1024));
  const size_t smem = (size_t)6 * 3 * (F0 + 2) * sizeof(float);
  DISPATCH_T(dtype, TFASR_KLAUNCH(conv1_gram_kernel<float>, dim3(grid), dim3(256), smem, s, (const float*)x, gram, B, T0, F0, T1, F1),
             TFASR_KLAUNCH(conv1_gram_kernel<bf16_t>, dim3(grid), dim3(256), smem, s, (const bf16_t*)x, gram, B, T0, F0, T1, F1));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_conv1_stats_from_gram(const double* gram, const float* w, const float* bias, float* stats, int C, void* stream_) {
  if (!gram || !w || !stats || C <= 0) return TFASR_STATUS_INVALID_VALUE;
  TFASR_KLAUNCH(conv1_stats_from_gram_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream_, gram, w, bias, stats, C);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
extern "C" int tfasr_conv1_bn_bwd_onepass_s2d(const void* x, const float* w, const float* bias, const float* fin, const void* dy, float* bstats,
                                              float* pbuf, int B, int T0, int F0, int C, int dtype, void* stream_) {
  if (!fin || !dy || !bstats || !pbuf) return TFASR_STATUS_INVALID_VALUE;
  return conv1_bn_launch<4>(x, w, bias, fin, nullptr, 0.f, nullptr, dy, bstats, pbuf, B, T0, F0, C, dtype, stream_);
}
extern "C" int tfasr_conv1_bn_bwd_finalize(const double* gram, const float* w, const float* bias, const float* fin, const float* bstats,
                                           float count, const float* pbuf, float* dw, float* db, int C, void* stream_) {
  if (!gram || !w || !fin || !bstats || !pbuf || !dw || C <= 0 || count <= 0.f) return TFASR_STATUS_INVALID_VALUE;
  TFASR_KLAUNCH(conv1_bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream_, gram, w, bias, fin, bstats,
                     1.f / count, pbuf, dw, db, C);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_halo_zero(void* x, int B, int T2, int F2, int W, int dtype, void* stream_) {
  if (!x || B <= 0 || T2 <= 0 || F2 <= 0 || W <= 0 || W % 8) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid((long)B * ((F2 + 1) + T2) * (W / 8));
  DISPATCH_T(dtype, TFASR_KLAUNCH(halo_zero_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)x, B, T2, F2, W),
             TFASR_KLAUNCH(halo_zero_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t*)x, B, T2, F2, W));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_s2d_edge_zero(void* x, int B, int T1, int F1, int C, int dtype, void* stream_) {
  if (!x || B <= 0 || T1 <= 0 || F1 <= 0 || C <= 0 || C % 8) return TFASR_STATUS_INVALID_VALUE;
  if (!(T1 & 1) && !(F1 & 1)) return TFASR_STATUS_SUCCESS;  // nothing past the edge
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid((long)B * (((T1 & 1) ? F2 * 2 : 0) + ((F1 & 1) ? T2 * 2 : 0)) * (C / 8));
  DISPATCH_T(dtype, TFASR_KLAUNCH(s2d_edge_zero_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)x, B, T1, F1, T2, F2, C),
             TFASR_KLAUNCH(s2d_edge_zero_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t*)x, B, T1, F1, T2, F2, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_im2col_3x3s2(const void* x, void* col, int B, int T1, int F1, int C, int dtype, void* stream_) {
  if (!x || !col || B <= 0 || T1 <= 0 || F1 <= 0 || C <= 0 || C % 8) return TFASR_STATUS_INVALID_VALUE;
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid((long)B * T2 * F2 * 9 * C / 8);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(im2col_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)col, B, T1, F1, T2, F2, C),
             TFASR_KLAUNCH(im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)col, B, T1, F1, T2, F2, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_col2im_3x3s2(const void* dcol, void* dx, int B, int T1, int F1, int C, int dtype, void* stream_) {
  if (!dcol || !dx || B <= 0 || T1 <= 0 || F1 <= 0 || C <= 0 || C % 8) return TFASR_STATUS_INVALID_VALUE;
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = flat_grid((long)B * T1 * F1 * C / 8);
  DISPATCH_T(dtype,
             TFASR_KLAUNCH(col2im_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dcol, (float*)dx, B, T1, F1, T2, F2, C),
             TFASR_KLAUNCH(col2im_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dcol, (bf16_t*)dx, B, T1, F1, T2, F2, C));
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
