// LayerNorm / BatchNorm (training-mode batch statistics) forward + backward for gfx950.
//
// Reference sites: keras.layers.LayerNormalization (epsilon 1e-3) at conformer.py:59-64,101-109,
// base_transducer.py:88-93; keras.layers.BatchNormalization(synchronized=True, momentum .99,
// epsilon 1e-3) at conformer.py:327-333 and subsampling.py:197-203; swish fused after BN
// (conformer.py:342, subsampling.py:214).  All are HBM-bound row/column reductions: one wave per
// row (LayerNorm) or lanes-own-columns accumulation with one f32 atomic per column per wave
// (BatchNorm statistics, gamma/beta gradients).
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int MAXC_PER_LANE = 16;  // C <= 1024

// ------------------------------------------------------------------------------------ LayerNorm
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long w0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long r = w0; r < rows; r += nw) {
    const T* xr = x + r * C;
    float v[MAXC_PER_LANE];
    float s = 0.f;
    int n = 0;
    for (int c = lane; c < C; c += 64, ++n) { v[n] = Num<T>::ld(xr + c); s += v[n]; }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
    for (int i = 0; i < n; ++i) { const float d = v[i] - mean; q += d * d; }
    const float var = wave_sum(q) / C;
    const float rstd = rsqrtf(var + eps);
    T* yr = y + r * C;
    n = 0;
    for (int c = lane; c < C; c += 64, ++n) Num<T>::st(yr + c, (v[n] - mean) * rstd * gamma[c] + beta[c]);
    if (lane == 0) { if (mean_out) mean_out[r] = mean; if (rstd_out) rstd_out[r] = rstd; }
  }
}

// dx = add + rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)); dgamma += dy*xhat; dbeta += dy
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* add, T* dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, long rows,
                                                     int C) {
  const int lane = threadIdx.x & 63;
  const long w0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  float ag[MAXC_PER_LANE], ab[MAXC_PER_LANE], g[MAXC_PER_LANE];
  int ncol = 0;
  for (int c = lane; c < C; c += 64, ++ncol) { ag[ncol] = 0.f; ab[ncol] = 0.f; g[ncol] = gamma[c]; }
  for (long r = w0; r < rows; r += nw) {
    const T* xr = x + r * C;
    const T* dyr = dy + r * C;
    const float m = mean[r], rs = rstd[r];
    float xh[MAXC_PER_LANE], dg[MAXC_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
    int n = 0;
    for (int c = lane; c < C; c += 64, ++n) {
      const float d = Num<T>::ld(dyr + c);
      xh[n] = (Num<T>::ld(xr + c) - m) * rs;
      dg[n] = d * g[n];
      s1 += dg[n];
      s2 += dg[n] * xh[n];
      ag[n] += d * xh[n];
      ab[n] += d;
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
    n = 0;
    for (int c = lane; c < C; c += 64, ++n) {
      float v = rs * (dg[n] - s1 - xh[n] * s2);
      if (add) v += Num<T>::ld(add + r * C + c);
      Num<T>::st(dx + r * C + c, v);
    }
  }
  // block-level reduction of the per-wave column sums, then ONE atomic per column per block
  __shared__ float red[2][4][64 * MAXC_PER_LANE];
  const int w = threadIdx.x >> 6;
  int n = 0;
  for (int c = lane; c < C; c += 64, ++n) { red[0][w][c] = ag[n]; red[1][w][c] = ab[n]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    if (dgamma) atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    if (dbeta) atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// ------------------------------------------------------------------------------------ BatchNorm
// stats[0:C] += sum_rows f(x), stats[C:2C] += sum_rows g(x)
//   MODE 0: f = x,  g = x*x                                  (forward moments)
//   MODE 1: f = dz, g = dz*xhat  with z = x*scale+shift, dz = dy*act'(z), xhat = (x-mean)*rstd (backward sums)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const float* __restrict__ fin /*[4C] mean,rstd,scale,shift*/,
                                                       float* __restrict__ stats, long rows, int C, int act) {
  const int lane = threadIdx.x & 63;
  const long w0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  float a0[MAXC_PER_LANE], a1[MAXC_PER_LANE], mean[MAXC_PER_LANE], rstd[MAXC_PER_LANE], sc[MAXC_PER_LANE],
      sh[MAXC_PER_LANE];
  int ncol = 0;
  for (int c = lane; c < C; c += 64, ++ncol) {
    a0[ncol] = 0.f; a1[ncol] = 0.f;
    if (MODE == 1) { mean[ncol] = fin[c]; rstd[ncol] = fin[C + c]; sc[ncol] = fin[2 * C + c]; sh[ncol] = fin[3 * C + c]; }
  }
  for (long r = w0; r < rows; r += nw) {
    int n = 0;
    for (int c = lane; c < C; c += 64, ++n) {
      const float xv = Num<T>::ld(x + r * C + c);
      if (MODE == 0) { a0[n] += xv; a1[n] += xv * xv; }
      else {
        float d = Num<T>::ld(dy + r * C + c);
        if (act == TFASR_ACT_SWISH) d *= dswishf_(xv * sc[n] + sh[n]);
        a0[n] += d;
        a1[n] += d * (xv - mean[n]) * rstd[n];
      }
    }
  }
  __shared__ float red[2][4][64 * MAXC_PER_LANE];
  const int w = threadIdx.x >> 6;
  int n = 0;
  for (int c = lane; c < C; c += 64, ++n) { red[0][w][c] = a0[n]; red[1][w][c] = a1[n]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomicAdd(stats + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(stats + C + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// fin[0:C]=mean, [C:2C]=rstd, [2C:3C]=scale=gamma*rstd, [3C:4C]=shift=beta-mean*scale; moving stats updated
// as keras: moving = moving*momentum + batch*(1-momentum)   (biased batch variance)
__global__ void bn_finalize_kernel(const float* __restrict__ stats, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ fin, float* moving_mean,
                                   float* moving_var, float momentum, float eps, int C, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    mean = stats[c] / count;
    var = fmaxf(stats[C + c] / count - mean * mean, 0.f);
    if (moving_mean) moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
    if (moving_var) moving_var[c] = moving_var[c] * momentum + var * (1.f - momentum);
  } else {
    mean = moving_mean[c];
    var = moving_var[c];
  }
  const float rstd = rsqrtf(var + eps);
  const float scale = gamma[c] * rstd;
  fin[c] = mean;
  fin[C + c] = rstd;
  fin[2 * C + c] = scale;
  fin[3 * C + c] = beta[c] - mean * scale;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const T* __restrict__ x, const float* __restrict__ fin,
                                                           T* __restrict__ y, long n8, int C, int act) {
  // 8 elements per thread; C % 8 == 0
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 8;
    const int c = (int)(e % C);
    float v[8], sc[8], sh[8];
    ld8(x + e, v);
    ld8(fin + 2 * C + c, sc);
    ld8(fin + 3 * C + c, sh);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float z = v[k] * sc[k] + sh[k];
      if (act == TFASR_ACT_SWISH) z = swishf_(z);
      v[k] = z;
    }
    st8(y + e, v);
  }
}

// dx = scale*(dz - S0/n - xhat*S1/n)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ fin,
                                                           const float* __restrict__ bstats, float count, T* dx, long n8,
                                                           int C, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 8;
    const int c = (int)(e % C);
    float xv[8], d[8], mean[8], rstd[8], sc[8], sh[8], s0[8], s1[8];
    ld8(x + e, xv);
    ld8(dy + e, d);
    ld8(fin + c, mean); ld8(fin + C + c, rstd); ld8(fin + 2 * C + c, sc); ld8(fin + 3 * C + c, sh);
    ld8(bstats + c, s0); ld8(bstats + C + c, s1);
    const float inv = 1.f / count;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float dz = d[k];
      if (act == TFASR_ACT_SWISH) dz *= dswishf_(xv[k] * sc[k] + sh[k]);
      const float xh = (xv[k] - mean[k]) * rstd[k];
      d[k] = sc[k] * (dz - s0[k] * inv - xh * s1[k] * inv);
    }
    st8(dx + e, d);
  }
}


// Row-looping variants of the two apply kernels (C % 8 == 0, 256 % (C/8) == 0): a thread owns 8 channels and keeps their coefficients
// in registers while it walks the rows.  The flat kernels above load 2 (forward) / 6 (backward) 32-byte coefficient vectors per
// 16-byte activation vector - 12 load instructions beside 2 for the backward: 17.7 us for a [12480, 256] tensor whose three
// activation streams take ~8.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fwd_rows_kernel(const T* __restrict__ x, const float* __restrict__ fin, T* __restrict__ y,
                                                                long rows, int C, int act) {
  const int lpr = C >> 3, li = threadIdx.x % lpr, sub = threadIdx.x / lpr, rpb = 256 / lpr;
  const int c = li * 8;
  float sc[8], sh[8];
  ld8(fin + 2 * C + c, sc);
  ld8(fin + 3 * C + c, sh);
  const long step = (long)gridDim.x * rpb;
  for (long r0 = (long)blockIdx.x * rpb + sub; r0 < rows; r0 += 2 * step) {
    float v[2][8];
    const bool two = r0 + step < rows;
    ld8(x + r0 * C + c, v[0]);
    if (two) ld8(x + (r0 + step) * C + c, v[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float z = v[u][k] * sc[k] + sh[k];
        if (act == TFASR_ACT_SWISH) z = swishf_(z);
        v[u][k] = z;
      }
      st8(y + (r0 + u * step) * C + c, v[u]);
    }
  }
}
// bn_finalize_kernel + bn_apply_fwd_rows_kernel in one launch: every workgroup derives the coefficients of its lanes' 8 channels from the
// statistics itself (same arithmetic as bn_finalize_kernel), workgroup 0's first row of lanes also writes `fin` (the backward reads it)
// and the moving statistics.
// Statistics spread over `copies` copies [copies][2][C] by their producer (tfasr_dwconv_fwd_stats, the E_BNS epilogue of tfasr_gemm): the
// workgroup adds them up ONCE through LDS - thread i owns four consecutive floats of the 2C sums and requests its (up to 8) copies of them
// in one batch - instead of every lane walking the copies of its own eight channels (a chain of dependent round trips in front of a
// 12 us kernel: +3.9 us measured).  sred [2C] floats; call from every thread of the block.
constexpr int BN_COPIES_MAX = 8;
__device__ __forceinline__ void bn_sum_copies(const float* __restrict__ stats, int copies, int C, float* sred) {
  for (int u = threadIdx.x; u < (2 * C) / 4; u += blockDim.x) {
    float4 t[BN_COPIES_MAX];
#pragma unroll
    for (int q = 0; q < BN_COPIES_MAX; ++q) t[q] = *reinterpret_cast<const float4*>(stats + (size_t)min(q, copies - 1) * 2 * C + u * 4);
    float4 a = t[0];
#pragma unroll
    for (int q = 1; q < BN_COPIES_MAX; ++q)
      if (q < copies) { a.x += t[q].x; a.y += t[q].y; a.z += t[q].z; a.w += t[q].w; }
    *reinterpret_cast<float4*>(sred + u * 4) = a;
  }
  __syncthreads();
}
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_apply_fwd_rows_kernel(const T* __restrict__ x, const float* __restrict__ stats, float count,
                                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                         float* __restrict__ fin, float* moving_mean, float* moving_var, float momentum,
                                                                         float eps, T* __restrict__ y, long rows, int C, int act, int training, int copies) {
  const int lpr = C >> 3, li = threadIdx.x % lpr, sub = threadIdx.x / lpr, rpb = 256 / lpr;
  const int c = li * 8;
  __shared__ __attribute__((aligned(16))) float sred[2 * 2048];  // (the row kernels take C <= 2048)
  if (training && copies > 1) bn_sum_copies(stats, copies, C, sred);
  float sc[8], sh[8];
  {
    float mean[8], var[8], gm[8], bt[8], mm[8], mv[8];
    ld8(gamma + c, gm); ld8(beta + c, bt);
    const bool writer = blockIdx.x == 0 && sub == 0;
    if (training) {
      float s0[8], s1[8];
      if (copies > 1) { ld8(sred + c, s0); ld8(sred + C + c, s1); }
      else { ld8(stats + c, s0); ld8(stats + C + c, s1); }
      if (writer && moving_mean) { ld8(moving_mean + c, mm); ld8(moving_var + c, mv); }
#pragma unroll
      for (int k = 0; k < 8; ++k) { mean[k] = s0[k] / count; var[k] = fmaxf(s1[k] / count - mean[k] * mean[k], 0.f); }
      if (writer && moving_mean) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { mm[k] = mm[k] * momentum + mean[k] * (1.f - momentum); mv[k] = mv[k] * momentum + var[k] * (1.f - momentum); }
        st8(moving_mean + c, mm); st8(moving_var + c, mv);
      }
    } else {
      ld8(moving_mean + c, mean); ld8(moving_var + c, var);
    }
    float rstd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { rstd[k] = rsqrtf(var[k] + eps); sc[k] = gm[k] * rstd[k]; sh[k] = bt[k] - mean[k] * sc[k]; }
    if (writer) { st8(fin + c, mean); st8(fin + C + c, rstd); st8(fin + 2 * C + c, sc); st8(fin + 3 * C + c, sh); }
  }
  const long step = (long)gridDim.x * rpb;
  for (long r0 = (long)blockIdx.x * rpb + sub; r0 < rows; r0 += 2 * step) {
    float v[2][8];
    const bool two = r0 + step < rows;
    ld8(x + r0 * C + c, v[0]);
    if (two) ld8(x + (r0 + step) * C + c, v[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float z = v[u][k] * sc[k] + sh[k];
        if (act == TFASR_ACT_SWISH) z = swishf_(z);
        v[u][k] = z;
      }
      st8(y + (r0 + u * step) * C + c, v[u]);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ fin,
                                                                const float* __restrict__ bstats, float count, T* dx, long rows, int C, int act,
                                                                float* dgamma, float* dbeta, float gscale, int copies) {
  const int lpr = C >> 3, li = threadIdx.x % lpr, sub = threadIdx.x / lpr, rpb = 256 / lpr;
  const int c = li * 8;
  __shared__ __attribute__((aligned(16))) float sred[2 * 2048];
  if (copies > 1) bn_sum_copies(bstats, copies, C, sred);
  float sc[8], sh[8], cb[8], cd[8];  // dx = sc * dz + cb * x + cd  (dz = dy * act'(sc x + sh))
  {
    float mean[8], rstd[8], s0[8], s1[8];
    ld8(fin + c, mean); ld8(fin + C + c, rstd); ld8(fin + 2 * C + c, sc); ld8(fin + 3 * C + c, sh);
    if (copies > 1) { ld8(sred + c, s0); ld8(sred + C + c, s1); }
    else { ld8(bstats + c, s0); ld8(bstats + C + c, s1); }
    // the BatchNorm parameter gradients are the two statistics themselves: dbeta += gscale * sum dz, dgamma += gscale * sum dz xhat
    // (one writer: block 0's first row of lanes; they were two extra axpy launches per BatchNorm)
    if (blockIdx.x == 0 && sub == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (dbeta) dbeta[c + k] += gscale * s0[k];
        if (dgamma) dgamma[c + k] += gscale * s1[k];
      }
    }
    const float inv = 1.f / count;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float q = sc[k] * rstd[k] * s1[k] * inv;
      cb[k] = -q;
      cd[k] = q * mean[k] - sc[k] * s0[k] * inv;
    }
  }
  const long step = (long)gridDim.x * rpb;
  for (long r0 = (long)blockIdx.x * rpb + sub; r0 < rows; r0 += 2 * step) {
    float xv[2][8], d[2][8];
    const bool two = r0 + step < rows;
    ld8(x + r0 * C + c, xv[0]);
    ld8(dy + r0 * C + c, d[0]);
    if (two) { ld8(x + (r0 + step) * C + c, xv[1]); ld8(dy + (r0 + step) * C + c, d[1]); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float dz = d[u][k];
        if (act == TFASR_ACT_SWISH) dz *= dswishf_(xv[u][k] * sc[k] + sh[k]);
        d[u][k] = sc[k] * dz + cb[k] * xv[u][k] + cd[k];
      }
      st8(dx + (r0 + u * step) * C + c, d[u]);
    }
  }
}
inline bool rows_variant_ok(int C) { return (C % 8) == 0 && C >= 64 && C <= 2048 && (256 % (C / 8)) == 0; }
inline int rows_variant_grid(long rows, int C) {
  const long rpb = 256 / (C / 8);
  return (int)std::max<long>(1, std::min<long>((rows + 2 * rpb - 1) / (2 * rpb), 2048L));
}

// ------------------------------------------------------------------------- vec8 variants (C % 8 == 0, C <= 512)
// LPR lanes cover one row with 16-B accesses (8 channels per lane); 64/LPR rows per wave iteration.
// 8 consecutive f32 coefficients of a live lane: two 16-byte loads when the address allows, never 8 branchy dword loads
// (with 32 lanes x 32-B stride every scalar load instruction touches 8 cache lines: the parameter loads then cost more
// memory-pipeline work than the activation stream itself -- 18.9 us vs 6.3 us for a [23808,256] LayerNorm)
__device__ __forceinline__ void ldf8(const float* __restrict__ p, float (&v)[8], bool live) {
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.f;
  if (!live) return;
  if ((((uintptr_t)p) & 15) == 0) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[k];
  }
}

template <int LPR> __device__ __forceinline__ float seg_sum(float v) {
  v = row16_sum(v);  // DPP inside each 16-lane row, then the (1 or 2) cross-row steps through the permute unit
#pragma unroll
  for (int o = 16; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// U rows in flight per lane group: all U row loads (and the coefficient loads) are issued before anything waits, so a wave's life is one
// memory round trip + arithmetic + stores instead of a chain of them (one row at a time: 10.2 us for [23808, 256], the device copy 4.7).
template <typename T, int LPR, int U = 2>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         long rows, int C, float eps) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, li = lane % LPR, sub = lane / LPR;
  const int c0 = li * 8;
  const bool act = c0 < C;
  const long w0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + sub;
  const long step = (long)gridDim.x * (blockDim.x >> 6) * RPW;
  const float invC = 1.f / C;
  for (long r0 = w0; r0 < rows; r0 += U * step) {  // shuffles stay inside one LPR-lane segment (= one row)
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[u][k] = 0.f;
      const long r = r0 + u * step;
      if (act && r < rows) ld8(x + r * C + c0, v[u]);
    }
    float g[8], b[8];
    ldf8(gamma + c0, g, act);
    ldf8(beta + c0, b, act);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + u * step;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[u][k];
      const float mean = seg_sum<LPR>(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = act ? v[u][k] - mean : 0.f; q += d * d; }
      const float rstd = rsqrtf(seg_sum<LPR>(q) * invC + eps);
      if (r < rows) {
        if (act) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[u][k] = (v[u][k] - mean) * rstd * g[k] + b[k];
          st8(y + r * C + c0, v[u]);
        }
        if (li == 0) { if (mean_out) mean_out[r] = mean; if (rstd_out) rstd_out[r] = rstd; }
      }
    }
  }
}

// Row-reduction kernels below end in one global atomicAdd per column per BLOCK, and same-address atomics serialise
// (~20-30 ns each on this chip): with 512 blocks that tail alone cost ~10 us.  So: few (<= 128) fat 1024-thread blocks,
// two rows in flight per lane for bandwidth, partial sums combined with LDS atomics, then <= 128 global atomics per column.
// U = rows in flight per lane (bytes in flight are what the memory system rewards: 2 rows x 3 streams x 16 B per lane left a
// 512-thread block per CU at ~2.7 TB/s), NW = waves per block.
template <typename T, int LPR, int U = 2, int NW = 8>
__global__ __launch_bounds__(NW * 64) void ln_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const T* add, T* dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, long rows,
                                                          int C, T* dxm, float drop_p, uint64_t drop_seed, float* __restrict__ part = nullptr) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[2][NW * RPW][LPR * 8];
  const int lane = threadIdx.x & 63, li = lane % LPR, sub = lane / LPR, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c0 = li * 8;
  const bool act = c0 < C;
  const float drop_inv = 1.f / (1.f - drop_p);
  float g[8], ag[8], ab[8];
  ldf8(gamma + c0, g, act);
#pragma unroll
  for (int k = 0; k < 8; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
  const long w0 = ((long)blockIdx.x * nw + w) * RPW + sub;
  const long step = (long)gridDim.x * nw * RPW;
  const float invC = 1.f / C;
  for (long r0 = w0; r0 < rows; r0 += U * step) {
    float d[U][8], xv[U][8], o[U][8], m[U], rs[U];
    bool rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + u * step;
      rv[u] = act && r < rows;
#pragma unroll
      for (int k = 0; k < 8; ++k) { d[u][k] = 0.f; xv[u][k] = 0.f; o[u][k] = 0.f; }
      m[u] = 0.f; rs[u] = 0.f;
      if (rv[u]) {
        ld8(dy + r * C + c0, d[u]); ld8(x + r * C + c0, xv[u]); m[u] = mean[r]; rs[u] = rstd[r];
        if (add) ld8(add + r * C + c0, o[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        xv[u][k] = (xv[u][k] - m[u]) * rs[u];  // xhat
        const float dg = d[u][k] * g[k];
        s1 += dg;
        s2 += dg * xv[u][k];
        ag[k] += d[u][k] * xv[u][k];
        ab[k] += d[u][k];
      }
      s1 = seg_sum<LPR>(s1) * invC;
      s2 = seg_sum<LPR>(s2) * invC;
      if (rv[u]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[u][k] += rs[u] * (d[u][k] * g[k] - s1 - xv[u][k] * s2);
        const long base = (r0 + u * step) * C + c0;
        st8(dx + base, o[u]);
        if (dxm) {  // second output: dropout(dx) for the consumer's masked branch, computed from the ROUNDED dx (= tfasr_dropout(dx))
          float mv[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float rr = o[u][k];
            if (sizeof(T) == 2) rr = bf16_to_f32(f32_to_bf16(rr));
            mv[k] = drop_keep(drop_seed, (uint64_t)(base + k), drop_p) ? rr * drop_inv : 0.f;
          }
          st8(dxm + base, mv);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][w * RPW + sub][c0 + k] = ag[k]; red[1][w * RPW + sub][c0 + k] = ab[k]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sg = 0.f, sb = 0.f;
    for (int q = 0; q < nw * RPW; ++q) { sg += red[0][q][c]; sb += red[1][q][c]; }
    if (part) {  // per-block partial sums, folded later by ln_bwd_fold_kernel (no same-address atomic chain at the end of this kernel)
      part[(long)blockIdx.x * 2 * C + c] = sg;
      part[(long)blockIdx.x * 2 * C + C + c] = sb;
    } else {
      if (dgamma) atomicAdd(dgamma + c, sg);
      if (dbeta) atomicAdd(dbeta + c, sb);
    }
  }
}

// dgamma / dbeta of up to LNF_MAX LayerNorms from their per-block partial sums part[set][nblk][2C]: block = (64 columns of one set),
// 4 thread groups each take a quarter of the partial rows, LDS combine, ONE writer per column (plain +=, no atomics)
constexpr int LNF_MAX = 8;
struct LnFoldArgs { float* dg[LNF_MAX]; float* db[LNF_MAX]; };
// the fold of one (set, 64-column slice): p = the set's partial sums at this thread's column
__device__ __forceinline__ void ln_fold_slice(const float* __restrict__ p, int nblk, int C, int col, float* dg, float* db) {
  __shared__ float red[4][64];
  const int grp = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < 2 * C) {
    const int per = (nblk + 3) / 4, b0 = grp * per, b1 = min(nblk, b0 + per);
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
      s0 += p[(long)b * 2 * C]; s1 += p[(long)(b + 1) * 2 * C]; s2 += p[(long)(b + 2) * 2 * C]; s3 += p[(long)(b + 3) * 2 * C];
    }
    for (; b < b1; ++b) s0 += p[(long)b * 2 * C];
  }
  red[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && col < 2 * C) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (col < C) { if (dg) dg[col] += v; }
    else if (db) db[col - C] += v;
  }
}
__global__ __launch_bounds__(256) void ln_bwd_fold_kernel(const float* __restrict__ part, int nblk, int C, LnFoldArgs a) {
  const int set = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63);
  ln_fold_slice(part + (long)set * nblk * 2 * C + col, nblk, C, col, a.dg[set], a.db[set]);
}
// the same for sets that lie in different buffers (the LayerNorms of ALL Conformer blocks of a step: one launch instead of one per block);
// the table is a kernel argument (3 KiB of the 4 KiB a dispatch may carry)
constexpr int LNF_SETS_MAX = 128;
struct LnFoldSets { const float* part[LNF_SETS_MAX]; float* dg[LNF_SETS_MAX]; float* db[LNF_SETS_MAX]; };
__global__ __launch_bounds__(256) void ln_bwd_fold_sets_kernel(int nblk, int C, LnFoldSets a) {
  const int set = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63);
  ln_fold_slice(a.part[set] + col, nblk, C, col, a.dg[set], a.db[set]);
}

template <typename T, int MODE, int LPR>
__global__ __launch_bounds__(512) void bn_stats_vec_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            const float* __restrict__ fin, float* __restrict__ stats,
                                                            long rows, int C, int act_kind) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[2][8 * RPW][LPR * 8];  // up to 8 waves (512 threads)
  const int lane = threadIdx.x & 63, li = lane % LPR, sub = lane / LPR, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int cs = blockIdx.y * (LPR * 8);  // channel slab (C > LPR*8: ContextNet widths up to 1280)
  const int c0 = cs + li * 8;
  const bool act = c0 < C;
  float a0[8], a1[8], mean[8], rstd[8], sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a0[k] = 0.f; a1[k] = 0.f; }
  ldf8(fin + c0, mean, MODE == 1 && act);
  ldf8(fin + C + c0, rstd, MODE == 1 && act);
  ldf8(fin + 2 * C + c0, sc, MODE == 1 && act);
  ldf8(fin + 3 * C + c0, sh, MODE == 1 && act);
  const long w0 = ((long)blockIdx.x * nw + w) * RPW + sub;
  const long step = (long)gridDim.x * nw * RPW;
  if (act)
    for (long r0 = w0; r0 < rows; r0 += 2 * step) {
      float xv[2][8], d[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long r = r0 + u * step;
#pragma unroll
        for (int k = 0; k < 8; ++k) { xv[u][k] = 0.f; d[u][k] = 0.f; }
        if (r < rows) {
          ld8(x + r * C + c0, xv[u]);
          if (MODE == 1) ld8(dy + r * C + c0, d[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) { a0[k] += xv[u][k]; a1[k] += xv[u][k] * xv[u][k]; }
        } else if (r0 + u * step < rows) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float dz = d[u][k];
            if (act_kind == TFASR_ACT_SWISH) dz *= dswishf_(xv[u][k] * sc[k] + sh[k]);
            a0[k] += dz;
            a1[k] += dz * (xv[u][k] - mean[k]) * rstd[k];
          }
        }
      }
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][w * RPW + sub][li * 8 + k] = a0[k]; red[1][w * RPW + sub][li * 8 + k] = a1[k]; }
  __syncthreads();
  for (int cl = threadIdx.x; cl < LPR * 8 && cs + cl < C; cl += blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    for (int q = 0; q < nw * RPW; ++q) { s0 += red[0][q][cl]; s1 += red[1][q][cl]; }
    atomicAdd(stats + cs + cl, s0);
    atomicAdd(stats + C + cs + cl, s1);
  }
}

// launch shape of the fat row-reduction kernels: 1024 threads, <= 128 blocks, >= 2 row-pairs per wave
inline int red_threads() { return 512; }
inline int red_grid_cap() { return 192; }
inline int fat_grid(long rows, int rows_per_wave_iter) {
  const long per_block = (long)(red_threads() / 64) * rows_per_wave_iter;
  // the per-block atomic tail (~17 ns per block per column) is only worth limiting for small inputs
  const long cap = std::max<long>(red_grid_cap(), std::min<long>(rows / 4096, 1024L));
  return (int)std::max<long>(1, std::min<long>(rows / (2L * per_block) + 1, cap));
}

inline int rows_grid(long rows) { return (int)std::min<long>((rows + 3) / 4, 256L * 8); }
inline int flat_grid(long n) { return (int)std::min<long>((n + 255) / 256, 256L * 16); }

}  // namespace

extern "C" int tfasr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                   float* rstd, long rows, int C, float eps, int dtype, void* stream_) {
  if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0 || C > 64 * MAXC_PER_LANE) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16 && (C % 8) == 0 && C <= 512) {
    if (C <= 256) {
      static const int U = 4;
      static const long cap = 2048L;
      const int grid = (int)std::min<long>((rows + 8 * U - 1) / (8 * U), cap);
      if (U == 4) TFASR_KLAUNCH((ln_fwd_vec_kernel<bf16_t, 32, 4>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, C, eps);
      else if (U == 1) TFASR_KLAUNCH((ln_fwd_vec_kernel<bf16_t, 32, 1>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, C, eps);
      else TFASR_KLAUNCH((ln_fwd_vec_kernel<bf16_t, 32>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, C, eps);
    } else {
      const int grid = (int)std::min<long>((rows + 3) / 4, 2048L);
      TFASR_KLAUNCH((ln_fwd_vec_kernel<bf16_t, 64>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, C, eps);
    }
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(ln_fwd_kernel<float>, dim3(rows_grid(rows)), dim3(256), 0, s, (const float*)x, gamma, beta,
                       (float*)y, mean, rstd, rows, C, eps);
  else
    TFASR_KLAUNCH(ln_fwd_kernel<bf16_t>, dim3(rows_grid(rows)), dim3(256), 0, s, (const bf16_t*)x, gamma, beta,
                       (bf16_t*)y, mean, rstd, rows, C, eps);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

static int num_cus_norm() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n;
}
// blocks of the vectorised backward for a shape; the partial-sum variant has no atomic tail to limit, so it takes every CU
static int ln_bwd_vec_grid(long rows, int C, bool part) {
  const int rpw = C <= 256 ? 2 : 1;
  if (!part) return std::max(1, fat_grid(rows, rpw));
  const long per_block = 8L * rpw;  // rows per block pass (8 waves)
  return (int)std::max<long>(1, std::min<long>(rows / (2 * per_block) + 1, (long)num_cus_norm()));
}
static int ln_bwd_vec_launch(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* add, void* dx,
                             float* dgamma, float* dbeta, void* dx_dropped, float drop_p, long drop_seed, long rows, int C, float* part,
                             hipStream_t s, int nblk = 0) {
  // nblk > 0: the caller's partial-sum buffer has that many slots (shared with other producers): blocks without rows store zeros
  const int grid = nblk > 0 ? nblk : ln_bwd_vec_grid(rows, C, part != nullptr);
  if (C <= 256)
    TFASR_KLAUNCH((ln_bwd_vec_kernel<bf16_t, 32, 2, 8>), dim3(grid), dim3(512), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                       (const bf16_t*)add, (bf16_t*)dx, dgamma, dbeta, rows, C, (bf16_t*)dx_dropped, drop_p, (uint64_t)drop_seed, part);
  else
    TFASR_KLAUNCH((ln_bwd_vec_kernel<bf16_t, 64, 2, 8>), dim3(grid), dim3(512), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                       (const bf16_t*)add, (bf16_t*)dx, dgamma, dbeta, rows, C, (bf16_t*)dx_dropped, drop_p, (uint64_t)drop_seed, part);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_layernorm_bwd_part_blocks(long rows, int C, int dtype) {
  if (dtype != TFASR_BF16 || rows <= 0 || C <= 0 || (C % 8) || C > 512) return 0;
  return ln_bwd_vec_grid(rows, C, true);
}
extern "C" int tfasr_layernorm_bwd_part(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* add,
                                        void* dx, float* part, void* dx_dropped, float drop_p, long drop_seed, long rows, int C, int dtype,
                                        void* stream_) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !part || rows <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (dx_dropped && !(drop_p >= 0.f && drop_p < 1.f)) return TFASR_STATUS_INVALID_VALUE;
  if (dtype != TFASR_BF16 || (C % 8) || C > 512 || C <= 0) return TFASR_STATUS_UNSUPPORTED;
  return ln_bwd_vec_launch(dy, x, gamma, mean, rstd, add, dx, nullptr, nullptr, dx_dropped, drop_p, drop_seed, rows, C, part, (hipStream_t)stream_);
}
extern "C" int tfasr_layernorm_bwd_part_n(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* add,
                                          void* dx, float* part, int nblk, void* dx_dropped, float drop_p, long drop_seed, long rows, int C, int dtype,
                                          void* stream_) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !part || rows <= 0 || nblk <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (dx_dropped && !(drop_p >= 0.f && drop_p < 1.f)) return TFASR_STATUS_INVALID_VALUE;
  if (dtype != TFASR_BF16 || (C % 8) || C > 512 || C <= 0) return TFASR_STATUS_UNSUPPORTED;
  return ln_bwd_vec_launch(dy, x, gamma, mean, rstd, add, dx, nullptr, nullptr, dx_dropped, drop_p, drop_seed, rows, C, part, (hipStream_t)stream_, nblk);
}
extern "C" int tfasr_layernorm_bwd_fold(const float* part, int nsets, int nblk, int C, float* const* dgamma, float* const* dbeta, void* stream_) {
  if (!part || nsets <= 0 || nsets > LNF_MAX || nblk <= 0 || C <= 0 || !dgamma || !dbeta) return TFASR_STATUS_INVALID_VALUE;
  LnFoldArgs a;
  for (int i = 0; i < LNF_MAX; ++i) { a.dg[i] = i < nsets ? dgamma[i] : nullptr; a.db[i] = i < nsets ? dbeta[i] : nullptr; }
  TFASR_KLAUNCH(ln_bwd_fold_kernel, dim3((2 * C + 63) / 64, nsets), dim3(256), 0, (hipStream_t)stream_, part, nblk, C, a);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_layernorm_bwd_fold_sets(const float* const* part, int nsets, int nblk, int C, float* const* dgamma, float* const* dbeta,
                                             void* stream_) {
  if (!part || nsets <= 0 || nsets > LNF_SETS_MAX || nblk <= 0 || C <= 0 || !dgamma || !dbeta) return TFASR_STATUS_INVALID_VALUE;
  LnFoldSets a;
  for (int i = 0; i < LNF_SETS_MAX; ++i) {
    a.part[i] = i < nsets ? part[i] : nullptr;
    a.dg[i] = i < nsets ? dgamma[i] : nullptr;
    a.db[i] = i < nsets ? dbeta[i] : nullptr;
    if (i < nsets && !part[i]) return TFASR_STATUS_INVALID_VALUE;
  }
  TFASR_KLAUNCH(ln_bwd_fold_sets_kernel, dim3((2 * C + 63) / 64, nsets), dim3(256), 0, (hipStream_t)stream_, nblk, C, a);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                        const void* add, void* dx, float* dgamma, float* dbeta, void* dx_dropped, float drop_p,
                                        long drop_seed, long rows, int C, int dtype, void* stream_) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || rows <= 0 || C <= 0 || C > 64 * MAXC_PER_LANE)
    return TFASR_STATUS_INVALID_VALUE;
  if (dx_dropped && !(drop_p >= 0.f && drop_p < 1.f)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16 && (C % 8) == 0 && C <= 512)
    return ln_bwd_vec_launch(dy, x, gamma, mean, rstd, add, dx, dgamma, dbeta, dx_dropped, drop_p, drop_seed, rows, C, nullptr, s);
  const int grid = (int)std::min<long>((rows + 3) / 4, 256L);
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(ln_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, mean,
                       rstd, (const float*)add, (float*)dx, dgamma, dbeta, rows, C);
  else
    TFASR_KLAUNCH(ln_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma,
                       mean, rstd, (const bf16_t*)add, (bf16_t*)dx, dgamma, dbeta, rows, C);
  TFASR_CHECK_LAUNCH();
  // shapes outside the vectorised kernel: the dropped copy is a separate pass (same mask, same rounding)
  if (dx_dropped) return tfasr_dropout(dx, dx_dropped, rows * C, drop_p, drop_seed, dtype, stream_);
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                   const float* rstd, const void* add, void* dx, float* dgamma, float* dbeta, long rows,
                                   int C, int dtype, void* stream_) {
  return tfasr_layernorm_bwd_drop(dy, x, gamma, mean, rstd, add, dx, dgamma, dbeta, nullptr, 0.f, 0, rows, C, dtype, stream_);
}

extern "C" int tfasr_bn_stats(const void* x, float* stats, long rows, int C, int dtype, void* stream_) {
  if (!x || !stats || rows <= 0 || C <= 0 || (C > 64 * MAXC_PER_LANE && (dtype != TFASR_BF16 || (C % 8)))) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16 && (C % 8) == 0) {
    if (C <= 256) TFASR_KLAUNCH((bn_stats_vec_kernel<bf16_t, 0, 32>), dim3(fat_grid(rows, 2)), dim3(red_threads()), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, stats, rows, C, 0);
    else TFASR_KLAUNCH((bn_stats_vec_kernel<bf16_t, 0, 64>), dim3(fat_grid(rows, 1), (C + 511) / 512), dim3(red_threads()), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, stats, rows, C, 0);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  const int grid = (int)std::min<long>((rows + 3) / 4, 1024L);
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH((bn_stats_kernel<float, 0>), dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)nullptr,
                       (const float*)nullptr, stats, rows, C, 0);
  else
    TFASR_KLAUNCH((bn_stats_kernel<bf16_t, 0>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)nullptr, (const float*)nullptr, stats, rows, C, 0);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_bn_finalize(const float* stats, float count, const float* gamma, const float* beta, float* fin,
                                 float* moving_mean, float* moving_var, float momentum, float eps, int C, int training,
                                 void* stream_) {
  if (!gamma || !beta || !fin || C <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (training && !stats) return TFASR_STATUS_INVALID_VALUE;
  if (!training && (!moving_mean || !moving_var)) return TFASR_STATUS_INVALID_VALUE;
  TFASR_KLAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream_, stats, count, gamma,
                     beta, fin, moving_mean, moving_var, momentum, eps, C, training);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_bn_apply_fwd(const void* x, const float* fin, void* y, long rows, int C, int act, int dtype,
                                  void* stream_) {
  if (!x || !fin || !y || rows <= 0 || C <= 0 || (C % 8) != 0) return TFASR_STATUS_INVALID_VALUE;
  const long n8 = rows * C / 8;
  hipStream_t s = (hipStream_t)stream_;
  if (rows_variant_ok(C)) {
    const int grid = rows_variant_grid(rows, C);
    if (dtype == TFASR_F32) TFASR_KLAUNCH(bn_apply_fwd_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, fin, (float*)y, rows, C, act);
    else TFASR_KLAUNCH(bn_apply_fwd_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, fin, (bf16_t*)y, rows, C, act);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(bn_apply_fwd_kernel<float>, dim3(flat_grid(n8)), dim3(256), 0, s, (const float*)x, fin, (float*)y,
                       n8, C, act);
  else
    TFASR_KLAUNCH(bn_apply_fwd_kernel<bf16_t>, dim3(flat_grid(n8)), dim3(256), 0, s, (const bf16_t*)x, fin,
                       (bf16_t*)y, n8, C, act);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

// (block.hip: whether tfasr_bn_finalize_apply_fwd_copies will take the channel count - it decides before it spreads statistics over copies)
bool tfasr_bn_rows_kernel_ok(int C) { return (C % 8) == 0 && rows_variant_ok(C); }

extern "C" int tfasr_bn_finalize_apply_fwd_copies(const void* x, const float* stats, int copies, float count, const float* gamma, const float* beta, float* fin,
                                           float* moving_mean, float* moving_var, float momentum, float eps, void* y, long rows, int C, int act,
                                           int training, int dtype, void* stream_) {
  if (!x || !gamma || !beta || !fin || !y || rows <= 0 || C <= 0 || copies <= 0 || copies > BN_COPIES_MAX) return TFASR_STATUS_INVALID_VALUE;
  if (training && !stats) return TFASR_STATUS_INVALID_VALUE;
  if (!training && (!moving_mean || !moving_var)) return TFASR_STATUS_INVALID_VALUE;
  if ((C % 8) != 0 || !rows_variant_ok(C) || (dtype != TFASR_F32 && dtype != TFASR_BF16)) return TFASR_STATUS_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = rows_variant_grid(rows, C);
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(bn_finalize_apply_fwd_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, stats, count, gamma, beta, fin, moving_mean,
                       moving_var, momentum, eps, (float*)y, rows, C, act, training, copies);
  else
    TFASR_KLAUNCH(bn_finalize_apply_fwd_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, stats, count, gamma, beta, fin, moving_mean,
                       moving_var, momentum, eps, (bf16_t*)y, rows, C, act, training, copies);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_bn_finalize_apply_fwd(const void* x, const float* stats, float count, const float* gamma, const float* beta, float* fin,
                                           float* moving_mean, float* moving_var, float momentum, float eps, void* y, long rows, int C, int act,
                                           int training, int dtype, void* stream_) {
  return tfasr_bn_finalize_apply_fwd_copies(x, stats, 1, count, gamma, beta, fin, moving_mean, moving_var, momentum, eps, y, rows, C, act, training, dtype, stream_);
}

extern "C" int tfasr_bn_bwd_stats(const void* x, const void* dy, const float* fin, float* bstats, long rows, int C,
                                  int act, int dtype, void* stream_) {
  if (!x || !dy || !fin || !bstats || rows <= 0 || C <= 0 || (C > 64 * MAXC_PER_LANE && (dtype != TFASR_BF16 || (C % 8)))) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == TFASR_BF16 && (C % 8) == 0) {
    if (C <= 256) TFASR_KLAUNCH((bn_stats_vec_kernel<bf16_t, 1, 32>), dim3(fat_grid(rows, 2)), dim3(red_threads()), 0, s, (const bf16_t*)x, (const bf16_t*)dy, fin, bstats, rows, C, act);
    else TFASR_KLAUNCH((bn_stats_vec_kernel<bf16_t, 1, 64>), dim3(fat_grid(rows, 1), (C + 511) / 512), dim3(red_threads()), 0, s, (const bf16_t*)x, (const bf16_t*)dy, fin, bstats, rows, C, act);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  const int grid = (int)std::min<long>((rows + 3) / 4, 1024L);
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH((bn_stats_kernel<float, 1>), dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)dy, fin,
                       bstats, rows, C, act);
  else
    TFASR_KLAUNCH((bn_stats_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy,
                       fin, bstats, rows, C, act);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_bn_apply_bwd_grads_copies(const void* x, const void* dy, const float* fin, const float* bstats, int copies, float count, void* dx, long rows,
                                        int C, int act, float* dgamma, float* dbeta, float grad_scale, int dtype, void* stream_) {
  if (!x || !dy || !fin || !bstats || !dx || rows <= 0 || C <= 0 || (C % 8) != 0 || copies <= 0 || copies > BN_COPIES_MAX) return TFASR_STATUS_INVALID_VALUE;
  if (copies > 1 && !rows_variant_ok(C)) return TFASR_STATUS_UNSUPPORTED;  // only the row kernel adds copies up
  const long n8 = rows * C / 8;
  hipStream_t s = (hipStream_t)stream_;
  if (rows_variant_ok(C)) {
    const int grid = rows_variant_grid(rows, C);
    if (dtype == TFASR_F32) TFASR_KLAUNCH(bn_apply_bwd_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)dy, fin, bstats, count, (float*)dx, rows, C, act, dgamma, dbeta, grad_scale, copies);
    else TFASR_KLAUNCH(bn_apply_bwd_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, fin, bstats, count, (bf16_t*)dx, rows, C, act, dgamma, dbeta, grad_scale, copies);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  // widths outside the row kernel: the parameter gradients as separate passes
  if (dbeta) { const int st = tfasr_axpy(dbeta, bstats, grad_scale, C, stream_); if (st != TFASR_STATUS_SUCCESS) return st; }
  if (dgamma) { const int st = tfasr_axpy(dgamma, bstats + C, grad_scale, C, stream_); if (st != TFASR_STATUS_SUCCESS) return st; }
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(bn_apply_bwd_kernel<float>, dim3(flat_grid(n8)), dim3(256), 0, s, (const float*)x,
                       (const float*)dy, fin, bstats, count, (float*)dx, n8, C, act);
  else
    TFASR_KLAUNCH(bn_apply_bwd_kernel<bf16_t>, dim3(flat_grid(n8)), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)dy, fin, bstats, count, (bf16_t*)dx, n8, C, act);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_bn_apply_bwd_grads(const void* x, const void* dy, const float* fin, const float* bstats, float count, void* dx, long rows,
                                        int C, int act, float* dgamma, float* dbeta, float grad_scale, int dtype, void* stream_) {
  return tfasr_bn_apply_bwd_grads_copies(x, dy, fin, bstats, 1, count, dx, rows, C, act, dgamma, dbeta, grad_scale, dtype, stream_);
}

extern "C" int tfasr_bn_apply_bwd(const void* x, const void* dy, const float* fin, const float* bstats, float count,
                                  void* dx, long rows, int C, int act, int dtype, void* stream_) {
  return tfasr_bn_apply_bwd_grads(x, dy, fin, bstats, count, dx, rows, C, act, nullptr, nullptr, 0.f, dtype, stream_);
}
