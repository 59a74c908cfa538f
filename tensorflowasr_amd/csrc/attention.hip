// Relative-position attention softmax (forward / backward) for gfx950.
//
// Reference: MultiHeadRelativeAttention._compute_attention (multihead_attention.py:543-582):
//   scores = content + rel_left_shift(positional)[..., -T:]            (:554-569)
//   probs  = softmax(masked_fill(scores, mask, -1e9))                  (general.py:30-41, math_util.py:229-246)
// rel_left_shift (multihead_attention.py:27-77, causal=False) is pure index arithmetic:
//   shifted[i, j] = positional[i, T-1-i+j]                             
// and the per-sample roll/mask of RelativeSinusoidalPositionalEncoding.call (positional_encoding.py:152-172)
//   pe_b[r] = pe[(r + T - len_b) mod R]  for r < 2*len_b-1, else 0     (R = 2T-1)
// is folded into the same gather: the positional score matrix `pos` is computed ONCE against the shared,
// un-rolled projected table (R rows) plus one extra column R that holds (q+v).bias (the projection of a
// zeroed encoding row), so pos has R1 = R+1 columns:
//   pos_score(b,h,i,j) = pos[b,h,i, r < 2*len_b-1 ? (r + T - len_b) mod R : R],  r = T-1-i+j.
// Auto mask: only padded QUERY rows are masked (whole row -> uniform 1/T); keys are not.
// One wave per (b,h,i) row; the row lives in registers (T <= 64*MAXJ).
#include "common.h"
#include <algorithm>

namespace {

constexpr int MAXJ = 16;  // T <= 1024

__device__ __forceinline__ int pos_col(int i, int j, int T, int R, int len) {
  const int r = T - 1 - i + j;
  if (r >= 2 * len - 1) return R;
  int rr = r + (T - len);
  if (rr >= R) rr -= R;
  return rr;
}

template <typename T_>
__global__ __launch_bounds__(256) void relattn_softmax_fwd_kernel(const T_* content, const T_* __restrict__ pos,
                                                                  const int32_t* __restrict__ lengths, T_* probs, int B,
                                                                  int H, int T, int ldc, int ldp, int use_mask, int chunk, int hist) {
  const int lane = threadIdx.x & 63;
  const int R = 2 * T - 1;
  const long nrows = (long)B * H * T;
  const long w0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long row = w0; row < nrows; row += nw) {
    const int i = (int)(row % T);
    const int b = (int)(row / ((long)H * T));
    const int len = lengths ? min(lengths[b], T) : T;
    T_* out = probs + row * ldc;
    if (use_mask && i >= len) {  // masked query row: every score == -1e9 -> uniform
      const float uval = 1.f / T;
      for (int j = lane; j < T; j += 64) Num<T_>::st(out + j, uval);
      continue;
    }
    const T_* crow = content + row * ldc;
    const T_* prow = pos + row * ldp;
    float s[MAXJ];
    float mx = -INFINITY;
    int n = 0;
    // streaming window of query i (compute_streaming_mask, multihead_attention.py:104-143): keys outside it carry -1e9 in the
    // reference = probability exactly 0 (a window is never empty: it always contains the query's own chunk)
    int jlo = 0, jhi = T;
    if (chunk > 0) {
      const int index = (i / chunk) * chunk;
      jlo = hist < 0 ? 0 : max(0, index - hist);
      jhi = min(T, index + chunk);
    }
    for (int j = lane; j < T; j += 64, ++n) {
      s[n] = (j >= jlo && j < jhi) ? Num<T_>::ld(crow + j) + Num<T_>::ld(prow + pos_col(i, j, T, R, len)) : -INFINITY;
      mx = fmaxf(mx, s[n]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = 0; k < n; ++k) { s[k] = __expf(s[k] - mx); sum += s[k]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    n = 0;
    for (int j = lane; j < T; j += 64, ++n) Num<T_>::st(out + j, s[n] * inv);
  }
}

// dS = P * (dP - sum_j dP*P); dcontent = dS (may alias dP); dpos[i, col] = gathered dS, bias column = leftover sum
template <typename T_>
__global__ __launch_bounds__(256) void relattn_softmax_bwd_kernel(const T_* __restrict__ probs, const T_* dprobs,
                                                                  const int32_t* __restrict__ lengths, T_* dcontent,
                                                                  T_* __restrict__ dpos, int B, int H, int T,
                                                                  int ldc, int ldp, int use_mask) {
  extern __shared__ float sh[];  // 4 waves * T floats
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ds = sh + (long)w * T;
  const int R = 2 * T - 1, R1 = R + 1;
  const long nrows = (long)B * H * T;
  const long w0 = (long)blockIdx.x * (blockDim.x >> 6) + w;
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long row = w0; row < nrows; row += nw) {
    const int i = (int)(row % T);
    const int b = (int)(row / ((long)H * T));
    const int len = lengths ? min(lengths[b], T) : T;
    T_* dc = dcontent + row * ldc;
    T_* dp = dpos + row * ldp;
    if (use_mask && i >= len) {  // constant scores: zero gradient
      for (int j = lane; j < T; j += 64) Num<T_>::st(dc + j, 0.f);
      for (int r = lane; r < ldp; r += 64) Num<T_>::st(dp + r, 0.f);
      continue;
    }
    const T_* p = probs + row * ldc;
    const T_* d = dprobs + row * ldc;
    float pv[MAXJ], dv[MAXJ];
    float dot = 0.f;
    int n = 0;
    for (int j = lane; j < T; j += 64, ++n) {
      pv[n] = Num<T_>::ld(p + j);
      dv[n] = Num<T_>::ld(d + j);
      dot += pv[n] * dv[n];
    }
    dot = wave_sum(dot);
    float left = 0.f;  // gradient mass that went through the bias column
    n = 0;
    for (int j = lane; j < T; j += 64, ++n) {
      const float g = pv[n] * (dv[n] - dot);
      ds[j] = g;
      Num<T_>::st(dc + j, g);
      if (pos_col(i, j, T, R, len) == R) left += g;
    }
    left = wave_sum(left);
    // make this wave's LDS writes visible to its own lanes (wave-synchronous: s_waitcnt via fence)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // inverse gather: column rr <- j  where rr = (T-1-i+j + T-len) mod R and r = T-1-i+j < 2len-1
    const int shift = T - len;
    for (int rr = lane; rr < R; rr += 64) {
      int r = rr - shift;
      if (r < 0) r += R;
      const int j = r - (T - 1 - i);
      float g = 0.f;
      if (r < 2 * len - 1 && j >= 0 && j < T) g = ds[j];
      Num<T_>::st(dp + rr, g);
    }
    if (lane == 0) Num<T_>::st(dp + R, left);
    for (int r = R1 + lane; r < ldp; r += 64) Num<T_>::st(dp + r, 0.f);  // keep the row padding finite
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

inline int rows_grid(long rows) { return (int)std::max<long>(1, std::min<long>((rows + 3) / 4, 256L * 16)); }

}  // namespace

extern "C" int tfasr_relattn_softmax_fwd(const void* content, const void* pos, const int32_t* lengths, void* probs,
                                         int B, int H, int T, int ldc, int ldp, int use_mask, int dtype, void* stream_) {
  return tfasr_relattn_softmax_fwd_streaming(content, pos, lengths, probs, B, H, T, ldc, ldp, use_mask, 0, 0, dtype, stream_);
}

extern "C" int tfasr_relattn_softmax_fwd_streaming(const void* content, const void* pos, const int32_t* lengths, void* probs,
                                                   int B, int H, int T, int ldc, int ldp, int use_mask, int chunk, int hist, int dtype,
                                                   void* stream_) {
  if (!content || !pos || !probs || B <= 0 || H <= 0 || T <= 0 || T > 64 * MAXJ || ldc < T || ldp < 2 * T) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = rows_grid((long)B * H * T);
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(relattn_softmax_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)content,
                       (const float*)pos, lengths, (float*)probs, B, H, T, ldc, ldp, use_mask, chunk, hist);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(relattn_softmax_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)content,
                       (const bf16_t*)pos, lengths, (bf16_t*)probs, B, H, T, ldc, ldp, use_mask, chunk, hist);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_relattn_softmax_bwd(const void* probs, const void* dprobs, const int32_t* lengths, void* dcontent,
                                         void* dpos, int B, int H, int T, int ldc, int ldp, int use_mask, int dtype, void* stream_) {
  if (!probs || !dprobs || !dcontent || !dpos || B <= 0 || H <= 0 || T <= 0 || T > 64 * MAXJ || ldc < T || ldp < 2 * T)
    return TFASR_STATUS_INVALID_VALUE;
  hipStream_t s = (hipStream_t)stream_;
  const int grid = rows_grid((long)B * H * T);
  const size_t shmem = 4 * (size_t)T * sizeof(float);
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(relattn_softmax_bwd_kernel<float>, dim3(grid), dim3(256), shmem, s, (const float*)probs,
                       (const float*)dprobs, lengths, (float*)dcontent, (float*)dpos, B, H, T, ldc, ldp, use_mask);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(relattn_softmax_bwd_kernel<bf16_t>, dim3(grid), dim3(256), shmem, s, (const bf16_t*)probs,
                       (const bf16_t*)dprobs, lengths, (bf16_t*)dcontent, (bf16_t*)dpos, B, H, T, ldc, ldp, use_mask);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
