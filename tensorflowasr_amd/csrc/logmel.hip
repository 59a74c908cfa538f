// Log-mel frontend for gfx950: pre-emphasis -> framing (pad_end) -> periodic Hann -> rFFT(512) -> |.|^2 ->
// mel filterbank -> ln(. + eps), one wave per frame, everything after the single read of the PCM stays in LDS.
//
// Reference: FeatureExtraction.call (feature_extraction.py:255-303): preemphasis_signal (:170-175),
// stft = tf.signal.stft(frame_length=400, frame_step=160, fft_length=512, pad_end=True), abs, square (:192-212),
// log_mel_spectrogram = matmul(S, linear_to_mel_weight_matrix(80, 257, 16000, 0, 8000)), log(S + 1e-6) (:214-231).
// The window, the dense mel matrix W[257,80] and its non-zero band per mel bin are built on the host exactly as
// the oracle builds them (HTK mel) and passed in, so the filterbank weights are bit-identical.
// HBM-bound: reads 4*B*N bytes, writes B*T0*F*s bytes; FFT is radix-2 in LDS (9 stages, 4 butterflies/lane/stage).
#include "common.h"
// NOTE: built with -ffp-contract=off (build.py FILE_FLAGS): the front end is held to <= 1e-4 abs in the log domain against
// the reference arithmetic (separate mul / add in the FFT butterflies); the rest of the library uses -ffp-contract=fast.
#include <algorithm>

namespace {

constexpr int NFFT = 512, NBIN = 257, LOG2N = 9;

template <typename T>
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ sig, int B, int N, float preemph,
                                                     const float* __restrict__ window, int frame_len, int frame_step,
                                                     const float* __restrict__ melw, const int32_t* __restrict__ band,
                                                     int F, float eps, T* __restrict__ out, int T0) {
  __shared__ float tw[NFFT / 2][2];
  __shared__ float buf[4][2][NFFT];
  __shared__ float pw[4][NBIN + 3];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < NFFT / 2; k += blockDim.x) {
    float s, c;
    sincospif(2.0f * (float)k / (float)NFFT, &s, &c);
    tw[k][0] = c;
    tw[k][1] = -s;
  }
  __syncthreads();
  float* re = buf[w][0];
  float* im = buf[w][1];
  const long nframes = (long)B * T0;
  const long w0 = (long)blockIdx.x * 4 + w, nw = (long)gridDim.x * 4;
  for (long fr = w0; fr < nframes; fr += nw) {
    const int b = (int)(fr / T0), ti = (int)(fr % T0);
    const float* x = sig + (long)b * N;
    const long g0 = (long)ti * frame_step;
    // load + pre-emphasis + window, bit-reversed placement
    for (int n = lane; n < NFFT; n += 64) {
      float v = 0.f;
      const long g = g0 + n;
      if (n < frame_len && g < N) {
        v = x[g];
        if (preemph > 0.f && g > 0) v -= preemph * x[g - 1];
        v *= window[n];
      }
      const int r = (int)(__brev((unsigned)n) >> (32 - LOG2N));
      re[r] = v;
      im[r] = 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int s = 1; s <= LOG2N; ++s) {
      const int len = 1 << s, half = len >> 1, tstep = NFFT >> s;
#pragma unroll
      for (int q0 = 0; q0 < NFFT / 2; q0 += 64) {
        const int q = q0 + lane;
        const int pos = q & (half - 1);
        const int i = ((q >> (s - 1)) << s) + pos, j = i + half;
        const float wr = tw[pos * tstep][0], wi = tw[pos * tstep][1];
        const float xr = re[j], xi = im[j];
        const float tr = wr * xr - wi * xi, tim = wr * xi + wi * xr;
        const float ur = re[i], ui = im[i];
        re[j] = ur - tr; im[j] = ui - tim;
        re[i] = ur + tr; im[i] = ui + tim;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    // power spectrum into re[0..256]  (abs then square: feature_extraction.py:204-209)
    for (int k = lane; k < NBIN; k += 64) {
      const float a = sqrtf(re[k] * re[k] + im[k] * im[k]);
      pw[w][k] = a * a;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int m = lane; m < F; m += 64) {
      const int lo = band[2 * m], hi = band[2 * m + 1];
      float acc = 0.f;
      for (int k = lo; k <= hi; ++k) acc += pw[w][k] * melw[k * F + m];
      Num<T>::st(out + fr * F + m, logf(acc + eps));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

}  // namespace

extern "C" int tfasr_logmel(const float* signal, int B, int N, float preemph, const float* window, int frame_len,
                            int frame_step, int nfft, const float* melw, const int32_t* band, int F, float eps, void* out,
                            int T0, int dtype, void* stream_) {
  if (!signal || !window || !melw || !band || !out || B <= 0 || N <= 0 || F <= 0 || T0 <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (nfft != NFFT || frame_len > NFFT || frame_len <= 0 || frame_step <= 0) return TFASR_STATUS_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream_;
  const long nframes = (long)B * T0;
  const int grid = (int)std::max<long>(1, std::min<long>((nframes + 3) / 4, 256L * 8));
  if (dtype == TFASR_F32)
    hipLaunchKernelGGL(logmel_kernel<float>, dim3(grid), dim3(256), 0, s, signal, B, N, preemph, window, frame_len,
                       frame_step, melw, band, F, eps, (float*)out, T0);
  else if (dtype == TFASR_BF16)
    hipLaunchKernelGGL(logmel_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, signal, B, N, preemph, window, frame_len,
                       frame_step, melw, band, F, eps, (bf16_t*)out, T0);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
