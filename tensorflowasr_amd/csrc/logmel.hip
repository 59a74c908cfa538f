// Log-mel frontend for gfx950: pre-emphasis -> framing (pad_end) -> periodic Hann -> rFFT(512) -> |.|^2 ->
// mel filterbank -> ln(. + eps), one wave per frame, everything after the single read of the PCM stays in LDS.
//
// Reference: FeatureExtraction.call (feature_extraction.py:255-303): preemphasis_signal (:170-175),
// stft = tf.signal.stft(frame_length=400, frame_step=160, fft_length=512, pad_end=True), abs, square (:192-212),
// log_mel_spectrogram = matmul(S, linear_to_mel_weight_matrix(80, 257, 16000, 0, 8000)), log(S + 1e-6) (:214-231).
// The window, the dense mel matrix W[257,80] and its non-zero band per mel bin are built on the host exactly as
// the oracle builds them (HTK mel) and passed in, so the filterbank weights are bit-identical.
// HBM-bound in principle (reads 4*B*N bytes, writes B*T0*F*s bytes).  The 512-point FFT is three radix-8 passes IN REGISTERS
// (512 = 8 x 8 x 8: a lane holds 8 complex points) with two conflict-free exchanges through LDS, in DOUBLE precision: f64 adds and
// fused multiply-adds issue at the plain f32 rate on this chip, and the spectrum then equals the oracle's f64 rFFT to ~1e-15 relative
// instead of carrying an f32 FFT's ~1e-7 x |frame| absolute error into the near-silent bins (2e-4 in the log domain there).
// The first version ran nine radix-2 f32 stages in LDS (455 LDS instructions per frame, half of their array cycles bank conflicts: the
// LDS was 62 % busy) and walked the mel bands with one dependent global load per tap: 292 us per Conformer-M batch.
#include "common.h"
// NOTE: built with -ffp-contract=off (build.py FILE_FLAGS): pre-emphasis, window and the mel sums are the reference's separate f32
// mul / add; the twiddle products use explicit fused multiply-adds.
#include <algorithm>

namespace {

constexpr int NFFT = 512, NBIN = 257;
constexpr int EX_ROW = 72;                 // exchange image: 8 rows of 64 points, pitch 72 (both exchanges conflict-free, see below)
constexpr int EX_PTS = 8 * EX_ROW;
constexpr int MELW_TAPS = 16;              // taps of a mel band kept in LDS, zero-padded (pitch 20 floats: the 16-byte reads of 16 consecutive
constexpr int TAP_PITCH = 20;              // bins fall into 16 different 16-byte bank groups); longer bands read their tail from global
constexpr int MAX_PASS = 4;                // mel bins are walked 64 at a time; passes beyond the fourth use all 16 LDS taps
constexpr int PW_LEN = NBIN + 4 * (NBIN >> 5) + 3;

struct cd { double r, i; };
__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ cd cmul(cd a, cd w) { return cd{__builtin_fma(a.r, w.r, -(a.i * w.i)), __builtin_fma(a.r, w.i, a.i * w.r)}; }
__device__ __forceinline__ cd mul_mi(cd a) { return cd{a.i, -a.r}; }  // a * (-i)
// 4-point DFT, outputs in natural order
__device__ __forceinline__ void dft4(cd t0, cd t1, cd t2, cd t3, cd& y0, cd& y1, cd& y2, cd& y3) {
  const cd a0 = t0 + t2, a1 = t0 - t2, b0 = t1 + t3, b1 = mul_mi(t1 - t3);
  y0 = a0 + b0; y1 = a1 + b1; y2 = a0 - b0; y3 = a1 - b1;
}
// 8-point DFT (w = exp(-2 pi i / 8)), decimation in frequency, in place, outputs in natural order
__device__ __forceinline__ void dft8(cd (&x)[8]) {
  constexpr double H = 0.70710678118654752440;
  const cd s0 = x[0] + x[4], s1 = x[1] + x[5], s2 = x[2] + x[6], s3 = x[3] + x[7];
  const cd e0 = x[0] - x[4], e1 = x[1] - x[5], e2 = x[2] - x[6], e3 = x[3] - x[7];
  const cd d0 = e0;
  const cd d1 = cd{(e1.r + e1.i) * H, (e1.i - e1.r) * H};    // e1 * (1 - i) / sqrt 2
  const cd d2 = mul_mi(e2);
  const cd d3 = cd{(e3.i - e3.r) * H, -(e3.r + e3.i) * H};   // e3 * (-1 - i) / sqrt 2
  dft4(s0, s1, s2, s3, x[0], x[2], x[4], x[6]);
  dft4(d0, d1, d2, d3, x[1], x[3], x[5], x[7]);
}
// Output stores the compiler does not book-keep.  With ordinary stores its wait for the NEXT frame's prefetched samples becomes
// s_waitcnt vmcnt(0) - "every pending store too" (the store sits in a loop it cannot count) - and each frame paid the ~2 us of a store
// acknowledgement.  vmcnt retires in order, so a wait counted without these stores can only wait longer than intended, never shorter.
__device__ __forceinline__ void st_untracked(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_untracked(bf16_t* p, float v) {
  const uint32_t b = f32_to_bf16(v);
  asm volatile("global_store_short %0, %1, off" ::"v"(p), "v"(b) : "memory");
}
__device__ __forceinline__ int pw_index(int f) { return f + 4 * (f >> 5); }  // spreads the stride-8 writes of the last pass over the banks

// One wave per frame.  Point n = l + 64 k sits in lane l, register k.  With f = k2 + 8 b2 + 64 a2 and l = a + 8 b:
//   pass 1: DFT over k            -> y1[l][k2] * w512^(l k2)      exchange: lane l writes row k2, lane (k2, a) reads b = 0..7
//   pass 2: DFT over b            -> y2[k2][a][b2] * w64^(a b2)   exchange: lane (k2, a) writes [k2][9 a + b2], lane (k2, b2) reads a = 0..7
//   pass 3: DFT over a            -> X[k2 + 8 b2 + 64 a2]
// Real and imaginary parts live in separate f64 images; exchange addresses in 8-byte units: write 72 k2 + l / read 72 k2 + a + 8 b,
// then write 72 k2 + 9 a + b2 / read 72 k2 + 9 a + b2: inside every 32-lane group each access hits 32 different bank pairs.
template <typename T>
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ sig, int B, int N, float preemph,
                                                     const float* __restrict__ window, int frame_len, int frame_step,
                                                     const float* __restrict__ melw, const int32_t* __restrict__ band,
                                                     int F, float eps, T* __restrict__ out, int T0) {
  __shared__ __attribute__((aligned(16))) double ex[4][2][EX_PTS];   // first the block's twiddle table (512 x 2 doubles), then the exchange images
  __shared__ float pw[4][PW_LEN];
  __shared__ int snq[MAX_PASS];                                      // tap quads the widest band of each 64-bin pass needs
  extern __shared__ __attribute__((aligned(16))) char dyn[];         // F x 20 band taps, F x 2 band limits
  float* taps = reinterpret_cast<float*>(dyn);
  int* sband = reinterpret_cast<int*>(taps + F * TAP_PITCH);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: frame arithmetic on the scalar unit
  if (threadIdx.x < MAX_PASS) snq[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < F * MELW_TAPS; i += blockDim.x) {
    const int m = i / MELW_TAPS, t = i % MELW_TAPS, k = band[2 * m] + t;
    taps[m * TAP_PITCH + t] = (k <= band[2 * m + 1]) ? melw[k * F + m] : 0.f;
  }
  for (int i = threadIdx.x; i < F; i += blockDim.x) {
    const int lo = band[2 * i], hi = band[2 * i + 1];
    sband[2 * i] = lo; sband[2 * i + 1] = hi;
    if ((i >> 6) < MAX_PASS) atomicMax(&snq[i >> 6], (min(max(hi - lo + 1, 0), MELW_TAPS) + 3) >> 2);
  }
  double* twt = &ex[0][0][0];
  for (int k = threadIdx.x; k < NFFT; k += blockDim.x) {
    double sn, cs;
    sincospi(2.0 * (double)k / (double)NFFT, &sn, &cs);
    twt[2 * k] = cs;
    twt[2 * k + 1] = -sn;
  }
  __syncthreads();
  // twiddles of this lane: w512^(l k2) after pass 1, w64^(a b2) = w512^(8 a b2) after pass 2
  cd tw1[8], tw2[8];
#pragma unroll
  for (int q = 1; q < 8; ++q) {
    const int e1 = lane * q, e2 = 8 * (lane & 7) * q;
    tw1[q] = cd{twt[2 * e1], twt[2 * e1 + 1]};
    tw2[q] = cd{twt[2 * e2], twt[2 * e2 + 1]};
  }
  float win[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) win[k] = (lane + 64 * k < frame_len) ? window[lane + 64 * k] : 0.f;
  __syncthreads();
  double* er = ex[w][0];
  double* ei = ex[w][1];
  float* pwr = pw[w];
  const int k2 = lane >> 3, lo3 = lane & 7;
  const int nframes = B * T0;  // < 2^31 (checked by the host entry): 32-bit frame arithmetic, a 64-bit division is ~100 instructions
  const int w0 = blockIdx.x * 4 + w, nw = gridDim.x * 4;
  // the NEXT frame's samples are requested before the current frame is transformed (one wave = ~30 frames: the 1-2 us of an HBM
  // round trip per frame was most of a frame's time at three waves per SIMD)
  float cur[8], prev[8];
  // raw loads from clamped addresses, no test on the loaded value here: the 16 requests leave back to back and nothing waits for them
  // until the next iteration masks them (a select right behind each load made the compiler wait for every pair in turn)
  // (sample indices are 32-bit: N < 2^31 - 512, checked by the host entry)
  int rows_ok = 0;  // bit k: point lane + 64 k lies inside the frame
#pragma unroll
  for (int k = 0; k < 8; ++k) rows_ok |= (lane + 64 * k < frame_len) ? (1 << k) : 0;
  auto fetch = [&](int fr, float (&c)[8], float (&pv)[8]) {
    const int b = fr / T0, ti = fr - b * T0;
    const float* x = sig + (long)b * N;
    const int g0 = ti * frame_step + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int g = g0 + 64 * k;
      const int gc = (((rows_ok >> k) & 1) && g < N) ? g : 0;
      c[k] = x[gc];
      pv[k] = x[max(gc - 1, 0)];
    }
  };
  // pre-emphasis + window in f32 as the reference does, widened to f64 for the transform
  auto prepare = [&](int fr, const float (&c)[8], const float (&pv)[8], double (&o)[8]) {
    const int g0 = (fr % T0) * frame_step + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int g = g0 + 64 * k;
      const bool ok = ((rows_ok >> k) & 1) && g < N;
      const float cc = ok ? c[k] : 0.f, pp = (ok && g > 0 && preemph > 0.f) ? pv[k] : 0.f;
      o[k] = (double)((cc - preemph * pp) * win[k]);
    }
  };
  double xin[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) xin[k] = 0.0;
  if (w0 < nframes) { fetch(w0, cur, prev); prepare(w0, cur, prev, xin); }
  for (int fr = w0; fr < nframes; fr += nw) {
    cd v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = cd{xin[k], 0.0};
    // (unconditional, on a clamped frame index: under `if (more)` the compiler sees a path fetch -> no prepare -> loop top and guards the
    // loop top with vmcnt waits - which are waits for the previous frame's stores)
    const int nfr = min(fr + nw, nframes - 1);
    fetch(nfr, cur, prev);
    dft8(v);
    er[lane] = v[0].r; ei[lane] = v[0].i;
#pragma unroll
    for (int q = 1; q < 8; ++q) { const cd t = cmul(v[q], tw1[q]); er[EX_ROW * q + lane] = t.r; ei[EX_ROW * q + lane] = t.i; }
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int a = EX_ROW * k2 + lo3 + 8 * q; v[q] = cd{er[a], ei[a]}; }
    dft8(v);
    er[EX_ROW * k2 + 9 * lo3] = v[0].r; ei[EX_ROW * k2 + 9 * lo3] = v[0].i;
#pragma unroll
    for (int q = 1; q < 8; ++q) { const cd t = cmul(v[q], tw2[q]); const int a = EX_ROW * k2 + 9 * lo3 + q; er[a] = t.r; ei[a] = t.i; }
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int a = EX_ROW * k2 + 9 * q + lo3; v[q] = cd{er[a], ei[a]}; }
    dft8(v);
    // power spectrum of bins 0..256 (|X|^2 in f64, stored f32: feature_extraction.py:204-209); this lane holds f = k2 + 8 lo3 + 64 a2
    const int fb = k2 + 8 * lo3;
#pragma unroll
    for (int a2 = 0; a2 < 4; ++a2) pwr[pw_index(fb + 64 * a2)] = (float)__builtin_fma(v[a2].r, v[a2].r, v[a2].i * v[a2].i);
    if (lane == 0) pwr[pw_index(NFFT / 2)] = (float)__builtin_fma(v[4].r, v[4].r, v[4].i * v[4].i);
    // the next frame's samples are consumed HERE, in front of this frame's stores: a wait for them at the top of the next iteration is a
    // wait for vmcnt(0), i.e. for the stores' acknowledgements too (~2 us per frame)
    prepare(nfr, cur, prev, xin);
    // pin the consumption here: volatile asm statements keep their order, so the stores below cannot move in front of it (left alone the
    // compiler sinks `prepare` to the loop latch, behind the stores)
    asm volatile("" : "+v"(xin[0]), "+v"(xin[1]), "+v"(xin[2]), "+v"(xin[3]), "+v"(xin[4]), "+v"(xin[5]), "+v"(xin[6]), "+v"(xin[7]));
    // mel bands: a pass over 64 bins reads its taps four at a time for as many quads as its widest band needs (uniform count, so the
    // reads of a quad leave together; the taps beyond a band's end are zero and the spectrum index is clamped).  A per-lane tap loop
    // was one LDS round trip per tap: 27 in a row per frame.
    for (int m = lane, pass = 0; m - lane < F; m += 64, ++pass) {
      const int nq = __builtin_amdgcn_readfirstlane(pass < MAX_PASS ? snq[pass] : MELW_TAPS / 4);
      const bool on = m < F;
      const int mm = on ? m : 0;
      const int lo = sband[2 * mm], hi = sband[2 * mm + 1];
      const float* tp = taps + mm * TAP_PITCH;
      float acc = 0.f;
      for (int q = 0; q < nq; ++q) {  // uniform count; a quad's five reads leave together
        const float4 t4 = *reinterpret_cast<const float4*>(tp + 4 * q);
        const float p0 = pwr[pw_index(min(lo + 4 * q, NBIN - 1))], p1 = pwr[pw_index(min(lo + 4 * q + 1, NBIN - 1))];
        const float p2 = pwr[pw_index(min(lo + 4 * q + 2, NBIN - 1))], p3 = pwr[pw_index(min(lo + 4 * q + 3, NBIN - 1))];
        acc += p0 * t4.x; acc += p1 * t4.y; acc += p2 * t4.z; acc += p3 * t4.w;
      }
      if (!on) continue;
      for (int k = lo + MELW_TAPS; k <= hi; ++k) acc += pwr[pw_index(k)] * melw[k * F + m];
      st_untracked(out + (long)fr * F + m, logf(acc + eps));
    }
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
  if (nframes > 0x7fffffffL - (1L << 20) || (long)N > 0x7fffffffL - 1024 || (long)T0 * frame_step > 0x7fffffffL - 1024) return TFASR_STATUS_UNSUPPORTED;
  const size_t dyn = (size_t)F * TAP_PITCH * sizeof(float) + (size_t)F * 2 * sizeof(int);
  if (dyn > 48 * 1024) return TFASR_STATUS_UNSUPPORTED;
  // one round of resident workgroups (each wave walks ~30 frames and keeps its twiddles / window in registers)
  static int occ[2] = {0, 0}, cus = 0;
  const int oi = dtype == TFASR_F32 ? 0 : 1;
  if (!occ[oi]) {
    int dev = 0, o = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (oi == 0) hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, logmel_kernel<float>, 256, dyn);
    else hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, logmel_kernel<bf16_t>, 256, dyn);
    occ[oi] = std::max(o, 1);
  }
  const int grid = (int)std::max<long>(1, std::min<long>((nframes + 3) / 4, (long)std::max(cus, 1) * occ[oi]));
  if (dtype == TFASR_F32)
    TFASR_KLAUNCH(logmel_kernel<float>, dim3(grid), dim3(256), dyn, s, signal, B, N, preemph, window, frame_len,
                       frame_step, melw, band, F, eps, (float*)out, T0);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(logmel_kernel<bf16_t>, dim3(grid), dim3(256), dyn, s, signal, B, N, preemph, window, frame_len,
                       frame_step, melw, band, F, eps, (bf16_t*)out, T0);
  else return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
