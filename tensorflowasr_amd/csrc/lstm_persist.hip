// Persistent LSTM recurrence of the transducer prediction network for gfx950: ONE launch per direction instead of
// 2 x U1 (a recurrent GEMM + a cell kernel per step).
//
// Reference: keras.layers.LSTM(units=P, return_sequences=True, zero_output_for_mask=True) in TransducerPrediction
// (tensorflow_asr/models/transducer/base_transducer.py:71-85,123-132): gates i,f,c,o; recurrent kernel R [P,4P]; masked steps
// (t >= length) carry the state and emit zeros.  The input projection x @ W + b for all steps is one GEMM before this kernel.
//
// Forward.  Workgroup j owns 16 hidden units (all four gates of them): its slice of R (P x 64) stays in REGISTERS for the whole
// sequence (wave q = gate q: P/32 MFMA B fragments), its cell state in registers.  Per step: wait until every workgroup has
// published h_{t-1} (one monotonic device-scope counter, relaxed polling by one lane, then ONE agent-scope acquire), stage h_{t-1}
// [B, P] into LDS, z = h_{t-1} R_slice on the matrix cores, gate math for the owned units, store h_t WRITE-THROUGH (sc1), drain,
// and add 1 to the counter.  The sequence buffers the backward needs (gates, c, h) are the hand-off buffers themselves.
// Backward (BPTT).  Workgroup j owns the same 16 units: rows u of R (16 x 4P, k-contiguous) in registers (wave q = the k range of
// gate q), dh / dc carries in registers.  Per step: gather dz_{t+1} [B, 4P] (published by all workgroups) straight into MFMA A
// fragments, dhr = dz_{t+1} R^T for the owned units (4 partial k ranges summed through LDS), cell backward, publish dz_t.
// Every spin is bounded by the wall clock; a timeout raises a flag (second word of the caller's 64-byte `sync` record) that makes every
// workgroup leave - after overwriting what it owns of the remaining steps with NaN (no hang, and no silently wrong result: the step's loss /
// gradients are NaN); the caller can also check the word after a stream synchronisation.  The launch itself is refused unless the whole
// grid can be resident at once (`grid_fits`).
// Measured (B = 32, U1 = 111, P = 640): 8.7 us per forward step, 13.4 us per backward step - the hand-off itself (drain of the
// write-through stores, 40 arrivals on one counter, poll, acquire, re-load of the exchanged vector), not the loads' issue order:
// batching every load of a step in front of its first use changed nothing.
#include "common.h"

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;
#define AGENT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int PW = 16;          // hidden units per workgroup
constexpr int MAXKS = 32;       // k-steps of 32 per wave: P <= 1024
constexpr long long SPIN_TICKS = 100000000LL;  // 1 s of the 100 MHz wall clock

struct Sync { unsigned int count; unsigned int abort; unsigned int pad[14]; };

// wait until *count >= target (one lane polls, relaxed); false on timeout / abort
__device__ __forceinline__ bool wait_count(Sync* s, unsigned target) {
  bool ok = true;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    gu32* cnt = (gu32*)&s->count;
    gu32* abt = (gu32*)&s->abort;
    while (__hip_atomic_load(cnt, AGENT_RLX) < target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0) {
        if (__hip_atomic_load(abt, AGENT_RLX) != 0u || wall_clock64() - t0 > SPIN_TICKS) {
          __hip_atomic_store(abt, 1u, AGENT_RLX);
          ok = false;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ONE buffer_inv sc1 after the match: drops this CU's stale L1 lines
  }
  return __syncthreads_and(ok ? 1 : 0) != 0;
}

// every storing wave has drained its write-through stores -> one arrival
__device__ __forceinline__ void arrive(Sync* s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32*)&s->count, 1u, AGENT_RLX);
}

// write-through (sc1) store of two adjacent bf16 values: visible at device scope once the storing wave has drained its stores
__device__ __forceinline__ void store_wt2(bf16_t* p, float lo, float hi) {
  __hip_atomic_store((gu32*)p, pack2_bf16(lo, hi), AGENT_RLX);
}
__device__ __forceinline__ void ld2(const bf16_t* p, float& lo, float& hi) {
  const unsigned v = *reinterpret_cast<const unsigned*>(p);
  lo = __uint_as_float(v << 16);
  hi = __uint_as_float(v & 0xffff0000u);
}

// A timed-out / aborted launch must not pass for a result: the leaving workgroup overwrites what it owns of the REMAINING
// steps with NaN, so the loss / the gradients of that step are NaN (what the reference's TerminateOnNaN watches) - no host read needed.
__device__ __forceinline__ void poison_rows(bf16_t* base, long row_stride, long step_stride, int t_lo, int t_hi, int B, int col0, int width) {
  for (int t = t_lo; t < t_hi; ++t)
    for (int i = threadIdx.x; i < B * width; i += blockDim.x)
      base[(long)(i / width) * row_stride + (long)t * step_stride + col0 + i % width] = (bf16_t)0x7FC0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int MT>  // MT = ceil(B / 16) batch row tiles
__global__ __launch_bounds__(256) void lstm_persist_fwd_kernel(
    const bf16_t* __restrict__ xg, const bf16_t* __restrict__ rk, const bf16_t* __restrict__ h0, long h0_stride_b,
    const float* __restrict__ c0, long c0_stride_b, const int32_t* __restrict__ lengths, bf16_t* __restrict__ gates,
    float* __restrict__ cseq, bf16_t* hseq, bf16_t* __restrict__ yseq, int B, int U1, int P, Sync* sync,
    const bf16_t* __restrict__ rk_t = nullptr, int t_begin = 0, int t_end = -1) {
  // rk_t / [t_begin, t_end): the SAME step body as ONE LAUNCH PER STEP (round 5; measured slower beside the encoder, its entry points removed in round 6: profiles/r05_ab/lstm_fused_step.txt): a range of one step never waits and
  // never arrives; the carried state comes from the sequence buffers (cseq / hseq hold the CARRIED state at every step, masked ones
  // included); rk_t = R transposed [4P, P] so that the wave's B fragments are twenty 16-byte loads instead of 160 strided 2-byte ones.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * PW;
  const int nwg = gridDim.x;
  const int Bp = MT * 16;
  if (t_end < 0) t_end = U1;
  const int ldh = P * 2 + 16;                       // LDS row stride of the staged h tile in bytes (+16: rows land 4 banks apart)
  char* sH = smem;                                  // [Bp][P] bf16
  float* sZ = reinterpret_cast<float*>(smem + (long)Bp * ldh);  // [4 gates][Bp][16] f32
  const int ks = P / 32;

  // this wave's gate: B fragments of R[:, q*P + u0 + n], k = hidden index (strided 2-byte loads, once)
  short8_t bw[MAXKS];
#pragma unroll
  for (int k = 0; k < MAXKS; ++k) {
    if (k < ks) {
      if (rk_t) bw[k] = *reinterpret_cast<const short8_t*>(rk_t + (long)(w * P + u0 + r) * P + k * 32 + g * 8);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bw[k][e] = (short)rk[(long)(k * 32 + g * 8 + e) * 4 * P + w * P + u0 + r];
      }
    }
  }
  // owned items: (batch row b, unit pair up) -> units u0 + 2 up, u0 + 2 up + 1; item = threadIdx.x + 256 * i
  constexpr int NIT = (MT * 16 * (PW / 2) + 255) / 256;
  float c_st[NIT][2], h_st[NIT][2];
  int len_b[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
    const bool in = it < Bp * (PW / 2) && b < B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (t_begin > 0) {  // the carried state of the step in front of the range
        c_st[i][j] = in ? cseq[((long)b * U1 + (t_begin - 1)) * P + u0 + u + j] : 0.f;
        h_st[i][j] = in ? bf16_to_f32(hseq[((long)b * U1 + (t_begin - 1)) * P + u0 + u + j]) : 0.f;
      } else {
        c_st[i][j] = (in && c0) ? c0[b * c0_stride_b + u0 + u + j] : 0.f;
        h_st[i][j] = (in && h0) ? bf16_to_f32(h0[b * h0_stride_b + u0 + u + j]) : 0.f;
      }
    }
    len_b[i] = in ? (lengths ? lengths[b] : U1) : 0;
  }

  for (int t = t_begin; t < t_end; ++t) {
    // input-projection terms of this step for the owned items (independent of the recurrence: in flight during the wait)
    float xz[NIT][4][2];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
      const bool in = it < Bp * (PW / 2) && b < B;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xz[i][q][0] = xz[i][q][1] = 0.f;
        if (in) ld2(xg + ((long)b * U1 + t) * 4 * P + q * P + u0 + u, xz[i][q][0], xz[i][q][1]);
      }
    }
    if (t > t_begin && !wait_count(sync, (unsigned)nwg * (unsigned)(t - t_begin))) {
      poison_rows(hseq, (long)U1 * P, P, t, U1, B, u0, PW);
      if (yseq) poison_rows(yseq, (long)U1 * P, P, t, U1, B, u0, PW);
      return;
    }
    // stage h_{t-1} [Bp][P] (rows >= B: zeros): every load of the tile is issued before the first LDS store, so the step pays ONE
    // memory round trip here, not one per 4 KiB
    const int chunks = P / 8;  // 16-B chunks per row
    constexpr int NST = MT * 16 * (32 * MAXKS / 8) / 256;  // 16-B chunks per thread at P = 1024
    uint4 hv[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int c = threadIdx.x + 256 * i;
      hv[i] = make_uint4(0, 0, 0, 0);
      if (c < Bp * chunks) {
        const int b = c / chunks, ch = c % chunks;
        if (b < B) {
          if (t > 0) hv[i] = *reinterpret_cast<const uint4*>(hseq + ((long)b * U1 + (t - 1)) * P + ch * 8);
          else if (h0) hv[i] = *reinterpret_cast<const uint4*>(h0 + b * h0_stride_b + ch * 8);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int c = threadIdx.x + 256 * i;
      if (c < Bp * chunks) *reinterpret_cast<uint4*>(sH + (long)(c / chunks) * ldh + (c % chunks) * 16) = hv[i];
    }
    __syncthreads();
    float4_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MAXKS; ++k) {
      if (k < ks) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const short8_t a = *reinterpret_cast<const short8_t*>(sH + (long)(m * 16 + r) * ldh + (k * 32 + g * 8) * 2);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw[k], acc[m], 0, 0, 0);
        }
      }
    }
    // C layout: row = g*4+e (batch row within the tile), col = r (unit)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) sZ[(w * Bp + m * 16 + g * 4 + e) * PW + r] = acc[m][e];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
      if (it < Bp * (PW / 2) && b < B) {
        const long so = ((long)b * U1 + t) * P + u0 + u;
        const long go = ((long)b * U1 + t) * 4 * P + u0 + u;
        float gt[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, y[2] = {0.f, 0.f};
        if (t < len_b[i]) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = xz[i][q][j] + sZ[(q * Bp + b) * PW + u + j];
            const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), gg = tanh_fast(z[2]), og = sigmoidf_(z[3]);
            const float c = fg * c_st[i][j] + ig * gg;
            const float h = og * tanh_fast(c);
            c_st[i][j] = c;
            h_st[i][j] = bf16_to_f32(f32_to_bf16(h));  // the carried state is what the sequence buffer holds
            gt[0][j] = ig; gt[1][j] = fg; gt[2][j] = gg; gt[3][j] = og;
            y[j] = h;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<unsigned*>(gates + go + q * P) = pack2_bf16(gt[q][0], gt[q][1]);
        if (yseq) *reinterpret_cast<unsigned*>(yseq + so) = pack2_bf16(y[0], y[1]);
        *reinterpret_cast<float2*>(cseq + so) = make_float2(c_st[i][0], c_st[i][1]);
        store_wt2(hseq + so, h_st[i][0], h_st[i][1]);
      }
    }
    if (t + 1 < t_end) arrive(sync);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void lstm_persist_bwd_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ rk, const bf16_t* __restrict__ gates, const float* __restrict__ cseq,
    const int32_t* __restrict__ lengths, bf16_t* dz, float* __restrict__ dh_carry, float* __restrict__ dc_carry, int B, int U1, int P,
    Sync* sync, int t_begin = 0, int t_end = -1) {
  // [t_begin, t_end): steps t_end - 1 down to t_begin (one launch per step: round 5, entry points removed in round 6); the carries live in dh_carry / dc_carry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * PW;
  const int nwg = gridDim.x;
  const int Bp = MT * 16;
  if (t_end < 0) t_end = U1;
  float* sZ = reinterpret_cast<float*>(smem);  // [4 k ranges][Bp][16] f32 partial products
  const int ks = P / 32;
  // B fragments: B[k][n] = R[u0 + n][w*P + k] (row u0 + n of R, k-contiguous: 16-B loads)
  short8_t bw[MAXKS];
#pragma unroll
  for (int k = 0; k < MAXKS; ++k)
    if (k < ks) bw[k] = *reinterpret_cast<const short8_t*>(rk + (long)(u0 + r) * 4 * P + w * P + k * 32 + g * 8);
  constexpr int NIT = (MT * 16 * (PW / 2) + 255) / 256;
  float dh_c[NIT][2], dc_c[NIT][2];
  int len_b[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
    const bool in = it < Bp * (PW / 2) && b < B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dh_c[i][j] = in ? dh_carry[(long)b * P + u0 + u + j] : 0.f;
      dc_c[i][j] = in ? dc_carry[(long)b * P + u0 + u + j] : 0.f;
    }
    len_b[i] = in ? (lengths ? lengths[b] : U1) : 0;
  }
  for (int t = t_end - 1; t >= t_begin; --t) {
    // operands of this step's cell backward (independent of the recurrence)
    float gq[NIT][4][2], cc[NIT][2], cp[NIT][2], dyv[NIT][2];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
      const bool in = it < Bp * (PW / 2) && b < B;
      const long so = ((long)b * U1 + t) * P + u0 + u;
      const long go = ((long)b * U1 + t) * 4 * P + u0 + u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gq[i][q][0] = gq[i][q][1] = 0.f;
        if (in) ld2(gates + go + q * P, gq[i][q][0], gq[i][q][1]);
      }
      cc[i][0] = cc[i][1] = cp[i][0] = cp[i][1] = dyv[i][0] = dyv[i][1] = 0.f;
      if (in) {
        const float2 c2 = *reinterpret_cast<const float2*>(cseq + so);
        cc[i][0] = c2.x; cc[i][1] = c2.y;
        if (t > 0) { const float2 p2 = *reinterpret_cast<const float2*>(cseq + so - P); cp[i][0] = p2.x; cp[i][1] = p2.y; }
        ld2(dy + so, dyv[i][0], dyv[i][1]);
      }
    }
    float dhr[NIT][2];
#pragma unroll
    for (int i = 0; i < NIT; ++i) dhr[i][0] = dhr[i][1] = 0.f;
    if (t < U1 - 1) {
      if (t < t_end - 1 && !wait_count(sync, (unsigned)nwg * (unsigned)(t_end - 1 - t))) {
        for (int q = 0; q < 4; ++q) poison_rows(dz, (long)U1 * 4 * P, 4 * P, 0, t + 1, B, q * P + u0, PW);
        return;
      }
      // dhr = dz_{t+1} @ R^T for the owned units: this wave's k range = gate w's columns of dz
      float4_t acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = float4_t{0.f, 0.f, 0.f, 0.f};
      // the A fragments come straight from global memory (dz_{t+1}, 16 bytes per lane): batches of KB k-steps, every load of a batch
      // issued before its first MFMA (one round trip per batch instead of a dependent load -> MFMA chain)
      constexpr int KB = 8;
#pragma unroll
      for (int k0 = 0; k0 < MAXKS; k0 += KB) {
        if (k0 < ks) {
          short8_t af[KB][MT];
#pragma unroll
          for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const int b = m * 16 + r;
              af[kk][m] = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
              if (k0 + kk < ks && b < B)
                af[kk][m] = *reinterpret_cast<const short8_t*>(dz + ((long)b * U1 + (t + 1)) * 4 * P + w * P + (k0 + kk) * 32 + g * 8);
            }
#pragma unroll
          for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              if (k0 + kk < ks) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][m], bw[k0 + kk], acc[m], 0, 0, 0);  // bw[k >= ks] was never loaded
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) sZ[(w * Bp + m * 16 + g * 4 + e) * PW + r] = acc[m][e];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
        if (it < Bp * (PW / 2)) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            dhr[i][j] = sZ[(0 * Bp + b) * PW + u + j] + sZ[(1 * Bp + b) * PW + u + j] + sZ[(2 * Bp + b) * PW + u + j] + sZ[(3 * Bp + b) * PW + u + j];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
      if (it < Bp * (PW / 2) && b < B) {
        const long go = ((long)b * U1 + t) * 4 * P + u0 + u;
        float d[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float dh_total = dh_c[i][j] + dhr[i][j];
          if (t < len_b[i]) {
            const float dh = dh_total + dyv[i][j];
            const float ig = gq[i][0][j], fg = gq[i][1][j], gg = gq[i][2][j], og = gq[i][3][j];
            const float tc = tanhf(cc[i][j]);
            const float dc = dc_c[i][j] + dh * og * (1.f - tc * tc);
            d[0][j] = dc * gg * ig * (1.f - ig);
            d[1][j] = dc * cp[i][j] * fg * (1.f - fg);
            d[2][j] = dc * ig * (1.f - gg * gg);
            d[3][j] = dh * tc * og * (1.f - og);
            dc_c[i][j] = dc * fg;
            dh_c[i][j] = 0.f;
          } else {
            dh_c[i][j] = dh_total;  // the state passes straight through a masked step
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) store_wt2(dz + go + q * P, d[q][0], d[q][1]);
      }
    }
    if (t > t_begin) {
      __syncthreads();  // sZ is reused by the next step
      arrive(sync);
    }
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = threadIdx.x + 256 * i, b = it / (PW / 2), u = (it % (PW / 2)) * 2;
    if (it < Bp * (PW / 2) && b < B) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { dh_carry[(long)b * P + u0 + u + j] = dh_c[i][j]; dc_carry[(long)b * P + u0 + u + j] = dc_c[i][j]; }
    }
  }
}

bool persist_ok(int B, int P, int dtype) { return dtype == TFASR_BF16 && B >= 1 && B <= 64 && P % 32 == 0 && P >= 32 && P <= 32 * MAXKS; }

// The workgroups of a launch wait for each other, so ALL of them must be resident at once (an ordinary launch gives no such promise):
// the launch is refused (UNSUPPORTED -> the caller's per-step kernels) unless resident-blocks-per-CU x CUs covers the grid.
// NECESSARY, NOT SUFFICIENT: the occupancy query knows nothing about kernels of OTHER streams that hold CUs when the launch starts (the
// encoder beside the prediction network); a workgroup that is not scheduled in time is caught by the kernels' wall-clock bound and abort
// flag (results poisoned, `sync` word 1 set), never by a hang.
// The answer depends on the kernel INSTANCE (the MT variants differ in registers, hence in blocks per CU), the device, the LDS size and the
// grid: a small per-thread table keyed on all four (the function-pointer type is the same for every MT, so a function-local static would
// be ONE entry shared by all instantiations - ADVICE r04).
template <typename KERNEL>
bool grid_fits(KERNEL kernel, int grid, size_t smem) {
  struct Entry { const void* fn; int dev; size_t smem; int grid; int ok; };
  static thread_local Entry cache[16];
  static thread_local int used = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const void* fn = reinterpret_cast<const void*>(kernel);
  for (int i = 0; i < used; ++i)
    if (cache[i].fn == fn && cache[i].dev == dev && cache[i].smem == smem && cache[i].grid == grid) return cache[i].ok != 0;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, smem) != hipSuccess) return false;
  const int ok = (long)per_cu * cus >= grid ? 1 : 0;
  Entry& e = cache[used < 16 ? used++ : (used = 1, 0)];  // (full: start over - 16 distinct launch shapes per thread do not occur in one model)
  e.fn = fn; e.dev = dev; e.smem = smem; e.grid = grid; e.ok = ok;
  return ok != 0;
}

// R [P, 4P] -> R^T [4P, P] (bf16), 32 x 32 tiles through LDS: once per sequence, for the per-step forward launches

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// C ABI.  `sync`: 64 bytes of device memory owned by the caller (tfasr_lstm_persist_sync_bytes), zeroed here (memset node in front of
// the launch); after a stream synchronisation word [1] != 0 means a spin timed out (results invalid).  UNSUPPORTED when the shape is
// outside the persistent kernels' range (bf16, B <= 64, P a multiple of 32, P <= 1024): use tfasr_lstm_seq_fwd / _bwd's step path.
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" size_t tfasr_lstm_persist_sync_bytes(void) { return sizeof(Sync); }

extern "C" int tfasr_lstm_persist_fwd(const void* xg, const void* rk, const void* h0, long h0_stride_b, const float* c0, long c0_stride_b,
                                      const int32_t* lengths, void* gates, float* cseq, void* hseq, void* yseq, int B, int U1, int P, int dtype,
                                      void* sync, void* stream_) {
  if (!xg || !rk || !gates || !cseq || !hseq || !sync || B <= 0 || U1 <= 0 || P <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (!persist_ok(B, P, dtype)) return TFASR_STATUS_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream_;
  const int MT = (B + 15) / 16, Bp = MT * 16;
  const size_t smem = (size_t)Bp * (P * 2 + 16) + (size_t)4 * Bp * PW * 4;
  const dim3 grid(P / PW);
#define TFASR_LAUNCH(M) TFASR_KLAUNCH(lstm_persist_fwd_kernel<M>, grid, dim3(256), smem, s, (const bf16_t*)xg, (const bf16_t*)rk, (const bf16_t*)h0, \
                                           h0_stride_b, c0, c0_stride_b, lengths, (bf16_t*)gates, cseq, (bf16_t*)hseq, (bf16_t*)yseq, B, U1, P, (Sync*)sync)
  const bool fits = MT == 1 ? grid_fits(lstm_persist_fwd_kernel<1>, grid.x, smem) : MT == 2 ? grid_fits(lstm_persist_fwd_kernel<2>, grid.x, smem)
                    : MT == 3 ? grid_fits(lstm_persist_fwd_kernel<3>, grid.x, smem) : grid_fits(lstm_persist_fwd_kernel<4>, grid.x, smem);
  if (!fits) { (void)hipGetLastError(); return TFASR_STATUS_UNSUPPORTED; }
  if (hipMemsetAsync(sync, 0, sizeof(Sync), s) != hipSuccess) return TFASR_STATUS_EXECUTION_FAILED;
  switch (MT) { case 1: TFASR_LAUNCH(1); break; case 2: TFASR_LAUNCH(2); break; case 3: TFASR_LAUNCH(3); break; default: TFASR_LAUNCH(4); }
#undef TFASR_LAUNCH
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_lstm_persist_bwd(const void* dy, const void* rk, const void* gates, const float* cseq, const int32_t* lengths, void* dz,
                                      float* dh_carry, float* dc_carry, int B, int U1, int P, int dtype, void* sync, void* stream_) {
  if (!dy || !rk || !gates || !cseq || !dz || !dh_carry || !dc_carry || !sync || B <= 0 || U1 <= 0 || P <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (!persist_ok(B, P, dtype)) return TFASR_STATUS_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream_;
  const int MT = (B + 15) / 16, Bp = MT * 16;
  const size_t smem = (size_t)4 * Bp * PW * 4;
  const dim3 grid(P / PW);
#define TFASR_LAUNCH(M) TFASR_KLAUNCH(lstm_persist_bwd_kernel<M>, grid, dim3(256), smem, s, (const bf16_t*)dy, (const bf16_t*)rk, (const bf16_t*)gates, cseq, \
                                           lengths, (bf16_t*)dz, dh_carry, dc_carry, B, U1, P, (Sync*)sync)
  const bool fits = MT == 1 ? grid_fits(lstm_persist_bwd_kernel<1>, grid.x, smem) : MT == 2 ? grid_fits(lstm_persist_bwd_kernel<2>, grid.x, smem)
                    : MT == 3 ? grid_fits(lstm_persist_bwd_kernel<3>, grid.x, smem) : grid_fits(lstm_persist_bwd_kernel<4>, grid.x, smem);
  if (!fits) { (void)hipGetLastError(); return TFASR_STATUS_UNSUPPORTED; }
  if (hipMemsetAsync(sync, 0, sizeof(Sync), s) != hipSuccess) return TFASR_STATUS_EXECUTION_FAILED;
  switch (MT) { case 1: TFASR_LAUNCH(1); break; case 2: TFASR_LAUNCH(2); break; case 3: TFASR_LAUNCH(3); break; default: TFASR_LAUNCH(4); }
#undef TFASR_LAUNCH
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
