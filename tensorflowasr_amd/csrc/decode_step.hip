// One greedy-search step of the transducer (Transducer.call_next, base_transducer.py:437-464, inside the loops of :496-712) in
// FOUR launches instead of ten: the step's products are [B <= 64, <= 1280] x [<= 1280, <= 2560] in f32 - far too skinny for the
// 128x128 MFMA tiles (20 workgroups walking 40 k-slabs each took ~43 us per product) - so they run as many-workgroup "skinny" dot
// products on the vector ALUs with the surrounding pointwise work fused in:
//   decode_lstm_kernel  : loop condition (decode_prepare's) + embedding lookup + x W + h R + b + LSTM cell     -> h_new, c_new
//   decode_joint_kernel : LayerNorm(h_new) + prediction projection + encoder frame gather + tanh              -> z [B, J]
//   decode_vocab_kernel : vocabulary projection                                                             -> logits [B, V] f32
//   tfasr_decode_update (decode.hip): log-softmax, arg-max, token / frame / state bookkeeping of the reference loops
// All arithmetic is f32 on the f32 master weights (the reference's CPU path), whatever the model's storage type.
// Thread mapping of the dot products: 256 threads = NB batch slots x KS k-slices; a thread accumulates NC output columns of its
// batch row over its k-slice, the slices meet in LDS.  The weight rows a workgroup reads are its own NC columns only, so the
// matrices stay resident in the L2 of the XCD that owns those columns across the iterations of the search.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int MAXB = 64;
#ifdef TFASR_DECODE_TIMING
// probe builds (tools/decode_timing.sh): shader-clock stamps of thread 0 of the middle workgroup of each step kernel, [kernel][stamp];
// stamp 15 = the 100 MHz wall clock at entry, 14 = at exit
__device__ long long g_dec_t[4][16];
#define DEC_T(kern, k) { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) g_dec_t[kern][k] = (k) >= 14 ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_readcyclecounter(); }
#else
#define DEC_T(kern, k)
#endif
constexpr int NT = 1024;  // threads per workgroup: 16 k-slices at B = 64, 32 at B = 32 (short dependent load chains per thread)

// acc[c] += sum_{k in slice} x(k) * W[k * ldw + col0 + c]   for c < NC (NC % 4 == 0, col0 % 4 == 0 -> float4 loads)
template <int NC, typename XF>
__device__ __forceinline__ void slice_dot(float (&acc)[NC], XF x, const float* __restrict__ W, long ldw, int k0, int k1) {
#pragma unroll 8
  for (int k = k0; k < k1; ++k) {
    const float xv = x(k);
    const float* wr = W + (long)k * ldw;
#pragma unroll
    for (int c = 0; c < NC; c += 4) {
      const float4 wv = *reinterpret_cast<const float4*>(wr + c);
      acc[c] += xv * wv.x; acc[c + 1] += xv * wv.y; acc[c + 2] += xv * wv.z; acc[c + 3] += xv * wv.w;
    }
  }
}

struct Map { int nb, ks, b, s; };
__device__ __forceinline__ Map make_map(int B) {
  Map m;
  m.nb = B <= 8 ? 8 : (B <= 16 ? 16 : (B <= 32 ? 32 : 64));
  m.ks = NT / m.nb;
  m.b = threadIdx.x % m.nb;
  m.s = threadIdx.x / m.nb;
  return m;
}

// the while_loop condition of the reference loops, evaluated by every workgroup (B values); see decode_prepare_kernel
__device__ __forceinline__ bool loop_active(const int32_t* nframes, const int32_t* frame_idx, const int32_t* tok_idx, int B, int max_tokens,
                                            int mode) {
  __shared__ int all_frames, all_tokens;
  if (threadIdx.x == 0) { all_frames = 1; all_tokens = 1; }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (mode == 0) {
      if (!(frame_idx[b] >= nframes[b] - 1)) atomicAnd(&all_frames, 0);
      if (!(tok_idx[b] >= max_tokens - 1)) atomicAnd(&all_tokens, 0);
    } else {
      if (frame_idx[b] < nframes[b]) atomicAnd(&all_frames, 0);
      atomicAnd(&all_tokens, 0);
    }
  }
  __syncthreads();
  return !(all_frames || all_tokens);
}

// ---- 1. embedding + LSTM cell for UPW hidden units per workgroup ----------------------------------------------------------------
template <int UPW>
__global__ __launch_bounds__(NT) void decode_lstm_kernel(
    const float* __restrict__ emb, const float* __restrict__ Wk, const float* __restrict__ Wr, const float* __restrict__ bias,
    const int32_t* __restrict__ prev_tok, const float* __restrict__ h, const float* __restrict__ c, const int32_t* __restrict__ nframes,
    const int32_t* __restrict__ frame_idx, const int32_t* __restrict__ tok_idx, int32_t* __restrict__ active, float* __restrict__ h_new,
    float* __restrict__ c_new, int B, int E, int P, int V, int max_tokens, int mode) {
  constexpr int NC = 4 * UPW;
  __shared__ float red[NT][NC + 1];
  const bool act = loop_active(nframes, frame_idx, tok_idx, B, max_tokens, mode);
  if (blockIdx.x == 0 && threadIdx.x == 0) active[0] = act ? 1 : 0;
  if (!act) return;
  const Map m = make_map(B);
  const int u0 = blockIdx.x * UPW;  // first hidden unit of this workgroup (P % UPW == 0 checked by the host)
  const int K = E + P;
  float acc[4][UPW];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < UPW; ++u) acc[q][u] = 0.f;
  if (m.b < B) {
    const int per = (K + m.ks - 1) / m.ks, k0 = m.s * per, k1 = min(K, k0 + per);
    const int tok = min(max(prev_tok[m.b], 0), V - 1);
    const float* er = emb + (long)tok * E;
    const float* hr = h + (long)m.b * P;
    auto run = [&](const float* xr, const float* Wm, int ka, int kb) {  // k in [ka, kb) of one operand (embedding row x Wk, h x Wr)
#pragma unroll 8
      for (int k = ka; k < kb; ++k) {
        const float xv = xr[k];
        const float* wr = Wm + (long)k * 4 * P + u0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (UPW == 4) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + (long)q * P);
            acc[q][0] += xv * wv.x; acc[q][1] += xv * wv.y; acc[q][2] += xv * wv.z; acc[q][3] += xv * wv.w;
          } else {
#pragma unroll
            for (int u = 0; u < UPW; ++u) acc[q][u] += xv * wr[(long)q * P + u];
          }
        }
      }
    };
    run(er, Wk, min(k0, E), min(k1, E));
    run(hr, Wr, max(k0, E) - E, max(k1, E) - E);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < UPW; ++u) red[threadIdx.x][q * UPW + u] = acc[q][u];
  __syncthreads();
  // one thread per (b, unit): sum the k-slices, add the bias, run the cell (gate order i, f, c, o; lstm.hip)
  for (int i = threadIdx.x; i < B * UPW; i += blockDim.x) {
    const int b = i / UPW, u = i % UPW;
    if (u0 + u >= P) continue;
    float z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sum = bias[q * P + u0 + u];
      for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + b][q * UPW + u];
      z[q] = sum;
    }
    const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), gg = tanh_fast(z[2]), og = sigmoidf_(z[3]);
    const float cn = fg * c[(long)b * P + u0 + u] + ig * gg;
    c_new[(long)b * P + u0 + u] = cn;
    h_new[(long)b * P + u0 + u] = og * tanh_fast(cn);
  }
}

// ---- 2. LayerNorm + prediction projection + tanh(enc + pred) for NC joint columns per workgroup -------------------------------
template <int NC>
__global__ __launch_bounds__(NT) void decode_joint_kernel(
    const float* __restrict__ h_new, const float* __restrict__ ln_g, const float* __restrict__ ln_b, const float* __restrict__ Wjp,
    const float* __restrict__ bjp, const float* __restrict__ encj, const int32_t* __restrict__ nframes,
    const int32_t* __restrict__ frame_idx, const int32_t* __restrict__ active, float* __restrict__ z, int B, int T, int P, int J,
    float ln_eps) {
  if (!active[0]) return;
  __shared__ float red[NT][NC + 1];
  __shared__ float s_mean[MAXB], s_rstd[MAXB];
  const Map m = make_map(B);
  const int j0 = blockIdx.x * NC;
  const int per = (P + m.ks - 1) / m.ks, k0 = m.s * per, k1 = min(P, k0 + per);
  const float* hr = h_new + (long)min(m.b, B - 1) * P;
  if (ln_g) {  // keras LayerNormalization (eps 1e-3): two passes over the row for the moments
    float part = 0.f;
    if (m.b < B) for (int k = k0; k < k1; ++k) part += hr[k];
    red[threadIdx.x][0] = part;
    __syncthreads();
    if (threadIdx.x < B) {
      float sum = 0.f;
      for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + threadIdx.x][0];
      s_mean[threadIdx.x] = sum / P;
    }
    __syncthreads();
    const float mu = s_mean[min(m.b, B - 1)];
    part = 0.f;
    if (m.b < B) for (int k = k0; k < k1; ++k) { const float dlt = hr[k] - mu; part += dlt * dlt; }
    __syncthreads();
    red[threadIdx.x][0] = part;
    __syncthreads();
    if (threadIdx.x < B) {
      float sum = 0.f;
      for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + threadIdx.x][0];
      s_rstd[threadIdx.x] = rsqrtf(sum / P + ln_eps);
    }
    __syncthreads();
  }
  float acc[NC];
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) acc[cidx] = 0.f;
  if (m.b < B) {
    if (ln_g) {
      const float mu = s_mean[m.b], rs = s_rstd[m.b];
      slice_dot<NC>(acc, [&](int k) { return (hr[k] - mu) * rs * ln_g[k] + ln_b[k]; }, Wjp + j0, J, k0, k1);
    } else {
      slice_dot<NC>(acc, [&](int k) { return hr[k]; }, Wjp + j0, J, k0, k1);
    }
  }
  __syncthreads();
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) red[threadIdx.x][cidx] = acc[cidx];
  __syncthreads();
  for (int i = threadIdx.x; i < B * NC; i += blockDim.x) {
    const int b = i / NC, cidx = i % NC;
    if (j0 + cidx >= J) continue;
    float sum = bjp[j0 + cidx];
    for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + b][cidx];
    int f = min(frame_idx[b], nframes[b] - 1);
    f = max(min(f, T - 1), 0);
    z[(long)b * J + j0 + cidx] = tanhf(encj[((long)b * T + f) * J + j0 + cidx] + sum);  // TransducerJointMerge add + tanh (:199-207,291)
  }
}

// ---- 3. vocabulary projection for NC classes per workgroup -----------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(NT) void decode_vocab_kernel(const float* __restrict__ z, const float* __restrict__ Wv, const float* __restrict__ bv,
                                                           const int32_t* __restrict__ active, float* __restrict__ logits, int B, int J, int V) {
  if (!active[0]) return;
  __shared__ float red[NT][NC + 1];
  const Map m = make_map(B);
  const int v0 = blockIdx.x * NC;
  float acc[NC];
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) acc[cidx] = 0.f;
  if (m.b < B) {
    const int per = (J + m.ks - 1) / m.ks, k0 = m.s * per, k1 = min(J, k0 + per);
    const float* zr = z + (long)m.b * J;
    slice_dot<NC>(acc, [&](int k) { return zr[k]; }, Wv + v0, V, k0, k1);
  }
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) red[threadIdx.x][cidx] = acc[cidx];
  __syncthreads();
  for (int i = threadIdx.x; i < B * NC; i += blockDim.x) {
    const int b = i / NC, cidx = i % NC;
    if (v0 + cidx >= V) continue;
    float sum = bv[v0 + cidx];
    for (int sl = 0; sl < m.ks; ++sl) sum += red[sl * m.nb + b][cidx];
    logits[(long)b * V + v0 + cidx] = sum;
  }
}


// =================================================================================================================================
// Round 4: the same three products on the MATRIX cores in exact f32 (v_mfma_f32_16x16x4_f32).  The vector-ALU kernels above spent their
// time in the operand path, not the arithmetic: a thread = (batch row, k slice) loads one activation float per k from 32 DIFFERENT rows
// (64 cache lines per wave instruction) and every weight 16-byte piece 32 times (once per batch lane): 30 / 34 / 12.5 us per launch
// for 13 MB of weights, ~83 us per search iteration; these kernels on the packed weights: 10.1 / 11.1 / 7.4 us (rocprofv3,
// profiles/r04_decode_kernel_stats.md).  A step is three skinny GEMMs
// [B <= 64, K] x [K, 16 columns per workgroup]: as MFMA tiles every lane loads 16 contiguous bytes of its OWN activation row (A: row =
// lane & 15, four consecutive k per lane) and each weight element is loaded once per workgroup (B: column = lane & 15), the k range is
// split over the four waves of a workgroup and the partial tiles meet in LDS.  Products are exact f32, the sums run in another order
// than the sequential loops above (last-ulp differences in the logits; the token decisions are checked against the same goldens).
// One search iteration = the same four launches; a persistent one-launch search was NOT built: on this chip a device-wide barrier costs
// 4-7 us and a dependent kernel boundary 1.5-1.9 us (MI355X_MICROARCH.md, barrier-xcd / boundary rows) - four barriers per token would
// be slower than four launches.
// k assignment inside a group of 16: MFMA j (0..3) multiplies k = 4 g + j of the group (g = lane >> 4), so a lane's four A values are one
// float4 and its four B values four rows of the weight matrix at the same column.
template <int MT>
__device__ __forceinline__ void mfma_group(float4_t (&acc)[MT], const float4 (&a)[MT], const float (&b)[4]) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[0], acc[m], 0, 0, 0);
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b[1], acc[m], 0, 0, 0);
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b[2], acc[m], 0, 0, 0);
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b[3], acc[m], 0, 0, 0);
  }
}
// These launches are LATENCY-bound: a kernel of the search is a handful of dependent memory round trips (~1-2 us each at this occupancy),
// so every load that does not depend on another load is issued up front, in ONE batch per wave: 16 waves per workgroup, each with at most
// GPW groups of 16 k (operands of a whole launch in flight at once), partial tiles meet in LDS.
constexpr int NWV = 16, GPW = 5;
// partial C tiles of the 16 waves -> LDS part[w][MT*16][16]; returns after the barrier
template <int MT>
__device__ __forceinline__ void stash_partials(float* part, const float4_t (&acc)[MT], int w, int r, int g) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[((w * MT + m) * 16 + g * 4 + e) * 16 + r] = acc[m][e];
  __syncthreads();
}
template <int MT>
__device__ __forceinline__ float sum_partials(const float* part, int b, int cidx) {
  float sum = 0.f;
#pragma unroll
  for (int ww = 0; ww < NWV; ++ww) sum += part[((ww * MT + (b >> 4)) * 16 + (b & 15)) * 16 + cidx];
  return sum;
}

// ---- weight re-layout for the kernels below (tfasr_decode_pack): tile t = the 16 output columns of workgroup t, inside a tile the
// operands of one wave instruction are contiguous: float index ((grp * 4 + j) * 4 + g) * 16 + r holds W[k = grp * 16 + 4 g + j][column r
// of the tile] - a wave's load of MFMA j's B operand is 256 contiguous bytes.  In the [K, N] row-major masters a workgroup's columns are
// 16-byte (LSTM: 4 units x 4 gates, gate stride P) or 64-byte pieces of a row, every piece in another cache line than the next k's: the
// fetches moved 4-8x the bytes they used and the 13 MB of weights cost 20+ us per iteration.
// Sections of `packed`: [recurrent kernel tiles (P/4) x P x 16] [joint tiles] [vocabulary tiles] [G = emb @ Wk, [V][P/4][16]] [Wk with
// its columns in G's order, [E][P/4][16]].  G is the input half of the LSTM pre-activation for EVERY token (keras: x @ kernel, a product
// of its own, lstm.py [ext]): the step then gathers 64 bytes per (row, workgroup) instead of multiplying the embedding row again, the
// k range of the step's product shrinks from E + P to P, and the token -> embedding dependent load leaves the critical path (it is
// consumed by the cell epilogue only).  Column order inside a G tile: u * 4 + q (unit-major: a cell thread's four gates = one float4).
struct PackDims { long nl, nj, nv, ng, nk; };
__host__ __device__ inline PackDims pack_dims(int E, int P, int J, int V) {
  PackDims d;
  d.nl = (long)(P / 4) * P * 16; d.nj = (long)((J + 15) / 16) * P * 16; d.nv = (long)((V + 15) / 16) * J * 16;
  d.ng = (long)V * 4 * P; d.nk = (long)E * 4 * P;
  return d;
}
__global__ __launch_bounds__(256) void decode_pack_kernel(const float* __restrict__ Wk, const float* __restrict__ Wr, const float* __restrict__ Wjp,
                                                          const float* __restrict__ Wv, float* __restrict__ out, int E, int P, int J, int V) {
  const PackDims dm = pack_dims(E, P, J, V);
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= dm.nl + dm.nj + dm.nv + dm.nk) return;
  if (idx >= dm.nl + dm.nj + dm.nv) {  // Wk, columns in tile order: [e][t][u * 4 + q] = Wk[e][q * P + 4 t + u]
    const long o = idx - (dm.nl + dm.nj + dm.nv);
    const int e = (int)(o / (4 * P)), cidx = (int)(o % (4 * P)), t = cidx >> 4, u = (cidx >> 2) & 3, q = cidx & 3;
    out[dm.nl + dm.nj + dm.nv + dm.ng + o] = Wk[(long)e * 4 * P + (long)q * P + t * 4 + u];
    return;
  }
  const int sec = idx < dm.nl ? 0 : (idx < dm.nl + dm.nj ? 1 : 2);
  const long o = idx - (sec == 0 ? 0 : (sec == 1 ? dm.nl : dm.nl + dm.nj));
  const int K = sec == 2 ? J : P;
  const long tile = o / ((long)K * 16);
  const int rem = (int)(o - tile * K * 16);
  const int grp = rem >> 8, j = (rem >> 6) & 3, g = (rem >> 4) & 3, r = rem & 15;
  const int k = grp * 16 + 4 * g + j;
  float v = 0.f;
  if (sec == 0) {
    v = Wr[(long)k * 4 * P + (long)(r >> 2) * P + tile * 4 + (r & 3)];  // column r of the tile = gate r / 4, unit 4 tile + r % 4
  } else if (sec == 1) {
    const long col = tile * 16 + r;
    if (col < J) v = Wjp[(long)k * J + col];
  } else {
    const long col = tile * 16 + r;
    if (col < V) v = Wv[(long)k * V + col];
  }
  out[idx] = v;
}

// the while_loop condition of the reference loops (see loop_active above) without LDS or barriers: every wave evaluates the B <= 64 rows
// itself, one row per lane
__device__ __forceinline__ bool loop_active_wave(int fi, int nf, int ti, int lane, int B, int max_tokens, int mode) {
  const bool in = lane < B;
  const bool frames_left = in && (mode == 0 ? !(fi >= nf - 1) : (fi < nf));
  const bool tokens_left = in && (mode == 0 ? !(ti >= max_tokens - 1) : true);
  const bool all_frames = __ballot(frames_left) == 0, all_tokens = __ballot(tokens_left) == 0;
  return !(all_frames || all_tokens);
}

// ---- 1'. embedding + LSTM cell for 4 hidden units (16 gate columns) per workgroup; side job of workgroups 0 .. B-1: the encoder
// frame of row b for the joint kernel (ecur [B, J], a dependent gather that kernel no longer waits for) ----
template <int MT>
__global__ __launch_bounds__(1024) void decode_lstm_mfma_kernel(
    const float* __restrict__ G, const float* __restrict__ pk, const float* __restrict__ bias, const int32_t* __restrict__ prev_tok,
    const float* __restrict__ h, const float* __restrict__ c, const float* __restrict__ encj, const int32_t* __restrict__ nframes,
    const int32_t* __restrict__ frame_idx, const int32_t* __restrict__ tok_idx, int32_t* __restrict__ active, float* __restrict__ h_new,
    float* __restrict__ c_new, float* __restrict__ ecur, int B, int T, int P, int J, int V, int max_tokens, int mode) {
  __shared__ float part[NWV * MT * 16 * 16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * 4;
  const float* tile = pk + (long)blockIdx.x * P * 16;
  DEC_T(0, 15) DEC_T(0, 0)
  const int ngr = P / 16, per = (ngr + NWV - 1) / NWV;  // per <= GPW (checked by the host)
  const int g0 = w * per, g1 = min(ngr, g0 + per);
  // ONE batch of independent loads: the previous tokens of the cell threads (the G rows hang off them), the loop condition's counters,
  // the weights of this wave's groups, the h rows, the cell operands
  const int eb = threadIdx.x >> 2, eu = threadIdx.x & 3;
  const bool cell = (int)threadIdx.x < B * 4;
  const int etok = cell ? prev_tok[eb] : 0;
  int fi = 0, nf = 1, ti = 0;
  if (lane < B) { fi = frame_idx[lane]; nf = nframes[lane]; ti = tok_idx[lane]; }
  float bw[GPW][4];
  float4 a[GPW][MT];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int gr = min(g0 + i, max(g1 - 1, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[i][j] = tile[(gr * 4 + j) * 64 + lane];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[i][m] = *reinterpret_cast<const float4*>(h + (long)min(m * 16 + r, B - 1) * P + gr * 16 + g * 4);
  }
  float cb[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
  if (cell) {
#pragma unroll
    for (int q = 0; q < 4; ++q) cb[q] = bias[q * P + u0 + eu];
    cprev = c[(long)eb * P + u0 + eu];
  }
  // second round trip, consumed by the cell epilogue only: x @ Wk of the previous token = one float4 of G (gates i, f, g, o of this unit)
  float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cell) xg = *reinterpret_cast<const float4*>(G + ((long)min(max(etok, 0), V - 1) * (P / 4) + blockIdx.x) * 16 + eu * 4);
  DEC_T(0, 1)
  const bool act = loop_active_wave(fi, nf, ti, lane, B, max_tokens, mode);
  DEC_T(0, 2)
  if (blockIdx.x == 0 && threadIdx.x == 0) active[0] = act ? 1 : 0;
  if (!act) return;
  // side job: ecur[b, :] = encj[b, min(frame, nframes - 1), :] for rows b = blockIdx.x, + gridDim.x, ... (loads here, stores at the end)
  constexpr int EC = 2;  // J <= 2048
  float ev[EC];
  const int eb0 = blockIdx.x;
  if (eb0 < B) {
    int f = min(frame_idx[eb0], nframes[eb0] - 1);
    f = max(min(f, T - 1), 0);
#pragma unroll
    for (int t = 0; t < EC; ++t) { const int j = threadIdx.x + 1024 * t; ev[t] = j < J ? encj[((long)eb0 * T + f) * J + j] : 0.f; }
  }
  float4_t acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GPW; ++i)
    if (g0 + i < g1) mfma_group<MT>(acc, a[i], bw[i]);
  DEC_T(0, 3)
  stash_partials<MT>(part, acc, w, r, g);
  DEC_T(0, 4)
  if (cell) {
    // keras LSTMCell: z = x @ kernel + h @ recurrent_kernel, then + bias
    const float zi = (xg.x + sum_partials<MT>(part, eb, 0 * 4 + eu)) + cb[0], zf = (xg.y + sum_partials<MT>(part, eb, 1 * 4 + eu)) + cb[1];
    const float zg = (xg.z + sum_partials<MT>(part, eb, 2 * 4 + eu)) + cb[2], zo = (xg.w + sum_partials<MT>(part, eb, 3 * 4 + eu)) + cb[3];
    const float ig = sigmoidf_(zi), fg = sigmoidf_(zf), gg = tanh_fast(zg), og = sigmoidf_(zo);
    const float cn = fg * cprev + ig * gg;
    c_new[(long)eb * P + u0 + eu] = cn;
    h_new[(long)eb * P + u0 + eu] = og * tanh_fast(cn);
  }
  if (eb0 < B) {
#pragma unroll
    for (int t = 0; t < EC; ++t) { const int j = threadIdx.x + 1024 * t; if (j < J) ecur[(long)eb0 * J + j] = ev[t]; }
    for (int bb = eb0 + gridDim.x; bb < B; bb += gridDim.x) {  // (fewer workgroups than rows: P < 4 B)
      int f = min(frame_idx[bb], nframes[bb] - 1);
      f = max(min(f, T - 1), 0);
      for (int j = threadIdx.x; j < J; j += 1024) ecur[(long)bb * J + j] = encj[((long)bb * T + f) * J + j];
    }
  }
  DEC_T(0, 5) DEC_T(0, 14)
}

// ---- 2'. LayerNorm + prediction projection + tanh(enc + pred) for 16 joint columns per workgroup ----
// The LayerNorm statistics come from the MFMA A fragments the waves hold anyway (a wave = its k slice of EVERY row): two LDS
// reductions (sum, then centred squares - keras' two passes) instead of a second copy of the rows in other registers, which doubled the
// bytes every CU pulls through its L1 (only J / 16 workgroups run, so the launch is bound by per-CU load bandwidth).  `ecur` [B, J] =
// the encoder frames the LSTM kernel gathered (may alias z: a thread reads its element before it writes it).
template <int MT>
__global__ __launch_bounds__(1024) void decode_joint_mfma_kernel(
    const float* __restrict__ h_new, const float* __restrict__ ln_g, const float* __restrict__ ln_b, const float* __restrict__ pk,
    const float* __restrict__ bjp, const float* ecur, const int32_t* __restrict__ active, float* z, int B, int P, int J, float ln_eps) {
  __shared__ float part[NWV * MT * 16 * 16];
  __shared__ float s_red[2][NWV][MAXB];
  __shared__ __attribute__((aligned(16))) float s_g[1024], s_b[1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int j0 = blockIdx.x * 16;
  const float* tile = pk + (long)blockIdx.x * P * 16;
  DEC_T(1, 15) DEC_T(1, 0)
  const int ngr = P / 16, per = (ngr + NWV - 1) / NWV;
  const int g0 = w * per, g1 = min(ngr, g0 + per);
  // one batch of loads: weights, raw prediction rows, LayerNorm coefficients (via LDS), the encoder frame, the bias
  float bw[GPW][4];
  float4 a[GPW][MT];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int gr = min(g0 + i, max(g1 - 1, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[i][j] = tile[(gr * 4 + j) * 64 + lane];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[i][m] = *reinterpret_cast<const float4*>(h_new + (long)min(m * 16 + r, B - 1) * P + gr * 16 + g * 4);
  }
  float gl = 1.f, bl = 0.f;
  if (ln_g && (int)threadIdx.x < P) { gl = ln_g[threadIdx.x]; bl = ln_b[threadIdx.x]; }
  const int eb = threadIdx.x >> 4, ec = threadIdx.x & 15;
  float ebias = 0.f, eenc = 0.f;
  const bool ethr = threadIdx.x < B * 16 && j0 + ec < J;
  if (ethr) { ebias = bjp[j0 + ec]; eenc = ecur[(long)eb * J + j0 + ec]; }
  DEC_T(1, 1)
  if (!active[0]) return;
  DEC_T(1, 2)
  float mu[MT], rs[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) { mu[m] = 0.f; rs[m] = 1.f; }
  if (ln_g) {  // keras LayerNormalization (eps 1e-3): mean, then the centred second moment
    if ((int)threadIdx.x < P) { s_g[threadIdx.x] = gl; s_b[threadIdx.x] = bl; }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < GPW; ++i)
        if (g0 + i < g1) ps += (a[i][m].x + a[i][m].y) + (a[i][m].z + a[i][m].w);
      ps = xor32_sum(xor16_sum(ps));
      if (g == 0) s_red[0][w][m * 16 + r] = ps;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float sum = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWV; ++ww) sum += s_red[0][ww][m * 16 + r];
      mu[m] = sum / P;
      float qs = 0.f;
#pragma unroll
      for (int i = 0; i < GPW; ++i)
        if (g0 + i < g1) {
          const float d0 = a[i][m].x - mu[m], d1 = a[i][m].y - mu[m], d2 = a[i][m].z - mu[m], d3 = a[i][m].w - mu[m];
          qs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      qs = xor32_sum(xor16_sum(qs));
      if (g == 0) s_red[1][w][m * 16 + r] = qs;
    }
    DEC_T(1, 3)
    __syncthreads();
    DEC_T(1, 4)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float sum = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWV; ++ww) sum += s_red[1][ww][m * 16 + r];
      rs[m] = rsqrtf(sum / P + ln_eps);
    }
  }
  float4_t acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    if (g0 + i >= g1) continue;
    if (ln_g) {
      const int k0 = (g0 + i) * 16 + g * 4;
      const float4 gv = *reinterpret_cast<const float4*>(s_g + k0), bv = *reinterpret_cast<const float4*>(s_b + k0);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 x = a[i][m];
        a[i][m] = make_float4((x.x - mu[m]) * rs[m] * gv.x + bv.x, (x.y - mu[m]) * rs[m] * gv.y + bv.y, (x.z - mu[m]) * rs[m] * gv.z + bv.z,
                              (x.w - mu[m]) * rs[m] * gv.w + bv.w);
      }
    }
    mfma_group<MT>(acc, a[i], bw[i]);
  }
  DEC_T(1, 5)
  stash_partials<MT>(part, acc, w, r, g);
  DEC_T(1, 6)
  if (ethr) z[(long)eb * J + j0 + ec] = tanhf(eenc + (ebias + sum_partials<MT>(part, eb, ec)));  // TransducerJointMerge add + tanh (:199-207,291)
  DEC_T(1, 7) DEC_T(1, 14)
}

// ---- 3'. vocabulary projection for 16 classes per workgroup ----
template <int MT>
__global__ __launch_bounds__(1024) void decode_vocab_mfma_kernel(const float* __restrict__ z, const float* __restrict__ pk, const float* __restrict__ bv,
                                                                 const int32_t* __restrict__ active, float* __restrict__ logits, int B, int J, int V) {
  __shared__ float part[NWV * MT * 16 * 16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int v0 = blockIdx.x * 16;
  const float* tile = pk + (long)blockIdx.x * J * 16;
  DEC_T(2, 15) DEC_T(2, 0)
  const int ngr = J / 16, per = (ngr + NWV - 1) / NWV;
  const int g0 = w * per, g1 = min(ngr, g0 + per);
  float bw[GPW][4];
  float4 a[GPW][MT];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int gr = min(g0 + i, max(g1 - 1, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[i][j] = tile[(gr * 4 + j) * 64 + lane];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[i][m] = *reinterpret_cast<const float4*>(z + (long)min(m * 16 + r, B - 1) * J + gr * 16 + g * 4);
  }
  const int eb = threadIdx.x >> 4, ec = threadIdx.x & 15;
  const bool ethr = threadIdx.x < B * 16 && v0 + ec < V;
  const float ebias = ethr ? bv[v0 + ec] : 0.f;
  DEC_T(2, 1)
  if (!active[0]) return;
  DEC_T(2, 2)
  float4_t acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GPW; ++i)
    if (g0 + i < g1) mfma_group<MT>(acc, a[i], bw[i]);
  DEC_T(2, 3)
  stash_partials<MT>(part, acc, w, r, g);
  DEC_T(2, 4)
  if (ethr) logits[(long)eb * V + v0 + ec] = ebias + sum_partials<MT>(part, eb, ec);
  DEC_T(2, 5) DEC_T(2, 14)
}

template <int MT>
int launch_decode_mfma(const float* packed, const float* lstm_b, const float* ln_g, const float* ln_b, const float* joint_pred_b,
                       const float* vocab_b, const float* encj, const int32_t* nframes, const int32_t* frame_idx, const int32_t* tok_idx,
                       const int32_t* prev_tok, const float* h, const float* c, int32_t* active, float* h_new, float* c_new, float* z,
                       float* logits, int B, int T, int E, int P, int J, int V, int max_tokens, int mode, float ln_eps, hipStream_t s) {
  const PackDims dm = pack_dims(E, P, J, V);
  const float* pj = packed + dm.nl;
  const float* pv = pj + dm.nj;
  const float* G = pv + dm.nv;
  TFASR_KLAUNCH(decode_lstm_mfma_kernel<MT>, dim3(P / 4), dim3(1024), 0, s, G, packed, lstm_b, prev_tok, h, c, encj, nframes, frame_idx, tok_idx,
                     active, h_new, c_new, z, B, T, P, J, V, max_tokens, mode);  // (z doubles as the gathered encoder frames until the joint kernel)
  TFASR_CHECK_LAUNCH();
  TFASR_KLAUNCH(decode_joint_mfma_kernel<MT>, dim3((J + 15) / 16), dim3(1024), 0, s, h_new, ln_g, ln_b, pj, joint_pred_b, z, active, z, B, P, J, ln_eps);
  TFASR_CHECK_LAUNCH();
  TFASR_KLAUNCH(decode_vocab_mfma_kernel<MT>, dim3((V + 15) / 16), dim3(1024), 0, s, z, pv, vocab_b, active, logits, B, J, V);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

// shapes the MFMA kernels take (packed weights): whole groups of 16 k, a wave's share of the k range within its register budget
bool mfma_shapes(int E, int P, int J) {
  return E > 0 && (P % 16) == 0 && (J % 16) == 0 && P <= 1024 && P <= 16 * NWV * GPW && J <= 16 * NWV * GPW;
}

}  // namespace

extern "C" size_t tfasr_decode_pack_floats(int E, int P, int J, int V) {
  if (E <= 0 || P <= 0 || J <= 0 || V <= 1 || !mfma_shapes(E, P, J)) return 0;
  const PackDims dm = pack_dims(E, P, J, V);
  return (size_t)(dm.nl + dm.nj + dm.nv + dm.ng + dm.nk);
}

extern "C" int tfasr_decode_pack(const float* emb, const float* lstm_k, const float* lstm_rk, const float* joint_pred_w, const float* vocab_w,
                                 float* packed, int E, int P, int J, int V, void* stream_) {
  if (!emb || !lstm_k || !lstm_rk || !joint_pred_w || !vocab_w || !packed) return TFASR_STATUS_INVALID_VALUE;
  const size_t n = tfasr_decode_pack_floats(E, P, J, V);
  if (n == 0) return TFASR_STATUS_UNSUPPORTED;
  const PackDims dm = pack_dims(E, P, J, V);
  const long nk = dm.nl + dm.nj + dm.nv + dm.nk;  // (threads: every section except G, which the product below fills)
  TFASR_KLAUNCH(decode_pack_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, lstm_k, lstm_rk, joint_pred_w, vocab_w,
                     packed, E, P, J, V);
  TFASR_CHECK_LAUNCH();
  // G = emb @ Wk (exact f32, columns in tile order): the input half of the LSTM pre-activation of every token
  tfasr_gemm_args ga;
  memset(&ga, 0, sizeof(ga));
  float* G = packed + dm.nl + dm.nj + dm.nv;
  ga.A = emb; ga.B = G + dm.ng; ga.D = G; ga.M = V; ga.N = 4 * P; ga.K = E; ga.lda = E; ga.ldb = 4 * P; ga.ldd = 4 * P;
  ga.nb1 = 1; ga.nb2 = 1; ga.alpha = 1.f; ga.beta = 0.f; ga.dtype = TFASR_F32; ga.split_k = 1;
  return tfasr_gemm(&ga, stream_);
}

extern "C" int tfasr_decode_step(const float* emb, const float* lstm_k, const float* lstm_rk, const float* lstm_b, const float* ln_g,
                                 const float* ln_b, const float* joint_pred_w, const float* joint_pred_b, const float* vocab_w,
                                 const float* vocab_b, const float* packed, const float* encj, const int32_t* nframes,
                                 const int32_t* frame_idx, const int32_t* tok_idx, const int32_t* prev_tok, const float* h, const float* c,
                                 int32_t* active, float* h_new, float* c_new, float* z, float* logits, int B, int T, int E, int P, int J, int V,
                                 int max_tokens, int mode, float ln_eps, void* stream_) {
  if (!emb || !lstm_k || !lstm_rk || !lstm_b || !joint_pred_w || !joint_pred_b || !vocab_w || !vocab_b || !encj || !nframes || !frame_idx ||
      !tok_idx || !prev_tok || !h || !c || !active || !h_new || !c_new || !z || !logits)
    return TFASR_STATUS_INVALID_VALUE;
  if (B <= 0 || B > MAXB || T <= 0 || E <= 0 || P <= 0 || J <= 0 || V <= 1) return TFASR_STATUS_INVALID_VALUE;
  if ((P % 4) || (J % 4) || (V % 8) || (((uintptr_t)lstm_k | (uintptr_t)lstm_rk | (uintptr_t)joint_pred_w | (uintptr_t)vocab_w) & 15))
    return TFASR_STATUS_UNSUPPORTED;  // float4 weight loads
  hipStream_t s = (hipStream_t)stream_;
  static const bool mfma_off = false;  // A/B probe: the vector-ALU kernels
  if (packed && !mfma_off && mfma_shapes(E, P, J) && ((((uintptr_t)h | (uintptr_t)h_new | (uintptr_t)z | (uintptr_t)packed) & 15) == 0)) {
    const int mt = (B + 15) / 16;
#define TFASR_DM(M) return launch_decode_mfma<M>(packed, lstm_b, ln_g, ln_b, joint_pred_b, vocab_b, encj, nframes, frame_idx, tok_idx, prev_tok, h, c, \
                                                 active, h_new, c_new, z, logits, B, T, E, P, J, V, max_tokens, mode, ln_eps, s)
    if (mt == 1) TFASR_DM(1);
    if (mt == 2) TFASR_DM(2);
    if (mt == 3) TFASR_DM(3);
    TFASR_DM(4);
#undef TFASR_DM
  }
  TFASR_KLAUNCH(decode_lstm_kernel<4>, dim3(P / 4), dim3(NT), 0, s, emb, lstm_k, lstm_rk, lstm_b, prev_tok, h, c, nframes, frame_idx, tok_idx,
                     active, h_new, c_new, B, E, P, V, max_tokens, mode);
  TFASR_CHECK_LAUNCH();
  TFASR_KLAUNCH(decode_joint_kernel<4>, dim3((J + 3) / 4), dim3(NT), 0, s, h_new, ln_g, ln_b, joint_pred_w, joint_pred_b, encj, nframes,
                     frame_idx, active, z, B, T, P, J, ln_eps);
  TFASR_CHECK_LAUNCH();
  TFASR_KLAUNCH(decode_vocab_kernel<8>, dim3((V + 7) / 8), dim3(NT), 0, s, z, vocab_w, vocab_b, active, logits, B, J, V);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

// `iters` search iterations (fused step + bookkeeping each) queued by ONE host call: the Python host costs ~100 us per iteration
// in ctypes argument marshalling alone, more than the four launches themselves.  Iterations after the loop condition turned false
// are no-ops on the device (active[0] == 0), so the caller checks `active` only every `iters` iterations.
extern "C" int tfasr_decode_steps(const float* emb, const float* lstm_k, const float* lstm_rk, const float* lstm_b, const float* ln_g,
                                  const float* ln_b, const float* joint_pred_w, const float* joint_pred_b, const float* vocab_w,
                                  const float* vocab_b, const float* packed, const float* encj, const int32_t* nframes, int32_t* frame_idx,
                                  int32_t* tok_idx, int32_t* prev_tok, float* h, float* c, int32_t* active, float* h_new, float* c_new, float* z,
                                  float* logits, int32_t* tokens, int32_t* per_frame, int B, int T, int E, int P, int J, int V, int max_tokens,
                                  int blank, int mode, int max_tokens_per_frame, float ln_eps, int iters, void* stream_) {
  if (iters <= 0 || !tokens) return TFASR_STATUS_INVALID_VALUE;
  for (int i = 0; i < iters; ++i) {
    int st = tfasr_decode_step(emb, lstm_k, lstm_rk, lstm_b, ln_g, ln_b, joint_pred_w, joint_pred_b, vocab_w, vocab_b, packed, encj, nframes,
                               frame_idx, tok_idx, prev_tok, h, c, active, h_new, c_new, z, logits, B, T, E, P, J, V, max_tokens, mode, ln_eps,
                               stream_);
    if (st != TFASR_STATUS_SUCCESS) return st;
    st = tfasr_decode_update(logits, active, nframes, frame_idx, prev_tok, tok_idx, tokens, per_frame, h_new, c_new, h, c, B, V, P, max_tokens, blank,
                             mode, max_tokens_per_frame, TFASR_F32, stream_);
    if (st != TFASR_STATUS_SUCCESS) return st;
  }
#ifdef TFASR_DECODE_TIMING
  {  // probe build (-DTFASR_DECODE_TIMING): always dumps
    long long h[4][16];
    if (hipStreamSynchronize((hipStream_t)stream_) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dec_t), sizeof(h)) == hipSuccess) {
      for (int k = 0; k < 3; ++k) {
        fprintf(stderr, "[decode_timing] kernel %d: wall entry %lld exit %lld (10 ns units; d = %lld) | clocks since entry:", k, h[k][15], h[k][14], h[k][14] - h[k][15]);
        for (int q = 1; q < 8; ++q) fprintf(stderr, " %lld", h[k][q] ? h[k][q] - h[k][0] : 0LL);
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "[decode_timing] wall gaps: lstm exit -> joint entry %lld, joint exit -> vocab entry %lld (10 ns units)\n", h[1][15] - h[0][14], h[2][15] - h[1][14]);
    }
  }
#endif
  return TFASR_STATUS_SUCCESS;
}
