// RNN-T (transducer) loss + gradient for gfx950.
//
// Semantics follow tensorflow_asr/losses/impl/rnnt.py:181-278 (compute_rnnt_loss_and_grad_helper):
//   lp = log_softmax(logits); blank[t,u] = lp[t,u,0]; truth[t,u] = lp[t,u,labels[u]]      (:94-105,:211)
//   alpha[t,u] = lse(alpha[t-1,u]+blank[t-1,u], alpha[t,u-1]+truth[t,u-1])                (:108-137)
//   beta[Tl-1,Ul] = blank[Tl-1,Ul]; beta[t,u] = lse(beta[t+1,u]+blank[t,u], beta[t,u+1]+truth[t,u]) (:140-178)
//   loss = -beta[0,0]                                                                     (:277)
//   g_blank[t,u] = -exp(alpha[t,u]+beta[t+1,u]+blank[t,u]-beta00), t<Tl-1,u<=Ul; -1 at (Tl-1,Ul) (:233-248)
//   g_truth[t,u] = -exp(alpha[t,u]+beta[t,u+1]+truth[t,u]-beta00), t<Tl, u<Ul             (:251-254)
//   dlogits[v]   = g[v] - softmax[v]*sum_v g[v]                                           (:267-275)
// (the anti-diagonal packing of :81-91 is only a vectorisation; per-node arithmetic is the 2-term
//  log-sum-exp reproduced here: SURVEY.md A.4 item 10).
//
// Three kernels, all HBM-bound except the (latency-bound) lattice scan:
//   1. rnnt_logprobs : one wave per lattice node, single pass over V (online max/sum), writes
//                      lse / blank / truth  (reads B*T*U1*V logits once; skips padded nodes)
//   2. rnnt_lattice  : one workgroup per (utterance, {alpha|beta}); one thread per u walks the
//                      anti-diagonals, neighbours exchanged through a double-buffered LDS row
//   3. rnnt_grad     : one wave per lattice node, rewrites the row with the gradient
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

struct RowStat { float m, s; };

__device__ __forceinline__ void online_add(RowStat& st, float x) {
  if (x > st.m) { st.s = st.s * __expf(st.m - x) + 1.f; st.m = x; }
  else          { st.s += __expf(x - st.m); }
}
__device__ __forceinline__ void online_merge(RowStat& a, float m2, float s2) {
  const float m = fmaxf(a.m, m2);
  if (m == -INFINITY) { a.m = m; a.s = 0.f; return; }
  a.s = a.s * __expf(a.m - m) + s2 * __expf(m2 - m);
  a.m = m;
}

// lattice node addressing: dense [B,T,U1] (cell_off == nullptr) or PACKED (only the valid nodes t < Tl_b, u <= Ul_b of
// every utterance, utterance b starting at row cell_off[b], row-major (t,u) with U1_b = Ul_b+1 columns).
struct Cell { int b, t, u, Tl, Ul; bool valid; };
__device__ __forceinline__ Cell locate(long r, const long* __restrict__ cell_off, int B, int Tm, int U1,
                                       const int32_t* __restrict__ label_len, const int32_t* __restrict__ logit_len) {
  Cell c;
  if (cell_off) {
    int lo = 0, hi = B;  // largest b with cell_off[b] <= r
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cell_off[mid] <= r) lo = mid; else hi = mid; }
    c.b = lo;
    c.Tl = min(logit_len[lo], Tm); c.Ul = min(label_len[lo], U1 - 1);
    const long q = r - cell_off[lo];
    c.t = (int)(q / (c.Ul + 1)); c.u = (int)(q % (c.Ul + 1));
    c.valid = true;
  } else {
    c.u = (int)(r % U1); c.t = (int)((r / U1) % Tm); c.b = (int)(r / ((long)U1 * Tm));
    c.Tl = min(logit_len[c.b], Tm); c.Ul = min(label_len[c.b], U1 - 1);
    c.valid = (c.t < c.Tl && c.u <= c.Ul);
  }
  return c;
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rnnt_logprobs_kernel(
    const T* __restrict__ logits, const int32_t* __restrict__ labels, const int32_t* __restrict__ label_len,
    const int32_t* __restrict__ logit_len, const long* __restrict__ cell_off, long nrows, int B, int Tm, int U1, int V,
    float* __restrict__ lse, float* __restrict__ blank_lp, float* __restrict__ truth_lp) {
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const bool vec_ok = (V % 8 == 0);
  for (long r = wave0; r < nrows; r += nwaves) {
    const Cell cl = locate(r, cell_off, B, Tm, U1, label_len, logit_len);
    const int u = cl.u, b = cl.b;
    if (!cl.valid) continue;  // padded node: never read by the lattice / grad kernels
    const T* row = logits + r * V;
    RowStat st{-INFINITY, 0.f};
    int lab = -1;  // issued before the row so that it is not a second dependent round trip
    if (u < U1 - 1) lab = min(max(labels[(long)b * (U1 - 1) + u], 0), V - 1);
    float x_blank = 0.f, x_truth = 0.f;
    bool from_regs = false;
    if (vec_ok && V <= 2 * 64 * 8) {
      // the whole row sits in registers (<= 16 values per lane): row max first, then one exp2 per element
      float x[2][8];
      bool on[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int v0 = lane * 8 + c * 512;
        on[c] = v0 < V;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[c][i] = -INFINITY;
        if (on[c]) ld8(row + v0, x[c]);
      }
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, x[c][i]);
      m = wave_max(m);
      float sum = 0.f;
      const float m2 = m * 1.4426950408889634f;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += __builtin_amdgcn_exp2f(x[c][i] * 1.4426950408889634f - m2);  // exp2(-inf) = 0 for the padding
      st.m = m;
      st.s = wave_sum(sum);
      // blank = column 0 (lane 0, first value); truth = column lab: picked out of its owner's registers
      from_regs = true;
      x_blank = __shfl(x[0][0], 0, 64);
      if (lab >= 0) {
        const int q = lab & 7, c = lab >> 9;
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) mine = (i == q) ? (c ? x[1][i] : x[0][i]) : mine;
        x_truth = __shfl(mine, (lab & 511) >> 3, 64);
      }
    } else {
      if (vec_ok) {
        for (int v0 = lane * 8; v0 < V; v0 += 64 * 8) {
          float x[8];
          ld8(row + v0, x);
#pragma unroll
          for (int i = 0; i < 8; ++i) online_add(st, x[i]);
        }
      } else {
        for (int v = lane; v < V; v += 64) online_add(st, Num<T>::ld(row + v));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(st.m, o, 64), s2 = __shfl_xor(st.s, o, 64);
        online_merge(st, m2, s2);
      }
    }
    if (lane == 0) {
      const float l = st.m + logf(st.s);
      lse[r] = l;
      blank_lp[r] = (from_regs ? x_blank : Num<T>::ld(row)) - l;
      float tr = -INFINITY;
      if (lab >= 0) tr = (from_regs ? x_truth : Num<T>::ld(row + lab)) - l;
      truth_lp[r] = tr;
    }
  }
}

// Label id of every lattice row (for the projection GEMM's fused statistics epilogue): labels[b,u] for u < Ul_b, else -1.
__global__ __launch_bounds__(256) void rnnt_row_labels_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ label_len,
                                                              const int32_t* __restrict__ logit_len, const long* __restrict__ cell_off,
                                                              long nrows, int B, int Tm, int U1, int V, int32_t* __restrict__ row_label) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long)gridDim.x * blockDim.x) {
    const Cell cl = locate(r, cell_off, B, Tm, U1, label_len, logit_len);
    int lab = -1;
    if (cl.valid && cl.u < cl.Ul) lab = min(max(labels[(long)cl.b * (U1 - 1) + cl.u], 0), V - 1);
    row_label[r] = lab;
  }
}

// The statistics the GEMM epilogue left per 64-column slice -> lse / blank / truth log-probabilities of every lattice node
// (what rnnt_logprobs_kernel computes with a full pass over the logits).
__global__ __launch_bounds__(256) void rnnt_stats_finalize_kernel(const float2* __restrict__ part, int nparts, const float* __restrict__ pick,
                                                                  const int32_t* __restrict__ label_len, const int32_t* __restrict__ logit_len,
                                                                  const long* __restrict__ cell_off, long nrows, int B, int Tm, int U1,
                                                                  float* __restrict__ lse, float* __restrict__ blank_lp, float* __restrict__ truth_lp) {
  if (nparts == 16) {
    // V = 1000: a row's 16 (max, sum) pairs are 128 contiguous bytes.  SIXTEEN LANES PER ROW, one pair each: a wave reads 4 rows = 512
    // contiguous bytes per instruction, the merge is two DPP row reductions.  (One thread per row - 8 x 16-byte loads 128 bytes apart
    // across the lanes - touched 64 cache lines per instruction: 115 us and 0.7 GB fetched for 47 MB of statistics, rocprofv3 PMC.)
    const int li = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4, gs = ((long)gridDim.x * blockDim.x) >> 4;
    const long nr16 = (nrows + gs - 1) / gs * gs;  // every lane of a row group runs the same trip count (DPP needs the whole row active)
    for (long r = g0; r < nr16; r += gs) {
      const bool in = r < nrows;
      const float2 v = in ? part[r * 16 + li] : make_float2(-INFINITY, 0.f);
      const float mx = row16_max(v.x);
      const float sm = row16_sum(v.y > 0.f ? v.y * __expf(v.x - mx) : 0.f);  // an empty slice is (-inf, 0)
      if (in && li == 0) {
        const Cell cl = locate(r, cell_off, B, Tm, U1, label_len, logit_len);
        if (cl.valid) {
          const float l = mx + logf(sm);
          const float2 pk = *reinterpret_cast<const float2*>(pick + 2 * r);
          lse[r] = l;
          blank_lp[r] = pk.x - l;
          truth_lp[r] = (cl.u < cl.Ul) ? pk.y - l : -INFINITY;
        }
      }
    }
    return;
  }
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long)gridDim.x * blockDim.x) {
    const Cell cl = locate(r, cell_off, B, Tm, U1, label_len, logit_len);
    if (!cl.valid) continue;
    RowStat st{-INFINITY, 0.f};
    for (int c = 0; c < nparts; ++c) {
      const float2 v = part[r * nparts + c];
      online_merge(st, v.x, v.y);
    }
    const float l = st.m + logf(st.s);
    lse[r] = l;
    blank_lp[r] = pick[2 * r] - l;
    truth_lp[r] = (cl.u < cl.Ul) ? pick[2 * r + 1] - l : -INFINITY;
  }
}

// ---------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for every prefetch
// in flight; the lattice threads exchange data through LDS alone (global alpha / beta cells are written, never re-read here).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// grid = (B, 2): y==0 -> alpha (forward), y==1 -> beta (backward). blockDim.x >= U1 (multiple of 64).
__global__ void rnnt_lattice_kernel(const float* __restrict__ blank_lp, const float* __restrict__ truth_lp,
                                    const int32_t* __restrict__ label_len, const int32_t* __restrict__ logit_len,
                                    const long* __restrict__ cell_off, int Tm, int U1m, float* __restrict__ alpha,
                                    float* __restrict__ beta, float* __restrict__ costs) {
  extern __shared__ float sh[];  // 2 * blockDim.x floats
  const int b = blockIdx.x;
  const int u = threadIdx.x;
  const int nthr = blockDim.x;
  const int Tl = min(logit_len[b], Tm), Ul = min(label_len[b], U1m - 1);
  const int U1 = cell_off ? Ul + 1 : U1m;  // row stride of this utterance's lattice
  const long base = cell_off ? cell_off[b] : (long)b * Tm * U1m;
  const float* bl = blank_lp + base;
  const float* tr = truth_lp + base;
  float* buf0 = sh;
  float* buf1 = sh + nthr;
  buf0[u] = -INFINITY;
  buf1[u] = -INFINITY;
  __syncthreads();
  if (Tl <= 0) { if (u == 0 && blockIdx.y == 1) costs[b] = 0.f; return; }
  const int ndiag = Tl + Ul;  // diagonals n = t+u in [0, Tl-1+Ul]
  const bool ucol = (u <= Ul);

  // Each thread walks one lattice column, so its operands are two strided streams (stride U1 floats).  They were written by
  // another kernel on other XCDs and come from the memory side (~1 us): with a one-diagonal prefetch every diagonal paid that
  // latency (0.95 us / diagonal measured).  A register ring of PF diagonals keeps PF loads in flight per thread instead.
  constexpr int PF = 32;
  float rb[PF], rt[PF];
  if (blockIdx.y == 0) {
    float* al = alpha + base;
    float self = -INFINITY;  // alpha[t-1,u]
    // cell (t,u) on diagonal n = t+u needs blank[t-1,u] and truth[t,u-1]
    // unconditional loads from clamped (always valid) cells: the consumer only uses a value where the cell exists
    const int uc = min(u, Ul);
    auto fetch = [&](int n, float& fb, float& ft) {
      const int tc = min(max(n - u, 0), Tl - 1);
      fb = bl[(long)max(tc - 1, 0) * U1 + uc];
      ft = tr[(long)tc * U1 + max(uc - 1, 0)];
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) fetch(j, rb[j], rt[j]);
    for (int n0 = 0; n0 < ndiag; n0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int n = n0 + j;  // diagonals past the lattice (padding up to a multiple of PF) have no active cell
        float* cur = (n & 1) ? buf1 : buf0;
        const float* prev = (n & 1) ? buf0 : buf1;
        const int t = n - u;
        const bool act = ucol && t >= 0 && t < Tl;
        const float pb = rb[j], pt = rt[j];
        fetch(n + PF, rb[j], rt[j]);
        if (act) {
          float a;
          if (t == 0 && u == 0) a = 0.f;
          else {
            const float xb = (t > 0) ? self + pb : -INFINITY;
            const float xt = (u > 0) ? prev[u - 1] + pt : -INFINITY;
            a = logaddexpf_(xb, xt);
          }
          al[(long)t * U1 + u] = a;
          self = a;
          cur[u] = a;
        }
        lds_barrier();
      }
    }
  } else {
    float* be = beta + base;
    float self = -INFINITY;  // beta[t+1,u]
    // step it handles diagonal n = ndiag-1-it; cell (t,u) needs blank[t,u] and truth[t,u]
    const int uc = min(u, Ul);
    auto fetch = [&](int it, float& fb, float& ft) {
      const int tc = min(max((ndiag - 1 - it) - u, 0), Tl - 1);
      fb = bl[(long)tc * U1 + uc];
      ft = tr[(long)tc * U1 + uc];
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) fetch(j, rb[j], rt[j]);
    for (int i0 = 0; i0 < ndiag; i0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int it = i0 + j;
        const int n = ndiag - 1 - it;  // n < 0 (padding): no active cell
        float* cur = (it & 1) ? buf1 : buf0;
        const float* prev = (it & 1) ? buf0 : buf1;
        const int t = n - u;
        const bool act = ucol && t >= 0 && t < Tl;
        const float pb = rb[j], pt = rt[j];
        fetch(it + PF, rb[j], rt[j]);
        if (act) {
          float v;
          if (t == Tl - 1 && u == Ul) v = pb;
          else {
            const float xb = (t + 1 < Tl) ? self + pb : -INFINITY;
            const float xt = (u < Ul) ? prev[u + 1] + pt : -INFINITY;
            v = logaddexpf_(xb, xt);
          }
          be[(long)t * U1 + u] = v;
          self = v;
          cur[u] = v;
          if (t == 0 && u == 0) costs[b] = -v;
        }
        lds_barrier();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wave-synchronous lattice (round 4).  The workgroup kernel above pays an LDS round trip + a cross-wave barrier per anti-diagonal
// (0.34 us per diagonal, 293 us per Conformer-M batch with the rest of the chip idle: VERDICT r03 weak 11).  Here ONE wave owns a
// whole (utterance, direction): lane l holds E adjacent lattice columns u = l*E .. l*E+E-1 in registers, the neighbour column's value of
// the previous diagonal arrives through ONE cross-lane move per diagonal (DPP wave shift, no LDS, no barrier), the E cells of a lane on
// a diagonal are independent (their log-add-exp chains overlap), operands come from the same register ring of prefetched diagonals.
// Same two terms per node as the kernel above; FAST = false evaluates them with the same libm calls (bitwise equal alpha / beta).  U1 <= 64 * E.
// FAST: the 2-term log-sum-exp through the hardware exp2 / log2 (1 ulp each) as m + log(1 + exp(-|a - b|)) - the form
// tf.math.reduce_logsumexp itself evaluates (impl/rnnt.py:126) - instead of libm's expf / log1pf (~4x the instructions of the chain).
template <bool FAST>
__device__ __forceinline__ float lae_(float a, float b) {
  if constexpr (!FAST) return logaddexpf_(a, b);
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  const float e = __builtin_amdgcn_exp2f(-fabsf(a - b) * 1.44269504088896340736f);
  return m + __builtin_amdgcn_logf(1.0f + e) * 0.69314718055994530942f;
}

template <int E, bool FAST>
__global__ __launch_bounds__(64) void rnnt_lattice_wave_kernel(const float* __restrict__ blank_lp, const float* __restrict__ truth_lp,
                                                               const int32_t* __restrict__ label_len, const int32_t* __restrict__ logit_len,
                                                               const long* __restrict__ cell_off, int Tm, int U1m, float* __restrict__ alpha,
                                                               float* __restrict__ beta, float* __restrict__ costs) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int Tl = min(logit_len[b], Tm), Ul = min(label_len[b], U1m - 1);
  const int U1 = cell_off ? Ul + 1 : U1m;
  const long base = cell_off ? cell_off[b] : (long)b * Tm * U1m;
  const float* bl = blank_lp + base;
  const float* tr = truth_lp + base;
  if (Tl <= 0) { if (lane == 0 && blockIdx.y == 1) costs[b] = 0.f; return; }
  const int ndiag = Tl + Ul;
  constexpr int PF = E == 1 ? 32 : (E == 2 ? 16 : 8);
  float rb[PF][E], rt[PF][E];
  int ue[E], uc[E];
  bool ucol[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { ue[e] = lane * E + e; uc[e] = min(ue[e], Ul); ucol[e] = ue[e] <= Ul; }
  float self[E];
#pragma unroll
  for (int e = 0; e < E; ++e) self[e] = -INFINITY;
  if (blockIdx.y == 0) {
    float* al = alpha + base;
    auto fetch = [&](int n, int e, float& fb, float& ft) {
      const int tc = min(max(n - ue[e], 0), Tl - 1);
      fb = bl[max(tc - 1, 0) * U1 + uc[e]];  // (32-bit cell index: an utterance's lattice has < 2^31 cells)
      ft = tr[tc * U1 + max(uc[e] - 1, 0)];
    };
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int e = 0; e < E; ++e) fetch(j, e, rb[j][e], rt[j][e]);
    for (int n0 = 0; n0 < ndiag; n0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int n = n0 + j;
        // alpha[t, u-1] of column u = l*E: the previous diagonal's value of lane l-1's LAST column (lane 0: no left neighbour)
        float left[E];
        left[0] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(self[E - 1]), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
#pragma unroll
        for (int e = 1; e < E; ++e) left[e] = self[e - 1];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int t = n - ue[e];
          const bool act = ucol[e] && t >= 0 && t < Tl;
          const float pb = rb[j][e], pt = rt[j][e];
          fetch(n + PF, e, rb[j][e], rt[j][e]);
          if (act) {
            float a;
            if (t == 0 && ue[e] == 0) a = 0.f;
            else {
              const float xb = (t > 0) ? self[e] + pb : -INFINITY;
              const float xt = (ue[e] > 0) ? left[e] + pt : -INFINITY;
              a = lae_<FAST>(xb, xt);
            }
            al[t * U1 + ue[e]] = a;
            self[e] = a;
          }
        }
      }
    }
  } else {
    float* be = beta + base;
    auto fetch = [&](int it, int e, float& fb, float& ft) {
      const int tc = min(max((ndiag - 1 - it) - ue[e], 0), Tl - 1);
      fb = bl[tc * U1 + uc[e]];
      ft = tr[tc * U1 + uc[e]];
    };
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int e = 0; e < E; ++e) fetch(j, e, rb[j][e], rt[j][e]);
    for (int i0 = 0; i0 < ndiag; i0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int it = i0 + j;
        const int n = ndiag - 1 - it;
        // beta[t, u+1] of column u = l*E+E-1: the previous step's value of lane l+1's FIRST column (lane 63: none)
        float right[E];
        right[E - 1] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(self[0]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
#pragma unroll
        for (int e = 0; e + 1 < E; ++e) right[e] = self[e + 1];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int t = n - ue[e];
          const bool act = ucol[e] && t >= 0 && t < Tl;
          const float pb = rb[j][e], pt = rt[j][e];
          fetch(it + PF, e, rb[j][e], rt[j][e]);
          if (act) {
            float v;
            if (t == Tl - 1 && ue[e] == Ul) v = pb;
            else {
              const float xb = (t + 1 < Tl) ? self[e] + pb : -INFINITY;
              const float xt = (ue[e] < Ul) ? right[e] + pt : -INFINITY;
              v = lae_<FAST>(xb, xt);
            }
            be[t * U1 + ue[e]] = v;
            self[e] = v;
            if (t == 0 && ue[e] == 0) costs[b] = -v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(
    const T* logits, T* grads, const int32_t* __restrict__ labels,
    const int32_t* __restrict__ label_len, const int32_t* __restrict__ logit_len,
    const float* __restrict__ grad_scale, const long* __restrict__ cell_off, long nrows, int B, int Tm, int U1, int V,
    const float* __restrict__ lse, const float* __restrict__ blank_lp, const float* __restrict__ truth_lp,
    const float* __restrict__ alpha, const float* __restrict__ beta) {
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const bool vec_ok = (V % 8 == 0);
  for (long r = wave0; r < nrows; r += nwaves) {
    const Cell cl = locate(r, cell_off, B, Tm, U1, label_len, logit_len);
    const int u = cl.u, t = cl.t, b = cl.b, Tl = cl.Tl, Ul = cl.Ul;
    const int ustride = cell_off ? Ul + 1 : U1;
    const T* row = logits + r * V;
    T* out = grads + r * V;
    if (!cl.valid) {  // outside the lattice: zero gradient (impl/rnnt.py masks :218-224)
      if (vec_ok) {
        const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int v0 = lane * 8; v0 < V; v0 += 64 * 8) st8(out + v0, z);
      } else {
        for (int v = lane; v < V; v += 64) Num<T>::st(out + v, 0.f);
      }
      continue;
    }
    const long lb = cell_off ? cell_off[b] : (long)b * Tm * U1;
    const float b00 = beta[lb];
    const float a = alpha[r];
    float gb = 0.f, gt = 0.f;
    if (t < Tl - 1) gb = -__expf(a + beta[r + ustride] + blank_lp[r] - b00);
    else if (u == Ul) gb = -1.f;
    int lab = -1;
    if (u < Ul) {
      gt = -__expf(a + beta[r + 1] + truth_lp[r] - b00);
      lab = min(max(labels[(long)b * (U1 - 1) + u], 0), V - 1);
    }
    const float sc = grad_scale ? grad_scale[b] : 1.f;
    const float l = lse[r];
    const float ssum = gb + gt;
    if (vec_ok) {
      // g_v = (-softmax_v * (gb+gt) + [v==0] gb + [v==lab] gt) * sc, with exp((x-l)) as one exp2 and the two special columns
      // patched per 8-value chunk (they are rare) instead of two compares per element.  (Issuing the row loads ahead of the
      // lattice operands, the whole row held in registers, was measured SLOWER: 611 vs 505 us - lower occupancy.)
      const float l2 = l * 1.4426950408889634f, k1 = -ssum * sc, gbs = gb * sc, gts = gt * sc;
      for (int v0 = lane * 8; v0 < V; v0 += 64 * 8) {
        float x[8];
        ld8(row + v0, x);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i] * 1.4426950408889634f - l2) * k1;
        if (v0 == 0) x[0] += gbs;
        if (lab >= 0 && (lab >> 3) == (v0 >> 3)) {
          const int q = lab & 7;
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] += (i == q) ? gts : 0.f;
        }
        st8(out + v0, x);
      }
    } else {
      for (int v = lane; v < V; v += 64) {
        float g = -__expf(Num<T>::ld(row + v) - l) * ssum;
        if (v == 0) g += gb;
        if (v == lab) g += gt;
        Num<T>::st(out + v, g * sc);
      }
    }
  }
}

// Per-row coefficients of the loss gradient for the variant that re-computes the logit tile instead of reading it back
// (tfasr_gemm_args.rgrad_coef): (lse * log2(e), -(gb + gt) * scale, gb * scale, gt * scale); rows outside the lattice: zero gradient.
__global__ __launch_bounds__(256) void rnnt_coef_kernel(
    const int32_t* __restrict__ label_len, const int32_t* __restrict__ logit_len, const float* __restrict__ grad_scale,
    const long* __restrict__ cell_off, long nrows, int B, int Tm, int U1, const float* __restrict__ lse, const float* __restrict__ blank_lp,
    const float* __restrict__ truth_lp, const float* __restrict__ alpha, const float* __restrict__ beta, float4* __restrict__ coef,
    const int32_t* __restrict__ labels = nullptr, int V = 0, int32_t* __restrict__ rlab = nullptr) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long)gridDim.x * blockDim.x) {
    const Cell cl = locate(r, cell_off, B, Tm, U1, label_len, logit_len);
    float4 c = make_float4(3.0e38f, 0.f, 0.f, 0.f);  // outside the lattice: exp2(x - 3e38) = 0, times 0: a zero gradient whatever the logit
    if (rlab) rlab[r] = (cl.valid && cl.u < cl.Ul) ? min(max(labels[(long)cl.b * (U1 - 1) + cl.u], 0), V - 1) : -1;
    if (cl.valid) {
      const int u = cl.u, t = cl.t, b = cl.b, Tl = cl.Tl, Ul = cl.Ul;
      const int ustride = cell_off ? Ul + 1 : U1;
      const long lb = cell_off ? cell_off[b] : (long)b * Tm * U1;
      const float b00 = beta[lb], a = alpha[r];
      float gb = 0.f, gt = 0.f;
      if (t < Tl - 1) gb = -__expf(a + beta[r + ustride] + blank_lp[r] - b00);
      else if (u == Ul) gb = -1.f;
      if (u < Ul) gt = -__expf(a + beta[r + 1] + truth_lp[r] - b00);
      const float sc = grad_scale ? grad_scale[b] : 1.f;
      c = make_float4(lse[r] * 1.4426950408889634f, -(gb + gt) * sc, gb * sc, gt * sc);
    }
    coef[r] = c;
  }
}

// Loss gradient as a pure stream over the logits: g = exp2(x log2(e) - c.x) c.y, + c.z in column 0, + c.w in the label's column, with the
// four per-row coefficients (and the row's label) computed beforehand by rnnt_coef_kernel.  rnnt_grad_kernel does the same in one launch
// but opens every row with a dependent chain (cell lookup -> alpha / beta / lse / log-probabilities -> exponentials) before it can
// touch the row: 2.6 TB/s.  Here every load of a row pair is issued up front and only the 16-byte coefficient record is per-row state.
// Same arithmetic in the same order: bitwise the same gradients.
template <typename T>
__global__ __launch_bounds__(256) void rnnt_grad_apply_kernel(const T* __restrict__ logits, T* __restrict__ grads, const float4* __restrict__ coef,
                                                              const int32_t* __restrict__ rlab, long nrows, int V) {
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  constexpr int U = 2, MAXP = 2;  // rows in flight per wave, 512-column passes held in registers (V <= 1024)
  for (long r0 = wave0; r0 < nrows; r0 += U * nwaves) {
    float x[U][MAXP][8];
    float4 c[U];
    int lab[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + u * nwaves;
      c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      lab[u] = -1;
      if (r < nrows) {
        c[u] = coef[r];
        lab[u] = rlab[r];
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
          const int v0 = p * 512 + lane * 8;
          if (v0 < V) ld8(logits + r * V + v0, x[u][p]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + u * nwaves;
      if (r >= nrows) continue;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        const int v0 = p * 512 + lane * 8;
        if (v0 >= V) continue;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[u][p][i] = __builtin_amdgcn_exp2f(x[u][p][i] * 1.4426950408889634f - c[u].x) * c[u].y;
        if (v0 == 0) x[u][p][0] += c[u].z;
        if (lab[u] >= 0 && (lab[u] >> 3) == (v0 >> 3)) {
          const int q = lab[u] & 7;
#pragma unroll
          for (int i = 0; i < 8; ++i) x[u][p][i] += (i == q) ? c[u].w : 0.f;
        }
        st8(grads + r * V + v0, x[u][p]);
      }
    }
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int tfasr_rnnt_loss_workspace_size(int B, int T, int U1, int V, size_t* bytes) {
  if (!bytes || B <= 0 || T <= 0 || U1 <= 0 || V <= 0) return TFASR_STATUS_INVALID_VALUE;
  const size_t n = (size_t)B * T * U1;
  *bytes = 10 * align256(n * sizeof(float));  // lse, blank, truth, alpha, beta + per-row gradient coefficients [n][4] + row labels
  return TFASR_STATUS_SUCCESS;
}

static int rnnt_impl(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                     const int32_t* logit_len, const float* grad_scale, const long* cell_off, long nrows, int B, int T, int U1,
                     int V, int blank, int dtype, float* costs, void* workspace, size_t workspace_bytes, void* stream_,
                     const float* lse_part = nullptr, int lse_parts = 0, const float* pick = nullptr, float* coef = nullptr) {
  if ((!logits && !(lse_part && coef && !grads)) || !labels || !label_len || !logit_len || !costs || !workspace) return TFASR_STATUS_INVALID_VALUE;
  if (blank != 0) return TFASR_STATUS_UNSUPPORTED;  // losses/base_loss.py:24
  if (B <= 0 || T <= 0 || U1 <= 0 || V <= 1 || U1 > 1024 || nrows <= 0) return TFASR_STATUS_INVALID_VALUE;
  const size_t n = (size_t)nrows;
  const size_t seg = align256(n * sizeof(float));
  if (workspace_bytes < 5 * seg) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  float* lse = (float*)(ws);
  float* blank_lp = (float*)(ws + seg);
  float* truth_lp = (float*)(ws + 2 * seg);
  float* alpha = (float*)(ws + 3 * seg);
  float* beta = (float*)(ws + 4 * seg);
  const int wpb = 4;
  int grid = (int)std::min<long>((nrows + wpb - 1) / wpb, 256L * 32);
  if (lse_part) {
    const int fg = (int)std::min<long>(((lse_parts == 16 ? nrows * 16 : nrows) + 255) / 256, 256L * 16);
    TFASR_KLAUNCH(rnnt_stats_finalize_kernel, dim3(fg), dim3(256), 0, stream, (const float2*)lse_part, lse_parts, pick, label_len, logit_len,
                       cell_off, nrows, B, T, U1, lse, blank_lp, truth_lp);
  } else if (dtype == TFASR_F32)
    TFASR_KLAUNCH(rnnt_logprobs_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)logits, labels,
                       label_len, logit_len, cell_off, nrows, B, T, U1, V, lse, blank_lp, truth_lp);
  else if (dtype == TFASR_BF16)
    TFASR_KLAUNCH(rnnt_logprobs_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)logits, labels,
                       label_len, logit_len, cell_off, nrows, B, T, U1, V, lse, blank_lp, truth_lp);
  else
    return TFASR_STATUS_INVALID_VALUE;
  TFASR_CHECK_LAUNCH();
  const int nthr = ((U1 + 63) / 64) * 64;
  static const bool wave_off = false;  // A/B probe: the workgroup kernel
  static const bool fast = !(false);
#define TFASR_LW(E) do { if (fast) TFASR_KLAUNCH((rnnt_lattice_wave_kernel<E, true>), dim3(B, 2), dim3(64), 0, stream, blank_lp, truth_lp, label_len, logit_len, cell_off, T, U1, alpha, beta, costs); \
                         else TFASR_KLAUNCH((rnnt_lattice_wave_kernel<E, false>), dim3(B, 2), dim3(64), 0, stream, blank_lp, truth_lp, label_len, logit_len, cell_off, T, U1, alpha, beta, costs); } while (0)
  if (!wave_off && U1 <= 64) TFASR_LW(1);
  else if (!wave_off && U1 <= 128) TFASR_LW(2);
  else if (!wave_off && U1 <= 256) TFASR_LW(4);
  else
    TFASR_KLAUNCH(rnnt_lattice_kernel, dim3(B, 2), dim3(nthr), 2 * nthr * sizeof(float), stream, blank_lp,
                       truth_lp, label_len, logit_len, cell_off, T, U1, alpha, beta, costs);
  TFASR_CHECK_LAUNCH();
  // gradient as coefficient pass + pure stream (rnnt_grad_apply_kernel) when the workspace has room for the coefficients (older callers
  // sized it for 5 segments: they keep the one-launch kernel); TFASR_RNNT_GRAD_STREAM=0 forces the one-launch kernel
  static const bool stream_off = false;
  const bool streamed = grads && !stream_off && (V % 8) == 0 && V <= 1024 && workspace_bytes >= 10 * seg && (dtype == TFASR_F32 || dtype == TFASR_BF16);
  float4* cbuf = coef ? (float4*)coef : (float4*)(ws + 5 * seg);
  int32_t* rlab = (int32_t*)(ws + 9 * seg);
  if (coef || streamed) {
    const int cg = (int)std::min<long>((nrows + 255) / 256, 256L * 8);
    TFASR_KLAUNCH(rnnt_coef_kernel, dim3(cg), dim3(256), 0, stream, label_len, logit_len, grad_scale, cell_off, nrows, B, T, U1, lse, blank_lp,
                       truth_lp, alpha, beta, cbuf, streamed ? labels : (const int32_t*)nullptr, V, streamed ? rlab : (int32_t*)nullptr);
    TFASR_CHECK_LAUNCH();
  }
  if (streamed) {
    const int ag = (int)std::min<long>((nrows + 7) / 8, 256L * 16);
    if (dtype == TFASR_F32)
      TFASR_KLAUNCH(rnnt_grad_apply_kernel<float>, dim3(ag), dim3(256), 0, stream, (const float*)logits, (float*)grads, cbuf, rlab, nrows, V);
    else
      TFASR_KLAUNCH(rnnt_grad_apply_kernel<bf16_t>, dim3(ag), dim3(256), 0, stream, (const bf16_t*)logits, (bf16_t*)grads, cbuf, rlab, nrows, V);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  if (grads) {
    if (dtype == TFASR_F32)
      TFASR_KLAUNCH(rnnt_grad_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)logits,
                         (float*)grads, labels, label_len, logit_len, grad_scale, cell_off, nrows, B, T, U1, V, lse, blank_lp,
                         truth_lp, alpha, beta);
    else
      TFASR_KLAUNCH(rnnt_grad_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)logits,
                         (bf16_t*)grads, labels, label_len, logit_len, grad_scale, cell_off, nrows, B, T, U1, V, lse, blank_lp,
                         truth_lp, alpha, beta);
    TFASR_CHECK_LAUNCH();
  }
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_rnnt_loss(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                               const int32_t* logit_len, const float* grad_scale, int B, int T, int U1, int V,
                               int blank, int dtype, float* costs, void* workspace, size_t workspace_bytes,
                               void* stream_) {
  return rnnt_impl(logits, grads, labels, label_len, logit_len, grad_scale, nullptr, (long)B * T * U1, B, T, U1, V, blank, dtype,
                   costs, workspace, workspace_bytes, stream_);
}

extern "C" int tfasr_rnnt_loss_packed(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                                      const int32_t* logit_len, const float* grad_scale, const long* cell_off, long total_cells,
                                      int B, int T, int U1, int V, int blank, int dtype, float* costs, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
  if (!cell_off) return TFASR_STATUS_INVALID_VALUE;
  return rnnt_impl(logits, grads, labels, label_len, logit_len, grad_scale, cell_off, total_cells, B, T, U1, V, blank, dtype,
                   costs, workspace, workspace_bytes, stream_);
}

extern "C" int tfasr_rnnt_loss_packed_stats(const void* logits, void* grads, const int32_t* labels, const int32_t* label_len,
                                            const int32_t* logit_len, const float* grad_scale, const long* cell_off, long total_cells,
                                            const float* lse_part, int lse_parts, const float* pick, int B, int T, int U1, int V, int blank,
                                            int dtype, float* costs, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!cell_off || !lse_part || !pick || lse_parts <= 0) return TFASR_STATUS_INVALID_VALUE;
  return rnnt_impl(logits, grads, labels, label_len, logit_len, grad_scale, cell_off, total_cells, B, T, U1, V, blank, dtype,
                   costs, workspace, workspace_bytes, stream_, lse_part, lse_parts, pick);
}

extern "C" int tfasr_rnnt_loss_packed_coef(const int32_t* labels, const int32_t* label_len, const int32_t* logit_len, const float* grad_scale,
                                           const long* cell_off, long total_cells, const float* lse_part, int lse_parts, const float* pick, int B,
                                           int T, int U1, int V, int blank, float* costs, float* coef, void* workspace, size_t workspace_bytes,
                                           void* stream_) {
  if (!cell_off || !lse_part || !pick || !coef || lse_parts <= 0) return TFASR_STATUS_INVALID_VALUE;
  return rnnt_impl(nullptr, nullptr, labels, label_len, logit_len, grad_scale, cell_off, total_cells, B, T, U1, V, blank, TFASR_BF16, costs,
                   workspace, workspace_bytes, stream_, lse_part, lse_parts, pick, coef);
}

extern "C" int tfasr_rnnt_row_labels(const int32_t* labels, const int32_t* label_len, const int32_t* logit_len, const long* cell_off,
                                     long total_cells, int B, int T, int U1, int V, int32_t* row_label, void* stream_) {
  if (!labels || !label_len || !logit_len || !row_label || total_cells <= 0 || B <= 0 || T <= 0 || U1 <= 0 || V <= 1) return TFASR_STATUS_INVALID_VALUE;
  const int g = (int)std::min<long>((total_cells + 255) / 256, 256L * 8);
  TFASR_KLAUNCH(rnnt_row_labels_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream_, labels, label_len, logit_len, cell_off, total_cells, B, T,
                     U1, V, row_label);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
