// bf16 MFMA GEMM, persistent 4-stage LDS-DMA pipeline (gfx950).  Same contract / epilogue as gemm.hip; tried first by
// tfasr_gemm for bf16 operands that satisfy the alignment rules of gemm_fast.hip.
//
//   128 x BN output tile (BN = 128, or 64 for narrow outputs), 4 waves (2x2), BK = 32, FOUR LDS stages filled by
//   global_load_lds_dwordx4: three K-slabs stay in flight ahead of the MFMAs and ONE s_barrier per slab orders both
//   "slab i has landed" and "everyone is done with slab i-1" (whose stage the barrier's successor DMA overwrites).
//   Workgroups are persistent: 2 per CU, each walks a strided list of tiles; the next tile's first three slabs are
//   issued BEFORE the current tile's epilogue, so the epilogue (bias / activation / dropout / residual / stores) runs
//   under the DMA latency instead of in front of it, and the pipeline never drains between tiles.
//   The DMA instructions of slab i+3 are issued BETWEEN the fragment reads and the MFMAs of slab i: measured with cycle
//   counters on gemm_fast.hip, issuing a slab's 8 global_load_lds per wave costs ~900 cycles of issue stall - more than
//   the slab's MFMAs - when it sits in its own phase.
//   Tile order is XCD-aware: in every round each XCD (workgroup id mod 8) owns one contiguous run of the n-fastest tile
//   sequence, so the n-tiles sharing an A row-block hit the same L2.
//   LDS images (bank-conflict free, swizzle applied on the DMA source address, undone on the fragment read):
//     k-contiguous operand: [rows][32 k] (64-B rows), 16-B chunk ^= (row >> 2) & 3, fragment = one ds_read_b128
//     k-strided operand   : [32 k][rows] (256 / 128-B rows), chunk ^= key(k), fragment = two ds_read_b64_tr_b16
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short short4_t;

namespace {

constexpr int BM = 128, BK = 32, NST = 4;
constexpr int A_BYTES = BM * BK * 2;  // 8 KiB

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ int key_d(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ int key_t(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }
__device__ __forceinline__ int key_t64(int k) { return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1; }

// ---- LDS-DMA issue (per wave: ROWS/64 instructions per operand per slab) -------------------------------------------
template <int ROWS>
__device__ __forceinline__ void issue_direct(char* s, const bf16_t* g, long ld, int rows0, int nrows, int kt, int w, int lane) {
#pragma unroll
  for (int i = 0; i < ROWS / 64; ++i) {
    const int q = w * (ROWS / 64) + i;  // 16 rows x 64 B per instruction
    const int row = q * 16 + (lane >> 2), p = lane & 3;
    const int gr = min(rows0 + row, nrows - 1);
    const bf16_t* src = g + (long)gr * ld + kt + ((p ^ key_d(row)) << 3);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(s + __builtin_amdgcn_readfirstlane(q * 1024)), 16, 0, 0);
  }
}
template <int ROWS>
__device__ __forceinline__ void issue_trans(char* s, const bf16_t* g, long ld, int rows0, int nrows, int kt, int w, int lane) {
  const int maxchunk = ((nrows + 7) >> 3) - 1;
  constexpr int CPR = ROWS / 8;   // 16-B chunks per k-row
  constexpr int KPI = 64 / CPR;   // k-rows per wave instruction
#pragma unroll
  for (int i = 0; i < ROWS / 64; ++i) {
    const int q = w * (ROWS / 64) + i;
    const int k = q * KPI + lane / CPR, p = lane % CPR;
    const int c = min((rows0 >> 3) + (p ^ (ROWS == 128 ? key_t(k) : key_t64(k))), maxchunk);
    const bf16_t* src = g + (long)(kt + k) * ld + ((long)c << 3);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(s + __builtin_amdgcn_readfirstlane(q * 1024)), 16, 0, 0);
  }
}
// K-tail slab (K % 32 != 0): plain stores with zero fill into the same images
template <int ROWS>
__device__ __forceinline__ void tail_direct(char* s, const bf16_t* g, long ld, int rows0, int nrows, int kt, int k_end) {
  for (int c = threadIdx.x; c < ROWS * 4; c += 256) {
    const int row = c >> 2, p = c & 3;
    const int gr = min(rows0 + row, nrows - 1);
    const int kc = (p ^ key_d(row)) << 3;
    uint32_t w4[4] = {0, 0, 0, 0};
    const bf16_t* src = g + (long)gr * ld + kt + kc;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (kt + kc + e < k_end) w4[e >> 1] |= ((uint32_t)src[e]) << ((e & 1) * 16);
    *reinterpret_cast<uint4*>(s + row * 64 + p * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
  }
}
template <int ROWS>
__device__ __forceinline__ void tail_trans(char* s, const bf16_t* g, long ld, int rows0, int nrows, int kt, int k_end) {
  const int maxchunk = ((nrows + 7) >> 3) - 1;
  constexpr int CPR = ROWS / 8;
  for (int c = threadIdx.x; c < BK * CPR; c += 256) {
    const int k = c / CPR, p = c % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (kt + k < k_end) {
      const int ch = min((rows0 >> 3) + (p ^ (ROWS == 128 ? key_t(k) : key_t64(k))), maxchunk);
      v = *reinterpret_cast<const uint4*>(g + (long)(kt + k) * ld + ((long)ch << 3));
    }
    *reinterpret_cast<uint4*>(s + k * (ROWS * 2) + p * 16) = v;
  }
}

// ---- fragment reads ---------------------------------------------------------------------------------------------
__device__ __forceinline__ short8_t frag_direct(const char* s, int row, int c) {
  return *reinterpret_cast<const short8_t*>(s + row * 64 + ((c ^ key_d(row)) << 4));
}
template <int ROWS>
__device__ __forceinline__ short8_t frag_trans(const char* s, int rowbase, int kbase, int r) {
  const int col = rowbase + ((r & 3) << 2);
  const int chunk = col >> 3, half = (col >> 2) & 1;
  const int k0 = kbase + (r >> 2), k1 = k0 + 4;
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) short4_t*)(s + k0 * (ROWS * 2) + ((chunk ^ (ROWS == 128 ? key_t(k0) : key_t64(k0))) << 4) + half * 8));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) short4_t*)(s + k1 * (ROWS * 2) + ((chunk ^ (ROWS == 128 ? key_t(k1) : key_t64(k1))) << 4) + half * 8));
  short8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return v;
}

template <bool TA, bool TB, int BN_, typename F>
__device__ __forceinline__ void mma_slab(const char* sA, const char* sB, int wm, int wn, int lane, float4_t (&acc)[4][BN_ / 32], F&& between) {
  constexpr int NJ = BN_ / 32, WN = BN_ / 2;
  const int r = lane & 15, g = lane >> 4;
  short8_t a[4], b[NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (TA) a[i] = frag_trans<128>(sA, wm * 64 + i * 16, g * 8, r);
    else    a[i] = frag_direct(sA, wm * 64 + i * 16 + r, g);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (TB) b[j] = frag_direct(sB, wn * WN + j * 16 + r, g);
    else    b[j] = frag_trans<BN_>(sB, wn * WN + j * 16, g * 8, r);
  }
  __builtin_amdgcn_sched_barrier(0);
  between();  // the next slab's DMA issue rides under the LDS latency of the fragment reads
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
}

__device__ __forceinline__ float act_f(float v, int act) {
  switch (act) {
    case TFASR_ACT_SWISH: return swishf_(v);
    case TFASR_ACT_TANH: return tanhf(v);
    case TFASR_ACT_SIGMOID: return sigmoidf_(v);
    default: return v;
  }
}
__device__ __forceinline__ float dact_f(float z, int act) {
  switch (act) {
    case TFASR_ACT_SWISH: return dswishf_(z);
    case TFASR_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
    case TFASR_ACT_SIGMOID: { const float s = sigmoidf_(z); return s * (1.f - s); }
    default: return 1.f;
  }
}

enum : int { E_ACT = 1 /* swish(+prez) */, E_DACT = 2 /* * swish'(dact_z) */, E_DROP = 4, E_RES = 8, E_GEN = 256 };


// wait until at most `groups` DMA groups (GI wave-instructions each) are still in flight
template <int GI>
__device__ __forceinline__ void wait_groups(int groups) {
  if (groups >= 2) {
    if (GI == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else if (groups == 1) {
    if (GI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <bool TA, bool TB, int BN_, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_pipe_kernel(const tfasr_gemm_args p, const int gx, const int gy, const int gz, const int ntiles) {
  constexpr bool GEN = (EPI & E_GEN) != 0;
  constexpr bool C_ACT = GEN || (EPI & E_ACT), C_DACT = GEN || (EPI & E_DACT), C_DROP = GEN || (EPI & E_DROP), C_RES = GEN || (EPI & E_RES);
  constexpr int BN = BN_, NJ = BN_ / 32, WN = BN_ / 2;
  constexpr int B_BYTES = BN_ * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int GI = 2 + BN_ / 64;  // DMA wave-instructions per slab per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NST stages + epilogue strips
  const int split = p.split_k > 1 ? p.split_k : 1;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int r = lane & 15, g = lane >> 4;
  const int G = gridDim.x;
  int kchunk = (p.K + split - 1) / split;
  kchunk = ((kchunk + BK - 1) / BK) * BK;

  struct Tile { const bf16_t* A; const bf16_t* B; long doff; int m0, n0, k_begin, k_end, nfull, tail, ks; };
  // XCD-aware tile order (hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own L2):
  //  * plain / batched: in every round each XCD owns one contiguous run of the n-fastest tile sequence, so the n-tiles
  //    sharing an A row-block hit the same L2 instead of fetching it over the fabric once per XCD;
  //  * split-K with gz % 8 == 0 (weight gradients): a whole k-slice (all its M x N tiles re-read the same slabs) stays
  //    on ONE XCD - 5x less HBM traffic measured on the joint weight gradient.
  auto tile_of = [&](int it) {
    Tile T;
    T.nfull = -1;
    const int tpp = gx * gy;
    int tx, ty, tz;
    if ((G & 7) == 0) {
      const int x = blockIdx.x & 7, j = (blockIdx.x >> 3) + it * (G >> 3);
      if (split > 1 && (gz & 7) == 0) {
        if (j >= (gz >> 3) * tpp) return T;
        tz = x + 8 * (j / tpp);
        const int t = j % tpp;
        tx = t % gx; ty = t / gx;
      } else {
        // round `it` covers tiles [it*G, (it+1)*G); XCD x takes the x-th eighth of it
        const int t = it * G + x * (G >> 3) + (blockIdx.x >> 3);
        if (t >= ntiles) return T;
        tx = t % gx; const int rest = t / gx; ty = rest % gy; tz = rest / gy;
      }
    } else {  // G == ntiles (fewer tiles than resident slots): one round, ragged but bijective runs
      if (it > 0) return T;
      const int q = G >> 3, rem = G & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      const int t = G >= 16 ? (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot : (int)blockIdx.x;
      tx = t % gx; const int rest = t / gx; ty = rest % gy; tz = rest / gy;
    }
    const int ks = tz % split, bidx = tz / split;
    const int b1 = bidx / p.nb2, b2 = bidx % p.nb2;
    T.A = (const bf16_t*)p.A + b1 * p.sA1 + b2 * p.sA2;
    T.B = (const bf16_t*)p.B + b1 * p.sB1 + b2 * p.sB2;
    T.doff = b1 * p.sD1 + b2 * p.sD2;
    T.m0 = ty * BM;
    T.n0 = tx * BN;
    T.ks = ks;
    T.k_begin = ks * kchunk;
    T.k_end = min(p.K, T.k_begin + kchunk);
    const int len = max(T.k_end - T.k_begin, 0);
    T.nfull = len / BK;
    T.tail = (len % BK) != 0;
    return T;
  };
  auto issue = [&](const Tile& T, int slab, int ring) {
    char* sA = smem + (ring & (NST - 1)) * STAGE_BYTES;
    char* sB = sA + A_BYTES;
    const int kt = T.k_begin + slab * BK;
    if (TA) issue_trans<128>(sA, T.A, p.lda, T.m0, p.M, kt, w, lane); else issue_direct<128>(sA, T.A, p.lda, T.m0, p.M, kt, w, lane);
    if (TB) issue_direct<BN_>(sB, T.B, p.ldb, T.n0, p.N, kt, w, lane); else issue_trans<BN_>(sB, T.B, p.ldb, T.n0, p.N, kt, w, lane);
  };

  int ring = 0;  // ring position of the current tile's slab 0
  Tile cur = tile_of(0);
  if (cur.nfull < 0) return;
  {
    const int pre = min(cur.nfull, NST - 1);
    for (int i = 0; i < pre; ++i) issue(cur, i, ring + i);
  }
  bool drained = false;  // true: every DMA issued so far has been waited for (after a tile-boundary drain)

  for (int it = 0;; ++it) {
    float4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int n = cur.nfull;
    for (int i = 0; i < n; ++i) {
      // slab i landed?  groups issued after it so far: slabs i+1, i+2 (when they exist)
      if (!drained || i >= NST - 1) wait_groups<GI>(min(2, n - 1 - i));
      __builtin_amdgcn_s_barrier();  // slab i visible to all waves; all waves finished slab i-1 (whose stage slab i+3 reuses)
      const char* st = smem + ((ring + i) & (NST - 1)) * STAGE_BYTES;
      mma_slab<TA, TB, BN_>(st, st + A_BYTES, wm, wn, lane, acc, [&] { if (i + NST - 1 < n) issue(cur, i + NST - 1, ring + i + NST - 1); });
    }
    int used = n;
    if (cur.tail) {
      // the stage after the last DMA slab: last read NST slabs ago; no DMA targets it (the prefetch never reaches a tail)
      char* sA = smem + ((ring + n) & (NST - 1)) * STAGE_BYTES;
      char* sB = sA + A_BYTES;
      const int kt = cur.k_begin + n * BK;
      __syncthreads();
      if (TA) tail_trans<128>(sA, cur.A, p.lda, cur.m0, p.M, kt, cur.k_end); else tail_direct<128>(sA, cur.A, p.lda, cur.m0, p.M, kt, cur.k_end);
      if (TB) tail_direct<BN_>(sB, cur.B, p.ldb, cur.n0, p.N, kt, cur.k_end); else tail_trans<BN_>(sB, cur.B, p.ldb, cur.n0, p.N, kt, cur.k_end);
      __syncthreads();
      mma_slab<TA, TB, BN_>(sA, sB, wm, wn, lane, acc, [] {});
      used = n + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with this tile's last stages
    ring += used;
    // ---- cross-tile prefetch: the next tile's first slabs go out before this tile's epilogue ----
    const Tile nxt = tile_of(it + 1);
    {
      const int pre = nxt.nfull > 0 ? min(nxt.nfull, NST - 1) : 0;
      for (int i = 0; i < pre; ++i) issue(nxt, i, ring + i);
    }

    // ---- epilogue ----
    const int m0 = cur.m0, n0 = cur.n0;
    const long doff = cur.doff;
    const bool first_split = (cur.ks == 0);
    bf16_t* Dt = (bf16_t*)p.D + doff;
    float* Df = (float*)p.D + doff;
    const bf16_t* res = p.res ? (const bf16_t*)p.res + doff : nullptr;
    const bf16_t* dz = p.dact_z ? (const bf16_t*)p.dact_z + doff : nullptr;
    bf16_t* prez = p.prez ? (bf16_t*)p.prez + doff : nullptr;
    if (p.accumulate) {
      // split-K / gradient accumulation: f32 atomics straight from the MFMA fragment layout (16 consecutive
      // columns x 4 rows per instruction = 4 cache lines), no other epilogue terms are legal here
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn * WN + j * 16 + r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = m0 + wm * 64 + i * 16 + g * 4 + e;
            if (col < p.N && row < p.M) {
              float v = p.alpha * acc[i][j][e];
              if (p.bias && first_split) v += p.bias[col];
              atomicAdd(Df + (long)row * p.ldd + col, v);
            }
          }
        }
    } else {
      // Each wave turns its 16 x WN fragment strip into row-major order through LDS (RPP rows at a time), then every store /
      // load instruction of the epilogue covers WHOLE rows: 8 lanes x 16 B = one 128-B line per row (BN 128), so HBM sees
      // full lines (16-B pieces at a 32-B stride cost ~5000 extra cycles per tile in store-issue stalls).
      constexpr int SLD = WN + 4;
      constexpr int LPRW = WN / 8;        // lanes per strip row: 8 (BN 128) or 4 (BN 64), 8 columns each
      constexpr int RPP = 64 / LPRW;      // rows per pass: 8 or 16
      constexpr int NPASS = 16 / RPP;     // 2 or 1
      float* sc = reinterpret_cast<float*>(smem + NST * STAGE_BYTES) + w * (RPP * SLD);
      const int prow = lane / LPRW, c8 = (lane % LPRW) * 8;
      const int col0 = n0 + wn * WN + c8;
      const bool vec_ok = ((p.ldd & 7) == 0) && ((((uintptr_t)p.D) & 15) == 0) && ((doff & 7) == 0);
      const bool full = vec_ok && (col0 + 8 <= p.N);
      float bv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) bv[q] = 0.f;
      if (p.bias && first_split && col0 < p.N) {
        const float* bp = p.bias + col0;
        if (col0 + 8 <= p.N && ((((uintptr_t)bp) & 15) == 0)) {
          const float4 t0 = *reinterpret_cast<const float4*>(bp), t1 = *reinterpret_cast<const float4*>(bp + 4);
          bv[0] = t0.x; bv[1] = t0.y; bv[2] = t0.z; bv[3] = t0.w; bv[4] = t1.x; bv[5] = t1.y; bv[6] = t1.z; bv[7] = t1.w;
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) if (col0 + q < p.N) bv[q] = bp[q];
        }
      }
      const uint32_t dthr = drop_thr(p.drop_p);
      const float dinv = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
      auto strip = [&](auto I_, auto H_) {
        constexpr int i = decltype(I_)::value, h = decltype(H_)::value;
        if ((g * 4) / RPP == h) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[(g * 4 + e - h * RPP) * SLD + j * 16 + r] = acc[i][j][e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float x[8];
        {
          const float4 t0 = *reinterpret_cast<const float4*>(sc + prow * SLD + c8);
          const float4 t1 = *reinterpret_cast<const float4*>(sc + prow * SLD + c8 + 4);
          x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; x[4] = t1.x; x[5] = t1.y; x[6] = t1.z; x[7] = t1.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int row = m0 + wm * 64 + i * 16 + h * RPP + prow;
        if (row < p.M && col0 < p.N) {
          const long idx0 = (long)row * p.ldd + col0;
#pragma unroll
          for (int q = 0; q < 8; ++q) x[q] = p.alpha * x[q] + bv[q];
          if constexpr (C_ACT) {
            if (prez) {
              if (full) st8(prez + idx0, x);
              else
_Pragma("unroll")
                for (int q = 0; q < 8; ++q) if (col0 + q < p.N) prez[idx0 + q] = f32_to_bf16(x[q]);
            }
            if constexpr (GEN) {
              if (p.act != TFASR_ACT_NONE) {
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = act_f(x[q], p.act);
              }
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) x[q] = swishf_(x[q]);
            }
          }
          if constexpr (C_DACT) if (dz) {
            float z[8];
            if (full) ld8(dz + idx0, z);
            else
_Pragma("unroll")
              for (int q = 0; q < 8; ++q) z[q] = (col0 + q < p.N) ? bf16_to_f32(dz[idx0 + q]) : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] *= GEN ? dact_f(z[q], p.dact) : dswishf_(z[q]);
          }
          if constexpr (C_DROP) if (p.drop_p > 0.f) {
            const uint64_t e0 = (uint64_t)(doff + idx0);
            if ((e0 & 1) == 0) {  // one hash per even/odd element pair
#pragma unroll
              for (int q = 0; q < 8; q += 2) {
                const uint32_t hh = drop_hash((uint64_t)p.drop_seed, (e0 >> 1) + (q >> 1));
                x[q] = (hh & 0xffffu) >= dthr ? x[q] * dinv : 0.f;
                x[q + 1] = (hh >> 16) >= dthr ? x[q + 1] * dinv : 0.f;
              }
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) x[q] = drop_keep((uint64_t)p.drop_seed, e0 + q, p.drop_p) ? x[q] * dinv : 0.f;
            }
          }
          if constexpr (C_RES) if (res) {
            float z[8];
            if (full) ld8(res + idx0, z);
            else
_Pragma("unroll")
              for (int q = 0; q < 8; ++q) z[q] = (col0 + q < p.N) ? bf16_to_f32(res[idx0 + q]) : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = z[q] + p.beta * x[q];
          }
          if (GEN && p.out_f32) {
            if (full) st8(Df + idx0, x);
            else
_Pragma("unroll")
              for (int q = 0; q < 8; ++q) if (col0 + q < p.N) Df[idx0 + q] = x[q];
          } else {
            if (full) st8(Dt + idx0, x);
            else
_Pragma("unroll")
              for (int q = 0; q < 8; ++q) if (col0 + q < p.N) Dt[idx0 + q] = f32_to_bf16(x[q]);
          }
        }
      };
      auto strips = [&](auto I_) {
        strip(I_, std::integral_constant<int, 0>{});
        if constexpr (NPASS == 2) strip(I_, std::integral_constant<int, 1>{});
      };
      strips(std::integral_constant<int, 0>{});
      strips(std::integral_constant<int, 1>{});
      strips(std::integral_constant<int, 2>{});
      strips(std::integral_constant<int, 3>{});
    }
    if (nxt.nfull < 0) break;
    // tile boundary: epilogue stores and loads are mixed into the vector-memory queue, so counted waits are void until
    // it drains once (the prefetched slabs had the whole epilogue to land)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drained = true;
    cur = nxt;
  }
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

template <bool TA, bool TB, int BN_, int EPI>
int launch_epi(const tfasr_gemm_args& a, dim3 tiles, hipStream_t stream) {
  const long ntiles = (long)tiles.x * tiles.y * tiles.z;
  const int slots = 2 * num_cus();  // 2 resident workgroups per CU
  int G = (int)(ntiles < slots ? ntiles : slots);
  if (ntiles >= slots) G &= ~7;
  constexpr int SMEM = NST * (A_BYTES + BN_ * BK * 2) + 4 * (64 / (BN_ / 16)) * (BN_ / 2 + 4) * 4;  // stages + 4 waves' strips
  hipLaunchKernelGGL((gemm_pipe_kernel<TA, TB, BN_, EPI>), dim3(G), dim3(256), SMEM, stream, a, (int)tiles.x, (int)tiles.y, (int)tiles.z, (int)ntiles);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

template <bool TA, bool TB>
int launch_one(const tfasr_gemm_args& a, hipStream_t stream) {
  const int split = a.split_k > 1 ? a.split_k : 1;
  const bool narrow = a.N <= 64;  // per-head attention products etc.: halve the wasted B tile
  const int bn = narrow ? 64 : 128;
  dim3 grid((a.N + bn - 1) / bn, (a.M + BM - 1) / BM, a.nb1 * a.nb2 * split);
  if ((long)grid.x * grid.y * grid.z > 0x7fffffffL) return TFASR_STATUS_INVALID_VALUE;
  // epilogue terms this call needs; accumulate (atomics from fragments) needs none of them
  int need = 0;
  bool generic = false;
  if (!a.accumulate) {
    if (a.out_f32) generic = true;
    if (a.act != TFASR_ACT_NONE || a.prez) { if (a.act == TFASR_ACT_SWISH) need |= E_ACT; else generic = true; }
    if (a.dact_z) { if (a.dact == TFASR_ACT_SWISH) need |= E_DACT; else generic = true; }
    if (a.drop_p > 0.f) need |= E_DROP;
    if (a.res) need |= E_RES;
  }
  if (narrow) return generic || need ? launch_epi<TA, TB, 64, E_GEN>(a, grid, stream) : launch_epi<TA, TB, 64, 0>(a, grid, stream);
  if (!generic) {
    if (need == 0) return launch_epi<TA, TB, 128, 0>(a, grid, stream);
    if constexpr (!TA && !TB) {  // forward Dense layers
      if (need == E_RES) return launch_epi<TA, TB, 128, E_RES>(a, grid, stream);
      if (need == (E_RES | E_DROP)) return launch_epi<TA, TB, 128, E_RES | E_DROP>(a, grid, stream);
      if (need == E_ACT) return launch_epi<TA, TB, 128, E_ACT>(a, grid, stream);
      if (need == (E_ACT | E_DROP)) return launch_epi<TA, TB, 128, E_ACT | E_DROP>(a, grid, stream);
    }
    if constexpr (!TA && TB) {  // data gradients (dy @ W^T)
      if (need == E_DACT) return launch_epi<TA, TB, 128, E_DACT>(a, grid, stream);
      if (need == (E_DACT | E_DROP)) return launch_epi<TA, TB, 128, E_DACT | E_DROP>(a, grid, stream);
    }
  }
  return launch_epi<TA, TB, 128, E_GEN>(a, grid, stream);
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// returns TFASR_STATUS_UNSUPPORTED when the preconditions do not hold (caller falls back)
int tfasr_gemm_pipe_try(const tfasr_gemm_args& a, hipStream_t stream) {
  if (a.dtype != TFASR_BF16 || a.colsum) return TFASR_STATUS_UNSUPPORTED;
  if (!al16(a.A) || !al16(a.B) || (a.lda & 7) || (a.ldb & 7)) return TFASR_STATUS_UNSUPPORTED;
  if ((a.sA1 & 7) || (a.sA2 & 7) || (a.sB1 & 7) || (a.sB2 & 7)) return TFASR_STATUS_UNSUPPORTED;
  if (a.trans_a && ((a.M + 7) & ~7) > a.lda) return TFASR_STATUS_UNSUPPORTED;
  if (!a.trans_b && ((a.N + 7) & ~7) > a.ldb) return TFASR_STATUS_UNSUPPORTED;
  if (a.K < 8) return TFASR_STATUS_UNSUPPORTED;
  if (a.trans_a) return a.trans_b ? launch_one<true, true>(a, stream) : launch_one<true, false>(a, stream);
  return a.trans_b ? launch_one<false, true>(a, stream) : launch_one<false, false>(a, stream);
}
