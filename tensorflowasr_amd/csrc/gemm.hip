// MFMA GEMM family for gfx950 (wave64).  See include/tfasr_hip.h for the contract.
//
// Structure (round-1 version): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// wave a 64x64 sub-tile = 4x4 MFMA 16x16 fragments, f32 accumulators), K staged through LDS in
// BK-deep slabs, both operands stored k-contiguous in LDS so every MFMA fragment is one 16-byte
// ds_read (bf16) / one dword (f32).  Operands whose k index is the strided one in memory
// (B of an NN product, A and B of a weight-gradient TN product) are transposed in registers on the
// way in (8x8 bf16 blocks) so global loads stay 16 B per lane and coalesced.
//   bf16 : __builtin_amdgcn_mfma_f32_16x16x32_bf16   (dense peak ~2.5 PFLOP/s)
//   f32  : __builtin_amdgcn_mfma_f32_16x16x4f32      (exact f32, 157 TFLOP/s)  -- the parity mode
// C/D fragment map (both): col = lane & 15, row = (lane >> 4) * 4 + reg.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> { static constexpr int BK = 64, LD = 72; };
template <> struct Cfg<float>  { static constexpr int BK = 16, LD = 17; };  // (32-deep slabs measured slower: encoder 22.5 vs 18.3 ms)

__device__ __forceinline__ bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- tile loaders --------------------------------------------------------------------------
// "direct": operand stored [rows, K] (ld), k contiguous.  rows0 = first row of the tile.
__device__ __forceinline__ void load_direct(bf16_t* s, const bf16_t* g, long ld, int rows0, int nrows_total,
                                            int kt, int k_end) {
  constexpr int BK = Cfg<bf16_t>::BK, LD = Cfg<bf16_t>::LD;
  for (int c = threadIdx.x; c < BM * BK / 8; c += 256) {
    const int row = c / (BK / 8), kc = (c % (BK / 8)) * 8;
    const int gr = rows0 + row, gk = kt + kc;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < nrows_total) {
      const bf16_t* p = g + (long)gr * ld + gk;
      if (gk + 8 <= k_end && aligned16(p)) v = *reinterpret_cast<const uint4*>(p);
      else {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (gk + i < k_end) w[i >> 1] |= ((uint32_t)p[i]) << ((i & 1) * 16);
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    *reinterpret_cast<uint4*>(s + row * LD + kc) = v;
  }
}
__device__ __forceinline__ void load_direct(float* s, const float* g, long ld, int rows0, int nrows_total, int kt,
                                            int k_end) {
  constexpr int BK = Cfg<float>::BK, LD = Cfg<float>::LD;
  for (int c = threadIdx.x; c < BM * BK / 4; c += 256) {
    const int row = c / (BK / 4), kc = (c % (BK / 4)) * 4;
    const int gr = rows0 + row, gk = kt + kc;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gr < nrows_total) {
      const float* p = g + (long)gr * ld + gk;
      if (gk + 4 <= k_end && aligned16(p)) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (gk + i < k_end) v[i] = p[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[row * LD + kc + i] = v[i];
  }
}

// "transposed": operand stored [K, rows] (ld), rows contiguous.  8x8 bf16 blocks transposed in registers.
__device__ __forceinline__ void load_trans(bf16_t* s, const bf16_t* g, long ld, int rows0, int nrows_total, int kt,
                                           int k_end) {
  constexpr int BK = Cfg<bf16_t>::BK, LD = Cfg<bf16_t>::LD;
  constexpr int NRB = BM / 8;  // row blocks
  for (int blk = threadIdx.x; blk < (BK / 8) * NRB; blk += 256) {
    const int kb = (blk / NRB) * 8, rb = (blk % NRB) * 8;
    const int gr = rows0 + rb;
    uint32_t in[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gk = kt + kb + j;
      in[j][0] = in[j][1] = in[j][2] = in[j][3] = 0;
      if (gk < k_end && gr < nrows_total) {
        const bf16_t* p = g + (long)gk * ld + gr;
        if (gr + 8 <= nrows_total && aligned16(p)) {
          const uint4 q = *reinterpret_cast<const uint4*>(p);
          in[j][0] = q.x; in[j][1] = q.y; in[j][2] = q.z; in[j][3] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (gr + i < nrows_total) in[j][i >> 1] |= ((uint32_t)p[i]) << ((i & 1) * 16);
        }
      }
    }
    // out row i (0..7) holds k = 0..7 : word w = (k=2w, k=2w+1)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t o[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t lo = in[2 * w][i >> 1], hi = in[2 * w + 1][i >> 1];
        o[w] = (i & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
      }
      *reinterpret_cast<uint4*>(s + (rb + i) * LD + kb) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}
__device__ __forceinline__ void load_trans(float* s, const float* g, long ld, int rows0, int nrows_total, int kt,
                                           int k_end) {
  constexpr int BK = Cfg<float>::BK, LD = Cfg<float>::LD;
  for (int c = threadIdx.x; c < BK * (BM / 4); c += 256) {
    const int k = c / (BM / 4), r4 = (c % (BM / 4)) * 4;
    const int gk = kt + k, gr = rows0 + r4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gk < k_end && gr < nrows_total) {
      const float* p = g + (long)gk * ld + gr;
      if (gr + 4 <= nrows_total && aligned16(p)) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (gr + i < nrows_total) v[i] = p[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[(r4 + i) * LD + k] = v[i];
  }
}

// f32 slabs in two steps - global -> registers, registers -> LDS - so that slab k+1 can be in flight while slab k is multiplied
// (same element placement as load_direct / load_trans above; 128 x 16 floats = 2 float4 per thread and operand)
struct F32Stage { float v[128 * Cfg<float>::BK / 4 / 256][4]; };
template <int ROWS>
__device__ __forceinline__ void fetch_direct(F32Stage& st, const float* g, long ld, int rows0, int nrows_total, int kt, int k_end) {
  constexpr int BK = Cfg<float>::BK;
#pragma unroll
  for (int it = 0; it < ROWS * Cfg<float>::BK / 4 / 256; ++it) {
    const int c = threadIdx.x + it * 256;
    const int row = c / (BK / 4), kc = (c % (BK / 4)) * 4;
    const int gr = rows0 + row, gk = kt + kc;
    float* v = st.v[it];
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (gr < nrows_total) {
      const float* p = g + (long)gr * ld + gk;
      if (gk + 4 <= k_end && aligned16(p)) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (gk + i < k_end) v[i] = p[i];
      }
    }
  }
}
template <int ROWS>
__device__ __forceinline__ void store_direct(float* s, const F32Stage& st) {
  constexpr int BK = Cfg<float>::BK, LD = Cfg<float>::LD;
#pragma unroll
  for (int it = 0; it < ROWS * Cfg<float>::BK / 4 / 256; ++it) {
    const int c = threadIdx.x + it * 256;
    const int row = c / (BK / 4), kc = (c % (BK / 4)) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) s[row * LD + kc + i] = st.v[it][i];
  }
}
template <int ROWS>
__device__ __forceinline__ void fetch_trans(F32Stage& st, const float* g, long ld, int rows0, int nrows_total, int kt, int k_end) {
#pragma unroll
  for (int it = 0; it < ROWS * Cfg<float>::BK / 4 / 256; ++it) {
    const int c = threadIdx.x + it * 256;
    const int k = c / (ROWS / 4), r4 = (c % (ROWS / 4)) * 4;
    const int gk = kt + k, gr = rows0 + r4;
    float* v = st.v[it];
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (gk < k_end && gr < nrows_total) {
      const float* p = g + (long)gk * ld + gr;
      if (gr + 4 <= nrows_total && aligned16(p)) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (gr + i < nrows_total) v[i] = p[i];
      }
    }
  }
}
template <int ROWS>
__device__ __forceinline__ void store_trans(float* s, const F32Stage& st) {
  constexpr int LD = Cfg<float>::LD;
#pragma unroll
  for (int it = 0; it < ROWS * Cfg<float>::BK / 4 / 256; ++it) {
    const int c = threadIdx.x + it * 256;
    const int k = c / (ROWS / 4), r4 = (c % (ROWS / 4)) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) s[(r4 + i) * LD + k] = st.v[it][i];
  }
}

// ---- MFMA inner product over one LDS slab ---------------------------------------------------
__device__ __forceinline__ void mma_slab(const bf16_t* sA, const bf16_t* sB, int wm, int wn, int lane,
                                         float4_t (&acc)[4][4]) {
  constexpr int BK = Cfg<bf16_t>::BK, LD = Cfg<bf16_t>::LD;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < BK / 32; ++kk) {
    short8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      a[i] = *reinterpret_cast<const short8_t*>(sA + (wm * 64 + i * 16 + r) * LD + kk * 32 + g * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      b[j] = *reinterpret_cast<const short8_t*>(sB + (wn * 64 + j * 16 + r) * LD + kk * 32 + g * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}
template <int NJ>  // 16-column fragments per wave: 4 (128-column tile) or 2 (64-column tile)
__device__ __forceinline__ void mma_slab(const float* sA, const float* sB, int wm, int wn, int lane,
                                         float4_t (&acc)[4][4]) {
  constexpr int BK = Cfg<float>::BK, LD = Cfg<float>::LD;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < BK / 4; ++kk) {
    float a[4], b[NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = sA[(wm * 64 + i * 16 + r) * LD + kk * 4 + g];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = sB[(wn * (NJ * 16) + j * 16 + r) * LD + kk * 4 + g];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case TFASR_ACT_SWISH: return swishf_(v);
    case TFASR_ACT_TANH: return tanhf(v);
    case TFASR_ACT_SIGMOID: return sigmoidf_(v);
    default: return v;
  }
}
__device__ __forceinline__ float apply_dact(float z, int act) {
  switch (act) {
    case TFASR_ACT_SWISH: return dswishf_(z);
    case TFASR_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
    case TFASR_ACT_SIGMOID: { const float s = sigmoidf_(z); return s * (1.f - s); }
    case TFASR_ACT_TANH_OUT: return 1.f - z * z;
    case TFASR_ACT_FACTOR: return z;
    default: return 1.f;
  }
}

// BN_: columns per tile.  64 only for f32: exact-f32 products with few 128-wide tiles (N = 256 at 8 000 rows: 126 tiles for 256 CUs) run as
// twice as many 128 x 64 tiles - same slabs, same k order per element: bitwise the same results.
template <typename T, bool TA, bool TB, int BN_ = BN>
__global__ __launch_bounds__(256) void gemm_kernel(const tfasr_gemm_args p) {
  constexpr int BK = Cfg<T>::BK, LD = Cfg<T>::LD, NJ = BN_ / 32;
  static_assert(BN_ == BN || sizeof(T) == 4, "narrow tiles: f32 only");
  __shared__ __attribute__((aligned(16))) T smem[(BM + BN_) * LD];
  T* sA = smem;
  T* sB = smem + BM * LD;

  const int split = p.split_k > 1 ? p.split_k : 1;
  const int ks = blockIdx.z % split;
  const int bidx = blockIdx.z / split;
  const int b1 = bidx / p.nb2, b2 = bidx % p.nb2;
  const T* A = (const T*)p.A + b1 * p.sA1 + b2 * p.sA2;
  const T* Bm = (const T*)p.B + b1 * p.sB1 + b2 * p.sB2;
  const long doff = b1 * p.sD1 + b2 * p.sD2;

  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN_;
  int kchunk = (p.K + split - 1) / split;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  const int k_begin = ks * kchunk;
  const int k_end = min(p.K, k_begin + kchunk);

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;

  float4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

  if constexpr (sizeof(T) == 4) {
    // exact-f32 path (parity mode, token-exact inference): slab k+1 is fetched into registers BEFORE slab k's 64 MFMAs per wave and stored to
    // LDS behind them - the load latency (one round trip per 16-deep slab) used to sit between every two slabs.  Same slabs, same k order:
    // bitwise the same results.
    F32Stage pa, pb;
    auto fetch = [&](int kt) {
      if (TA) fetch_trans<BM>(pa, (const float*)A, p.lda, m0, p.M, kt, k_end); else fetch_direct<BM>(pa, (const float*)A, p.lda, m0, p.M, kt, k_end);
      if (TB) fetch_direct<BN_>(pb, (const float*)Bm, p.ldb, n0, p.N, kt, k_end); else fetch_trans<BN_>(pb, (const float*)Bm, p.ldb, n0, p.N, kt, k_end);
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int kt = k_begin; kt < k_end; kt += BK) {
      if (TA) store_trans<BM>((float*)sA, pa); else store_direct<BM>((float*)sA, pa);
      if (TB) store_direct<BN_>((float*)sB, pb); else store_trans<BN_>((float*)sB, pb);
      __syncthreads();
      if (kt + BK < k_end) fetch(kt + BK);
      mma_slab<NJ>((const float*)sA, (const float*)sB, wm, wn, lane, acc);
      __syncthreads();
    }
  } else {
  for (int kt = k_begin; kt < k_end; kt += BK) {
    if (TA) load_trans(sA, A, p.lda, m0, p.M, kt, k_end); else load_direct(sA, A, p.lda, m0, p.M, kt, k_end);
    // B: trans_b==1 -> stored [N,K] (k contiguous) = "direct"; trans_b==0 -> stored [K,N] = needs transpose
    if (TB) load_direct(sB, Bm, p.ldb, n0, p.N, kt, k_end); else load_trans(sB, Bm, p.ldb, n0, p.N, kt, k_end);
    __syncthreads();
    mma_slab(sA, sB, wm, wn, lane, acc);
    __syncthreads();
  }
  }

  // ---- epilogue ----
  const int r = lane & 15, g = lane >> 4;
  const bool first_split = (ks == 0);
  T* Dt = (T*)p.D + doff;
  float* Df = (float*)p.D + doff;
  const T* res = p.res ? (const T*)p.res + doff : nullptr;
  const T* dz = p.dact_z ? (const T*)p.dact_z + doff : nullptr;
  T* prez = p.prez ? (T*)p.prez + doff : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn * (NJ * 16) + j * 16 + r;
      const bool col_ok = col < p.N;
      const float bias = (p.bias && first_split && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = m0 + wm * 64 + i * 16 + g * 4 + e;
        if (col_ok && row < p.M) {
        const long idx = (long)row * p.ldd + col;
        float v = p.alpha * acc[i][j][e] + bias;
        if (prez) Num<T>::st(prez + idx, v);
        v = apply_act(v, p.act);
        if (dz) v *= apply_dact(Num<T>::ld(dz + idx), p.dact);
        if (p.drop_p > 0.f) v = drop_keep((uint64_t)p.drop_seed, (uint64_t)(doff + idx), p.drop_p) ? v * (1.f / (1.f - p.drop_p)) : 0.f;
        if (res) v = Num<T>::ld(res + idx) + p.beta * v;
        if (p.out_f32) {
          if (p.accumulate) atomicAdd(Df + idx, v);
          else Df[idx] = v;
        } else {
          Num<T>::st(Dt + idx, v);
        }
        }
      }
    }
  }
}

template <typename T>
int launch(const tfasr_gemm_args& a, hipStream_t stream) {
  const int split = a.split_k > 1 ? a.split_k : 1;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.nb1 * a.nb2 * split);
  if (grid.y > 65535 || grid.z > 65535) return TFASR_STATUS_INVALID_VALUE;
  dim3 block(256);
  if constexpr (sizeof(T) == 4) {
    // f32: 128 x 64 tiles while the 128-wide tiling leaves CUs without a workgroup of their own (TFASR_GEMM_F32_NARROW=0: never)
    static const bool narrow_off = false;
    static int ncu = 0;
    if (ncu == 0) {
      int dev = 0, v = 0;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    if (!narrow_off && a.N > 64 && (long)grid.x * grid.y * grid.z < 2L * ncu) {
      dim3 g2((a.N + 63) / 64, grid.y, grid.z);
      if (a.trans_a) {
        if (a.trans_b) TFASR_KLAUNCH((gemm_kernel<T, true, true, 64>), g2, block, 0, stream, a);
        else           TFASR_KLAUNCH((gemm_kernel<T, true, false, 64>), g2, block, 0, stream, a);
      } else {
        if (a.trans_b) TFASR_KLAUNCH((gemm_kernel<T, false, true, 64>), g2, block, 0, stream, a);
        else           TFASR_KLAUNCH((gemm_kernel<T, false, false, 64>), g2, block, 0, stream, a);
      }
      TFASR_CHECK_LAUNCH();
      return TFASR_STATUS_SUCCESS;
    }
  }
  if (a.trans_a) {
    if (a.trans_b) TFASR_KLAUNCH((gemm_kernel<T, true, true>), grid, block, 0, stream, a);
    else           TFASR_KLAUNCH((gemm_kernel<T, true, false>), grid, block, 0, stream, a);
  } else {
    if (a.trans_b) TFASR_KLAUNCH((gemm_kernel<T, false, true>), grid, block, 0, stream, a);
    else           TFASR_KLAUNCH((gemm_kernel<T, false, false>), grid, block, 0, stream, a);
  }
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

}  // namespace

int tfasr_gemm_fast_try(const tfasr_gemm_args& a, hipStream_t stream);  // gemm_fast.hip (2-stage BK=64, one tile per workgroup)

static bool use_fast_path() {
  static int v = -1;
  if (v < 0) v = 1;
  return v == 1;
}

int tfasr_gemm_group_fast_try(const tfasr_gemm_args* a, int n, hipStream_t stream);

extern "C" int tfasr_gemm_group(const tfasr_gemm_args* args, int n, void* stream_) {
  if (!args || n <= 0) return TFASR_STATUS_INVALID_VALUE;
  static const bool off = false;
  int i = 0;
  while (i < n) {
    const int m = n - i < 10 ? n - i : 10;
    int st = TFASR_STATUS_UNSUPPORTED;
    if (!off && use_fast_path()) st = tfasr_gemm_group_fast_try(args + i, m, (hipStream_t)stream_);
    if (st == TFASR_STATUS_UNSUPPORTED) {
      for (int j = 0; j < m; ++j) {
        st = tfasr_gemm(args + i + j, stream_);
        if (st != TFASR_STATUS_SUCCESS) return st;
      }
    } else if (st != TFASR_STATUS_SUCCESS) {
      return st;
    }
    i += m;
  }
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_gemm(const tfasr_gemm_args* args, void* stream_) {
  if (!args || !args->A || !args->B || (!args->D && !args->lse_part)) return TFASR_STATUS_INVALID_VALUE;
  tfasr_gemm_args a = *args;
  if (a.rgrad_coef && (!a.row_label || !a.D || a.lse_part)) return TFASR_STATUS_INVALID_VALUE;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (a.nb1 < 1) a.nb1 = 1;
  if (a.nb2 < 1) a.nb2 = 1;
  if (a.split_k < 1) a.split_k = 1;
  if (a.accumulate && !a.out_f32) return TFASR_STATUS_INVALID_VALUE;
  if (a.split_k > 1 && !a.accumulate) return TFASR_STATUS_INVALID_VALUE;
  if (a.split_k > 1 && (a.res || a.dact_z || a.prez || a.act != TFASR_ACT_NONE || a.drop_p > 0.f)) return TFASR_STATUS_INVALID_VALUE;
  if (a.drop_p < 0.f || a.drop_p >= 1.f) return TFASR_STATUS_INVALID_VALUE;
  if (a.colsum && !(a.accumulate && a.trans_a && !a.trans_b && a.nb1 * a.nb2 == 1)) return TFASR_STATUS_INVALID_VALUE;
  hipStream_t stream = (hipStream_t)stream_;
  if (a.dtype == TFASR_BF16 && use_fast_path()) {
    const int st = tfasr_gemm_fast_try(a, stream);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  if (a.lse_part || a.rgrad_coef || a.bns_out) return TFASR_STATUS_UNSUPPORTED;  // only the bf16 fast path's epilogues produce the row statistics / the loss gradient / the BatchNorm sums
  if (a.seg_a_off) return TFASR_STATUS_UNSUPPORTED;  // K-segments: bf16 fast path only (the f32 host path issues one product per segment)
  if (a.colsum) {  // the generic kernels do not fuse the bias gradient: one extra pass over B = dy
    const int st = tfasr_colsum(a.B, a.ldb, a.colsum, a.K, a.N, a.alpha, a.dtype, stream_);
    if (st != TFASR_STATUS_SUCCESS) return st;
    a.colsum = nullptr;
  }
  if (a.dtype == TFASR_BF16) return launch<bf16_t>(a, stream);
  if (a.dtype == TFASR_F32) return launch<float>(a, stream);
  return TFASR_STATUS_INVALID_VALUE;
}
