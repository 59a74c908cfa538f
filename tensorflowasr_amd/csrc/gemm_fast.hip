// bf16 MFMA GEMM, LDS-DMA pipelined variant (gfx950).  Same contract / epilogue as gemm.hip, selected by
// tfasr_gemm when dtype == bf16 and the operands satisfy the alignment rules below; everything else falls back.
//
//   128x128 output tile, 4 waves (2x2), each wave 4x4 MFMA 16x16x32 bf16 fragments, BK = 64.
//   Both operand slabs go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip, 1 KiB per wave
//   instruction), two LDS stages (64 KiB), counted vmcnt + raw s_barrier so the next slab's DMA stays in
//   flight under the current slab's MFMAs.
//   LDS-DMA writes lane-linear, so bank-conflict swizzles are applied on the per-lane SOURCE address and undone
//   on the fragment read (guide rule 21):
//     k-contiguous operand ("direct"): image [128 rows][64 k]   (128 B rows), 16-B chunk ^= (row>>1)&7,
//                                      fragment = one ds_read_b128
//     k-strided operand   ("trans") : image [64 k][128 rows]    (256 B rows), 16-B chunk ^= key(k),
//                                      fragment = two ds_read_b64_tr_b16 (hardware transpose read)
//   so all four storage layouts (NN / NT / TN / TT) run at the same rate without register transposes.
//   The K tail (K % 64) is staged by plain stores with zero fill.
// Requirements for this path: A,B 16-B aligned, lda/ldb % 8 == 0, batch strides % 8 == 0, and for a k-strided
// operand its row extent (M for A, N for B) rounded up to 8 must fit inside the row stride.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short short4_t;

namespace {

constexpr int BM = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;
// BN_ = 128 (default) or 64 (narrow outputs such as the per-head attention products, N = head size)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ int key_d(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int key_t(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }
// 64-row trans image: 128-B k-rows, two per 256-B bank row -> spread rows {0..3, 8..11} over the 8 32-B windows
__device__ __forceinline__ int key_t64(int k) { return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1; }

// ---- LDS-DMA issue: 4 wave-instructions per operand per wave ------------------------------------------
// direct: operand stored [rows, K] (ld); rows0.. clamp to nrows-1; slab k range [kt, kt+64)
template <int ROWS>
__device__ __forceinline__ void issue_direct(char* s, const bf16_t* g, long ld, int rows0, int nrows, int kt, int w, int lane) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int q = w * (ROWS / 32) + i;
    const int row = q * 8 + (lane >> 3), p = lane & 7;
    const int gr = min(rows0 + row, nrows - 1);
    const bf16_t* src = g + (long)gr * ld + kt + ((p ^ key_d(row)) << 3);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(s + __builtin_amdgcn_readfirstlane(q * 1024)), 16, 0, 0);
  }
}
// trans: operand stored [K, rows] (ld); image [64 k][128 rows]
template <int ROWS>
__device__ __forceinline__ void issue_trans(char* s, const bf16_t* g, long ld, int rows0, int nrows, int kt, int w, int lane) {
  const int maxchunk = ((nrows + 7) >> 3) - 1;  // the row padding up to a multiple of 8 must exist (ld >= roundup8)
  constexpr int CPR = ROWS / 8;        // 16-B chunks per k-row
  constexpr int KPI = 64 / CPR;        // k-rows per wave instruction
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int q = w * (ROWS / 32) + i;
    const int k = q * KPI + lane / CPR, p = lane % CPR;
    const int c = min((rows0 >> 3) + (p ^ (ROWS == 128 ? key_t(k) : key_t64(k))), maxchunk);
    const bf16_t* src = g + (long)(kt + k) * ld + ((long)c << 3);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(s + __builtin_amdgcn_readfirstlane(q * 1024)), 16, 0, 0);
  }
}
// K-tail staging with zero fill (plain stores into the same swizzled images)
template <int ROWS>
__device__ __forceinline__ void tail_direct(char* s, const bf16_t* g, long ld, int rows0, int nrows, int kt, int k_end) {
  for (int c = threadIdx.x; c < ROWS * 8; c += 256) {
    const int row = c >> 3, p = c & 7;
    const int gr = min(rows0 + row, nrows - 1);
    const int kc = (p ^ key_d(row)) << 3;
    uint32_t w4[4] = {0, 0, 0, 0};
    const bf16_t* src = g + (long)gr * ld + kt + kc;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (kt + kc + e < k_end) w4[e >> 1] |= ((uint32_t)src[e]) << ((e & 1) * 16);
    *reinterpret_cast<uint4*>(s + row * 128 + p * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
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

// ---- fragment reads ---------------------------------------------------------------------------------
__device__ __forceinline__ short8_t frag_direct(const char* s, int row, int c) {
  return *reinterpret_cast<const short8_t*>(s + row * 128 + ((c ^ key_d(row)) << 4));
}
template <int ROWS>
__device__ __forceinline__ short8_t frag_trans(const char* s, int rowbase, int kbase, int r) {
  // 16-lane group reads the 4(k) x 16(row) block; lane gets column (rowbase + r), k = kbase..kbase+3 then +4..+7
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

template <bool TA, bool TB, int BN_>
__device__ __forceinline__ void mma_slab(const char* sA, const char* sB, int wm, int wn, int lane, float4_t (&acc)[4][BN_ / 32]) {
  constexpr int NJ = BN_ / 32, WN = BN_ / 2;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    short8_t a[4], b[NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (TA) a[i] = frag_trans<128>(sA, wm * 64 + i * 16, kk * 32 + g * 8, r);
      else    a[i] = frag_direct(sA, wm * 64 + i * 16 + r, kk * 4 + g);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (TB) b[j] = frag_direct(sB, wn * WN + j * 16 + r, kk * 4 + g);
      else    b[j] = frag_trans<BN_>(sB, wn * WN + j * 16, kk * 32 + g * 8, r);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
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

template <bool TA, bool TB, int BN_>
__global__ __launch_bounds__(256, 2) void gemm_fast_kernel(const tfasr_gemm_args p) {
  constexpr int BN = BN_, NJ = BN_ / 32, WN = BN_ / 2;
  constexpr int STAGE_BYTES = A_BYTES + BN_ * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (16 KiB A + 16|8 KiB B)
  const int split = p.split_k > 1 ? p.split_k : 1;
  // XCD-aware tile order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2),
  // so give every XCD one contiguous run of the n-fastest tile sequence: the n-tiles that share an A row-block then hit
  // the same L2 instead of fetching it over the fabric once per XCD (bijective remap, guide "XCD swizzle").
  int tile_x = blockIdx.x, tile_y = blockIdx.y, tile_z = blockIdx.z;
  {
    const int gx = gridDim.x, total = gx * gridDim.y;
    if (split > 1 && (gridDim.z & 7) == 0) {
      // split-K (weight gradients): every k-slice's tiles re-read the same A / B slabs, so put a whole slice on ONE XCD
      // (hardware deals linear workgroup ids round-robin over the 8 XCDs) - the slab is then fetched once per slice
      // instead of once per tile (5x less HBM traffic measured on the joint weight gradient).  Bijective for gz % 8 == 0.
      const int L = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = L & 7, slot = L >> 3;
      tile_z = xcd + 8 * (slot / total);
      const int t = slot % total;
      tile_x = t % gx;
      tile_y = t / gx;
    } else if (total >= 16) {
      const int L = blockIdx.x + gx * blockIdx.y;
      const int q = total >> 3, rem = total & 7, xcd = L & 7, slot = L >> 3;
      const int Lp = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
      tile_x = Lp % gx;
      tile_y = Lp / gx;
    }
  }
  const int ks = tile_z % split;
  const int bidx = tile_z / split;
  const int b1 = bidx / p.nb2, b2 = bidx % p.nb2;
  const bf16_t* A = (const bf16_t*)p.A + b1 * p.sA1 + b2 * p.sA2;
  const bf16_t* Bm = (const bf16_t*)p.B + b1 * p.sB1 + b2 * p.sB2;
  const long doff = b1 * p.sD1 + b2 * p.sD2;
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  int kchunk = (p.K + split - 1) / split;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  const int k_begin = ks * kchunk;
  const int k_end = min(p.K, k_begin + kchunk);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;

  float4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int nfull = (k_end - k_begin) / BK;  // slabs served by LDS-DMA
  const bool has_tail = (k_begin + nfull * BK) < k_end;

  auto issue = [&](int slab, int stage) {
    char* sA = smem + stage * STAGE_BYTES;
    char* sB = sA + A_BYTES;
    const int kt = k_begin + slab * BK;
    if (TA) issue_trans<128>(sA, A, p.lda, m0, p.M, kt, w, lane); else issue_direct<128>(sA, A, p.lda, m0, p.M, kt, w, lane);
    if (TB) issue_direct<BN_>(sB, Bm, p.ldb, n0, p.N, kt, w, lane); else issue_trans<BN_>(sB, Bm, p.ldb, n0, p.N, kt, w, lane);
  };

  if (nfull > 0) issue(0, 0);
  for (int s = 0; s < nfull; ++s) {
    const int stage = s & 1;
    if (s + 1 < nfull) {
      issue(s + 1, stage ^ 1);
      // this wave's DMA pieces of slab s (4 for A + BN/32 for B) have landed; the next slab's stay in flight
      if (BN_ == 128) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    mma_slab<TA, TB, BN_>(smem + stage * STAGE_BYTES, smem + stage * STAGE_BYTES + A_BYTES, wm, wn, lane, acc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading this stage before it is refilled
  }
  if (has_tail) {
    char* sA = smem;
    char* sB = smem + A_BYTES;
    const int kt = k_begin + nfull * BK;
    if (TA) tail_trans<128>(sA, A, p.lda, m0, p.M, kt, k_end); else tail_direct<128>(sA, A, p.lda, m0, p.M, kt, k_end);
    if (TB) tail_direct<BN_>(sB, Bm, p.ldb, n0, p.N, kt, k_end); else tail_trans<BN_>(sB, Bm, p.ldb, n0, p.N, kt, k_end);
    __syncthreads();
    mma_slab<TA, TB, BN_>(sA, sB, wm, wn, lane, acc);
  }

  // ---- epilogue: C fragments -> per-wave LDS strip (16 rows x 64 cols f32) -> row-major, 16 columns per lane,
  //      so bias / activation / residual / stores all move 16-32 B per lane on whole 128-B lines ----
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  const bool first_split = (ks == 0);
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
    return;
  }
  constexpr int SLD = WN + 4;
  constexpr int CPL = WN / 4;  // columns per lane on the way out: 16 (BN 128) or 8 (BN 64)
  float* sc = reinterpret_cast<float*>(smem) + w * (16 * SLD);
  const int rr = lane >> 2, cseg = (lane & 3) * CPL;
  const bool vec_ok = ((p.ldd & 7) == 0) && ((((uintptr_t)p.D) & 15) == 0) && ((doff & 7) == 0);
  // bias for this lane's CPL columns (the same in all four strips)
  float bv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) bv[q] = 0.f;
  {
    const int col0 = n0 + wn * WN + cseg;
    if (p.bias && first_split && col0 < p.N) {
      const float* bp = p.bias + col0;
      if (col0 + CPL <= p.N && ((((uintptr_t)bp) & 15) == 0)) {
#pragma unroll
        for (int q = 0; q < CPL / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(bp + q * 4);
          bv[q * 4] = t.x; bv[q * 4 + 1] = t.y; bv[q * 4 + 2] = t.z; bv[q * 4 + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < CPL; ++q) if (col0 + q < p.N) bv[q] = bp[q];
      }
    }
  }
  const uint32_t dthr = drop_thr(p.drop_p);
  auto strip = [&](auto I_) {
    constexpr int i = decltype(I_)::value;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[(g * 4 + e) * SLD + j * 16 + r] = acc[i][j][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int row = m0 + wm * 64 + i * 16 + rr;
    const int col0 = n0 + wn * WN + cseg;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = 0.f;  // (array stays 16 wide; CPL of it are live)
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(sc + rr * SLD + cseg + q * 4);
      v[q * 4] = t.x; v[q * 4 + 1] = t.y; v[q * 4 + 2] = t.z; v[q * 4 + 3] = t.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (row < p.M && col0 < p.N) {
      const long idx0 = (long)row * p.ldd + col0;
      const bool full = vec_ok && (col0 + CPL <= p.N);
#pragma unroll
      for (int q = 0; q < CPL; ++q) v[q] = p.alpha * v[q] + bv[q];
      if (prez) {
        if (full) { st8(prez + idx0, *reinterpret_cast<const float(*)[8]>(v)); if (CPL == 16) st8(prez + idx0 + 8, *reinterpret_cast<const float(*)[8]>(v + 8)); }
        else
_Pragma("unroll")
          for (int q = 0; q < CPL; ++q) if (col0 + q < p.N) prez[idx0 + q] = f32_to_bf16(v[q]);
      }
      if (p.act != TFASR_ACT_NONE) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) v[q] = act_f(v[q], p.act);
      }
      if (dz) {
        float z[16];
        if (full) { ld8(dz + idx0, *reinterpret_cast<float(*)[8]>(z)); if (CPL == 16) ld8(dz + idx0 + 8, *reinterpret_cast<float(*)[8]>(z + 8)); }
        else
_Pragma("unroll")
          for (int q = 0; q < CPL; ++q) z[q] = (col0 + q < p.N) ? bf16_to_f32(dz[idx0 + q]) : 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) v[q] *= dact_f(z[q], p.dact);
      }
      if (p.drop_p > 0.f) {
        const float inv = 1.f / (1.f - p.drop_p);
        const uint64_t e0 = (uint64_t)(doff + idx0);
        if ((e0 & 1) == 0) {  // one hash per even/odd element pair
#pragma unroll
          for (int q = 0; q < CPL; q += 2) {
            const uint32_t h = drop_hash((uint64_t)p.drop_seed, (e0 >> 1) + (q >> 1));
            v[q] = (h & 0xffffu) >= dthr ? v[q] * inv : 0.f;
            v[q + 1] = (h >> 16) >= dthr ? v[q + 1] * inv : 0.f;
          }
        } else {
#pragma unroll
          for (int q = 0; q < CPL; ++q) v[q] = drop_keep((uint64_t)p.drop_seed, e0 + q, p.drop_p) ? v[q] * inv : 0.f;
        }
      }
      if (res) {
        float z[16];
        if (full) { ld8(res + idx0, *reinterpret_cast<float(*)[8]>(z)); if (CPL == 16) ld8(res + idx0 + 8, *reinterpret_cast<float(*)[8]>(z + 8)); }
        else
_Pragma("unroll")
          for (int q = 0; q < CPL; ++q) z[q] = (col0 + q < p.N) ? bf16_to_f32(res[idx0 + q]) : 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) v[q] = z[q] + p.beta * v[q];
      }
      if (p.out_f32) {
        if (p.accumulate) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) if (col0 + q < p.N) atomicAdd(Df + idx0 + q, v[q]);
        } else if (full) {
          st8(Df + idx0, *reinterpret_cast<const float(*)[8]>(v)); if (CPL == 16) st8(Df + idx0 + 8, *reinterpret_cast<const float(*)[8]>(v + 8));
        } else {
_Pragma("unroll")
          for (int q = 0; q < CPL; ++q) if (col0 + q < p.N) Df[idx0 + q] = v[q];
        }
      } else {
        if (full) { st8(Dt + idx0, *reinterpret_cast<const float(*)[8]>(v)); if (CPL == 16) st8(Dt + idx0 + 8, *reinterpret_cast<const float(*)[8]>(v + 8)); }
        else
_Pragma("unroll")
          for (int q = 0; q < CPL; ++q) if (col0 + q < p.N) Dt[idx0 + q] = f32_to_bf16(v[q]);
      }
    }
  };
  strip(std::integral_constant<int, 0>{});
  strip(std::integral_constant<int, 1>{});
  strip(std::integral_constant<int, 2>{});
  strip(std::integral_constant<int, 3>{});
}

template <bool TA, bool TB>
int launch_one(const tfasr_gemm_args& a, hipStream_t stream) {
  const int split = a.split_k > 1 ? a.split_k : 1;
  const bool narrow = a.N <= 64;  // per-head attention products etc.: halve the wasted B tile
  const int bn = narrow ? 64 : 128;
  dim3 grid((a.N + bn - 1) / bn, (a.M + BM - 1) / BM, a.nb1 * a.nb2 * split);
  if (grid.y > 65535 || grid.z > 65535) return TFASR_STATUS_INVALID_VALUE;
  if (narrow) hipLaunchKernelGGL((gemm_fast_kernel<TA, TB, 64>), grid, dim3(256), 2 * (A_BYTES + 64 * BK * 2), stream, a);
  else        hipLaunchKernelGGL((gemm_fast_kernel<TA, TB, 128>), grid, dim3(256), 2 * (A_BYTES + 128 * BK * 2), stream, a);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// returns TFASR_STATUS_UNSUPPORTED when the fast path's preconditions do not hold (caller falls back)
int tfasr_gemm_fast_try(const tfasr_gemm_args& a, hipStream_t stream) {
  if (a.dtype != TFASR_BF16) return TFASR_STATUS_UNSUPPORTED;
  if (!al16(a.A) || !al16(a.B) || (a.lda & 7) || (a.ldb & 7)) return TFASR_STATUS_UNSUPPORTED;
  if ((a.sA1 & 7) || (a.sA2 & 7) || (a.sB1 & 7) || (a.sB2 & 7)) return TFASR_STATUS_UNSUPPORTED;
  if (a.trans_a && ((a.M + 7) & ~7) > a.lda) return TFASR_STATUS_UNSUPPORTED;  // 16-B chunks must stay inside the row
  if (!a.trans_b && ((a.N + 7) & ~7) > a.ldb) return TFASR_STATUS_UNSUPPORTED;
  if (a.K < 8) return TFASR_STATUS_UNSUPPORTED;
  if (a.trans_a) return a.trans_b ? launch_one<true, true>(a, stream) : launch_one<true, false>(a, stream);
  return a.trans_b ? launch_one<false, true>(a, stream) : launch_one<false, false>(a, stream);
}
