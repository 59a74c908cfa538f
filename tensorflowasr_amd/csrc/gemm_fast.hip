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
#include <string.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) short short4_t;

#ifdef TFASR_GEMM_TIMING
__device__ long long g_gemm_timing[16 * 32768];  // per workgroup: start, first slab landed, mainloop end, end (cycles)
#endif

namespace {

constexpr int BM = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;
// BN_ = 128 (default) or 64 (narrow outputs such as the per-head attention products, N = head size)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// One LDS-DMA piece (1 KiB per wave instruction): 16 bytes per lane from `g` to LDS byte offset (uniform) `lds` + 16 * lane.
// Issued through inline asm ON PURPOSE: for the builtin the compiler knows an LDS write is pending and puts `s_waitcnt vmcnt(0)` in
// front of the wave's next ds_read (it cannot prove the two do not alias) - i.e. it drained the whole prefetch queue in front of every
// slab's fragment reads, and the main loop ran at one DMA round trip per slab with nothing in flight under the MFMAs (what the cycle
// counters of round 2 showed and blamed on the hardware).  The pipeline's RAW / WAR ordering is the hand-counted `s_waitcnt vmcnt(N)` +
// `s_barrier` pairs in the kernels; stores and plain loads the compiler issues keep their own (conservative: vmcnt retires in order)
// waits.  TFASR_GLDS_BUILTIN=1 (compile time) restores the builtin for A/B runs.
__device__ __forceinline__ uint32_t lds_u32(const char* s) {
  // a generic pointer into LDS = the shared aperture (high half) + the LDS byte offset (low half)
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)s);
}
__device__ __forceinline__ void glds16(const void* g, const char* lds_uniform) {
#ifdef TFASR_GLDS_BUILTIN
  __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(lds_uniform), 16, 0, 0);
#else
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_u32(lds_uniform)) : "memory");
#endif
}
// the same with a uniform 64-bit base and a 32-bit per-lane byte offset (saddr form: no 64-bit vector add per piece)
template <bool ASM = true>
__device__ __forceinline__ void glds16_s(const char* base_uniform, uint32_t off, const char* lds_uniform) {
#ifdef TFASR_GLDS_BUILTIN
  constexpr bool BUILTIN = true;
#else
  constexpr bool BUILTIN = !ASM;
#endif
  if constexpr (BUILTIN) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(base_uniform + off), LDS_PTR(lds_uniform), 16, 0, 0);
    return;
  }
  const uint64_t b = (uint64_t)base_uniform;
  const uint64_t ub = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  // s_nop 4: the base usually reaches its SGPR pair through v_readfirstlane, and a VALU write of an SGPR needs 5 wait states before a
  // vector-memory instruction reads it - the compiler's hazard recogniser does not look inside the asm (without them the piece went out
  // with the previous base: memory faults at base 0 + lane offset)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(ub), "s"(lds_u32(lds_uniform)) : "memory");
}

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
    glds16(src, s + __builtin_amdgcn_readfirstlane(q * 1024));
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
    const int c = min((rows0 >> 3) + (p ^ (ROWS >= 128 ? key_t(k) : key_t64(k))), maxchunk);
    const bf16_t* src = g + (long)(kt + k) * ld + ((long)c << 3);
    glds16(src, s + __builtin_amdgcn_readfirstlane(q * 1024));
  }
}
// Per-tile source pointers for the DMA (slab 0), so that issuing a slab costs one 64-bit add per instruction instead of the
// whole address computation (row clamp, swizzle key, 64-bit multiply): measured, the 8 DMA instructions of a slab took ~950
// cycles to issue per wave - more than the slab's 32 MFMAs - with the addresses recomputed every slab.
template <int ROWS>
__device__ __forceinline__ void prep_direct(const bf16_t* (&src)[ROWS / 32], const bf16_t* g, long ld, int rows0, int nrows, int kt, int w, int lane) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int q = w * (ROWS / 32) + i;
    const int row = q * 8 + (lane >> 3), p = lane & 7;
    const int gr = min(rows0 + row, nrows - 1);
    src[i] = g + (long)gr * ld + kt + ((p ^ key_d(row)) << 3);
  }
}
template <int ROWS>
__device__ __forceinline__ void prep_trans(const bf16_t* (&src)[ROWS / 32], const bf16_t* g, long ld, int rows0, int nrows, int kt, int w, int lane) {
  const int maxchunk = ((nrows + 7) >> 3) - 1;
  constexpr int CPR = ROWS / 8, KPI = 64 / CPR;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int q = w * (ROWS / 32) + i;
    const int k = q * KPI + lane / CPR, p = lane % CPR;
    const int c = min((rows0 >> 3) + (p ^ (ROWS >= 128 ? key_t(k) : key_t64(k))), maxchunk);
    src[i] = g + (long)(kt + k) * ld + ((long)c << 3);
  }
}
// issue one slab: src + delta elements (delta = slab*BK for a k-contiguous operand, slab*BK*ld for a k-strided one)
template <int ROWS>
__device__ __forceinline__ void issue_from(char* s, const bf16_t* const (&src)[ROWS / 32], long delta, int w) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int q = w * (ROWS / 32) + i;
    glds16(src[i] + delta, s + __builtin_amdgcn_readfirstlane(q * 1024));
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
      const int ch = min((rows0 >> 3) + (p ^ (ROWS >= 128 ? key_t(k) : key_t64(k))), maxchunk);
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
      (__attribute__((address_space(3))) short4_t*)(s + k0 * (ROWS * 2) + ((chunk ^ (ROWS >= 128 ? key_t(k0) : key_t64(k0))) << 4) + half * 8));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) short4_t*)(s + k1 * (ROWS * 2) + ((chunk ^ (ROWS >= 128 ? key_t(k1) : key_t64(k1))) << 4) + half * 8));
  short8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return v;
}

struct NoHook { __device__ __forceinline__ void reads_done() const {} __device__ __forceinline__ void between(int) const {} };
// Hook: reads_done() runs once every fragment read of the slab has been ISSUED (the caller may wait for them and release the stage),
// between(q) after the q-th of the 8 groups of MFMAs (the caller's DMA pieces for the slab after next go there).
template <bool TA, bool TB, int BN_, bool CS = false, typename Hook = NoHook>
__device__ __forceinline__ void mma_slab(const char* sA, const char* sB, int wm, int wn, int lane, float4_t (&acc)[4][BN_ / 32],
                                         float4_t* accb = nullptr, bool do_cs = false, const Hook& hook = Hook()) {
  constexpr int NJ = BN_ / 32, WN = BN_ / 2;
  const int r = lane & 15, g = lane >> 4;
  // All fragment reads of the slab are issued first (LDS returns in order, so the first MFMAs start as soon as their
  // operands land and the rest of the reads fly under them).  Left to itself the compiler reused one A register quad and
  // serialised "ds_read -> wait -> 4 MFMA" eight times per slab: ~2000 cycles per slab for 512 cycles of MFMA.
  short8_t a[2][4], b[2][NJ];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (TA) a[kk][i] = frag_trans<128>(sA, wm * 64 + i * 16, kk * 32 + g * 8, r);
      else    a[kk][i] = frag_direct(sA, wm * 64 + i * 16 + r, kk * 4 + g);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (TB) b[kk][j] = frag_direct(sB, wn * WN + j * 16 + r, kk * 4 + g);
      else    b[kk][j] = frag_trans<BN_>(sB, wn * WN + j * 16, kk * 32 + g * 8, r);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  hook.reads_done();
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
      if constexpr (!std::is_same<Hook, NoHook>::value) {
        hook.between(kk * 4 + i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  if constexpr (CS) {
    if (do_cs) {  // all-ones A fragment: every row of the result is sum_k B[k, n] (the bias gradient's share of this slab)
      const short one = (short)0x3F80;
      const short8_t ones = {one, one, one, one, one, one, one, one};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < NJ; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b[kk][j], accb[j], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void unpack8(const uint4& a, float (&v)[8]) {  // 8 bf16 -> f32
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u); v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u); v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
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
    case TFASR_ACT_TANH_OUT: return 1.f - z * z;
    case TFASR_ACT_FACTOR: return z;
    default: return 1.f;
  }
}

// EPI selects which epilogue terms are COMPILED IN.  The fully generic epilogue (every term behind a runtime branch, tanh /
// sigmoid / f32 outputs included) is ~20k instructions and thrashes the instruction cache: a plain bias epilogue took 1700
// cycles per 16-row strip.  The step's common combinations get lean instantiations; anything else falls back to E_GEN.
enum : int { E_ACT = 1 /* swish(+prez) */, E_DACT = 2 /* * swish'(dact_z) */, E_DROP = 4, E_RES = 8, E_WS = 16 /* split-K partial -> workspace */, E_CSUM = 32 /* + column sums of B (bias gradient) */, E_LSE = 64 /* + log-softmax statistics of the output rows */, E_RGRAD = 128 /* re-computed logits -> RNN-T loss gradient */, E_GEN = 256, E_BNS = 512 /* + BatchNorm backward statistics of the output */, E_MUL = 1024 /* * dact_z (the stored derivative factor) */ };

// Persistent workgroups (2 per CU) walk a strided list of tiles.  Measured on [23808,256]x[256,1024] (cycle counters,
// tools/hwprobe/gemm_timing.hip): a tile spent 1900 cycles waiting for its first slab, ~2400 per further slab (the LDS-DMA
// round trip; the 512 cycles of MFMA per slab hide nothing) and 4300 in the epilogue.  So the NEXT tile's first two slabs
// are issued into the two (then idle) stages BEFORE this tile's epilogue: they land under it, and the epilogue's strip
// lives in its own 8.5 KiB so nothing waits for it.
// `bid` / `G` = this workgroup's index and the number of workgroups working on THIS product (blockIdx.x / gridDim.x for a plain
// launch; a slice of the grid inside a grouped launch, whose slices start at multiples of 8 so that bid & 7 is still the XCD).
// NST = LDS stages: 2 (two barriers per slab: the stage just read is refilled after the MFMAs) or 3 (one barrier per slab: slab s+2
// goes into the stage slab s-1 was read from, TWO slabs in flight under the MFMAs - for long-K products on HBM-cold operands
// (weight gradients), whose main loop runs at the DMA round trip per slab with a single slab in flight).
// FIXED: the caller chose this workgroup's tile itself: bid = tile index inside the gx x gy grid, G = its k-slice (grouped launch).
// SEG: K-segments (tfasr_gemm_args.seg_*): slab s of the k loop reads its operands at a table offset instead of s * BK.
// ONE: every workgroup owns exactly ONE tile (the grid is the tile list): no cross-tile prefetch, so the epilogue strips may lie OVER the
// (then dead) stage buffers - 48 KiB of LDS for the 64-column variant = 39 granules = THREE workgroups per CU (with its own strips it is
// 46 granules = two; 596 tiles on 512 persistent slots also ran as two rounds).
template <bool TA, bool TB, int BN_, int EPI, int NST = 2, bool FIXED = false, bool SEG = false, bool ONE = false>
__device__ __forceinline__ void gemm_fast_body(const tfasr_gemm_args& p, const int gx, const int gy, const int gz, const int ntiles, const int bid, const int G) {
  constexpr bool GEN = (EPI & E_GEN) != 0;
  constexpr bool C_ACT = GEN || (EPI & E_ACT), C_DACT = GEN || (EPI & E_DACT), C_DROP = GEN || (EPI & E_DROP), C_RES = GEN || (EPI & E_RES);
  constexpr bool C_WS = (EPI & E_WS) != 0;
  constexpr bool C_CS = (EPI & E_CSUM) != 0;
  constexpr bool C_LSE = (EPI & E_LSE) != 0;
  constexpr bool C_BNS = (EPI & E_BNS) != 0;
  constexpr bool C_MUL = (EPI & E_MUL) != 0;  // like E_DACT with the activation derivative already in dact_z (TFASR_ACT_FACTOR)
  constexpr int BN = BN_, NJ = BN_ / 32, WN = BN_ / 2;
  constexpr int STAGE_BYTES = A_BYTES + BN_ * BK * 2;
  constexpr int GI = 4 + BN_ / 32;  // DMA wave-instructions per slab per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (16 KiB A + 16|8 KiB B) + epilogue strips
  const int split = p.split_k > 1 ? p.split_k : 1;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int r = lane & 15, g = lane >> 4;
  int kchunk = (p.K + split - 1) / split;
  kchunk = ((kchunk + BK - 1) / BK) * BK;

  __shared__ long seg_tab[SEG ? 2 : 1][SEG ? 64 : 1];  // per-slab element offsets of op(A) / op(B) (K-segment mode)
  if constexpr (SEG) {
    const int nsl = p.K / BK;
    if ((int)threadIdx.x < nsl) {
      const int k = threadIdx.x * BK, sg = k / p.seg_k, within = k - sg * p.seg_k;
      seg_tab[0][threadIdx.x] = p.seg_a_off[sg] + (TA ? (long)within * p.lda : (long)within);
      seg_tab[1][threadIdx.x] = p.seg_b_off ? p.seg_b_off[sg] + (TB ? (long)within : (long)within * p.ldb) : (TB ? (long)k : (long)k * p.ldb);
    }
    __syncthreads();
  }
  struct Tile { const bf16_t* A; const bf16_t* B; long doff; int m0, n0, k_begin, k_end, nfull, tail, ks; const bf16_t* sa[4]; const bf16_t* sb[BN_ / 32]; };
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
    if constexpr (FIXED) {
      if (it > 0) return T;
      tx = bid % gx; ty = bid / gx; tz = G;
    } else if ((G & 7) == 0) {
      const int x = bid & 7, j = (bid >> 3) + it * (G >> 3);
      if (split > 1 && (gz & 7) == 0) {
        if (j >= (gz >> 3) * tpp) return T;
        tz = x + 8 * (j / tpp);
        const int t = j % tpp;
        tx = t % gx; ty = t / gx;
      } else {
        // round `it` covers tiles [it*G, (it+1)*G); XCD x takes the x-th eighth of it.  A PARTIAL last round (R < G tiles left)
        // is dealt in ragged eighths of R instead: with eighths of G its tiles all land on the first XCDs, two per CU, while
        // the other XCDs idle (760 tiles on 512 workgroups: the second round ran on XCDs 0-3 only).
        const int round0 = it * G, R = ntiles - round0, slot = bid >> 3;
        if (R <= 0) return T;
        int t;
        if (R >= G) t = round0 + x * (G >> 3) + slot;
        else {
          const int q = R >> 3, rem = R & 7;
          if (slot >= q + (x < rem ? 1 : 0)) return T;
          t = round0 + (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + slot;
        }
        tx = t % gx; const int rest = t / gx; ty = rest % gy; tz = rest / gy;
      }
    } else {  // G == ntiles (fewer tiles than resident slots): one round, ragged but bijective runs
      if (it > 0) return T;
      const int q = G >> 3, rem = G & 7, xcd = bid & 7, slot = bid >> 3;
      const int t = G >= 16 ? (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot : (int)bid;
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
    if (TA) prep_trans<128>(T.sa, T.A, p.lda, T.m0, p.M, T.k_begin, w, lane); else prep_direct<128>(T.sa, T.A, p.lda, T.m0, p.M, T.k_begin, w, lane);
    if (TB) prep_direct<BN_>(T.sb, T.B, p.ldb, T.n0, p.N, T.k_begin, w, lane); else prep_trans<BN_>(T.sb, T.B, p.ldb, T.n0, p.N, T.k_begin, w, lane);
    return T;
  };
  const long stepA = TA ? (long)BK * p.lda : (long)BK, stepB = TB ? (long)BK : (long)BK * p.ldb;  // elements per slab
  auto issue = [&](const Tile& T, int slab, int stage) {
    char* sA = smem + stage * STAGE_BYTES;
    char* sB = sA + A_BYTES;
    if constexpr (SEG) {
      issue_from<128>(sA, T.sa, seg_tab[0][slab], w);
      issue_from<BN_>(sB, T.sb, seg_tab[1][slab], w);
    } else {
      issue_from<128>(sA, T.sa, slab * stepA, w);
      issue_from<BN_>(sB, T.sb, slab * stepB, w);
    }
  };

  // piece q of this wave's GI pieces of a slab: q < 4 -> op(A), else op(B)
  auto issue_piece = [&](const Tile& T, int slab, int stage, int q) {
    char* sA = smem + stage * STAGE_BYTES;
    char* sB = sA + A_BYTES;
    long dA_, dB_;
    if constexpr (SEG) { dA_ = seg_tab[0][slab]; dB_ = seg_tab[1][slab]; }
    else { dA_ = slab * stepA; dB_ = slab * stepB; }
    if (q < 4) glds16(T.sa[q] + dA_, sA + __builtin_amdgcn_readfirstlane((w * 4 + q) * 1024));
    else glds16(T.sb[q - 4] + dB_, sB + __builtin_amdgcn_readfirstlane((w * (BN_ / 32) + (q - 4)) * 1024));
  };
  Tile cur = tile_of(0);
  if (cur.nfull < 0) return;
  if (cur.nfull > 0) issue(cur, 0, 0);
  if (cur.nfull > 1) issue(cur, 1, 1);
  bool drained = false;  // true after a tile boundary: slabs 0 and 1 of `cur` have landed

  for (int it = 0;; ++it) {
#ifdef TFASR_GEMM_TIMING
    const long long t_start = __builtin_readcyclecounter();
#endif
    float4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    float4_t accb[C_CS ? NJ : 1];
#pragma unroll
    for (int j = 0; j < (C_CS ? NJ : 1); ++j) accb[j] = float4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = C_CS && p.colsum && cur.m0 == 0 && wm == 0;  // first row of tiles, the two waves that cover its columns

    // ---- the swish' argument of the whole tile: inside each 16-row strip it cost one dependent global round trip per strip, 8 per tile (FFN
    // data gradient 43.1 -> 37.8 us once it was loaded behind the slab loop, round 4).  ONE (a workgroup owns one tile): requested in FRONT of
    // the slab loop, behind the tile's first two slabs - the compiler-visible vmcnt(0) below waits for all of them together, so the round
    // trip lies under the first slabs' DMA instead of between the slab loop and the epilogue (16 registers held through the loop).
    // Only the compiled swish' epilogues: the residual variants did not gain and the generic one started to spill.
    constexpr int PF_LPRW = (BN_ / 2) / 8, PF_RPP = 64 / PF_LPRW, PF_NPASS = 16 / PF_RPP;
    constexpr bool PF_Z = (C_DACT || C_MUL) && !GEN && !C_WS;
    [[maybe_unused]] uint4 pf_z[PF_Z ? 4 : 1][PF_Z ? PF_NPASS : 1];
    [[maybe_unused]] bool pf_ok = false;
    auto prefetch_z = [&]() {
      if constexpr (PF_Z) {
        if (!p.accumulate) {
          const int prow = lane / PF_LPRW, col0 = cur.n0 + wn * (BN_ / 2) + (lane % PF_LPRW) * 8;
          pf_ok = ((p.ldd & 7) == 0) && ((cur.doff & 7) == 0) && (col0 + 8 <= p.N) &&
                  p.dact_z && ((((uintptr_t)p.dact_z) & 15) == 0);
          if (pf_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int h = 0; h < PF_NPASS; ++h) {
                const int row = min(cur.m0 + wm * 64 + i * 16 + h * PF_RPP + prow, p.M - 1);
                const long idx0 = cur.doff + (long)row * p.ldd + col0;
                pf_z[i][h] = *reinterpret_cast<const uint4*>((const bf16_t*)p.dact_z + idx0);
              }
          }
        }
      }
    };
    if constexpr (ONE) prefetch_z();
    const int n = cur.nfull;
    // compiler-visible vmcnt(0) in front of the slab loop (its bookkeeping otherwise carries a pending vector-memory event into the loop
    // and puts its own s_waitcnt vmcnt(0) behind the first fragment reads of EVERY slab, draining the prefetch); costs the first tile the
    // wait for its second slab, later tiles nothing (the tile boundary has drained the queue already)
    __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef TFASR_GEMM_TIMING
    long long ph[5] = {0, 0, 0, 0, 0};
#define TFASR_TICK(k) { const long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - tp; tp = t_; }
#else
#define TFASR_TICK(k)
#endif
    if constexpr (NST == 3) {
      for (int s = 0; s < n; ++s) {
        const int stage = s % 3;
        if (!(drained && s < 2)) {
          if (s + 1 < n) { if (GI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // slab s is visible to every wave, and every wave is done reading slab s-1's stage
        if (s + 2 < n) issue(cur, s + 2, (s + 2) % 3);
        mma_slab<TA, TB, BN_, C_CS>(smem + stage * STAGE_BYTES, smem + stage * STAGE_BYTES + A_BYTES, wm, wn, lane, acc, accb, do_cs);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    } else {
      for (int s = 0; s < n; ++s) {
        const int stage = s & 1;
  #ifdef TFASR_GEMM_TIMING
        long long tp = __builtin_readcyclecounter();
  #endif
        if (!(drained && s < 2)) {
          // this wave's DMA pieces of slab s have landed; slab s+1 (when it exists) stays in flight
          if (s + 1 < n) { if (GI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        TFASR_TICK(0)
        __builtin_amdgcn_s_barrier();
        TFASR_TICK(1)
#ifdef TFASR_FAST_NO_ILV
        mma_slab<TA, TB, BN_, C_CS>(smem + stage * STAGE_BYTES, smem + stage * STAGE_BYTES + A_BYTES, wm, wn, lane, acc, accb, do_cs);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TFASR_TICK(2)
        __builtin_amdgcn_s_barrier();  // every wave is done reading this stage before it is refilled
        TFASR_TICK(3)
        if (s + 2 < n) issue(cur, s + 2, stage);
        TFASR_TICK(4)
#else
        // Every fragment of the slab is read into registers FIRST; once all waves' reads have returned the stage is free, and the pieces
        // of slab s+2 are issued BETWEEN the slab's MFMAs (a piece costs its wave ~90 clocks of issue time: 8 of them in a row after the
        // MFMAs were a serial 730 clocks per slab next to 512 clocks of matrix work)
        struct Hook {
          const decltype(issue_piece)& ip; const Tile& T; int slab, stage; bool on;
          __device__ __forceinline__ void reads_done() const {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave has its fragments: the stage may be refilled
          }
          __device__ __forceinline__ void between(int q) const { if (on && q < GI) ip(T, slab, stage, q); }
        };
        const Hook hook{issue_piece, cur, s + 2, stage, s + 2 < n};
        mma_slab<TA, TB, BN_, C_CS, Hook>(smem + stage * STAGE_BYTES, smem + stage * STAGE_BYTES + A_BYTES, wm, wn, lane, acc, accb, do_cs, hook);
        TFASR_TICK(2)
#endif
      }
    }
    if (cur.tail) {
      char* sA = smem;
      char* sB = smem + A_BYTES;
      const int kt = cur.k_begin + n * BK;
      if (TA) tail_trans<128>(sA, cur.A, p.lda, cur.m0, p.M, kt, cur.k_end); else tail_direct<128>(sA, cur.A, p.lda, cur.m0, p.M, kt, cur.k_end);
      if (TB) tail_direct<BN_>(sB, cur.B, p.ldb, cur.n0, p.N, kt, cur.k_end); else tail_trans<BN_>(sB, cur.B, p.ldb, cur.n0, p.N, kt, cur.k_end);
      __syncthreads();
      mma_slab<TA, TB, BN_, C_CS>(sA, sB, wm, wn, lane, acc, accb, do_cs);
      __syncthreads();
    }
    if constexpr (!ONE) prefetch_z();
    // ---- cross-tile prefetch: both stages are idle now ----
    Tile nxt;
    if constexpr (ONE) nxt.nfull = -1;
    else {
      nxt = tile_of(it + 1);
      if (nxt.nfull > 0) issue(nxt, 0, 0);
      if (nxt.nfull > 1) issue(nxt, 1, 1);
    }
#ifdef TFASR_GEMM_TIMING
    const long long t_main = __builtin_readcyclecounter();
#endif

    // ---- epilogue ----
    const int m0 = cur.m0, n0 = cur.n0;
    const long doff = cur.doff;
    const bool first_split = (cur.ks == 0);
    bf16_t* Dt = (bf16_t*)p.D + doff;
    float* Df = (float*)p.D + doff;
    const bf16_t* res = p.res ? (const bf16_t*)p.res + doff : nullptr;
    const bf16_t* dz = p.dact_z ? (const bf16_t*)p.dact_z + doff : nullptr;
    bf16_t* prez = p.prez ? (bf16_t*)p.prez + doff : nullptr;
    if constexpr (C_CS) {
      if (do_cs && g == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn * WN + j * 16 + r;
          if (col < p.N) atomicAdd(p.colsum + col, p.alpha * accb[j][0]);
        }
      }
    }
    if (!C_WS && p.accumulate) {
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
      float* sc = reinterpret_cast<float*>(smem + (ONE ? 0 : NST * STAGE_BYTES)) + w * (RPP * SLD);  // (ONE: over the dead stages)
      const int prow = lane / LPRW, c8 = (lane % LPRW) * 8;
      const int col0 = n0 + wn * WN + c8;
      const bool vec_ok = C_WS ? ((p.N & 7) == 0) : (((p.ldd & 7) == 0) && ((((uintptr_t)p.D) & 15) == 0) && ((doff & 7) == 0));
      const bool full = vec_ok && (col0 + 8 <= p.N);
      float* wsp = C_WS ? p.ws + (long)cur.ks * p.M * p.N : nullptr;  // this k-slice's [M, N] partial (row stride N)
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
      int lse_lab[4][2];
      if constexpr (C_LSE) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int hh = 0; hh < NPASS; ++hh) {
            const int rw = m0 + wm * 64 + ii * 16 + hh * RPP + prow;
            lse_lab[ii][hh] = (rw < p.M) ? p.row_label[rw] : -1;
          }
      }
      // E_BNS: BatchNorm backward statistics of this lane's 8 columns over the rows it passes (tfasr_gemm_args.bns_*); the BatchNorm inputs
      // of all its rows are requested in one batch in front of the strips
      [[maybe_unused]] float bs0[C_BNS ? 8 : 1], bs1[C_BNS ? 8 : 1], fsc[C_BNS ? 8 : 1], fsh[C_BNS ? 8 : 1], frs[C_BNS ? 8 : 1], fm2[C_BNS ? 8 : 1];
      [[maybe_unused]] uint4 bnx[C_BNS ? 4 : 1][C_BNS ? NPASS : 1];
      // (bns_c channels: column n of the product belongs to channel n % bns_c - the subsampling's linear layer, whose columns are
      // (frequency, channel) pairs; 0 = one channel per column)
      [[maybe_unused]] const int bnc = C_BNS ? (p.bns_c > 0 ? p.bns_c : p.N) : 0;
      [[maybe_unused]] const int bcol = C_BNS ? col0 % bnc : 0;
      if constexpr (C_BNS) {
        const bool cok = col0 + 8 <= p.N;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          bs0[q] = 0.f; bs1[q] = 0.f;
          const int cc = cok ? bcol + q : 0;
          const float mean = p.bns_fin[cc], rstd = p.bns_fin[bnc + cc];
          fsc[q] = p.bns_fin[2 * bnc + cc]; fsh[q] = p.bns_fin[3 * bnc + cc]; frs[q] = rstd; fm2[q] = -mean * rstd;
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int hh = 0; hh < NPASS; ++hh) {
            const int rw = m0 + wm * 64 + ii * 16 + hh * RPP + prow;
            bnx[ii][hh] = make_uint4(0u, 0u, 0u, 0u);
            if (rw < p.M && cok) bnx[ii][hh] = *reinterpret_cast<const uint4*>((const bf16_t*)p.bns_x + doff + (long)rw * p.ldd + col0);
          }
      }
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
          if constexpr (C_BNS) {
            if (full) {
              float xv[8];
              unpack8(bnx[i][h], xv);
#pragma unroll
              for (int q = 0; q < 8; q += 2) {
                const uint32_t pk = pack2_bf16(x[q], x[q + 1]);  // the gradient as the second pass would read it back
                const float d0 = __uint_as_float(pk << 16), d1 = __uint_as_float(pk & 0xffff0000u);
                const float dz0 = d0 * dswishf_(xv[q] * fsc[q] + fsh[q]), dz1 = d1 * dswishf_(xv[q + 1] * fsc[q + 1] + fsh[q + 1]);
                bs0[q] += dz0; bs1[q] += dz0 * (xv[q] * frs[q] + fm2[q]);
                bs0[q + 1] += dz1; bs1[q + 1] += dz1 * (xv[q + 1] * frs[q + 1] + fm2[q + 1]);
              }
            }
          }
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
          if constexpr (C_DACT || C_MUL) if (dz) {
            float z[8];
            bool got = false;
            if constexpr (PF_Z) {
              if (pf_ok) { unpack8(pf_z[i][h], z); got = true; }
            }
            if (!got) {
              if (full) ld8(dz + idx0, z);
              else
_Pragma("unroll")
                for (int q = 0; q < 8; ++q) z[q] = (col0 + q < p.N) ? bf16_to_f32(dz[idx0 + q]) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] *= GEN ? dact_f(z[q], p.dact) : (C_MUL ? z[q] : dswishf_(z[q]));
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
          if constexpr (C_WS) {
            const long wi = (long)row * p.N + col0;
            if (full) st8(wsp + wi, x);
            else
_Pragma("unroll")
              for (int q = 0; q < 8; ++q) if (col0 + q < p.N) wsp[wi + q] = x[q];
          } else if (GEN && p.out_f32) {
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
        if constexpr (C_LSE) {
          // log-softmax statistics of this row's 64-column slice, from the f32 values (alpha * acc + bias): row max over the
          // slice's 8 lanes first, then one exp2 per element against the common max and a plain sum (no exp in the merge).
          // Every lane of the row takes part in the shuffles (lanes past N or M contribute an empty set).
          const bool rok = row < p.M;
          const float L2E = 1.4426950408889634f;
          float mloc = -INFINITY;
          if (rok && full) {
#pragma unroll
            for (int q = 0; q < 8; ++q) mloc = fmaxf(mloc, x[q]);
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) if (rok && col0 + q < p.N) mloc = fmaxf(mloc, x[q]);
          }
#pragma unroll
          for (int o = 1; o < LPRW; o <<= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, o, 64));
          const float mb = mloc * L2E;
          float sloc = 0.f;
          if (rok && full) {
#pragma unroll
            for (int q = 0; q < 8; ++q) sloc += __builtin_amdgcn_exp2f(x[q] * L2E - mb);
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) if (rok && col0 + q < p.N) sloc += __builtin_amdgcn_exp2f(x[q] * L2E - mb);
          }
#pragma unroll
          for (int o = 1; o < LPRW; o <<= 1) sloc += __shfl_xor(sloc, o, 64);
          if (rok) {
            if ((lane % LPRW) == 0) {
              float2* dst = reinterpret_cast<float2*>(p.lse_part) + (long)row * p.lse_parts + ((n0 + wn * WN) >> 6);
              *dst = make_float2(mloc, sloc);
            }
            if (col0 == 0) p.pick[2L * row] = x[0];
            const int lab = lse_lab[i][h];  // preloaded for all strips (a load inside the strip would stall it for a round trip)
            if (lab >= col0 && lab < col0 + 8) {
              float v = x[0];
#pragma unroll
              for (int q = 1; q < 8; ++q) v = (lab - col0 == q) ? x[q] : v;
              p.pick[2L * row + 1] = v;
            }
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
      if constexpr (C_BNS) {
        // over the RPP lanes that share the columns (lane = row * LPRW + column group): inside a 16-lane row by rotations, across rows by swaps
        static_assert(LPRW == 4 || LPRW == 8, "rows of 4 or 8 lanes");
        float* out = p.bns_out + (size_t)(bid % (p.bns_copies > 0 ? p.bns_copies : 1)) * 2 * bnc;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float a0 = bs0[q], a1 = bs1[q];
          a0 += dpp_mov<0x128>(a0); a1 += dpp_mov<0x128>(a1);
          if constexpr (LPRW == 4) { a0 += dpp_mov<0x124>(a0); a1 += dpp_mov<0x124>(a1); }
          a0 = xor32_sum(xor16_sum(a0)); a1 = xor32_sum(xor16_sum(a1));
          if (lane < LPRW && col0 + 8 <= p.N) { atomicAdd(out + bcol + q, a0); atomicAdd(out + bnc + bcol + q, a1); }
        }
      }
    }
#ifdef TFASR_GEMM_TIMING
    if (threadIdx.x == 0 && it == 0) {
      const long long t_end = __builtin_readcyclecounter();
      long long* o = g_gemm_timing + 4L * bid;
      o[0] = t_start; o[1] = 0; o[2] = t_main - t_start; o[3] = t_end - t_start;
    }
    if (threadIdx.x == 0 && it == 1) {
      const long long t_end = __builtin_readcyclecounter();
      long long* o = g_gemm_timing + 4L * 32768 + 2L * bid;
      o[0] = t_main - t_start; o[1] = t_end - t_start;
    }
    if (threadIdx.x == 0 && it == 0) {
      long long* o = g_gemm_timing + 6L * 32768 + 5L * bid;
      for (int k = 0; k < 5; ++k) o[k] = ph[k];
    }
#endif
    if (nxt.nfull < 0) break;
    // tile boundary: the epilogue's stores / loads are mixed into the vector-memory queue, so counted waits are void until
    // it drains once (the prefetched slabs had the whole epilogue to land)
    // (the BUILTIN form, so that the compiler's own wait-count bookkeeping sees the queue empty here: with an asm wait it still believed
    // the epilogue's stores pending and put its own s_waitcnt vmcnt(0) - for the re-use of their data registers - INSIDE the next tile's
    // slab loop, where it drained the DMA prefetch once per slab)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("" ::: "memory");
    drained = true;
    cur = nxt;
  }
}

template <bool TA, bool TB, int BN_, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_fast_kernel(const tfasr_gemm_args p, const int gx, const int gy, const int gz, const int ntiles) {
  gemm_fast_body<TA, TB, BN_, EPI>(p, gx, gy, gz, ntiles, (int)blockIdx.x, (int)gridDim.x);
}
// one tile per workgroup, three workgroups per CU (64-column tiles: <= 168 registers, 48 KiB of LDS)
template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(256, 3) void gemm_fast_one_kernel(const tfasr_gemm_args p, const int gx, const int gy, const int gz, const int ntiles) {
  gemm_fast_body<TA, TB, 64, EPI, 2, false, false, true>(p, gx, gy, gz, ntiles, (int)blockIdx.x, (int)gridDim.x);
}
// K-segmented A / B operands (convolution taps as row shifts): same body, slab offsets from a table
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_seg_kernel(const tfasr_gemm_args p, const int gx, const int gy, const int gz, const int ntiles) {
  gemm_fast_body<TA, TB, 128, 0, 2, false, true>(p, gx, gy, gz, ntiles, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped weight gradients: the ~9 Dense-layer gradients  gW += x^T dy  of one Conformer block (K = B*T rows, a few 128x64 output
// tiles each) in ONE launch.  Launched one by one each of them has to split K 8-16 ways to occupy the chip - and pays
// `tiles x split` f32 atomics at a flat 320 G/s plus a launch ramp/drain - while together their ~190 tiles fill the 512
// resident workgroup slots with a split of 2.  Every workgroup of the grid belongs to one product (a contiguous slice of
// blockIdx.x starting at a multiple of 8) and runs gemm_fast_body on it unchanged.
constexpr int GROUP_MAX = 10;
constexpr int GROUP_UNITS = 20;  // (product, k-slice) units per XCD
struct GroupArgs {
  tfasr_gemm_args p[GROUP_MAX];
  int gx[GROUP_MAX], gy[GROUP_MAX];
  // XCD x (= blockIdx.x & 7) runs nunit[x] units; unit u covers slots [j0[x][u], j0[x][u+1]) of that XCD (slot = blockIdx.x >> 3)
  short uq[8][GROUP_UNITS], uks[8][GROUP_UNITS];
  int j0[8][GROUP_UNITS + 1];
  int nunit[8];
};
// A unit = ALL output tiles of one product for one k-slice, resident on ONE XCD at the same time: the tiles sharing an operand
// panel march through k together, so the panel comes out of HBM once and its re-reads hit that XCD's L2.  (With the tiles of a
// product dealt over all XCDs the panels were re-fetched per XCD: 1.5 GB of fabric traffic per block for 0.32 GB of operands,
// and the launch ran at the fabric's rate, 156 us; 3 LDS stages instead of 2 changed nothing.)
template <int EPI, int NST, int BN_>
__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(const GroupArgs ga) {
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int nu = ga.nunit[x];
  int u = -1;
  for (int i = 0; i < nu; ++i)
    if (j >= ga.j0[x][i] && j < ga.j0[x][i + 1]) u = i;
  u = __builtin_amdgcn_readfirstlane(u);
  if (u < 0) return;
  const int q = ga.uq[x][u], ks = ga.uks[x][u], t = j - ga.j0[x][u];
  const tfasr_gemm_args* pp = ga.p + q;
  gemm_fast_body<true, false, BN_, EPI, NST, true>(*pp, ga.gx[q], ga.gy[q], 1, ga.gx[q] * ga.gy[q], t, ks);
}

// second pass of the workspace split-K: D[m, n] += sum_s ws[s][m][n]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ D, int M, int N, int ldd, int split) {
  const long mn = (long)M * N;
  if ((N & 3) == 0 && (ldd & 3) == 0 && ((((uintptr_t)D) & 15) == 0)) {
    const int n4 = N >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < mn / 4; i += (long)gridDim.x * blockDim.x) {
      float4 acc = *reinterpret_cast<const float4*>(ws + 4 * i);
      for (int s2 = 1; s2 < split; ++s2) {
        const float4 t = *reinterpret_cast<const float4*>(ws + s2 * mn + 4 * i);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
      const long m = i / n4;
      float4* d = reinterpret_cast<float4*>(D + m * ldd + 4 * (i - m * n4));
      float4 o = *d;
      o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
      *d = o;
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (long)gridDim.x * blockDim.x) {
      float acc = 0.f;
      for (int s2 = 0; s2 < split; ++s2) acc += ws[s2 * mn + i];
      const long m = i / N;
      D[m * ldd + (i - m * N)] += acc;
    }
  }
}

int num_cus();

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
  // 2 resident workgroups per CU; TFASR_GEMM_SLOTS=1 is an experiment hook: one persistent workgroup per CU walking more tiles
  static const int per_cu = 2;
  const int slots = per_cu * num_cus();
  if constexpr (BN_ == 64 && (EPI & (E_WS | E_CSUM | E_LSE)) == 0) {
    // every tile its own workgroup, three per CU, when the tile list fits three per CU at once (TFASR_GEMM_ONE=0: the persistent kernel)
    static const bool one_off = false;
    static const long one_max = (1L << 30);  // probe: upper bound in tiles per CU
    if (!one_off && a.split_k <= 1 && ntiles > slots && ntiles <= one_max * num_cus()) {
      constexpr int SMEM1 = 2 * (A_BYTES + 64 * BK * 2);
      TFASR_KLAUNCH((gemm_fast_one_kernel<TA, TB, EPI>), dim3((unsigned)ntiles), dim3(256), SMEM1, stream, a, (int)tiles.x, (int)tiles.y, (int)tiles.z, (int)ntiles);
      TFASR_CHECK_LAUNCH();
      return TFASR_STATUS_SUCCESS;
    }
  }
  int G = (int)(ntiles < slots ? ntiles : slots);
  if (ntiles >= slots) G &= ~7;
  constexpr int SMEM = 2 * (A_BYTES + BN_ * BK * 2) + 4 * (64 / (BN_ / 16)) * (BN_ / 2 + 4) * 4;  // stages + 4 waves' strips
  TFASR_KLAUNCH((gemm_fast_kernel<TA, TB, BN_, EPI>), dim3(G), dim3(256), SMEM, stream, a, (int)tiles.x, (int)tiles.y, (int)tiles.z, (int)ntiles);
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

#include "gemm_big.h"
#include "ffn_fused.h"
#include "dense_ln.h"

template <bool TA, bool TB>
int launch_one(const tfasr_gemm_args& a, hipStream_t stream) {
  const int split = a.split_k > 1 ? a.split_k : 1;
  // 128x64 tiles: per-head attention products (N = head size: halves the wasted B tile) and every product whose 128x128 tiling
  // would leave at most one workgroup per CU (the N = 256 Dense layers of a Conformer block: 190 tiles): with a single resident
  // workgroup the DMA issue, the fragment reads and the MFMAs of a slab serialise (1530 cycles / slab measured); two 128x64
  // workgroups per CU overlap them (14.2 vs 16.9 us on [12096,256,1024]).  TFASR_GEMM_BN64=0 restores the old rule.
  static const bool bn64_off = false;
  const long t128 = (long)((a.N + 127) / 128) * ((a.M + BM - 1) / BM) * a.nb1 * a.nb2 * split;
  // Round 4: every product the one-tile kernel takes (no split-K, no accumulation / column sums; see bn64_lim below) runs as 64-column tiles, ONE per workgroup,
  // three workgroups per CU (gemm_fast_one_kernel), whatever its size: 744 128-wide tiles on 512 persistent slots are two rounds with half
  // the chip idle in the second, 1 488 narrow tiles on 768 dynamic slots are 1.94.  Same-box A/B: [rows,256]x[256,768|512] 17.8 -> 17.0 us,
  // the FFN data gradient with its swish' + dropout epilogue 38.6 -> 34.3, N = 256 products 17.0 -> 16.3 / 16.2 -> 13.0; M 23.23 -> 23.01 ms/step,
  // S 12.17 -> 12.01.  TFASR_GEMM_BN64_T=<128-wide tiles> restores a threshold (256 = round 3's rule).
  static const long bn64_thr = (1L << 40);
  // (above 1 x CUs only products the one-tile kernel takes)
  // ... and with K <= 256 or N <= 256 (the Conformer's d = 256 layers: four slabs per tile, or two 128-wide column tiles).  With both >= 512
  // (ContextNet's 512 - 1280-channel layers) the 128-wide persistent kernel amortises a tile better (cross-tile prefetch, half the operand
  // bytes per flop): 33.3 vs 35.5 us per launch there, ContextNet-L 45.0 vs 45.4 ms/step.
  const long bn64_lim = (split <= 1 && !a.accumulate && !a.colsum && std::min(a.K, a.N) <= 256) ? bn64_thr : std::min(bn64_thr, (long)num_cus());
  const bool narrow = !a.lse_part && !a.seg_a_off && (a.N <= 64 || (!bn64_off && t128 <= bn64_lim && a.N > 64 && !(a.accumulate && a.ws)));
  const int bn = narrow ? 64 : 128;
  dim3 grid((a.N + bn - 1) / bn, (a.M + BM - 1) / BM, a.nb1 * a.nb2 * split);
  if ((long)grid.x * grid.y * grid.z > 0x7fffffffL) return TFASR_STATUS_INVALID_VALUE;
  // epilogue terms this call needs; accumulate (atomics from fragments) needs none of them
  int need = 0;
  bool generic = false;
  if (!a.accumulate) {
    if (a.out_f32) generic = true;
    if (a.act != TFASR_ACT_NONE || a.prez) { if (a.act == TFASR_ACT_SWISH) need |= E_ACT; else generic = true; }
    if (a.dact_z) { if (a.dact == TFASR_ACT_SWISH) need |= E_DACT; else if (a.dact == TFASR_ACT_FACTOR) need |= E_MUL; else generic = true; }
    if (a.drop_p > 0.f) need |= E_DROP;
    if (a.res) need |= E_RES;
  }
  if (a.bns_out) {  // BatchNorm backward statistics in the epilogue: the plain NT product on 64-column tiles, or nothing at all
    if constexpr (!TA && TB) {
      if (narrow && !generic && need == 0 && !a.accumulate && !a.bias && split == 1 && a.nb1 * a.nb2 == 1 && a.bns_x && a.bns_fin && (a.N & 7) == 0 && (a.ldd & 7) == 0 &&
          (a.bns_c == 0 || (a.bns_c > 0 && (a.bns_c & 7) == 0 && a.N % a.bns_c == 0)) &&
          (((uintptr_t)a.D | (uintptr_t)a.bns_x) & 15) == 0 && !a.colsum && !a.lse_part && !a.seg_a_off && !a.rgrad_coef)
        return launch_epi<TA, TB, 64, E_BNS>(a, grid, stream);
    }
    return TFASR_STATUS_UNSUPPORTED;
  }
  if constexpr (!TA) {  // thousands of tiles: 256-row tiles, one 8-wave workgroup per CU (gemm_big.h)
    const bool tanh_out = !a.accumulate && a.dact_z && a.dact == TFASR_ACT_TANH_OUT;  // the only epilogue term gemm_big knows beyond bias
    const bool other = a.out_f32 || a.act != TFASR_ACT_NONE || a.prez || (a.dact_z && !tanh_out);
    const int st = launch_big<TB>(a, other, need, tanh_out, stream);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  if (a.rgrad_coef || !a.D) return TFASR_STATUS_UNSUPPORTED;  // gradient epilogue / statistics-only projection: 256-row kernel only
  // (measured: NOT faster than the atomics - 34.8 vs 33.0 us on the [256,1024,19040] weight gradient, +5 ms on the step from
  // the extra workspace traffic - so callers only pass a workspace when TFASR_SPLITK_WS=1; kept as the deterministic option)
  if (a.accumulate && split > 1 && a.nb1 * a.nb2 == 1 && a.ws && a.ws_elems >= (long)split * a.M * a.N && !narrow && !a.colsum) {
    const int st = launch_epi<TA, TB, 128, E_WS>(a, grid, stream);
    if (st != TFASR_STATUS_SUCCESS) return st;
    const long work = ((long)a.M * a.N + 3) / 4;
    const int rg = (int)(work / 256 + 1 < 2048 ? work / 256 + 1 : 2048);
    TFASR_KLAUNCH(splitk_reduce_kernel, dim3(rg), dim3(256), 0, stream, (const float*)a.ws, (float*)a.D, a.M, a.N, a.ldd, split);
    TFASR_CHECK_LAUNCH();
    return TFASR_STATUS_SUCCESS;
  }
  if (a.colsum) {
    if constexpr (TA && !TB) {
      if (a.accumulate && a.N > 64) return narrow ? launch_epi<TA, TB, 64, E_CSUM>(a, grid, stream) : launch_epi<TA, TB, 128, E_CSUM>(a, grid, stream);
    }
    return TFASR_STATUS_UNSUPPORTED;  // tfasr_gemm falls back to a separate column-sum pass
  }
  if (narrow) {
    if (!generic && need == 0) return launch_epi<TA, TB, 64, 0>(a, grid, stream);
    if constexpr (!TA && !TB) {
      if (!generic && need == E_RES) return launch_epi<TA, TB, 64, E_RES>(a, grid, stream);
      if (!generic && need == (E_RES | E_DROP)) return launch_epi<TA, TB, 64, E_RES | E_DROP>(a, grid, stream);
    }
    // Experiment hook (not reachable with the default threshold): 128x64 tiles for layers of MORE than one workgroup per CU need
    // the compiled activation epilogues, or they fall to the generic one and lose (measured with TFASR_GEMM_BN64_T=1024).
    if (t128 > num_cus() && a.N > 64 && !generic) {
      if constexpr (!TA && !TB) {
        if (need == E_ACT) return launch_epi<TA, TB, 64, E_ACT>(a, grid, stream);
        if (need == (E_ACT | E_DROP)) return launch_epi<TA, TB, 64, E_ACT | E_DROP>(a, grid, stream);
      }
      if constexpr (!TA && TB) {
        if (need == E_DACT) return launch_epi<TA, TB, 64, E_DACT>(a, grid, stream);
        if (need == (E_DACT | E_DROP)) return launch_epi<TA, TB, 64, E_DACT | E_DROP>(a, grid, stream);
        if (need == E_MUL) return launch_epi<TA, TB, 64, E_MUL>(a, grid, stream);
      }
    }
    return launch_epi<TA, TB, 64, E_GEN>(a, grid, stream);
  }
  if (a.seg_a_off) {  // K-segmented operands: conv2 forward (NN) / data gradient (NT) over the haloed space-to-depth layout
    if constexpr (!TA) {
      if (!generic && need == 0 && !a.accumulate && split == 1 && a.nb1 * a.nb2 == 1 && a.seg_k > 0 && (a.seg_k % BK) == 0 && (a.K % a.seg_k) == 0 &&
          a.K / BK <= 64 && !a.lse_part) {
        dim3 g128((a.N + 127) / 128, (a.M + BM - 1) / BM, 1);
        const long ntiles = (long)g128.x * g128.y;
        const int slots = 2 * num_cus();
        int G = (int)(ntiles < slots ? ntiles : slots);
        if (ntiles >= slots) G &= ~7;
        constexpr int SMEM = 2 * (A_BYTES + 128 * BK * 2) + 4 * (64 / (128 / 16)) * (128 / 2 + 4) * 4;
        TFASR_KLAUNCH((gemm_seg_kernel<TA, TB>), dim3(G), dim3(256), SMEM, stream, a, (int)g128.x, (int)g128.y, 1, (int)ntiles);
        TFASR_CHECK_LAUNCH();
        return TFASR_STATUS_SUCCESS;
      }
    }
    return TFASR_STATUS_UNSUPPORTED;
  }
  if (a.lse_part) {  // joint vocabulary projection with fused log-softmax statistics
    if constexpr (!TA && !TB) {
      if (!generic && need == 0 && !a.accumulate && a.nb1 * a.nb2 == 1 && a.row_label && a.pick && a.lse_parts == ((a.N + 127) / 128) * 2)
        return launch_epi<TA, TB, 128, E_LSE>(a, grid, stream);
    }
    return TFASR_STATUS_UNSUPPORTED;
  }
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
      if (need == E_MUL) return launch_epi<TA, TB, 128, E_MUL>(a, grid, stream);
    }
  }
  return launch_epi<TA, TB, 128, E_GEN>(a, grid, stream);
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool group_eligible(const tfasr_gemm_args& a) {
  return a.dtype == TFASR_BF16 && a.trans_a && !a.trans_b && a.accumulate && a.out_f32 && a.nb1 * a.nb2 <= 1 && !a.ws && !a.bias && !a.res &&
         !a.dact_z && !a.prez && a.act == TFASR_ACT_NONE && a.drop_p == 0.f && al16(a.A) && al16(a.B) && !(a.lda & 7) && !(a.ldb & 7) &&
         ((a.M + 7) & ~7) <= a.lda && ((a.N + 7) & ~7) <= a.ldb && a.K >= 8 && a.N > 64 && a.M > 0;
}

}  // namespace

// set around a launch that shares the chip with another stream's dependent chain (block.hip: the weight-gradient stream); one host thread
// per process queues the launches of a device, so a plain global is enough
int g_tfasr_group_beside = 0;

// One launch for up to GROUP_MAX weight-gradient products (see wgrad_group_kernel).  UNSUPPORTED -> the caller launches them one by one.
int tfasr_gemm_group_fast_try(const tfasr_gemm_args* a, int n, hipStream_t stream) {
  if (n < 2 || n > GROUP_MAX) return TFASR_STATUS_UNSUPPORTED;
  for (int i = 0; i < n; ++i)
    if (!group_eligible(a[i])) return TFASR_STATUS_UNSUPPORTED;
  GroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  long total = 0;
  static const int bn = 128;
  for (int i = 0; i < n; ++i) {
    ga.gx[i] = (a[i].N + bn - 1) / bn;
    ga.gy[i] = (a[i].M + BM - 1) / BM;
    total += (long)ga.gx[i] * ga.gy[i];
  }
  // k-split shared by the group: fill the resident slots (2 workgroups per CU) once; TFASR_GROUP_SLOTS overrides the slot count.
  // A group that runs BESIDE another stream's dependent chain (the block executor's weight-gradient stream: g_tfasr_group_beside, set
  // by the caller around the launch) takes 1.25 slots per CU (256 / 320 / 512 slots for those groups alone: -0.04 / -0.09 / 0 ms per step).
  // A group of LONG products in line (the nine shifted conv2 weight gradients: K = B T2 F2 = 500 k rows, 36 tiles) also wants about one
  // workgroup per CU: with 512 slots its 14 k-slices per tile were 8 M atomics and the slices' workgroups drifted apart in L2 - 256 / 320
  // slots for every group of the step: -0.28 / -0.33 ms per step against 512, of which the block groups beside the chain are 0.04.
  static const long env_slots = 0;
  long kmax = 0;
  for (int i = 0; i < n; ++i) kmax = a[i].K > kmax ? a[i].K : kmax;
  const long slots = env_slots > 0 ? env_slots : ((g_tfasr_group_beside || kmax >= 131072) ? num_cus() * 5L / 4 : 2L * num_cus());
  long split = slots / (total > 0 ? total : 1);
  if (split < 1) split = 1;
  {
    static const bool dbg = false;
    static int dbg_n = 0;
    if (dbg && dbg_n < 6) { fprintf(stderr, "[group] n %d tiles %ld beside %d slots %ld split %ld\n", n, total, g_tfasr_group_beside, slots, split); ++dbg_n; }
  }
  // units (product, k-slice), largest first, each onto the XCD with the fewest slots taken so far
  struct U { int q, ks, tiles; };
  U units[GROUP_MAX * 64];
  int nu = 0;
  for (int i = 0; i < n; ++i) {
    long sp = split;
    if (a[i].K / 512 < sp) sp = a[i].K / 512;
    if (sp < 1) sp = 1;
    if (sp > 64) sp = 64;
    ga.p[i] = a[i];
    ga.p[i].split_k = (int)sp;
    ga.p[i].nb1 = ga.p[i].nb2 = 1;
    for (int k2 = 0; k2 < (int)sp; ++k2) units[nu++] = U{i, k2, ga.gx[i] * ga.gy[i]};
  }
  for (int i = 1; i < nu; ++i) {  // insertion sort by tiles, descending
    const U v = units[i];
    int j = i - 1;
    while (j >= 0 && units[j].tiles < v.tiles) { units[j + 1] = units[j]; --j; }
    units[j + 1] = v;
  }
  int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nu; ++i) {
    int best = 0;
    for (int x = 1; x < 8; ++x)
      if (load[x] < load[best]) best = x;
    const int c = ga.nunit[best];
    if (c >= GROUP_UNITS) return TFASR_STATUS_UNSUPPORTED;
    ga.uq[best][c] = (short)units[i].q;
    ga.uks[best][c] = (short)units[i].ks;
    ga.j0[best][c] = load[best];
    load[best] += units[i].tiles;
    ga.j0[best][c + 1] = load[best];
    ga.nunit[best] = c + 1;
  }
  int maxload = 0;
  for (int x = 0; x < 8; ++x)
    if (load[x] > maxload) maxload = load[x];
  static const int nst = 2;
  // accumulate epilogue: atomics from the fragments, no strips in LDS
  if (bn == 64) {
    constexpr int STAGE = A_BYTES + 64 * BK * 2;
    if (nst == 3) TFASR_KLAUNCH((wgrad_group_kernel<E_CSUM, 3, 64>), dim3(8 * maxload), dim3(256), 3 * STAGE, stream, ga);
    else TFASR_KLAUNCH((wgrad_group_kernel<E_CSUM, 2, 64>), dim3(8 * maxload), dim3(256), 2 * STAGE, stream, ga);
  } else {
    constexpr int STAGE = A_BYTES + 128 * BK * 2;
    TFASR_KLAUNCH((wgrad_group_kernel<E_CSUM, 2, 128>), dim3(8 * maxload), dim3(256), 2 * STAGE, stream, ga);
  }
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}

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

// ---------------------------------------------------------------------------------------------------------------------
// FFModule forward in one launch (ffn_fused.h).  UNSUPPORTED outside its shape range: the caller keeps the three-launch route.
extern "C" int tfasr_ffn_fused_fwd2(const void* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                                    const float* b2, void* y, void* ln, float* mean, float* rstd, void* z, int z_factor, void* h, long rows, int d, int F,
                                    float ln_eps, float res_factor, float drop_p, long drop_seed1, long drop_seed2, int dtype, void* stream);
extern "C" int tfasr_ffn_fused_fwd(const void* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                                   const float* b2, void* y, void* ln, float* mean, float* rstd, void* z, void* h, long rows, int d, int F,
                                   float ln_eps, float res_factor, float drop_p, long drop_seed1, long drop_seed2, int dtype, void* stream) {
  return tfasr_ffn_fused_fwd2(x, gamma, beta, W1, b1, W2, b2, y, ln, mean, rstd, z, 0, h, rows, d, F, ln_eps, res_factor, drop_p, drop_seed1, drop_seed2, dtype, stream);
}
extern "C" int tfasr_ffn_fused_fwd2(const void* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                                    const float* b2, void* y, void* ln, float* mean, float* rstd, void* z, int z_factor, void* h, long rows, int d, int F,
                                    float ln_eps, float res_factor, float drop_p, long drop_seed1, long drop_seed2, int dtype, void* stream) {
  if (!x || !gamma || !beta || !W1 || !b1 || !W2 || !b2 || !y || !ln || !mean || !rstd || rows <= 0 || d <= 0 || F <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (!(drop_p >= 0.f && drop_p < 1.f)) return TFASR_STATUS_INVALID_VALUE;
  static const bool off = getenv("TFASR_FFN_FUSED") && getenv("TFASR_FFN_FUSED")[0] == '0';
  if (off || dtype != TFASR_BF16 || d != 256 || (F % 64) != 0 || F > 1024 || F < 128 || rows * (long)F >= (1L << 32)) return TFASR_STATUS_UNSUPPORTED;
  const uintptr_t al = (uintptr_t)x | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)y | (uintptr_t)ln | (uintptr_t)z | (uintptr_t)h | (uintptr_t)gamma | (uintptr_t)beta;
  if (al & 15) return TFASR_STATUS_UNSUPPORTED;
  FfnArgs a;
  a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.W1 = (const bf16_t*)W1; a.b1 = b1; a.W2 = (const bf16_t*)W2; a.b2 = b2;
  a.y = (bf16_t*)y; a.ln = (bf16_t*)ln; a.mean = mean; a.rstd = rstd; a.z = (bf16_t*)z; a.h = (bf16_t*)h;
  a.rows = rows; a.F = F; a.eps = ln_eps; a.res = res_factor; a.drop_p = drop_p; a.seed1 = drop_seed1; a.seed2 = drop_seed2;
  a.zfactor = z_factor ? 1 : 0;
  a.dbg = nullptr;
#ifdef TFASR_FFN_TIMING
  static long long* dbg_buf = nullptr;  // probe builds only (tools/hwprobe): 8 cycle sums of workgroup 0, printed by the caller via TFASR_FFN_DBG_DUMP
  if (!dbg_buf && hipMalloc((void**)&dbg_buf, 64) != hipSuccess) dbg_buf = nullptr;
  a.dbg = dbg_buf;
#endif
  const int st = launch_ffn_fused_fwd(a, (hipStream_t)stream);
#ifdef TFASR_FFN_TIMING
  if (dbg_buf) {  // probe build (-DTFASR_FFN_TIMING): always dumps
    long long h[8];
    if (hipStreamSynchronize((hipStream_t)stream) == hipSuccess && hipMemcpy(h, dbg_buf, 64, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[ffn_timing] prologue %lld | per-kernel sums: wait+barrier %lld dma-issue %lld gemm1 %lld z->lds %lld rowpass %lld gemm2 %lld | epilogue %lld\n",
              h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
#endif
  TFASR_CHECK_LAUNCH();
  return st;
}

// LayerNorm + Dense in one launch: out = LN(x) W + b, with ln / mean / rstd stored as tfasr_layernorm_fwd would (the Dense-only mode of the
// fused FFModule kernel, ffn_fused.h).  UNSUPPORTED outside bf16 / d = 256 / N % 64 == 0, N <= 1024 / aligned pointers.
extern "C" int tfasr_ln_dense_fwd(const void* x, const float* gamma, const float* beta, const void* W, const float* b, void* out, void* ln, float* mean,
                                  float* rstd, long rows, int d, int N, float ln_eps, int dtype, void* stream) {
  if (!x || !gamma || !beta || !W || !b || !out || !ln || !mean || !rstd || rows <= 0 || d <= 0 || N <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (dtype != TFASR_BF16 || d != 256 || (N % 64) != 0 || N > 1024 || N < 128 || rows * (long)N >= (1L << 32)) return TFASR_STATUS_UNSUPPORTED;
  const uintptr_t al = (uintptr_t)x | (uintptr_t)W | (uintptr_t)out | (uintptr_t)ln | (uintptr_t)gamma | (uintptr_t)beta;
  if (al & 15) return TFASR_STATUS_UNSUPPORTED;
  FfnArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.W1 = (const bf16_t*)W; a.b1 = b;
  a.y = nullptr; a.ln = (bf16_t*)ln; a.mean = mean; a.rstd = rstd; a.z = (bf16_t*)out; a.h = nullptr;
  a.rows = rows; a.F = N; a.eps = ln_eps;
  const int st = launch_ffn_fused_fwd(a, (hipStream_t)stream);
  TFASR_CHECK_LAUNCH();
  return st;
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense data gradient + the LayerNorm backward in front of it in one launch (dense_ln.h).  UNSUPPORTED outside its shape range.
extern "C" int tfasr_dense_ln_bwd(const void* dy, const void* W, int K, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  const void* add, void* dx, float* part, int nblk, void* dx_dropped, float drop_p, long drop_seed, long rows,
                                  int d, float alpha, int dtype, void* stream) {
  if (!dy || !W || !x || !gamma || !mean || !rstd || !dx || !part || rows <= 0 || d <= 0 || K <= 0 || nblk <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (dx_dropped && !(drop_p >= 0.f && drop_p < 1.f)) return TFASR_STATUS_INVALID_VALUE;
  if (dtype != TFASR_BF16 || d != DLN_D || (K % 64) != 0 || rows * (long)d >= (1L << 32)) return TFASR_STATUS_UNSUPPORTED;
  const uintptr_t al = (uintptr_t)dy | (uintptr_t)W | (uintptr_t)x | (uintptr_t)add | (uintptr_t)dx | (uintptr_t)dx_dropped | (uintptr_t)gamma;
  if (al & 15) return TFASR_STATUS_UNSUPPORTED;
  DenseLnArgs a;
  a.dy = (const bf16_t*)dy; a.ldy = K; a.W = (const bf16_t*)W; a.ldw = K; a.K = K; a.x = (const bf16_t*)x; a.gamma = gamma; a.mean = mean; a.rstd = rstd;
  a.add = (const bf16_t*)add; a.dx = (bf16_t*)dx; a.dxd = (bf16_t*)dx_dropped; a.drop_p = dx_dropped ? drop_p : 0.f; a.drop_seed = (uint64_t)drop_seed;
  a.part = part; a.rows = rows; a.alpha = alpha; a.ntiles = 0;
  const int st = launch_dense_ln_bwd(a, nblk, (hipStream_t)stream);
  if (st != TFASR_STATUS_SUCCESS) return st;
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
