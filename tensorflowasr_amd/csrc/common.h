// Shared device helpers for the tensorflowasr_amd HIP kernels (gfx950 / CDNA4 only).
// wave = 64 lanes everywhere in this code base.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/tfasr_hip.h"

#define TFASR_WAVE 64

// Every kernel of the library is launched through TFASR_KLAUNCH (the chevron launch + a host-side count: the statistic behind
// tfasr_launch_count(), what bench.py reports as `launches_per_step`; a kernel boundary costs 1.5-2.7 us on this chip, so the count is a
// quantity to watch).  Arguments as hipLaunchKernelGGL's; a templated kernel name goes in parentheses.
#include <atomic>
extern std::atomic<size_t> g_tfasr_launch_count;  // api.hip
#define TFASR_KLAUNCH(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                  \
  do {                                                                                                \
    g_tfasr_launch_count.fetch_add(1, std::memory_order_relaxed);                                     \
    kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);               \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same rule as TF / torch bfloat16 casts): gfx950's v_cvt_pk_bf16_f32, one
// instruction per element PAIR instead of ~7 VALU ops per element (the conversion was a third of a GEMM epilogue)
typedef __attribute__((ext_vector_type(2))) float tfasr_f2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 tfasr_b2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const tfasr_b2_t r = __builtin_convertvector(tfasr_f2_t{lo, hi}, tfasr_b2_t);
  return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack2_bf16(f, 0.f) & 0xffffu); }

template <typename T> struct Num;
template <> struct Num<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Num<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 8-element vector load/store (16 B for bf16, 32 B for f32). Pointer must be aligned to 16 B.
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  uint4 a;
  a.x = pack2_bf16(v[0], v[1]);
  a.y = pack2_bf16(v[2], v[3]);
  a.z = pack2_bf16(v[4], v[5]);
  a.w = pack2_bf16(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = a;
}

// All-reduce inside each row of 16 lanes with DPP row rotations (plain VALU, ~8 cycles each): __shfl_xor compiles to
// ds_bpermute_b32 (an LDS round trip, ~100 cycles, and these chains are dependent) - the 32 of them per key block were
// half of the fused attention forward's time.
// bound_ctrl on (every lane of a row rotation is valid, so it changes nothing): with old = 0 and bound_ctrl OFF the compiler
// materialises the zero and keeps a separate v_mov_dpp per step - 3 instructions per step instead of one v_add / v_max with a DPP
// operand (96 vs 32 instructions per key block in the fused attention forward)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0x128>(v);  // row_ror:8
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x122>(v);  // row_ror:2
  v += dpp_mov<0x121>(v);  // row_ror:1
  return v;
}
// v_max_f32 with a DPP operand, one instruction per step: fmaxf(v, dpp_mov(v)) compiles to v_mov_b32_dpp + a canonicalising v_max v, v, v
// (llvm.maxnum quiets a possible signalling NaN of the moved value first) + the v_max itself - 12 instructions for the four steps.
// The hazard recogniser does not look inside asm: a VGPR written by a VALU instruction needs 2 wait states before a DPP read of it
// (5 after a VALU write of EXEC: the first s_nop covers that case as well).
__device__ __forceinline__ float row16_max(float v) {
  asm("s_nop 4\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(v));
  return v;
}

// Cross-row steps of a wave reduction WITHOUT the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: an LDS round trip, ~100 cycles in
// a dependent chain): gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16-lane rows (the upper 32 lanes) of one register with
// the even rows (the lower lanes) of another.  Given the same value in both, the two results hold (own row group, partner row group) in
// every lane: one VALU op each for lane ^ 16 and lane ^ 32.
__device__ __forceinline__ float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// the value lane ^ 16 holds (one v_permlane16_swap + one select instead of a ds_bpermute round trip); `odd_row` = (lane >> 4) & 1
__device__ __forceinline__ uint32_t xor16_get(uint32_t v, bool odd_row) {
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return odd_row ? r[0] : r[1];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` = >= 16 floats of LDS scratch
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }  // v_rcp_f32: 1 ulp
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// tanh(x) = 1 - 2 / (exp(2x) + 1): v_exp_f32 + v_rcp_f32 (abs error ~1e-7; libm tanhf is ~25 instructions)
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.f); }
// d/dx [x * sigmoid(x)] = s + x*s*(1-s)
__device__ __forceinline__ float dswishf_(float x) { const float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }

// counter-based dropout mask: one 32-bit hash of (seed, element index / 2) serves an even/odd element pair, 16 bits each; keep iff the
// 16-bit uniform >= round(p * 65536).  Forward and backward regenerate the identical mask from (seed, index), nothing is stored.
// Round 5: the per-pair mix runs on 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24: FULL rate; a 32-bit v_mul_lo_u32 is quarter rate
// and the murmur3 finaliser of rounds 1-4 has two of them: 15 issue slots per pair against 10 now - the FFModule kernels are bound by
// exactly this vector work).  Two rounds of multiply + xor-shift over the pair index plus a per-launch key (the strongly mixed seed,
// loop invariant); statistics checked in tests/test_ops_gpu.py (keep rate, row / column variance, lag correlations, seed independence,
// avalanche >= 0.49 per input bit - measured before adoption).
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_thr(float p) { return (uint32_t)(p * 65536.f + 0.5f); }
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }  // v_mul_u32_u24: low 24 bits of each operand
// the seed's share: computed once per thread (uniform), so its two 32-bit multiplies do not matter
__device__ __forceinline__ uint32_t drop_key(uint64_t seed) { return fmix32((uint32_t)seed * 0x9E3779B1u ^ (uint32_t)(seed >> 32)); }
__device__ __forceinline__ uint32_t drop_mix(uint32_t key, uint32_t pair_lo, uint32_t pair_hi = 0) {
  uint32_t y = mul24(pair_lo, 0x9E3779u) + mul24(pair_lo >> 24, 0x85EBCBu) + key + mul24(pair_hi, 0xC2B2AFu);
  y ^= y >> 13;
  y = mul24(y, 0xC2B2AFu) ^ (y >> 11);
  y ^= y >> 15;
  return y;
}
__device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t pair) { return drop_mix(drop_key(seed), (uint32_t)pair, (uint32_t)(pair >> 32)); }
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, float p) {
  const uint32_t h = drop_hash(seed, idx >> 1);
  return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= drop_thr(p);
}

// log(exp(a)+exp(b)) with -inf handling (the 2-term log-sum-exp of losses/impl/rnnt.py:72-78,126)
__device__ __forceinline__ float logaddexpf_(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + log1pf(expf(-fabsf(a - b)));
}

// hipGetLastError() also reports stale, benign results of OTHER runtime calls made on this thread by the host framework
// (e.g. the caching allocator polling hipEventQuery -> hipErrorNotReady), so "not ready" is not a launch failure.
#include <stdio.h>
#define TFASR_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess && e_ != hipErrorNotReady) { \
    fprintf(stderr, "[tfasr_hip] %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return TFASR_STATUS_EXECUTION_FAILED; } } while (0)

static inline int tfasr_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
