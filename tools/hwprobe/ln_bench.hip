// Standalone micro-benchmark: how fast can a [rows, 256] bf16 row-normalisation stream on this chip?  (hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint16_t bf16_t;
__device__ __forceinline__ float b2f(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u); v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u); v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ uint32_t f2b(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; }
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  uint4 a; a.x = f2b(v[0]) | (f2b(v[1]) << 16); a.y = f2b(v[2]) | (f2b(v[3]) << 16); a.z = f2b(v[4]) | (f2b(v[5]) << 16); a.w = f2b(v[6]) | (f2b(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = a;
}
__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ x, uint4* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = x[i];
}
template <int LPR> __device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// v1: 32 lanes per row, one 16-B load per lane
__global__ __launch_bounds__(256) void ln_v1(const bf16_t* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, bf16_t* __restrict__ y,
                                             float* __restrict__ mo, float* __restrict__ ro, long rows) {
  const int lane = threadIdx.x & 63, li = lane & 31, sub = lane >> 5, c0 = li * 8;
  float gg[8], bb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { gg[k] = g[c0 + k]; bb[k] = b[c0 + k]; }
  const long w0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + sub, step = (long)gridDim.x * 8;
  for (long r = w0; r < rows; r += step) {
    float v[8]; ld8(x + r * 256 + c0, v);
    float s = 0; for (int k = 0; k < 8; ++k) s += v[k];
    const float mean = seg_sum<32>(s) * (1.f / 256);
    float q = 0; for (int k = 0; k < 8; ++k) { const float d = v[k] - mean; q += d * d; }
    const float rstd = rsqrtf(seg_sum<32>(q) * (1.f / 256) + 1e-3f);
    for (int k = 0; k < 8; ++k) v[k] = (v[k] - mean) * rstd * gg[k] + bb[k];
    st8(y + r * 256 + c0, v);
    if (li == 0) { mo[r] = mean; ro[r] = rstd; }
  }
}
// v2: 16 lanes per row (two 16-B loads per lane), 4 rows per wave, sum and sum-of-squares reduced together
__global__ __launch_bounds__(256) void ln_v2(const bf16_t* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, bf16_t* __restrict__ y,
                                             float* __restrict__ mo, float* __restrict__ ro, long rows) {
  const int lane = threadIdx.x & 63, li = lane & 15, sub = lane >> 4, c0 = li * 8;
  float g0[8], b0[8], g1[8], b1[8];
  {
    const float4* gp = reinterpret_cast<const float4*>(g); const float4* bp = reinterpret_cast<const float4*>(b);
    float4 t;
    t = gp[li * 2]; g0[0] = t.x; g0[1] = t.y; g0[2] = t.z; g0[3] = t.w; t = gp[li * 2 + 1]; g0[4] = t.x; g0[5] = t.y; g0[6] = t.z; g0[7] = t.w;
    t = gp[32 + li * 2]; g1[0] = t.x; g1[1] = t.y; g1[2] = t.z; g1[3] = t.w; t = gp[32 + li * 2 + 1]; g1[4] = t.x; g1[5] = t.y; g1[6] = t.z; g1[7] = t.w;
    t = bp[li * 2]; b0[0] = t.x; b0[1] = t.y; b0[2] = t.z; b0[3] = t.w; t = bp[li * 2 + 1]; b0[4] = t.x; b0[5] = t.y; b0[6] = t.z; b0[7] = t.w;
    t = bp[32 + li * 2]; b1[0] = t.x; b1[1] = t.y; b1[2] = t.z; b1[3] = t.w; t = bp[32 + li * 2 + 1]; b1[4] = t.x; b1[5] = t.y; b1[6] = t.z; b1[7] = t.w;
  }
  const long w0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub, step = (long)gridDim.x * 16;
  for (long r = w0; r < rows; r += step) {
    float v0[8], v1[8];
    ld8(x + r * 256 + c0, v0); ld8(x + r * 256 + 128 + c0, v1);
    float s = 0, q = 0;
    for (int k = 0; k < 8; ++k) { s += v0[k] + v1[k]; q += v0[k] * v0[k] + v1[k] * v1[k]; }
    s = seg_sum<16>(s); q = seg_sum<16>(q);
    const float mean = s * (1.f / 256);
    const float var = fmaxf(q * (1.f / 256) - mean * mean, 0.f);
    const float rstd = rsqrtf(var + 1e-3f);
    for (int k = 0; k < 8; ++k) { v0[k] = (v0[k] - mean) * rstd * g0[k] + b0[k]; v1[k] = (v1[k] - mean) * rstd * g1[k] + b1[k]; }
    st8(y + r * 256 + c0, v0); st8(y + r * 256 + 128 + c0, v1);
    if (li == 0) { mo[r] = mean; ro[r] = rstd; }
  }
}
template <typename F> float timeit(F f, int it = 50) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(a); for (int i = 0; i < it; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / it * 1e3f;
}
int main() {
  const long rows = 23808, C = 256, n = rows * C;
  bf16_t *x, *y; float *g, *b, *mo, *ro;
  hipMalloc(&x, n * 2); hipMalloc(&y, n * 2); hipMalloc(&g, C * 4); hipMalloc(&b, C * 4); hipMalloc(&mo, rows * 4); hipMalloc(&ro, rows * 4);
  hipMemset(x, 0x3c, n * 2); hipMemset(g, 0, C * 4); hipMemset(b, 0, C * 4);
  const double bytes = 2.0 * n * 2;
  for (int grid : {256, 512, 1024, 2048, 4096}) {
    float us = timeit([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)x, (uint4*)y, n / 8); });
    printf("copy16 grid %d: %.1f us  %.0f GB/s\n", grid, us, bytes / us / 1e3);
  }
  for (int grid : {512, 1024, 1488, 2048, 2976}) {
    float us = timeit([&] { hipLaunchKernelGGL(ln_v1, dim3(grid), dim3(256), 0, 0, x, g, b, y, mo, ro, rows); });
    printf("ln_v1 grid %d: %.1f us  %.0f GB/s\n", grid, us, bytes / us / 1e3);
  }
  for (int grid : {256, 372, 512, 744, 1024, 1488}) {
    float us = timeit([&] { hipLaunchKernelGGL(ln_v2, dim3(grid), dim3(256), 0, 0, x, g, b, y, mo, ro, rows); });
    printf("ln_v2 grid %d: %.1f us  %.0f GB/s\n", grid, us, bytes / us / 1e3);
  }
  float us = timeit([&] { hipLaunchKernelGGL(copy16, dim3(1), dim3(64), 0, 0, (const uint4*)x, (uint4*)y, 64L); });
  printf("tiny kernel: %.1f us\n", us);
  return 0;
}
