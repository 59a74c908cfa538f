// How fast is the split-K accumulate epilogue?  384 workgroups x 4 waves each add a 128x128 f32 tile into a 256x1024 output
// (the FFN weight-gradient shape: 16 tiles x 24 k-slices), with the k-slices of one tile spread over all XCDs (MAP 0, what
// gemm_fast does today) or kept on one XCD (MAP 1), with agent-scope or workgroup-scope (L2-resident) atomics.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int SCOPE, int MAP>
__global__ __launch_bounds__(256) void atomic_epi(float* out, int ldd, int tiles_n, int ntiles, int split) {
  const int id = blockIdx.x, x = id & 7, j = id >> 3;
  int tile, slice;
  if (MAP == 0) { const int per = split / 8; slice = x * per + j % per; tile = j / per; }       // slice per XCD
  else { const int tx = ntiles / 8; tile = x * tx + j % tx; slice = j / tx; }                   // tile per XCD
  (void)slice;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  float* base = out + (long)(tm * 128 + w * 32) * ldd + tn * 128;
  for (int r = 0; r < 32; ++r)
    for (int h = 0; h < 2; ++h) {
      float* p = base + (long)r * ldd + h * 64 + l;
      if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// the MFMA fragment layout gemm_fast uses for its accumulate epilogue: one instruction = 4 rows x 16 consecutive columns
__global__ __launch_bounds__(256) void atomic_frag(float* out, int ldd, int tiles_n, int ntiles, int split) {
  const int id = blockIdx.x, x = id & 7, j = id >> 3;
  const int per = split / 8, tile = j / per;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, g = l >> 4;
  const int wm = w >> 1, wn = w & 1;
  (void)x;
  for (int i = 0; i < 4; ++i)
    for (int jj = 0; jj < 4; ++jj)
      for (int e = 0; e < 4; ++e) {
        const int row = tm * 128 + wm * 64 + i * 16 + g * 4 + e, col = tn * 128 + wn * 64 + jj * 16 + r;
        atomicAdd(out + (long)row * ldd + col, 1.0f);
      }
}
void run_frag(float* out, int M, int N, int split) {
  const int tiles_n = N / 128, ntiles = (M / 128) * tiles_n, grid = ntiles * split;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) atomic_frag<<<grid, 256>>>(out, N, tiles_n, ntiles, split);
  (void)hipMemset(out, 0, (size_t)M * N * 4);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) atomic_frag<<<grid, 256>>>(out, N, tiles_n, ntiles, split);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  float* h = (float*)malloc((size_t)M * N * 4);
  (void)hipMemcpy(h, out, (size_t)M * N * 4, hipMemcpyDeviceToHost);
  long bad = 0; for (long i = 0; i < (long)M * N; ++i) bad += h[i] != 20.f * split;
  printf("%-34s %dx%d split %d: %.1f us/launch, wrong sums %ld\n", "fragment layout (4 rows x 64 B)", M, N, split, ms / 20 * 1e3, bad);
  free(h);
}
template <int SCOPE, int MAP>
void run(const char* name, float* out, int M, int N, int split) {
  const int tiles_n = N / 128, ntiles = (M / 128) * tiles_n, grid = ntiles * split;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) atomic_epi<SCOPE, MAP><<<grid, 256>>>(out, N, tiles_n, ntiles, split);
  hipMemset(out, 0, (size_t)M * N * 4);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) atomic_epi<SCOPE, MAP><<<grid, 256>>>(out, N, tiles_n, ntiles, split);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float* h = (float*)malloc((size_t)M * N * 4);
  hipMemcpy(h, out, (size_t)M * N * 4, hipMemcpyDeviceToHost);
  long bad = 0; for (long i = 0; i < (long)M * N; ++i) bad += h[i] != 20.f * split;
  printf("%-34s %dx%d split %d: %.1f us/launch, wrong sums %ld\n", name, M, N, split, ms / 20 * 1e3, bad);
  free(h);
}
int main() {
  float* out; hipMalloc(&out, (size_t)64 << 20);
  run<0, 0>("agent scope, slices over XCDs", out, 256, 1024, 24);
  run<0, 1>("agent scope, tile per XCD", out, 256, 1024, 24);
  // same number of atomics, fewer per address
  run<0, 0>("6.3M atomics, 6 per address", out, 1024, 1024, 8);   // (8 slices: 1 per XCD) 8.4M
  run<0, 0>("12 per address (3.1M)", out, 256, 1024, 16);
  run<0, 0>("48 per address (12.6M)", out, 256, 1024, 48);
  run<0, 0>("1 per address x 8 (8.4M)", out, 1024, 1024, 8);
  run<0, 0>("4096x1024 split 8 (33M)", out, 4096, 1024, 8);
  run_frag(out, 256, 1024, 24);
  run_frag(out, 256, 1024, 16);
  return 0;
}
