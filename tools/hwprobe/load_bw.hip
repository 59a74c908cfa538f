// Per-CU bandwidth of the two ways a GEMM slab can reach LDS on gfx950: global_load_lds (LDS-DMA, 16 B per lane) versus
// global_load_dwordx4 into VGPRs + ds_write_b128, with the source resident in L1 (every wave re-reads one 32-KB window) or
// only in L2 (each workgroup streams its own window, total footprint < L2).  Prints bytes / clock / CU at the measured time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef float float4_t __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA; MODE 1: VGPR load + ds_write; MODE 2: VGPR load only (accumulate)
template <int MODE>
__global__ __launch_bounds__(256) void bw_kernel(const char* src, long window, long wg_stride, int iters, float* sink) {
  extern __shared__ char smem[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* base = src + (long)blockIdx.x * wg_stride;
  float4_t acc = {0, 0, 0, 0};
  long off = 0;
  for (int it = 0; it < iters; ++it) {
    // one 32-KB slab per workgroup per iteration: 8 wave-instructions of 1 KB per wave
    float4_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = w * 8 + i;
      const char* p = base + off + q * 1024 + lane * 16;
      if (MODE == 0) __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(smem + (it & 1) * 32768 + __builtin_amdgcn_readfirstlane(q * 1024)), 16, 0, 0);
      else v[i] = *(const float4_t*)p;
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *(float4_t*)(smem + (it & 1) * 32768 + (w * 8 + i) * 1024 + lane * 16) = v[i];
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    off += 32768;
    if (off >= window) off = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 2) acc = *(float4_t*)(smem + threadIdx.x * 16);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
// GEMM "direct image" pattern: one wave-instruction = 8 rows x 128 B (8 lanes x 16 B per row), row stride ld bytes, optional XOR
// chunk swizzle; a slab = 128 A rows + 128 B rows of 128 B (32 KB); consecutive slabs advance 128 B along the rows.
template <int SWZ>
__global__ __launch_bounds__(256) void bw_rows_kernel(const char* src, long ld, long rows_total, int slabs, int iters, float* sink) {
  extern __shared__ char smem[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 256) % rows_total;
  const char* sp[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = w * 8 + i, row = q * 8 + (lane >> 3), p = lane & 7;
    sp[i] = src + (row0 + row) * ld + ((SWZ ? (p ^ ((row >> 1) & 7)) : p) << 4);
  }
  for (int it = 0; it < iters; ++it) {
    const long koff = (long)(it % slabs) * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(sp[i] + koff), LDS_PTR(smem + (it & 1) * 32768 + __builtin_amdgcn_readfirstlane((w * 8 + i) * 1024)), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float4_t acc = *(float4_t*)(smem + threadIdx.x * 16);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
template <int SWZ>
void run_rows(const char* name, const char* src, long ld, long rows_total, int grid) {
  const int iters = 400, slabs = (int)(ld / 128);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float* sink; (void)hipMalloc(&sink, 4096);
  for (int i = 0; i < 2; ++i) bw_rows_kernel<SWZ><<<grid, 256, 65536>>>(src, ld, rows_total, slabs, iters, sink);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) bw_rows_kernel<SWZ><<<grid, 256, 65536>>>(src, ld, rows_total, slabs, iters, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double bytes = (double)grid * iters * 32768;
  printf("%-46s grid %4d: %7.1f us, %6.2f TB/s aggregate, %5.1f B/clk/CU @2.4GHz (256 CUs)\n", name, grid, ms * 1e3, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / 256 / 2.4e9);
}
template <int MODE>
void run(const char* name, const char* src, long window, long wg_stride, int grid) {
  const int iters = 400;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float* sink; (void)hipMalloc(&sink, 4096);
  for (int i = 0; i < 2; ++i) bw_kernel<MODE><<<grid, 256, 65536>>>(src, window, wg_stride, iters, sink);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) bw_kernel<MODE><<<grid, 256, 65536>>>(src, window, wg_stride, iters, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double bytes = (double)grid * iters * 32768;
  printf("%-46s grid %4d: %7.1f us, %6.2f TB/s aggregate, %5.1f B/clk/CU @2.4GHz (256 CUs)\n", name, grid, ms * 1e3, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
  char* src; (void)hipMalloc(&src, (size_t)512 << 20); (void)hipMemset(src, 0, (size_t)512 << 20);
  // L1-resident: every workgroup re-reads the same 32 KB... the TCP is 32 KB, so use a 16-KB window?  keep 32 KB (2 iterations alias)
  run<0>("LDS-DMA, one shared 32-KB window (L1/L2 hot)", src, 32768, 0, 512);
  run<1>("VGPR+ds_write, one shared 32-KB window", src, 32768, 0, 512);
  run<2>("VGPR only, one shared 32-KB window", src, 32768, 0, 512);
  // L2-resident: 512 workgroups x 32 KB = 16 MB total = 2 MB per XCD (L2 4 MB per XCD)
  run<0>("LDS-DMA, own 32-KB window per WG (L2)", src, 32768, 32768, 512);
  run<1>("VGPR+ds_write, own 32-KB window per WG (L2)", src, 32768, 32768, 512);
  run<2>("VGPR only, own 32-KB window per WG (L2)", src, 32768, 32768, 512);
  // GEMM-like sharing: neighbouring workgroups read the same window (8 WGs per window)
  run<0>("LDS-DMA, window shared by 8 WGs, 256 KB each", src, 262144, 32768 / 8 * 0 + 0, 512);
  // streaming from HBM/MALL: own 1-MB window per WG (512 MB total)
  run<0>("LDS-DMA, own 1-MB window per WG (HBM)", src, 1 << 20, 1 << 20, 512);
  run<2>("VGPR only, own 1-MB window per WG (HBM)", src, 1 << 20, 1 << 20, 512);
  run<0>("LDS-DMA, 1 WG per CU, own 32-KB window (L2)", src, 32768, 32768, 256);
  run<2>("VGPR only, 1 WG per CU, own 32-KB window (L2)", src, 32768, 32768, 256);
  run_rows<0>("rows x128B, ld 512 B (K=256), 8k rows", src, 512, 8192, 512);
  run_rows<1>("rows x128B swizzled, ld 512 B, 8k rows", src, 512, 8192, 512);
  run_rows<0>("rows x128B, ld 2048 B (K=1024), 8k rows", src, 2048, 8192, 512);
  run_rows<1>("rows x128B swizzled, ld 2048 B, 8k rows", src, 2048, 8192, 512);
  run_rows<1>("rows swizzled, ld 2048 B, 128k rows (256 MB)", src, 2048, 131072, 512);
  run_rows<1>("rows swizzled, ld 2000 B (V=1000)", src, 2000, 131072, 512);
  return 0;
}
