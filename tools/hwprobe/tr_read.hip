// Probe: semantics of ds_read_b64_tr_b16 and global_load_lds on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(const short* g, short* out, int mode){
  __shared__ __attribute__((aligned(16))) short lds[4096];
  if (mode == 0) {
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  } else {
    // lane-linear destination check: each lane loads 8 shorts from g + (63-lane)*8
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (63 - threadIdx.x) * 8), (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  if (mode == 0) {
    // address per lane: row = lane>>2 (stride 64 shorts = 128 B), col = (lane&3)*4
    const int l = threadIdx.x;
    const int i = l & 15, grp = l >> 4;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + (grp * 4 + (i >> 2)) * 64 + (i & 3) * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
  } else {
    for (int j = 0; j < 8; ++j) out[threadIdx.x * 8 + j] = lds[threadIdx.x * 8 + j];
  }
}
int main(){
  short *g, *out; hipMalloc(&g, 8192); hipMalloc(&out, 8192);
  short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (short)i;
  hipMemcpy(g, h, 8192, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, out, mode);
    short r[512]; hipMemcpy(r, out, 1024, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < (mode ? 8 : 4); ++j) printf(" %4d", r[l * (mode ? 8 : 4) + j]); printf("\n"); }
  }
  return 0;
}
