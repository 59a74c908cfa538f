// Does hipExtAnyOrderLaunch (AQL barrier bit cleared) let two kernels of ONE stream overlap on gfx950, and what does a kernel boundary cost?
//   hipcc --offload-arch=gfx950 -O3 tools/hwprobe/anyorder_test.hip -o tools/hwprobe/anyorder_test && tools/hwprobe/anyorder_test
// (hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for the module-launch variant: measured, not assumed.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(long cycles, int* out) {
  const long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
__global__ void tiny_kernel(int* out) { if (out && threadIdx.x == 1000) out[0] = 1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int* d;
  CK(hipMalloc(&d, 64));
  for (int flags = 0; flags <= 1; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      const long c = 200000;
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, c, d);
      if (flags) hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, c, d);
      else hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, c, d);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("two spin kernels (64 workgroups each), second %s: %.1f us\n", flags ? "ANY-ORDER" : "in order ", ms * 1e3);
    }
  }
  {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, 200000L, d);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("one spin kernel: %.1f us\n", ms * 1e3);
  }
  for (int flags = 0; flags <= 1; ++flags) {
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) {
        if (flags) hipExtLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
        else hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, s, d);
      }
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%d empty kernels (256 workgroups), %s: %.2f us per launch\n", N, flags ? "ANY-ORDER" : "in order ", ms * 1e3 / N);
    }
  }
  return 0;
}
