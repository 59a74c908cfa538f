// Where does a gemm_fast tile spend its time?  Builds the production kernel with cycle-counter hooks (TFASR_GEMM_TIMING).
#define TFASR_GEMM_TIMING 1
#include "../../tensorflowasr_amd/csrc/gemm_fast.hip"
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 23808, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
  const int ta = argc > 4 ? atoi(argv[4]) : 0, tb = argc > 5 ? atoi(argv[5]) : 0, split = argc > 6 ? atoi(argv[6]) : 1;
  bf16_t *A, *B, *D;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)K * N * 2); hipMalloc(&D, (size_t)M * N * 4);
  hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(B, 0x3c, (size_t)K * N * 2);
  tfasr_gemm_args a; memset(&a, 0, sizeof(a));
  a.A = A; a.B = B; a.D = D; a.M = M; a.N = N; a.K = K; a.lda = ta ? M : K; a.ldb = tb ? K : N; a.ldd = N; a.trans_a = ta; a.trans_b = tb;
  if (split > 1) { a.split_k = split; a.accumulate = 1; a.out_f32 = 1; } a.nb1 = a.nb2 = 1; a.alpha = 1.f; a.beta = 1.f; a.dtype = TFASR_BF16; if (split <= 1) a.split_k = 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) tfasr_gemm_fast_try(a, 0);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) tfasr_gemm_fast_try(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int ntiles = ((N + 127) / 128) * ((M + 127) / 128) * (split > 1 ? split : 1);
  const int nblk = ntiles < 512 ? ntiles : 512;
  std::vector<long long> h(4L * nblk), h2(2L * nblk);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gemm_timing), h.size() * 8);
  hipMemcpyFromSymbol(h2.data(), HIP_SYMBOL(g_gemm_timing), h2.size() * 8, 4L * 32768 * 8);
  { std::vector<long long> h3(5L * nblk); hipMemcpyFromSymbol(h3.data(), HIP_SYMBOL(g_gemm_timing), h3.size() * 8, 6L * 32768 * 8);
    double ph[5] = {0, 0, 0, 0, 0}; for (int b = 0; b < nblk; ++b) for (int k = 0; k < 5; ++k) ph[k] += h3[5L * b + k];
    printf("first tile mainloop phases (cycles, summed over slabs): dma-wait %.0f barrier1 %.0f frag+mfma %.0f barrier2 %.0f dma-issue %.0f\n", ph[0] / nblk, ph[1] / nblk, ph[2] / nblk, ph[3] / nblk, ph[4] / nblk); }
  double m0 = 0, x0 = 0, m1 = 0, x1 = 0;
  for (int b = 0; b < nblk; ++b) { m0 += h[4L * b + 2]; x0 += h[4L * b + 3]; m1 += h2[2L * b]; x1 += h2[2L * b + 1]; }
  printf("M %d N %d K %d: %.1f us/launch, %d tiles on %d workgroups; cycles first tile: mainloop %.0f total %.0f | second tile: mainloop %.0f total %.0f\n",
         M, N, K, ms / 20 * 1e3, ntiles, nblk, m0 / nblk, x0 / nblk, m1 / nblk, x1 / nblk);
  return 0;
}
