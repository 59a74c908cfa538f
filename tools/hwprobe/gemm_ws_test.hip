// Weights-stationary GEMM (csrc/gemm_ws.h) against the tiled kernels of the library on the Conformer block's product shapes:
//   ./gemm_ws_test [M] [N] [K] [mul]      D[M,N] = A[M,K] @ Wt[N,K]^T (* mul[M,N])
// prints us/launch of both, max abs difference, and checks a sample of entries against a host f64 sum.
#include "../../tensorflowasr_amd/csrc/gemm_fast.hip"
#include "gemm_ws.h"
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <random>
std::atomic<size_t> g_tfasr_launch_count{0};

template <int K, int CG, int MT, bool MUL>
static float run_ws(const ws::Args& a0, int iters) {
  ws::Args a = a0;
  const int RP = 64 * MT;
  a.npanels = (a.M + RP - 1) / RP;
  a.ngroups = (a.N + CG - 1) / CG;
  int spx = 32 / a.ngroups; if (spx < 1) spx = 1;
  while (spx > 1 && 8 * (spx - 1) * 1 >= a.npanels) --spx;
  a.spx = spx;
  const int grid = 8 * a.ngroups * spx;
  const size_t smem = (size_t)CG * K * 2;
  auto kern = ws::gemm_ws_nt_kernel<K, CG, MT, MUL, false>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
  printf("  ws<K=%d,CG=%d,MT=%d> grid %d (groups %d, spx %d, panels %d) smem %zu: %.2f us\n", K, CG, MT, grid, a.ngroups, spx, a.npanels, smem, ms / iters * 1e3);
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 19264, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
  const int mul = argc > 4 ? atoi(argv[4]) : 0;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K), hG((size_t)M * N);
  auto tobf = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); };
  auto tof = [](bf16_t b) { uint32_t u = ((uint32_t)b) << 16; float f; memcpy(&f, &u, 4); return f; };
  for (auto& v : hA) v = tobf(nd(rng));
  for (auto& v : hW) v = tobf(nd(rng) * 0.06f);
  for (auto& v : hG) v = tobf(nd(rng));
  bf16_t *A, *W, *G, *D0, *D1;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&G, hG.size() * 2); hipMalloc(&D0, (size_t)M * N * 2); hipMalloc(&D1, (size_t)M * N * 2);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(G, hG.data(), hG.size() * 2, hipMemcpyHostToDevice);
  hipMemset(D0, 0, (size_t)M * N * 2); hipMemset(D1, 0, (size_t)M * N * 2);
  // library kernel: D = A @ W^T (trans_b = 1: B stored [N, K]) with the swish' epilogue standing in for the multiplier when mul is on
  tfasr_gemm_args a; memset(&a, 0, sizeof(a));
  a.A = A; a.B = W; a.D = D0; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldd = N; a.trans_a = 0; a.trans_b = 1;
  a.nb1 = a.nb2 = 1; a.alpha = 1.f; a.beta = 1.f; a.dtype = TFASR_BF16; a.split_k = 1;
  if (mul) { a.dact_z = G; a.dact = TFASR_ACT_SWISH; a.drop_p = 0.1f; a.drop_seed = 1234; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { int st = tfasr_gemm_fast_try(a, 0); if (st != 0 && i == 0) printf("library status %d\n", st); }
  hipEventRecord(e0); for (int i = 0; i < 50; ++i) tfasr_gemm_fast_try(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("M %d N %d K %d mul %d\n  library (tiled, %s epilogue): %.2f us\n", M, N, K, mul, mul ? "swish' + dropout" : "plain", ms / 50 * 1e3);
  ws::Args w; memset(&w, 0, sizeof(w));
  w.A = A; w.lda = K; w.Wt = W; w.ldw = K; w.D = D1; w.ldd = N; w.alpha = 1.f; w.M = M; w.N = N; w.mul = mul ? G : nullptr; w.beta = 1.f;
#define RUN(KK, CG, MT) do { if (mul) run_ws<KK, CG, MT, true>(w, 50); else run_ws<KK, CG, MT, false>(w, 50); } while (0)
  if (K == 256) { RUN(256, 128, 2); RUN(256, 128, 1); RUN(256, 64, 2); RUN(256, 256, 1); RUN(256, 64, 1); }
  else if (K == 512) { RUN(512, 128, 1); RUN(512, 64, 2); RUN(512, 64, 1); }
  else if (K == 768) { RUN(768, 64, 1); }
  else if (K == 1024) { RUN(1024, 64, 1); RUN(1024, 32, 1); }
  hipDeviceSynchronize();
  // correctness of the LAST ws run (plain product only) against the library and a host sum
  std::vector<bf16_t> h0((size_t)M * N), h1((size_t)M * N);
  hipMemcpy(h0.data(), D0, h0.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), D1, h1.size() * 2, hipMemcpyDeviceToHost);
  double maxd = 0, maxref = 0; long bad = 0;
  for (int s = 0; s < 4000; ++s) {
    const long m = (long)(rng() % M), n = (long)(rng() % N);
    double acc = 0; for (int k = 0; k < K; ++k) acc += (double)tof(hA[m * K + k]) * tof(hW[n * K + k]);
    if (mul) acc *= tof(hG[m * N + n]);
    const double d = fabs(acc - tof(h1[m * N + n]));
    if (d > maxref) maxref = d;
    if (d > 0.02 * (fabs(acc) + 1.0)) ++bad;
  }
  if (!mul) for (size_t i = 0; i < h0.size(); ++i) { const double d = fabs((double)tof(h0[i]) - tof(h1[i])); if (d > maxd) maxd = d; }
  printf("  ws vs host f64 (4000 samples): max abs diff %.4g, outside 2%%: %ld;  ws vs library (all): max abs diff %.4g\n", maxref, bad, maxd);
  return 0;
}
