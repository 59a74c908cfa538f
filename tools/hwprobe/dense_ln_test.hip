// Fused Dense data gradient + LayerNorm backward (csrc/dense_ln.h) against the library's plain product of the same shape, stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTFASR_DLN_TIMING tools/hwprobe/dense_ln_test.hip -o tools/hwprobe/dense_ln_test
//   ./dense_ln_test [rows] [K]     us/launch of both, and the phase clocks of wave 0 (median workgroup): prologue / waits / slab bodies / epilogue
#include "../../tensorflowasr_amd/csrc/gemm_fast.hip"
#include <vector>
#include <random>
#include <algorithm>
std::atomic<size_t> g_tfasr_launch_count{0};

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 23776;
  const int K = argc > 2 ? atoi(argv[2]) : 1024, d = 256;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto tobf = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); };
  std::vector<bf16_t> hdy((size_t)rows * K), hW((size_t)d * K), hx((size_t)rows * d);
  for (auto& v : hdy) v = tobf(nd(rng));
  for (auto& v : hW) v = tobf(nd(rng) * 0.06f);
  for (auto& v : hx) v = tobf(nd(rng));
  bf16_t *dy, *W, *x, *add, *dx, *dxd, *dln;
  float *gamma, *mean, *rstd, *part;
  hipMalloc(&dy, hdy.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&x, hx.size() * 2); hipMalloc(&add, hx.size() * 2);
  hipMalloc(&dx, hx.size() * 2); hipMalloc(&dxd, hx.size() * 2); hipMalloc(&dln, hx.size() * 2);
  hipMalloc(&gamma, d * 4); hipMalloc(&mean, rows * 4); hipMalloc(&rstd, rows * 4); hipMalloc(&part, 256 * 2 * d * 4);
  hipMemcpy(dy, hdy.data(), hdy.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(add, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> ones(std::max<long>(rows, d), 1.f);
  hipMemcpy(gamma, ones.data(), d * 4, hipMemcpyHostToDevice); hipMemset(mean, 0, rows * 4); hipMemcpy(rstd, ones.data(), rows * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  tfasr_gemm_args a; memset(&a, 0, sizeof(a));
  a.A = dy; a.B = W; a.D = dln; a.M = (int)rows; a.N = d; a.K = K; a.lda = K; a.ldb = K; a.ldd = d; a.trans_a = 0; a.trans_b = 1;
  a.nb1 = a.nb2 = 1; a.alpha = 1.f; a.beta = 1.f; a.dtype = TFASR_BF16; a.split_k = 1;
  for (int i = 0; i < 3; ++i) tfasr_gemm_fast_try(a, 0);
  hipEventRecord(e0); for (int i = 0; i < 50; ++i) tfasr_gemm_fast_try(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("rows %ld K %d\n  library product alone: %.2f us\n", rows, K, ms / 50 * 1e3);
  for (int drop = 0; drop < 2; ++drop) {
    auto run = [&]() { return tfasr_dense_ln_bwd(dy, W, K, x, gamma, mean, rstd, add, dx, part, 256, drop ? dxd : nullptr, 0.1f, 77, rows, d, 1.f, TFASR_BF16, nullptr); };
    for (int i = 0; i < 3; ++i) { const int st = run(); if (st && i == 0) printf("status %d\n", st); }
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("  dense_ln_bwd (%s second output): %.2f us\n", drop ? "with the dropped" : "no", ms / 50 * 1e3);
  }
#ifdef TFASR_DLN_TIMING
  hipDeviceSynchronize();
  std::vector<long long> t(8 * 1024);
  hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_dln_timing), t.size() * 8);
  const char* names[6] = {"prologue (slab 0)", "wait + barrier (sum)", "slab bodies (sum)", "-", "epilogue", "column sums"};
  for (int q = 0; q < 6; ++q) {
    std::vector<long long> v;
    for (int b = 0; b < 240; ++b) v.push_back(t[8 * b + q]);
    std::sort(v.begin(), v.end());
    printf("  %-26s median %8lld  max %8lld clocks\n", names[q], v[v.size() / 2], v.back());
  }
#endif
  return 0;
}
