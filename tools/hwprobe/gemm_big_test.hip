// gemm_big (256-row tiles) against gemm_fast (128-row tiles) on the joint-network shapes: bitwise comparison + timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Iinclude tools/hwprobe/gemm_big_test.hip -o tools/hwprobe/gemm_big_test
#ifndef TFASR_PROBE_PLAIN  // -DTFASR_PROBE_PLAIN: no cycle counters in the kernels (s_memtime turns every counted LDS wait into lgkmcnt(0))
#define TFASR_GEMM_TIMING 1
#endif
#include "../../tensorflowasr_amd/csrc/gemm_fast.hip"
#include <vector>
std::atomic<size_t> g_tfasr_launch_count{0};
#include <stdio.h>
static uint16_t rnd_bf16(uint64_t i, uint64_t seed) {
  uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull ^ seed * 0xD1B54A32D192ED03ull;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  const float f = ((int)(x & 0xffff) - 32768) / 32768.f * 0.5f;
  uint32_t u; memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static void fill(bf16_t* d, size_t n, uint64_t seed) {
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = rnd_bf16(i, seed);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}
static size_t diff(const void* a, const void* b, size_t bytes, const char* what) {
  std::vector<unsigned char> x(bytes), y(bytes);
  hipMemcpy(x.data(), a, bytes, hipMemcpyDeviceToHost); hipMemcpy(y.data(), b, bytes, hipMemcpyDeviceToHost);
  size_t n = 0, first = (size_t)-1;
  for (size_t i = 0; i < bytes; ++i) if (x[i] != y[i]) { if (first == (size_t)-1) first = i; ++n; }
  printf("  %s: %zu of %zu bytes differ%s", what, n, bytes, n ? "" : "  (bitwise equal)\n");
  if (n) printf(", first at byte %zu\n", first);
  return n;
}
static void dump_phases(const char* what) {
#ifdef TFASR_GEMM_TIMING
  std::vector<long long> h(16384 + 4L * 2 * 2 * 256);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gemm_timing), h.size() * 8);
  for (int it = 0; it < 2; ++it)
    for (int grp = 0; grp < 2; ++grp) {
      double s[12] = {0};
      for (int b = 0; b < 256; ++b) for (int k = 0; k < 11; ++k) s[k] += h[12L * ((b * 2 + it) * 2 + grp) + k];
      printf("  %s tile %d group %c: bar %.0f L0 %.0f bar %.0f M0 %.0f bar %.0f L1 %.0f bar %.0f M1 %.0f | mainloop %.0f epilogue %.0f total %.0f\n", what, it, 'A' + grp,
             s[0] / 256, s[1] / 256, s[2] / 256, s[3] / 256, s[4] / 256, s[5] / 256, s[6] / 256, s[7] / 256, s[8] / 256, s[9] / 256, s[10] / 256);
      double e[3] = {0, 0, 0};
      for (int b = 0; b < 256; ++b) for (int k = 0; k < 3; ++k) e[k] += h[16384 + 4L * ((b * 2 + it) * 2 + grp) + k];
      printf("      epilogue: setup (next-tile DMA issue, labels, bias) %.0f  statistics %.0f  transposition + stores %.0f\n", e[0] / 256, e[1] / 256, e[2] / 256);
    }
#endif
}
static float timeit(const tfasr_gemm_args& a, int mode) {
  g_gemm_big_mode = mode;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) tfasr_gemm_fast_try(a, 0);
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) tfasr_gemm_fast_try(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10 * 1e3f;
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 830000, J = argc > 2 ? atoi(argv[2]) : 320, V = argc > 3 ? atoi(argv[3]) : 1000;
  bf16_t *X, *W, *D0, *D1, *G0, *G1; float *bias, *l0, *l1, *p0, *p1; int* lab;
  hipMalloc(&X, (size_t)M * J * 2); hipMalloc(&W, (size_t)J * V * 2); hipMalloc(&D0, (size_t)M * V * 2); hipMalloc(&D1, (size_t)M * V * 2);
  hipMalloc(&G0, (size_t)M * J * 2); hipMalloc(&G1, (size_t)M * J * 2);
  hipMalloc(&bias, V * 4); hipMalloc(&l0, (size_t)M * 16 * 8); hipMalloc(&l1, (size_t)M * 16 * 8); hipMalloc(&p0, (size_t)M * 8); hipMalloc(&p1, (size_t)M * 8);
  hipMalloc(&lab, (size_t)M * 4);
  fill(X, (size_t)M * J, 1); fill(W, (size_t)J * V, 2);
  { std::vector<float> b(V); for (int i = 0; i < V; ++i) b[i] = 0.01f * (i % 17 - 8); hipMemcpy(bias, b.data(), V * 4, hipMemcpyHostToDevice);
    std::vector<int> l(M); for (int i = 0; i < M; ++i) l[i] = (int)((i * 2654435761u) % V); hipMemcpy(lab, l.data(), (size_t)M * 4, hipMemcpyHostToDevice); }
  hipMemset(D0, 0, (size_t)M * V * 2); hipMemset(D1, 0, (size_t)M * V * 2); hipMemset(G0, 0, (size_t)M * J * 2); hipMemset(G1, 0, (size_t)M * J * 2);
  hipMemset(l0, 0, (size_t)M * 128); hipMemset(l1, 0, (size_t)M * 128); hipMemset(p0, 0, (size_t)M * 8); hipMemset(p1, 0, (size_t)M * 8);
  tfasr_gemm_args a; memset(&a, 0, sizeof(a));
  a.nb1 = a.nb2 = 1; a.alpha = 1.f; a.beta = 1.f; a.dtype = TFASR_BF16; a.split_k = 1;
  // joint vocabulary projection with log-softmax statistics: D[M,V] = X[M,J] W[J,V] + bias
  a.A = X; a.B = W; a.M = M; a.N = V; a.K = J; a.lda = J; a.ldb = V; a.ldd = V; a.bias = bias; a.row_label = lab; a.lse_parts = 16;
  a.D = D0; a.lse_part = l0; a.pick = p0; const float t0 = timeit(a, 0);
  a.D = D1; a.lse_part = l1; a.pick = p1; const float t1 = timeit(a, 1);
  printf("joint forward [%d,%d,%d] + LSE: 128-row tiles %.1f us (%.0f TFLOP/s), 256-row tiles %.1f us (%.0f TFLOP/s)\n", M, V, J, t0, 2e-6 * M * V * J / t0, t1, 2e-6 * M * V * J / t1);
  dump_phases("fwd+lse");
  diff(D0, D1, (size_t)M * V * 2, "logits"); 
  { // the two kernels may report different (max, sum) pairs per slice (the 256-row one uses one reference maximum per 128 columns): what must
    // agree is the merged log-sum-exp of each row
    std::vector<float> x((size_t)M * 32), y((size_t)M * 32); hipMemcpy(x.data(), l0, (size_t)M * 128, hipMemcpyDeviceToHost); hipMemcpy(y.data(), l1, (size_t)M * 128, hipMemcpyDeviceToHost);
    double worst = 0; size_t nbad = 0;
    auto lse = [](const float* p) { double mx = -1e300; for (int c = 0; c < 16; ++c) if (p[2 * c + 1] > 0 && p[2 * c] > mx) mx = p[2 * c];
      double sm = 0; for (int c = 0; c < 16; ++c) if (p[2 * c + 1] > 0) sm += (double)p[2 * c + 1] * exp((double)p[2 * c] - mx); return mx + log(sm); };
    for (size_t i = 0; i < (size_t)M; ++i) { const double a = lse(&x[i * 32]), b = lse(&y[i * 32]); const double d = fabs(a - b) / (fabs(a) + 1e-6); if (d > worst) worst = d; if (!(d < 1e-5)) ++nbad; }
    printf("  merged row log-sum-exp: worst relative difference %.3g, %zu rows beyond 1e-5\n", worst, nbad); } diff(p0, p1, (size_t)M * 8, "picks");
  // plain (no statistics)
  a.lse_part = nullptr; a.pick = nullptr; a.row_label = nullptr; a.lse_parts = 0;
  hipMemset(D1, 0, (size_t)M * V * 2);
  a.D = D0; const float t2 = timeit(a, 0); a.D = D1; const float t3 = timeit(a, 1);
  printf("joint forward plain: %.1f us vs %.1f us\n", t2, t3); diff(D0, D1, (size_t)M * V * 2, "logits");
  // data gradient: G[M,J] = D[M,V] W^T
  memset(&a, 0, sizeof(a));
  a.nb1 = a.nb2 = 1; a.alpha = 1.f; a.beta = 1.f; a.dtype = TFASR_BF16; a.split_k = 1;
  a.A = D0; a.B = W; a.trans_b = 1; a.M = M; a.N = J; a.K = V; a.lda = V; a.ldb = V; a.ldd = J;
  a.D = G0; const float t4 = timeit(a, 0); a.D = G1; const float t5 = timeit(a, 1);
  printf("joint data gradient [%d,%d,%d]: 128-row tiles %.1f us (%.0f TFLOP/s), 256x320 tiles %.1f us (%.0f TFLOP/s)\n", M, J, V, t4, 2e-6 * M * V * J / t4, t5, 2e-6 * M * V * J / t5);
  dump_phases("dgrad");
  diff(G0, G1, (size_t)M * J * 2, "gradient (bitwise; the K tail is grouped differently)");
  { std::vector<uint16_t> x((size_t)M * J), y((size_t)M * J); hipMemcpy(x.data(), G0, (size_t)M * J * 2, hipMemcpyDeviceToHost); hipMemcpy(y.data(), G1, (size_t)M * J * 2, hipMemcpyDeviceToHost);
    double worst = 0, big = 0;
    for (size_t i = 0; i < (size_t)M * J; ++i) { uint32_t a = (uint32_t)x[i] << 16, b = (uint32_t)y[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
      worst = fmax(worst, fabs((double)fa - fb)); big = fmax(big, fabs((double)fa)); }
    printf("  gradient: worst absolute difference %.4g (largest value %.4g)\n", worst, big); }
  return 0;
}
