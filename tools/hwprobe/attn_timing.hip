// Phase timing of the fused attention forward (production kernel built with TFASR_ATTN_TIMING cycle-counter hooks).
#define TFASR_ATTN_TIMING 1
#include "../../tensorflowasr_amd/csrc/attn_fused.hip"
#include <vector>
#include <stdlib.h>
int main(int argc, char** argv) {
  const int B = 32, H = 4, T = argc > 1 ? atoi(argv[1]) : 595, HD = 256;
  bf16_t *qkv, *pext, *out; float *u, *v, *lse; int32_t* len;
  hipMalloc(&qkv, (size_t)B * T * 3 * HD * 2); hipMalloc(&pext, (size_t)2 * T * HD * 2); hipMalloc(&out, (size_t)B * T * HD * 2);
  hipMalloc(&u, HD * 4); hipMalloc(&v, HD * 4); hipMalloc(&lse, (size_t)B * H * T * 4); hipMalloc(&len, B * 4);
  hipMemset(qkv, 0x3c, (size_t)B * T * 3 * HD * 2); hipMemset(pext, 0x3c, (size_t)2 * T * HD * 2); hipMemset(u, 0, HD * 4); hipMemset(v, 0, HD * 4);
  std::vector<int32_t> hl(B, T);
  if (argc > 2 && atoi(argv[2]) == 1) {  // ragged: lengths spread over [T/6, T] (a LibriSpeech-shaped batch padded to its longest utterance)
    for (int b = 0; b < B; ++b) hl[b] = T / 6 + (int)((long)(T - T / 6) * ((b * 37) % B) / (B - 1));
  }
  hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) tfasr_relattn_fused_fwd(qkv, u, v, pext, len, out, lse, B, H, T, 64, 0.125f, 1, 0, -1, TFASR_BF16, 0);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) tfasr_relattn_fused_fwd(qkv, u, v, pext, len, out, lse, B, H, T, 64, 0.125f, 1, 0, -1, TFASR_BF16, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int nblk = ((T + 63) / 64) * H * B;
  std::vector<long long> h(5L * nblk);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_attn_timing), h.size() * 8);
  double ph[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < nblk; ++b) for (int k = 0; k < 5; ++k) ph[k] += h[5L * b + k];
  const int njb = (T + 63) / 64;
  printf("T %d: %.1f us/launch, %d blocks x %d key blocks; cycles per key block: load+wait %.0f, scores %.0f, softmax %.0f, PV %.0f, end barrier %.0f\n", T, ms / 20 * 1e3,
         nblk, njb, ph[0] / nblk / njb, ph[1] / nblk / njb, ph[2] / nblk / njb, ph[3] / nblk / njb, ph[4] / nblk / njb);
  // backward, query side
  {
    bf16_t *dout, *dqu, *dpos; float* dvec;
    const int ldp = (2 * T + 7) / 8 * 8;
    hipMalloc(&dout, (size_t)B * T * HD * 2); hipMalloc(&dqu, (size_t)B * T * HD * 2); hipMalloc(&dpos, (size_t)B * H * T * ldp * 2); hipMalloc(&dvec, (size_t)B * H * T * 4);
    hipMemset(dout, 0x3c, (size_t)B * T * HD * 2);
    for (int i = 0; i < 3; ++i) tfasr_relattn_fused_bwd_q(qkv, u, v, pext, len, out, dout, lse, dqu, dpos, dvec, B, H, T, 64, ldp, 0.125f, 1, TFASR_BF16, 0);
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) tfasr_relattn_fused_bwd_q(qkv, u, v, pext, len, out, dout, lse, dqu, dpos, dvec, B, H, T, 64, ldp, 0.125f, 1, TFASR_BF16, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_attn_timing), h.size() * 8);
    double q[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < nblk; ++b) for (int k = 0; k < 5; ++k) q[k] += h[5L * b + k];
    printf("bwd_q T %d: %.1f us/launch; cycles per key block: load+wait %.0f, score/dP/window MFMAs %.0f, dS + dpos stores %.0f, dS image + dQ MFMAs %.0f, end barrier %.0f\n", T,
           ms / 20 * 1e3, q[0] / nblk / njb, q[1] / nblk / njb, q[2] / nblk / njb, q[3] / nblk / njb, q[4] / nblk / njb);
  }
  return 0;
}
