// Fused Macaron feed-forward module, data gradient (included by gemm_fast.hip behind ffn_fused.h; same helpers).
//
// Backward of FFModule.call + Residual.call (tensorflow_asr/models/encoders/conformer.py:101-109, models/layers/residual.py:58-62) under keras
// autodiff, for  y = x + res * drop2( drop1( swish( LN(x) W1 + b1 ) ) W2 + b2 ):
//     dz  = res * (dyd W2^T) . swish'(z) . mask1 / (1 - p)          dyd = dy with the second dropout's mask applied (the caller's)
//     dln = dz W1^T
//     dx  = dy + LayerNorm'(dln; x, mean, rstd, gamma)              (+ the dropped copy of dx for the next module's backward)
// Before (round 3 / forward-only fusion): GEMM (d -> 4d, swish' + dropout epilogue, reads z, writes dz) + GEMM (4d -> d, reads dz, writes
// dln) + LayerNorm backward (reads dln, x, dy): three launches, 186 MB of HBM traffic per module at [19k, 256] x 1024, 67.6 us.
// Now ONE launch per 64-row tile (4 waves, 2 x 2; two workgroups per CU: 80 KiB = 64 LDS granules each):
//   prologue  the tile's dyd rows become MFMA A fragments in REGISTERS (64 rows x 256 k per workgroup, 64 VGPRs per wave) for the whole kernel;
//   16 (= 4d / 64) hidden chunks c:  t = dyd W2[c, :]^T (wave tile 32 x 32, K = 256)  ->  dz_c = res t swish'(z_c) mask in the MFMA C layout
//             (z_c comes from HBM by LDS-DMA beside the weights) -> bf16 tile [64][64] in LDS, stored to HBM from there in whole 128-B rows
//             (the weight gradient W1' = ln^T dz reads it) ->  dln += dz_c W1[:, c]^T (wave tile 32 x 128, K = 64);
//   weights   W2[c, :] (rows of the [4d, d] matrix: k-contiguous) and W1[:, c] (64-column slices of its rows: k-contiguous) are both "direct"
//             B images - no transposing reads; single buffered, three barriers per chunk, counted vmcnt waits (schedule below);
//   epilogue  LayerNorm backward on the f32 dln tile through per-wave LDS strips: row sums over the 256 columns meet across the two column
//             waves in LDS, dx (+ its dropped copy) leaves in whole 256-B row pieces, the gamma / beta sums of the tile go to part[tile][512]
//             (folded with the other LayerNorms of the block by ln_bwd_fold_kernel: no atomics).
// dln stays f32 until the LayerNorm backward has consumed it (the three-launch route rounded it to bf16 in between).
// Shapes: d = 256, 4d a multiple of 64 and <= 1024, bf16, rows * 4d < 2^32.  Everything else: TFASR_STATUS_UNSUPPORTED -> the caller's route.
//
// Schedule per chunk c (vector-memory operations a wave issues, in order: W1(c) [8 pieces], W2(c+1) [8], z(c+1) [2], dz stores(c) [2]):
//   top:  wait W2(c), z(c) [only the 2 stores of chunk c-1 were issued after them]; barrier T; issue W1(c)   (W1 buffer, dz tile released)
//         GEMM1(c); dz(c) -> LDS tile
//         barrier A; issue W2(c+1), z(c+1)                                                                  (W2 buffer, z tile released)
//         dz(c) -> HBM; wait W1(c) [W2(c+1), z(c+1), 2 stores issued after it = 12]; barrier B; GEMM2(c)

struct FfnBwdArgs {
  const bf16_t* dyd; const bf16_t* z; const bf16_t* W1; const bf16_t* W2; const bf16_t* x;
  const float* gamma; const float* mean; const float* rstd; const bf16_t* add;
  bf16_t* dz; bf16_t* dx; bf16_t* dxd; float* part;
  long rows; int F; float res, drop_p; long seed1, seed_next;
  long long* dbg;  // TFASR_FFN_TIMING builds: per-phase cycle sums of workgroup 0 / wave 0
};

__global__ __launch_bounds__(256, 2) void ffn_fused_bwd_kernel(const FfnBwdArgs p) {
#ifdef TFASR_FFN_TIMING
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tp = __builtin_readcyclecounter();
#endif
  constexpr int D = 256, MR = 2, BMR = 64, NW = 4, WR = 32;
  constexpr int W2_OFF = 0, W1_OFF = 32768, SH_OFF = 65536, SZ_OFF = 73728;  // [4 slabs][64 f][64 k] | [256 n][64 k] | dz tile | z tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = (long)blockIdx.x * BMR;
  const int F = p.F, NC = F / 64;
  const int nrow_tile = (int)min((long)BMR, p.rows - m0);
  const bool counted = nrow_tile == BMR;

  // per-lane byte offsets of this wave's DMA pieces (loop invariant; the chunk moves the uniform base)
  uint32_t offw2[8], offw1[8], offz[2];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int pid = w + n * NW, pp = lane & 7;
    { const int s = pid >> 3, q = pid & 7, fr = q * 8 + (lane >> 3);
      offw2[n] = (uint32_t)((fr * D + s * 64 + ((pp ^ key_d(fr)) << 3)) * 2); }
    { const int nrow = pid * 8 + (lane >> 3);
      offw1[n] = (uint32_t)((nrow * F + ((pp ^ key_d(nrow)) << 3)) * 2); }
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int pid = w + n * NW, pp = lane & 7, row = pid * 8 + (lane >> 3);
    offz[n] = (uint32_t)((min(row, nrow_tile - 1) * F + ((pp ^ key_d(row)) << 3)) * 2);
  }
  auto issue_w2 = [&](int cc) {  // rows 64 cc .. 64 cc + 63 of W2 [F, 256]: four 64-k slabs of [64 f][64 k] (128-B rows)
    const char* src = reinterpret_cast<const char*>(p.W2 + (long)min(cc, NC - 1) * 64 * D);
#pragma unroll
    for (int n = 0; n < 8; ++n) glds16_s(src, offw2[n], smem + W2_OFF + __builtin_amdgcn_readfirstlane((w + n * NW) * 1024));
  };
  auto issue_w1 = [&](int cc) {  // columns 64 cc .. 64 cc + 63 of every row of W1 [256, F]: image [256 n][64 k]
    const char* src = reinterpret_cast<const char*>(p.W1 + min(cc, NC - 1) * 64);
#pragma unroll
    for (int n = 0; n < 8; ++n) glds16_s(src, offw1[n], smem + W1_OFF + __builtin_amdgcn_readfirstlane((w + n * NW) * 1024));
  };
  auto issue_z = [&](int cc) {   // the tile's rows of z[:, 64 cc .. 64 cc + 63]: image [64 rows][64 k]
    const char* src = reinterpret_cast<const char*>(p.z + m0 * F + min(cc, NC - 1) * 64);
#pragma unroll
    for (int n = 0; n < 2; ++n) glds16_s(src, offz[n], smem + SZ_OFF + __builtin_amdgcn_readfirstlane((w + n * NW) * 1024));
  };
  issue_w2(0);
  issue_z(0);

  // ---- prologue: this wave's 32 rows of dyd as MFMA A fragments (lane (r, g): row r, 8 consecutive k), resident for the whole kernel ----
  short8_t dyf[MR][8];
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const long row = m0 + min(wr * WR + i * 16 + r, nrow_tile - 1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) dyf[i][ks] = *reinterpret_cast<const short8_t*>(p.dyd + row * D + ks * 32 + g * 8);
  }

  float4_t acc1[MR][2], acc2[MR][8];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc2[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
  const uint32_t dthr = drop_thr(p.drop_p);
  const float dinv = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const uint32_t dkey1 = drop_key((uint64_t)p.seed1);  // drop_hash's key (pair indices < 2^32 here)
  FFN_TICK(0)  // prologue (loads issued)
  // loop-invariant per-lane slots of the C layout (row = wr*32 + i*16 + g*4 + e, column = wc*32 + j*16 + r of the chunk): byte address in the
  // [64][64] bf16 tiles (z and dz share the layout) and the element's share of the dropout pair index
  int toff[MR][2][4];
  uint32_t pidx[MR][2][4];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = wr * WR + i * 16 + g * 4 + e, k = wc * 32 + j * 16 + r;
        toff[i][j][e] = row * 128 + (((k >> 3) ^ key_d(row)) << 4) + (k & 7) * 2;
        pidx[i][j][e] = (uint32_t)(row * F + k) >> 1;
      }
  const bool hi_half = (r & 1) != 0;  // this lane's elements are the odd members of their dropout pairs

  auto gemm1 = [&]() {  // acc1 = dyd W2[c, :]^T: wave tile 32 x 32, K = 256, A from registers, B = rows of the W2 chunk (k-contiguous)
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      short8_t bf[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[kk][j] = frag_direct(smem + W2_OFF + kp * 8192, wc * 32 + j * 16 + r, kk * 4 + g);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dyf[i][2 * kp + kk], bf[kk][j], acc1[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto dz_tile = [&](int cc) {  // dz = res * acc1 * swish'(z) * mask / (1 - p), in the C layout; bf16 -> the dz tile
    const uint32_t pbase = (uint32_t)((uint64_t)(m0 * F + cc * 64) >> 1);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zv = bf16_to_f32(*reinterpret_cast<const bf16_t*>(smem + SZ_OFF + toff[i][j][e]));
          float v = p.res * acc1[i][j][e];
          v *= dswishf_(zv);
          const uint32_t hh = drop_mix(dkey1, pbase + pidx[i][j][e]);
          v = (hi_half ? (hh >> 16) : (hh & 0xffffu)) >= dthr ? v * dinv : 0.f;  // (drop_p = 0: threshold 0 keeps everything, dinv = 1)
          *reinterpret_cast<bf16_t*>(smem + SH_OFF + toff[i][j][e]) = f32_to_bf16(v);
        }
  };
  auto dz_store = [&](int cc) {  // the dz tile to HBM: every thread two 16-byte pieces (8 hidden units of one row)
    bf16_t* base = p.dz + m0 * F + cc * 64;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = ps * 32 + (threadIdx.x >> 3), pp = threadIdx.x & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + SH_OFF + row * 128 + ((pp ^ key_d(row)) << 4));
      if (counted || row < nrow_tile) *reinterpret_cast<uint4*>(base + (row * F + pp * 8)) = v;
    }
  };
  auto gemm2 = [&]() {  // acc2 += dz(c) W1[:, c]^T: wave tile 32 x 128, K = 64, B = the W1 chunk's rows (k-contiguous)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      short8_t af[MR];
#pragma unroll
      for (int i = 0; i < MR; ++i) af[i] = frag_direct(smem + SH_OFF, wr * WR + i * 16 + r, kk * 4 + g);
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        short8_t bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = frag_direct(smem + W1_OFF, wc * 128 + (jh * 4 + j) * 16 + r, kk * 4 + g);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc2[i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc2[i][jh * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  for (int c = 0; c < NC; ++c) {
    if (counted && c > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_only_barrier();   // T: W2(c), z(c) visible; every wave is done with GEMM2(c-1) (W1 buffer, dz tile)
    FFN_TICK(1)
    issue_w1(c);
    gemm1();
    FFN_TICK(2)
    dz_tile(c);
    FFN_TICK(3)
    lds_only_barrier();   // A: dz(c) complete; W2 buffer and z tile released
    issue_w2(c + 1);
    issue_z(c + 1);
    dz_store(c);
    FFN_TICK(4)
    if (counted) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_only_barrier();   // B: W1(c) visible
    FFN_TICK(5)
    gemm2();
    FFN_TICK(6)
  }

  // ---- epilogue: LayerNorm backward of the tile (keras LayerNormalization under autodiff; arithmetic of ln_bwd_vec_kernel on the f32 dln) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (clamped) DMA pieces have landed: the buffers become strips
  __syncthreads();
  constexpr int SLD = 128 + 4;
  float* sc = reinterpret_cast<float*>(smem) + w * (WR * SLD);            // this wave's [32 rows][128 columns] of dln (16.5 KiB)
  float* srs = reinterpret_cast<float*>(smem) + NW * (WR * SLD);         // [2 column waves][64 rows][2] row sums
  float* scs = srs + 2 * BMR * 2;                                        // [2 row waves][512] column sums
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[(i * 16 + g * 4 + e) * SLD + j * 16 + r] = acc2[i][j][e];
  const int prow = lane >> 4, c8 = (lane & 15) * 8, col0 = wc * 128 + c8;
  constexpr int NP = WR / 4;  // passes of 4 rows
  float gm[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) gm[q] = p.gamma[col0 + q];
  uint4 xraw[NP], araw[NP];
  float mu[NP], rs[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const long row = m0 + min(wr * WR + ps * 4 + prow, nrow_tile - 1);
    xraw[ps] = *reinterpret_cast<const uint4*>(p.x + row * D + col0);
    araw[ps] = p.add ? *reinterpret_cast<const uint4*>(p.add + row * D + col0) : make_uint4(0, 0, 0, 0);
    mu[ps] = p.mean[row]; rs[ps] = p.rstd[row];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the strip is this wave's own: program order + the wait make it visible)
  // phase 1: this wave's share (128 of the 256 columns) of each row's sums  s1 = sum g dln,  s2 = sum g dln xhat
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int rl = ps * 4 + prow;
    const float4 t0 = *reinterpret_cast<const float4*>(sc + rl * SLD + c8), t1 = *reinterpret_cast<const float4*>(sc + rl * SLD + c8 + 4);
    const float dl[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    float xv[8];
    unpack8(xraw[ps], xv);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float xh = (xv[q] - mu[ps]) * rs[ps], dg = dl[q] * gm[q]; s1 += dg; s2 += dg * xh; }
    s1 = row16_sum(s1);
    s2 = row16_sum(s2);
    if ((lane & 15) == 0) { srs[(wc * BMR + wr * WR + rl) * 2] = s1; srs[(wc * BMR + wr * WR + rl) * 2 + 1] = s2; }
  }
  __syncthreads();
  // phase 2: dx = add + rstd (g dln - mean_c(g dln) - xhat mean_c(g dln xhat)); gamma / beta sums of the tile's rows
  float ag[8], ab[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { ag[q] = 0.f; ab[q] = 0.f; }
  const uint32_t dthr2 = dthr;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int rl = ps * 4 + prow, rowl = wr * WR + rl;
    const float s1 = (srs[rowl * 2] + srs[(BMR + rowl) * 2]) * (1.f / D), s2 = (srs[rowl * 2 + 1] + srs[(BMR + rowl) * 2 + 1]) * (1.f / D);
    const float4 t0 = *reinterpret_cast<const float4*>(sc + rl * SLD + c8), t1 = *reinterpret_cast<const float4*>(sc + rl * SLD + c8 + 4);
    const float dl[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    float xv[8], o[8];
    unpack8(xraw[ps], xv);
    unpack8(araw[ps], o);
    if (rowl < nrow_tile) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float xh = (xv[q] - mu[ps]) * rs[ps];
        o[q] += rs[ps] * (dl[q] * gm[q] - s1 - xh * s2);
        ag[q] += dl[q] * xh;
        ab[q] += dl[q];
      }
      const long idx0 = (m0 + rowl) * D + col0;
      uint4 pk;
      pk.x = pack2_bf16(o[0], o[1]); pk.y = pack2_bf16(o[2], o[3]); pk.z = pack2_bf16(o[4], o[5]); pk.w = pack2_bf16(o[6], o[7]);
      *reinterpret_cast<uint4*>(p.dx + idx0) = pk;
      if (p.dxd) {  // dropout(dx) for the consumer's masked branch, from the ROUNDED dx (= tfasr_dropout(dx)); idx0 is even: one hash per pair
        float rr[8];
        unpack8(pk, rr);
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          const uint32_t hh = drop_hash((uint64_t)p.seed_next, ((uint64_t)idx0 >> 1) + (q >> 1));
          rr[q] = (hh & 0xffffu) >= dthr2 ? rr[q] * dinv : 0.f;
          rr[q + 1] = (hh >> 16) >= dthr2 ? rr[q + 1] * dinv : 0.f;
        }
        st8(p.dxd + idx0, rr);
      }
    }
  }
  // column sums: over the 4 row lanes of a wave by shuffles, over the two row waves through LDS, one plain store per column
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    ag[q] += __shfl_xor(ag[q], 16, 64); ag[q] += __shfl_xor(ag[q], 32, 64);
    ab[q] += __shfl_xor(ab[q], 16, 64); ab[q] += __shfl_xor(ab[q], 32, 64);
  }
  if (prow == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { scs[wr * 512 + col0 + q] = ag[q]; scs[wr * 512 + D + col0 + q] = ab[q]; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * D; t += 256) p.part[(long)blockIdx.x * 2 * D + t] = scs[t] + scs[512 + t];
#ifdef TFASR_FFN_TIMING
  FFN_TICK(7)  // epilogue
  if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) p.dbg[k] = ph[k];
#endif
}

constexpr int FFN_BWD_SMEM = 81920;
static int launch_ffn_fused_bwd(const FfnBwdArgs& a, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)ffn_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_BWD_SMEM); attr_done = true; }
  const long tiles = (a.rows + 63) / 64;
  hipLaunchKernelGGL(ffn_fused_bwd_kernel, dim3((unsigned)tiles), dim3(256), FFN_BWD_SMEM, stream, a);
  return TFASR_STATUS_SUCCESS;
}
