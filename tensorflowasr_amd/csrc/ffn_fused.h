// Fused Macaron feed-forward module, forward (included by gemm_fast.hip; uses its LDS-DMA / fragment helpers).
//
// Reference: FFModule.call (tensorflow_asr/models/encoders/conformer.py:101-109) + Residual.call (models/layers/residual.py:58-62):
//     y = x + factor * dropout2( dropout1( swish( LN(x) W1 + b1 ) ) W2 + b2 )
// Before (round 3): LayerNorm kernel + GEMM (d -> 4d, swish epilogue, writes z and h) + GEMM (4d -> d, reads h): 168 MB of HBM traffic
// per module at [19k, 256] x 1024, three launches, 59 us (block GEMMs at 0.09-0.13 MFMA busy, 0.29 of HBM).
// Now ONE launch.  A workgroup owns BMR = 32 * NWR rows (NWR row-waves x 2 column-waves, 64 threads each):
//   prologue  LayerNorm of its rows (32 lanes per row, the arithmetic of ln_fwd_vec_kernel), ln / mean / rstd stored for the backward;
//             the normalised tile goes through LDS once and stays in REGISTERS as MFMA A fragments (32 rows x 256 k per wave = 64 VGPRs)
//             for the whole kernel - the first product re-reads only weights;
//   16 (= 4d / 64) hidden chunks c:  z_c = ln W1[:, c] + b1  (wave tile 32 x 32, K = 256)  ->  z_c and h_c = dropout(swish(z_c)) pass through
//             one LDS tile [BMR][64] (stored to HBM from there in whole 128-B rows: the backward reads them) ->  acc += h_c W2[c, :]
//             (wave tile 32 x 128, K = 64).  The 4d-wide hidden never exists as a GEMM operand in HBM;
//   weights   W1[:, c] (32 KiB) and W2[c, :] (32 KiB) stream L2 -> LDS by DMA (global_load_lds), double buffered: chunk c+1 is issued at the
//             top of chunk c and has the whole chunk to land;
//   epilogue  + b2, dropout2, x + factor * (.) through a per-wave LDS strip, whole 256-B row pieces to HBM.
// Same arithmetic as the three-kernel route (bf16 operands, f32 accumulation in the same k order, z = bf16(acc + b1), h = bf16(dropout(swish(
// f32 z))), identical dropout hash / element indices), so the saved tensors feed the unchanged backward.
// Shapes: d = 256 (K of the first product = 4 slabs of 64), 4d a multiple of 64, bf16.  Everything else: TFASR_STATUS_UNSUPPORTED -> caller's
// three-launch route.

struct FfnArgs {
  const bf16_t* x; const float* gamma; const float* beta; const bf16_t* W1; const float* b1; const bf16_t* W2; const float* b2;
  bf16_t* y; bf16_t* ln; float* mean; float* rstd; bf16_t* z; bf16_t* h;
  long rows; int F; float eps, res, drop_p; long seed1, seed2;
  int zfactor;     // 1: the `z` output holds the backward's factor swish'(z) * mask1 / (1 - p) instead of the pre-activation (tfasr_ffn_fused_fwd2)
  long long* dbg;  // TFASR_FFN_TIMING builds: per-phase cycle sums of workgroup 0 / wave 0 (tools/ffn_fused_check.py)
};

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef TFASR_FFN_TIMING
#define FFN_TICK(k) { const long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - tp; tp = t_; }
#else
#define FFN_TICK(k)
#endif

// Schedule (steady state, chunk c; W1(k) / W2(k) / h(k) live in buffer k % 2 of their double buffers):
//   P1   rowpass(c)  [VALU: z(c) -> HBM, h(c) = dropout(swish(z(c))) in place in sH[c%2] and -> HBM]   ||   GEMM1(c+1)  [MFMA, W1(c+1)]
//        wait own DMA pieces of W2(c); barrier B1; issue W1(c+3) into the buffer GEMM1(c+1) just released
//   P2   GEMM2(c)  [MFMA: acc += h(c) W2(c)]   ||   zlds(c+1)  [VALU: GEMM1(c+1)'s accumulators + b1 -> sH[(c+1)%2]]
//        wait own DMA pieces of W1(c+2); barrier B2; issue W2(c+2) into the buffer GEMM2(c) just released
// One wave per SIMD (the 512-register budget holds the normalised rows, both accumulators and the pieces' offsets without a spill - with two
// waves per SIMD, 256 registers each, part of the A fragments spilled, and a spill reload inside the chunk loop is a vector-memory LOAD that
// drains the DMA queue); the matrix instructions of a phase execute under the vector instructions issued behind them.  Two barriers per
// chunk; every DMA has more than a whole chunk to land.  Counted waits: the vector-memory
// operations a wave issues are, in order, stores(c) [S = 2 MR: MR 16-B stores per tensor], W1(c+3) [P pieces], W2(c+2) [P pieces] per chunk,
// retired in order - so before B1 the W2(c) pieces are complete once at most 2S + 2P operations are outstanding, before B2 the W1(c+2)
// pieces once at most S + 2P are.  Past the last chunk the issue points keep issuing (source clamped to the last chunk, target = a buffer
// nothing reads any more) so the counts stay valid; tail tiles / no z / the first chunk drain instead.
// WGS = 2: the two-workgroups-per-CU variant (MR = 2, 64 rows): weights SINGLE buffered (76 KiB of LDS per workgroup, <= 256 registers per
// wave), three barriers per chunk - inside one workgroup every wait is exposed, but the second workgroup of the CU runs its vector phase
// under the first one's matrix phase / DMA wait and vice versa (independent barriers: the two drift apart by themselves).
//   top:  wait W1(c); barrier T; issue W2(c)      (W2 buffer and sH released by GEMM2(c-1))
//         GEMM1(c); zlds(c)
//         barrier A; issue W1(c+1)                (W1 buffer released)
//         rowpass(c); wait W2(c); barrier B; GEMM2(c)
// ZF: the `z` tensor is written as the data gradient's FACTOR g = swish'(z) * mask1 / (1 - p) (bf16) instead of the pre-activation: the
// backward's d(4d -> d) product then multiplies by g in its epilogue (gemm_fast E_MUL) instead of recomputing swish' (v_exp + v_rcp + 6)
// and the dropout hash per element - what bounded that epilogue (53 us per launch against 23 for the bare product).  The row pass has
// sigmoid(z) and the mask at hand: five more vector instructions per element here.  h is unchanged (same expression as swishf_).
// DENSE: only the first half - out = LayerNorm(x) W1 + b1 (`z` = the output [rows, F]; W2 / b2 / h / y unused): the LayerNorm in front of the
// fused q/k/v projection and of the ConvModule's first pointwise conv with its Dense layer in one launch (tfasr_ln_dense_fwd).  The weight
// chunks alternate between the two 32-KiB buffers (the W2 buffer is idle), chunk c + 1 is issued at the top of chunk c: two barriers per chunk.
template <int MR, int WGS, bool ZF = false, bool DENSE = false>  // 16 * MR rows per wave; 4 waves (2 x 2): BMR = 32 * MR rows per workgroup
__global__ __launch_bounds__(256, WGS) void ffn_fused_fwd_kernel(const FfnArgs p) {
  static_assert(!DENSE || (WGS == 2 && !ZF), "the Dense-only mode is built on the two-workgroups-per-CU schedule");
#ifdef TFASR_FFN_TIMING
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tp = __builtin_readcyclecounter();
#endif
  constexpr int D = 256, BMR = 32 * MR, NW = 4, NT = 256, WR = 16 * MR;
  constexpr int PCS = 8;                         // DMA pieces per wave per operand chunk (32 pieces of 1 KiB)
  constexpr int NBUF = WGS == 2 ? 1 : 2;         // weight / h-tile buffers per kind
  constexpr int W1_OFF = 0, W2_OFF = NBUF * 32768;  // W1 chunk images NBUF x (4 slabs x 8 KiB) | W2 chunk images NBUF x (2 halves x 16 KiB)
  constexpr int SH_OFF = 2 * NBUF * 32768, SH_BYTES = BMR * 128;  // h / z tiles NBUF x [BMR][64] bf16 (direct A layout, XOR-swizzled 16-B chunks)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sB1 = reinterpret_cast<float*>(smem + SH_OFF + NBUF * SH_BYTES);  // b1 [F] (no vector-memory LOAD may sit in the chunk loop)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = (long)blockIdx.x * BMR;
  const int F = p.F, NC = F / 64;
  const int nrow_tile = (int)min((long)BMR, p.rows - m0);  // rows of this tile that exist
  const bool counted = nrow_tile == BMR && p.z != nullptr && (DENSE || p.h != nullptr);  // the store count per chunk is the S = 2 MR the counted waits assume

  // per-lane byte offsets of this wave's DMA pieces inside a chunk (loop invariant; the chunk moves the uniform base)
  uint32_t off1[PCS], off2[PCS];
#pragma unroll
  for (int n = 0; n < PCS; ++n) {
    const int pid = w + n * NW;
    { const int s = pid >> 3, q = pid & 7, k = q * 8 + (lane >> 3), pp = lane & 7;
      off1[n] = (uint32_t)(((s * 64 + k) * F + ((pp ^ key_t64(k)) << 3)) * 2); }
    { const int hf = pid >> 4, q = pid & 15, k = q * 4 + (lane >> 4), pp = lane & 15;
      off2[n] = (uint32_t)((k * D + ((hf * 16 + (pp ^ key_t(k))) << 3)) * 2); }
  }
  auto issue_w1 = [&](int cc) {  // W1[:, 64 cc .. 64 cc + 64) -> W1 buffer cc % 2: slab s = k / 64, image [64 k][64 n] (128-B k-rows)
    char* base = smem + (DENSE ? ((cc & 1) ? W2_OFF : W1_OFF) : W1_OFF + (cc & (NBUF - 1)) * 32768);
    const char* src = reinterpret_cast<const char*>(p.W1 + min(cc, NC - 1) * 64);
#pragma unroll
    for (int n = 0; n < PCS; ++n) glds16_s(src, off1[n], base + __builtin_amdgcn_readfirstlane((w + n * NW) * 1024));
  };
  auto issue_w2 = [&](int cc) {  // W2[64 cc .. 64 cc + 64, :] -> W2 buffer cc % 2: two column halves, image [64 k][128 n] (256-B k-rows)
    char* base = smem + W2_OFF + (cc & (NBUF - 1)) * 32768;
    const char* src = reinterpret_cast<const char*>(p.W2 + (long)min(cc, NC - 1) * 64 * D);
#pragma unroll
    for (int n = 0; n < PCS; ++n) glds16_s(src, off2[n], base + __builtin_amdgcn_readfirstlane((w + n * NW) * 1024));
  };
  issue_w1(0);  // land under the LayerNorm prologue
  if constexpr (WGS == 1) issue_w2(0);

  // ---- prologue: LayerNorm of the tile's rows (conformer.py:59-64,102; keras epsilon 1e-3) ----
  // temporary image of the normalised tile, 4 slabs [BMR][64 k] = BMR x 512 B <= 48 KiB: W2 buffer 1 + the h tiles behind it (both idle until the
  // fragments are in registers; W1(0) / W2(0) are landing in buffers 0 meanwhile)
  static_assert(WGS == 2 ? BMR * 512 <= 32768 : BMR * 512 <= 32768 + 2 * SH_BYTES, "LN image must fit W2 buffer 1 + the h tiles");
  char* sT = smem + W2_OFF + (WGS == 2 ? 0 : 32768);  // (WGS 2: the single W2 buffer - W2(0) is issued after the fragments are in registers)
  {
    const int li = threadIdx.x & 31, c0 = li * 8;
    float gm[8], bt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { gm[q] = p.gamma[c0 + q]; bt[q] = p.beta[c0 + q]; }
    constexpr int RP = NT / 32;  // rows per pass (8 passes)
    uint4 raw[BMR / RP];  // every row load of the tile is issued before anything is stored (a store between them would serialise the passes)
#pragma unroll
    for (int ps = 0; ps < BMR / RP; ++ps) {
      const int row = ps * RP + (threadIdx.x >> 5);
      raw[ps] = *reinterpret_cast<const uint4*>(p.x + m0 * D + (min(row, nrow_tile - 1) * D + c0));
    }
#pragma unroll
    for (int ps = 0; ps < BMR / RP; ++ps) {
      const int row = ps * RP + (threadIdx.x >> 5);
      float v[8];
      unpack8(raw[ps], v);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
      s = row16_sum(s);
      s = xor16_sum(s);
      const float mean = s * (1.f / D);
      float qq = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float dd = v[q] - mean; qq += dd * dd; }
      qq = row16_sum(qq);
      qq = xor16_sum(qq);
      const float rstd = rsqrtf(qq * (1.f / D) + p.eps);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (v[q] - mean) * rstd * gm[q] + bt[q];
      uint4 pk;
      pk.x = pack2_bf16(v[0], v[1]); pk.y = pack2_bf16(v[2], v[3]); pk.z = pack2_bf16(v[4], v[5]); pk.w = pack2_bf16(v[6], v[7]);
      *reinterpret_cast<uint4*>(sT + (li >> 3) * (BMR * 128) + row * 128 + (((li & 7) ^ key_d(row)) << 4)) = pk;
      if (row < nrow_tile) {
        *reinterpret_cast<uint4*>(p.ln + m0 * D + (row * D + c0)) = pk;
        if (li == 0) { (p.mean + m0)[row] = mean; (p.rstd + m0)[row] = rstd; }
      }
    }
  }
  for (int i = threadIdx.x; i < F; i += NT) sB1[i] = p.b1[i];
  __syncthreads();
  short8_t lnf[MR][8];  // this wave's 32 rows x 256 k as MFMA A fragments, resident for the whole kernel
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) lnf[i][ks] = frag_direct(sT + (ks >> 1) * (BMR * 128), wr * WR + i * 16 + r, (ks & 1) * 4 + g);
  FFN_TICK(0)  // prologue

  float4_t acc1[MR][2], acc2[MR][8];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc2[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
  const uint32_t dthr = drop_thr(p.drop_p);
  const float dinv = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const uint32_t dkey1 = drop_key((uint64_t)p.seed1);  // drop_hash's key (pair indices < 2^32 here)

  auto gemm1 = [&](int cc, bool fence = true) {  // acc1 = ln W1[:, cc]: wave tile 32 x 32, K = 256, A from registers
    const char* sW1 = smem + (DENSE ? ((cc & 1) ? W2_OFF : W1_OFF) : W1_OFF + (cc & (NBUF - 1)) * 32768);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {  // one 64-k slab at a time: 4 B fragments in flight (register budget: 2 waves per SIMD = 256 VGPRs each)
      short8_t bf[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[kk][j] = frag_trans<64>(sW1 + kp * 8192, wc * 32 + j * 16, kk * 32 + g * 8, r);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[kk][j], lnf[i][2 * kp + kk], acc1[i][j], 0, 0, 0);  // (transposed: see zlds)
      if (fence) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // GEMM1's accumulators are TRANSPOSED (the operand slots of the MFMA swapped: the A and B fragment layouts are the same): lane (r, g) of
  // tile (i, j) holds row i*16 + r, FOUR CONSECUTIVE hidden columns j*16 + g*4 + e - one 8-byte LDS store per tile (they were four 2-byte
  // stores into four rows: 16 per lane and chunk), the bias a 16-byte read
  auto zlds = [&](int cc) {  // acc1 + b1 -> z (bf16) -> sH[cc % 2]
    char* sH = smem + SH_OFF + (cc & (NBUF - 1)) * SH_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = wc * 32 + j * 16 + g * 4;
      const float4 bz = *reinterpret_cast<const float4*>(sB1 + cc * 64 + k);
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int row = wr * WR + i * 16 + r;
        uint2 v;
        v.x = pack2_bf16(acc1[i][j][0] + bz.x, acc1[i][j][1] + bz.y);
        v.y = pack2_bf16(acc1[i][j][2] + bz.z, acc1[i][j][3] + bz.w);
        *reinterpret_cast<uint2*>(sH + row * 128 + (((k >> 3) ^ key_d(row)) << 4) + (k & 7) * 2) = v;
      }
    }
  };
  // FAST (full tile, z and h stored): no branch inside, so the pass and the matrix instructions of the other product form ONE basic block
  auto rowpass = [&](int cc, auto FAST_) {  // every thread owns 16-byte pieces (8 hidden units of one row): z -> HBM, h = dropout(swish(z)) -> sH in place, -> HBM
    constexpr bool FAST = decltype(FAST_)::value;
    char* sH = smem + SH_OFF + (cc & (NBUF - 1)) * SH_BYTES;
    constexpr int RPP = NT / 8;
    bf16_t* zbase = p.z ? p.z + m0 * F + cc * 64 : nullptr;  // uniform bases; the per-lane part is a 32-bit offset
    bf16_t* hbase = p.h ? p.h + m0 * F + cc * 64 : nullptr;
    // dropout hash of an element PAIR (common.h drop_hash) with everything that does not depend on the pair hoisted: the key is the seed's
    // share of the hash, the pair index fits 32 bits (rows * F < 2^33 elements, checked by the caller), so a pair costs the finaliser alone
    const uint32_t pbase = (uint32_t)((uint64_t)(m0 * F + cc * 64) >> 1);
#pragma unroll
    for (int ps = 0; ps < BMR / RPP; ++ps) {
      const int row = ps * RPP + (threadIdx.x >> 3), pp = threadIdx.x & 7;
      char* sp = sH + row * 128 + ((pp ^ key_d(row)) << 4);
      const uint4 zv = *reinterpret_cast<const uint4*>(sp);
      const int off = row * F + pp * 8;
      if constexpr (DENSE) {
        if constexpr (FAST) *reinterpret_cast<uint4*>(zbase + off) = zv;
        else if (row < nrow_tile) *reinterpret_cast<uint4*>(zbase + off) = zv;
        continue;
      }
      if constexpr (!ZF) {
        if constexpr (FAST) *reinterpret_cast<uint4*>(zbase + off) = zv;
        else if (zbase && row < nrow_tile) *reinterpret_cast<uint4*>(zbase + off) = zv;
      }
      float hv[8];
      [[maybe_unused]] float gv[8];
      unpack8(zv, hv);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float sg = sigmoidf_(hv[q]);
        if constexpr (ZF) gv[q] = sg * (1.f + hv[q] * (1.f - sg));  // dswishf_
        hv[q] = hv[q] * sg;                                          // swishf_
      }
      {  // (drop_p = 0: threshold 0 keeps everything, dinv = 1 - no branch)
        const uint32_t pr0 = pbase + ((uint32_t)off >> 1);  // off is even: one dropout hash serves an element pair
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          const uint32_t hh = drop_mix(dkey1, pr0 + (q >> 1));
          const bool k0 = (hh & 0xffffu) >= dthr, k1 = (hh >> 16) >= dthr;
          hv[q] = k0 ? hv[q] * dinv : 0.f;
          hv[q + 1] = k1 ? hv[q + 1] * dinv : 0.f;
          if constexpr (ZF) { gv[q] = k0 ? gv[q] * dinv : 0.f; gv[q + 1] = k1 ? gv[q + 1] * dinv : 0.f; }
        }
      }
      if constexpr (ZF) {
        uint4 gp;
        gp.x = pack2_bf16(gv[0], gv[1]); gp.y = pack2_bf16(gv[2], gv[3]); gp.z = pack2_bf16(gv[4], gv[5]); gp.w = pack2_bf16(gv[6], gv[7]);
        if constexpr (FAST) *reinterpret_cast<uint4*>(zbase + off) = gp;
        else if (zbase && row < nrow_tile) *reinterpret_cast<uint4*>(zbase + off) = gp;
      }
      uint4 hp;
      hp.x = pack2_bf16(hv[0], hv[1]); hp.y = pack2_bf16(hv[2], hv[3]); hp.z = pack2_bf16(hv[4], hv[5]); hp.w = pack2_bf16(hv[6], hv[7]);
      *reinterpret_cast<uint4*>(sp) = hp;
      if constexpr (FAST) *reinterpret_cast<uint4*>(hbase + off) = hp;
      else if (hbase && row < nrow_tile) *reinterpret_cast<uint4*>(hbase + off) = hp;
    }
  };
  auto gemm2 = [&](int cc, bool fence = true) {  // acc2 += h(cc) W2[cc, :]: wave tile 32 x 128, K = 64
    const char* sH = smem + SH_OFF + (cc & (NBUF - 1)) * SH_BYTES;
    const char* sW2 = smem + W2_OFF + (cc & (NBUF - 1)) * 32768 + wc * 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      short8_t af[MR];
#pragma unroll
      for (int i = 0; i < MR; ++i) af[i] = frag_direct(sH, wr * WR + i * 16 + r, kk * 4 + g);
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {  // 4 B fragments in flight
        short8_t bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = frag_trans<128>(sW2, (jh * 4 + j) * 16, kk * 32 + g * 8, r);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc2[i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc2[i][jh * 4 + j], 0, 0, 0);
        if (fence) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if constexpr (DENSE) {
    // the normalised rows are in registers: the LN image (in the second buffer) is dead once every wave holds its fragments
    lds_only_barrier();
    if (NC > 1) issue_w1(1);
    for (int c = 0; c < NC; ++c) {
      // own pieces of W1(c): issued one chunk ago; only the previous chunk's MR stores were issued after them (c = 0 / 1: drain)
      if (counted && c > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MR) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_only_barrier();   // T: W1(c) visible; every wave is done with rowpass(c-1) (sH) and GEMM1(c-1) (the other buffer)
      if (c > 0 && c + 1 < NC) issue_w1(c + 1);
      gemm1(c);
      zlds(c);
      lds_only_barrier();   // A: z(c) complete
      if (counted) rowpass(c, std::true_type{}); else rowpass(c, std::false_type{});
    }
    return;
  } else if constexpr (WGS == 2) {
    for (int c = 0; c < NC; ++c) {
      // own pieces of W1(c): only the last chunk's z / h stores (2 MR instructions) were issued after them
      if (counted && c > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MR) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_only_barrier();   // T: W1(c) visible; every wave is done with GEMM2(c-1) (W2 buffer, sH) - at c = 0: the fragments are in registers
      issue_w2(c);
      FFN_TICK(1)
      gemm1(c);
      zlds(c);
      lds_only_barrier();   // A: z(c) complete; W1 buffer released
      issue_w1(c + 1);
      FFN_TICK(2)
      if (counted) rowpass(c, std::true_type{}); else rowpass(c, std::false_type{});
      FFN_TICK(3)
      // own pieces of W2(c): W1(c+1)'s 8 pieces and this chunk's 2 MR stores were issued after them
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PCS + 2 * MR) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_only_barrier();   // B: h(c) complete; W2(c) visible
      FFN_TICK(4)
      gemm2(c);
      FFN_TICK(5)
    }
  } else {
  // ---- start-up = "chunk -1": the normalised tile is in registers (buffers under sT are free after this barrier), GEMM1(0), zlds(0) ----
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // W1(0), W2(0) landed; every wave holds its fragments
  issue_w1(1);
  gemm1(0);
  lds_only_barrier();   // (B1 of chunk -1)
  issue_w1(2);
  zlds(0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PCS) : "memory");  // own pieces of W1(1) (only W1(2)'s PCS pieces were issued after them)
  lds_only_barrier();   // (B2 of chunk -1)
  issue_w2(1);

  auto b1_sync = [&](int c) {
    FFN_TICK(1)
    if (counted && c > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (2 * MR) + 2 * PCS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_only_barrier();   // B1: h(c) complete in sH[c%2]; W2(c) visible; W1 buffer (c+1)%2 released
    FFN_TICK(2)
    issue_w1(c + 3);
    FFN_TICK(3)
  };
  auto b2_sync = [&](int c) {
    FFN_TICK(4)
    if (counted && c > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MR + 2 * PCS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_only_barrier();   // B2: z(c+1) complete in sH[(c+1)%2]; W1(c+2) visible; W2 buffer c%2 released
    FFN_TICK(5)
    issue_w2(c + 2);
    FFN_TICK(6)
  };
  if (counted) {  // (full tile, z and h stored: branch-free row pass)
    for (int c = 0; c < NC; ++c) {
      const bool more = c + 1 < NC;
      if (more) gemm1(c + 1);
      rowpass(c, std::true_type{});
      b1_sync(c);
      gemm2(c);
      if (more) zlds(c + 1);
      b2_sync(c);
    }
  } else {
    for (int c = 0; c < NC; ++c) {
      const bool more = c + 1 < NC;
      if (more) gemm1(c + 1);
      rowpass(c, std::false_type{});
      b1_sync(c);
      gemm2(c);
      if (more) zlds(c + 1);
      b2_sync(c);
    }
  }
  }
  // ---- epilogue: y = x + res * dropout2(acc + b2), 16-row strips through a per-wave LDS strip (whole 256-B row pieces to HBM) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (clamped) DMA pieces have landed: the buffers become the strips
  __syncthreads();
  constexpr int SLD = 128 + 4;
  float* sc = reinterpret_cast<float*>(smem) + w * (16 * SLD);  // 16-row strip per wave (8.25 KiB)
  const int prow = lane >> 4, c8 = (lane & 15) * 8;
  const int col0 = wc * 128 + c8;
  float bv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bv[q] = p.b2[col0 + q];
  uint4 xres[MR][4];  // the residual rows of all strips, loaded up front (inside the strips each was a dependent HBM round trip)
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
      const int rowl = min(wr * WR + i * 16 + hp * 4 + prow, nrow_tile - 1);
      xres[i][hp] = *reinterpret_cast<const uint4*>(p.x + m0 * D + (rowl * D + col0));
    }
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    // the whole 16 x 128 fragment block of this wave in one go (every lane writes: 32 instructions, not 4 x 32 quarter-filled ones)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[(g * 4 + e) * SLD + j * 16 + r] = acc2[i][j][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
      float xv[8];
      {
        const float4 t0 = *reinterpret_cast<const float4*>(sc + (hp * 4 + prow) * SLD + c8);
        const float4 t1 = *reinterpret_cast<const float4*>(sc + (hp * 4 + prow) * SLD + c8 + 4);
        xv[0] = t0.x; xv[1] = t0.y; xv[2] = t0.z; xv[3] = t0.w; xv[4] = t1.x; xv[5] = t1.y; xv[6] = t1.z; xv[7] = t1.w;
      }
      const int rowl = wr * WR + i * 16 + hp * 4 + prow;
      if (rowl < nrow_tile) {
        const long idx0 = m0 * D + (rowl * D + col0);
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] += bv[q];
        if (p.drop_p > 0.f) {
#pragma unroll
          for (int q = 0; q < 8; q += 2) {  // idx0 is even: one hash per element pair
            const uint32_t hh = drop_hash((uint64_t)p.seed2, ((uint64_t)idx0 >> 1) + (q >> 1));
            xv[q] = (hh & 0xffffu) >= dthr ? xv[q] * dinv : 0.f;
            xv[q + 1] = (hh >> 16) >= dthr ? xv[q + 1] * dinv : 0.f;
          }
        }
        float xr[8];
        unpack8(xres[i][hp], xr);
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] = xr[q] + p.res * xv[q];
        st8(p.y + idx0, xv);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is read before the next block overwrites it
  }
#ifdef TFASR_FFN_TIMING
  FFN_TICK(7)  // epilogue
  if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) p.dbg[k] = ph[k];
#endif
}

// Two workgroups per CU with 64-row tiles (MR 2, WGS 2: weights single buffered).  Measured alternatives, removed in round 6 (the kernel is
// still a template over them): 96-row tiles with one workgroup per CU and double-buffered weights; 32-row tiles (round 5, per batch shape:
// a [14.8k, 256] batch is 231 64-row workgroups - one per CU - or 462 32-row ones = two per CU in one round; 50.9 vs ~48 us per launch in
// line, 21.76 vs 21.45 ms per step when forced everywhere: the second resident workgroup's overlap does not pay for the halved reuse of
// the weight images).
static int launch_ffn_fused_fwd(const FfnArgs& a, hipStream_t stream) {
  auto go = [&](auto kern, int MR, int NBUF) {
    const int smem = 2 * NBUF * 32768 + NBUF * 32 * MR * 128 + a.F * 4;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_done = true; }
    const long tiles = (a.rows + 32 * MR - 1) / (32 * MR);
    TFASR_KLAUNCH(kern, dim3((unsigned)tiles), dim3(256), smem, stream, a);
  };
  if (a.y == nullptr) go(ffn_fused_fwd_kernel<2, 2, false, true>, 2, 1);  // (Dense-only mode: tfasr_ln_dense_fwd)
  else if (a.zfactor && a.z) go(ffn_fused_fwd_kernel<2, 2, true>, 2, 1);
  else go(ffn_fused_fwd_kernel<2, 2, false>, 2, 1);
  return TFASR_STATUS_SUCCESS;
}
