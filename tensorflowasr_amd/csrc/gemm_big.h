// 256 x BN output tiles (BN = 256 | 320) for the products with thousands of tiles (joint vocabulary projection and its data gradient,
// conv2 over the haloed space-to-depth layout).  Included by gemm_fast.hip inside its anonymous namespace (shares the LDS images,
// the swizzles and the fragment readers).
//
// Why: every operand byte goes L2 -> LDS through the CU's vector-memory path (~64 B/clk/CU: one 1-KiB global_load_lds piece per
// ~16 clocks, whoever issues it).  A 128x128x64 slab needs 32 pieces for 512 clocks of MFMA - measured (tools/hwprobe/gemm_timing)
// 1630 clocks of DMA issue + 790 of fragment reads / MFMA per slab and workgroup, 31 % MFMA utilisation in the main loop with two
// workgroups per CU.  A 256x256x64 slab needs 64 pieces for 2176 clocks of MFMA: half the bytes per flop.
//   * ONE 8-wave workgroup per CU (two waves per SIMD, 256 registers each), wave tile 64 x BN/2: 128 accumulators pinned in
//     accumulation registers by inline-asm MFMAs (+ 32 in vector registers for BN 320; left to the compiler the accumulators were
//     renamed on every MFMA and shuffled between the two register classes);
//   * two 64 / 72 KiB LDS stages; the two waves of a SIMD take turns (group A = waves 0-3, group B = waves 4-7 one segment behind),
//     so every barrier-to-barrier segment pairs one wave's 32-40 MFMAs with its partner's fragment reads and DMA pieces (schedule in
//     the kernel body); the pieces of a slab are spread over three segments; each wave waits for its own pieces only;
//   * K tail: one more slab over k = [K - 64, K) with the already consumed columns' op(A) fragments zeroed on read;
//   * epilogue per 16-row fragment block: alpha / bias and - for the joint projection - the log-softmax statistics in the MFMA C
//     layout (DPP row reductions), then through a per-wave LDS strip to row-major bf16 stores; the next tile's first slab is
//     prefetched under it.
// A is always k-contiguous ([M, K] row major, M >= 256); B k-contiguous (TB, N a multiple of 8, >= BN) or k-strided (BN 256 only).
// Measured (tools/hwprobe/gemm_big_test, [400000, 1000, 640] / [400000, 640, 1000]): joint projection + statistics 1147 -> 1025 us,
// its data gradient 1113 -> 645 us (794 TFLOP/s) against the 128-row tiles; results bitwise equal (statistics: same to 5e-7).

// acc += a x b, accumulator pinned IN PLACE in an accumulation register (ACC) or a vector register.  Left to the compiler the accumulators
// of this software-pipelined loop were renamed on every MFMA (destination != addend) at the price of ~700 v_accvgpr moves per slab;
// the asm form also keeps the MFMA / DMA interleave exactly as written.  Hazards: an accumulator is re-read 64+ MFMAs after it was
// written and fragment registers are rewritten a barrier later, far beyond any required wait states.
template <bool ACC>
__device__ __forceinline__ void mfma_acc(float4_t& c, const short8_t& a, const short8_t& b) {
  if constexpr (ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// k-strided [64 k][256 n] image (512-B k-rows, 16-B chunk ^= key_t(k)): B fragment j of k half kk from ONE per-lane offset.  With
// k = kk*32 + g*8 + (r>>2) (+4) the key is ((r>>2)<<1) | ((g&1)<<3) for every fragment, so fragment j sits at tb ^ (j << 5):
//   tb = (g*8 + (r>>2))*512 + wn*256 + (((r&3)>>1) << 4) + ((key>>1) << 5) + (r&1)*8
__device__ __forceinline__ short8_t frag_trans_big(const char* sB, uint32_t tb, int j, int kk) {
  const char* q = sB + ((tb ^ (uint32_t)(j << 5)) + (uint32_t)(kk * 32 * 512));
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(q));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(q + 4 * 512));
  short8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return v;
}

// TR (round 5): TRANSPOSED accumulators - the two fragments of every MFMA swap their operand slots (A and B fragments have the same
// register layout: lane (r, g) = row / column r, k group g), so the product comes out as D^T: lane (r, g) of fragment (i, j) then holds
// FOUR CONSECUTIVE COLUMNS j*16 + g*4 + e of ONE row i*16 + r.  The main loop (LDS images, DMA, schedule) is untouched; the epilogue stores
// 8-byte pieces of a row straight from the accumulators - no LDS transposition strip, whose round trips bounded the row-oriented epilogue
// (25-30 k of a tile's ~67 k clocks) - and the log-softmax statistics of a row are in-lane work plus two cross-row-group shuffles.
template <bool TB, int BN_, int EPI, bool SEG, bool TR = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_big_kernel(const tfasr_gemm_args p, const int gx, const int gy, const int ntiles) {
  constexpr bool C_LSE = (EPI & E_LSE) != 0;
  static_assert(!TR || (BN_ == 256 && (EPI & ~(E_LSE | E_DACT)) == 0), "transposed accumulators: 256 columns; plain / log-softmax / tanh-out epilogues");
  static_assert(!TR || !((EPI & E_LSE) && (EPI & E_DACT)), "statistics and tanh-out never meet");
  // where the DMA pieces of the next slab are issued inside the fragment-read segments: 0 = in front of the reads, 2 = behind them (the
  // reads' latency then runs under the pieces' issue time: k-strided B, whose 16 transposing reads per k half make that segment the long
  // one, 878 -> 832 us on the joint projection; k-contiguous B measured 1 % slower), 1 = every piece between the MFMAs instead (probe
  // only: the partner's fragment reads then crawl under the DMA traffic; 10 % slower on the data gradient).  Cycle counters: whatever the
  // placement, group B's first fragment-read segment of a slab (the one during which the HBM-cold op(A) pieces issued a segment earlier
  // land in LDS) takes ~1600-1800 clocks against ~550-700 for the other three - the main loop's remaining slack.
  // k-strided B also issues its pieces through inline asm (see glds16 in gemm_fast.hip: no compiler-made vmcnt(0) in front of the fragment
  // reads: 780 -> 700 us on the joint projection); the k-contiguous-B kernels measured SLOWER that way (551 -> 627 us on the joint data
  // gradient, whose reads then overlap the landing of its own pieces), so they keep the builtin.
#ifdef TFASR_BIG_ASM_DMA
  constexpr bool ASM_DMA = TFASR_BIG_ASM_DMA != 0;
#else
  constexpr bool ASM_DMA = !TB;
#endif
#ifdef TFASR_BIG_SCHED
  constexpr int SCHED = TFASR_BIG_SCHED;
#else
  constexpr int SCHED = TB ? 0 : 2;
#endif
  constexpr int BMB = 256, WNC = BN_ / 2, NI = 4, NJ = WNC / 16;  // 8 waves = 4 (rows) x 2 (columns), wave tile 64 x BN/2
  constexpr int A_B = BMB * BK * 2, B_B = BN_ * BK * 2, STAGE = A_B + B_B;
  constexpr int NA = BMB / 64, NB = BN_ / 64, GI = NA + NB;  // DMA wave-instructions per slab per wave
  constexpr int CW = (WNC % 64 == 0) ? 64 : 32;     // columns per epilogue pass
  constexpr int LPRW = CW / 8, RPP = 64 / LPRW, NPASS = 16 / RPP, SLD = CW + 4, NCH = WNC / CW, JC = CW / 16;
  // BN 320: 2 x 72 KiB of stages leave no room for the strips - they live in stage 1, whose prefetch waits for the epilogue
  constexpr bool ALIAS = 2 * STAGE + 8 * RPP * SLD * 4 > 160 * 1024;
  static_assert(TB || BN_ == 256, "k-strided B images exist for 256 columns only");
  static_assert(!C_LSE || CW == 64, "log-softmax partials are per 64 columns");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int r = lane & 15, g = lane >> 4;
  const int bid = blockIdx.x, G = gridDim.x;
  const uint32_t tb0 = (uint32_t)((g * 8 + (r >> 2)) * 512 + wn * 256 + (((r & 3) >> 1) << 4) + ((((r >> 2) & 3) | ((g & 1) << 2)) << 5) + (r & 1) * 8);

  __shared__ long seg_tab[SEG ? 2 : 1][SEG ? 64 : 1];
  if constexpr (SEG) {
    const int nsl = p.K / BK;
    if ((int)threadIdx.x < nsl) {
      const int k = threadIdx.x * BK, sg = k / p.seg_k, within = k - sg * p.seg_k;
      seg_tab[0][threadIdx.x] = p.seg_a_off[sg] + (long)within;
      seg_tab[1][threadIdx.x] = p.seg_b_off ? p.seg_b_off[sg] + (TB ? (long)within : (long)within * p.ldb) : (TB ? (long)k : (long)k * p.ldb);
    }
    __syncthreads();
  }
  // DMA sources as a UNIFORM base (SGPRs) + 32-bit per-lane byte offsets (the global_load saddr form: no 64-bit vector adds, and
  // 2 + NB offset registers per tile instead of 2 x 18 pointer registers - the accumulators leave no room for those):
  //   A: rows are never clamped - the last row tile is shifted back to M - 256 (its first rows repeat the previous tile's and are not
  //      stored again), so instruction i = 2 ii + par reads offA[par] + ii * 16 rows (the swizzle key only depends on i's parity);
  //   B: likewise two offsets; a k-contiguous B shifts its last column tile back to N - BN, a k-strided one clamps the 16-B column
  //      chunks to the operand (the clamp only depends on i's parity as well).
  struct Tile { int m0, m0s, n0, n0s, ok; uint32_t offA[2]; uint32_t offB[2]; };
  // XCD-aware order as in gemm_fast: in every round XCD x (= bid & 7) owns one contiguous run of the n-fastest tile sequence
  auto tile_of = [&](int it) {
    Tile T;
    T.ok = 0;
    int t;
    if ((G & 7) == 0) {
      const int x = bid & 7, slot = bid >> 3, round0 = it * G, R = ntiles - round0;
      if (R <= 0) return T;
      if (R >= G) t = round0 + x * (G >> 3) + slot;
      else {
        const int q = R >> 3, rem = R & 7;
        if (slot >= q + (x < rem ? 1 : 0)) return T;
        t = round0 + (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + slot;
      }
    } else {
      if (it > 0) return T;
      t = bid;
    }
    const int tx = t % gx, ty = t / gx;
    T.ok = 1;
    T.m0 = ty * BMB;
    T.m0s = min(T.m0, p.M - BMB);
    T.n0 = tx * BN_;
    T.n0s = TB ? min(T.n0, p.N - BN_) : T.n0;  // k-contiguous B: the last column tile is shifted back like the last row tile
    int lt = lane;
    asm volatile("" : "+v"(lt));  // opaque: or every lane-dependent partial term below is kept in a register across the whole kernel
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int row = (w * NA + par) * 8 + (lt >> 3), pc = lt & 7;
      T.offA[par] = (uint32_t)(((long)row * p.lda + ((pc ^ key_d(row)) << 3)) * 2);
      if constexpr (TB) {
        const int rowb = (w * NB + par) * 8 + (lt >> 3);
        T.offB[par] = (uint32_t)(((long)rowb * p.ldb + ((pc ^ key_d(rowb)) << 3)) * 2);
      } else {
        // piece i covers k rows 2 (w NB + i) + (lane >> 5): the swizzle key and the column clamp only depend on i's parity
        constexpr int CPR = BN_ / 8, KPI = 64 / CPR;
        const int maxchunk = ((p.N + 7) >> 3) - 1;
        const int k = (w * NB + par) * KPI + lt / CPR, pcb = lt % CPR;
        const int c = min((T.n0 >> 3) + (pcb ^ key_t(k)), maxchunk);
        T.offB[par] = (uint32_t)(((long)k * p.ldb + ((long)c << 3)) * 2);
      }
    }
    return T;
  };
  const int nfull = p.K / BK;
  const bool tail = (p.K % BK) != 0;
  // K tail (V = 1000 = 15 * 64 + 40 in the joint's data gradient; K % 8 == 0 is a launch condition): one more slab over
  // k = [K - 64, K) - inside both operands, so nothing past a row's end is ever read - in which the op(A) fragments of the
  // 64 - K % 64 columns that slab nfull-1 already consumed are replaced by zeros as they are read (whole 8-column fragments).
  // (Staged with scalar loads and zero fill, as gemm_fast does, the tail cost 42k clocks per tile: 37 % of the main loop.)
  const int nsl = nfull + (tail ? 1 : 0);
  const int nzero = tail ? (BK - p.K % BK) / 8 : 0;
  const long stepA = BK, stepB = TB ? (long)BK : (long)BK * p.ldb;
  auto dA = [&](int slab) -> long {
    if constexpr (SEG) return seg_tab[0][slab];
    else return slab < nfull ? slab * stepA : (long)(p.K - BK);
  };
  auto dB = [&](int slab) -> long {
    if constexpr (SEG) return seg_tab[1][slab];
    else return slab < nfull ? slab * stepB : (TB ? (long)(p.K - BK) : (long)(p.K - BK) * p.ldb);
  };
  // one DMA instruction (i-th of its operand) of `slab` for tile T into the stage at `dst`
  auto dma_a = [&](const Tile& T, int i, long da, char* dst) {
    const char* base = (const char*)((const bf16_t*)p.A + ((long)T.m0s + (i >> 1) * 16) * p.lda + da);
    uint32_t o = T.offA[i & 1];
    asm volatile("" : "+v"(o));  // opaque: or the 64-bit sums base + offset are kept (and spilled) per piece across the slab loop
    glds16_s<ASM_DMA>(base, o, dst + __builtin_amdgcn_readfirstlane((w * NA + i) * 1024));
  };
  auto dma_b = [&](const Tile& T, int i, long db, char* dst) {
    // TB: piece i = rows 16 (i >> 1) further down than piece (i & 1); else: k rows 4 (i >> 1) further
    const char* base = (const char*)((const bf16_t*)p.B + db + (TB ? ((long)T.n0s + (i >> 1) * 16) * p.ldb : (long)(i >> 1) * 4 * p.ldb));
    uint32_t o = T.offB[i & 1];
    asm volatile("" : "+v"(o));
    glds16_s<ASM_DMA>(base, o, dst + A_B + __builtin_amdgcn_readfirstlane((w * NB + i) * 1024));
  };
  // this wave's pieces of a slab in two halves: 0 = its NA pieces of op(A), 1 = its NB pieces of op(B)
  auto issue_half = [&](const Tile& T, int slab, int stage, int half) {
    char* sA = smem + stage * STAGE;
    if (half == 0) {
      const long da = dA(slab);
#pragma unroll
      for (int i = 0; i < NA; ++i) dma_a(T, i, da, sA);
    } else {
      const long db = dB(slab);
#pragma unroll
      for (int i = 0; i < NB; ++i) dma_b(T, i, db, sA);
    }
  };
  auto issue = [&](const Tile& T, int slab, int stage) {
    issue_half(T, slab, stage, 0);
    issue_half(T, slab, stage, 1);
  };

  // ---- main loop: the two waves of a SIMD (w and w + 4) take turns.  Group A = waves 0-3, group B = waves 4-7, B one segment behind:
  //          P1(s)          P2(s)          P3(s)          P4(s)          P1(s+1)
  //   A  |  L0(s) + DMA  |   M0(s)      |   L1(s)      |   M1(s)      |  L0(s+1) ...         L = fragment reads of one k half
  //   B  |  M1(s-1)+DMA  |   L0(s)      |   M0(s)      |   L1(s)      |  M1(s) + DMA ...     M = its 32-40 MFMAs
  // so every segment pairs one wave's MFMAs with its partner's LDS reads (all eight waves reading at once left the matrix pipe idle
  // for the ~500 clocks 96 KiB of fragments take, twice per slab).  The same instruction stream for both groups, B enters it through
  // one extra barrier and A leaves it through one.  Stage s&1 is free once B has read L1(s), i.e. from P1(s+1): from there the pieces of
  // slab s+2 may be issued (schedule at the loop); each wave waits for its own pieces before it arrives at the barrier that opens the
  // slab (P1), so only one slab per wave is ever in flight and every vector-memory wait is vmcnt(0).
  const bool grpB = w >= 4;
  Tile cur = tile_of(0);
  if (!cur.ok) return;
  if (nfull > 0) issue(cur, 0, 0);
  if (nsl > 1 && grpB) { issue_half(cur, 1, 1, 0); if (SCHED == 1) issue_half(cur, 1, 1, 1); }

  for (int it = 0;; ++it) {
    float4_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#ifdef TFASR_GEMM_TIMING
    long long bph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long btp = __builtin_readcyclecounter();
#define BIG_TICK(k) { const long long t_ = __builtin_readcyclecounter(); bph[k] += t_ - btp; btp = t_; }
    const long long bt_start = btp;
#else
#define BIG_TICK(k)
#endif
    short8_t af[NI], bf[NJ];
    auto rd = [&](const char* sA, const char* sB, int kk, const int nz) {
      uint32_t tb = tb0;
      asm volatile("" : "+v"(tb));  // opaque: or the per-fragment addresses are hoisted out of the loop and spilled
      af[0] = frag_direct(sA, wm * 64 + r, kk * 4 + g);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if constexpr (TB) bf[j] = frag_direct(sB, wn * WNC + j * 16 + r, kk * 4 + g);
        else bf[j] = frag_trans_big(sB, tb, j, kk);
      }
#pragma unroll
      for (int i = 1; i < NI; ++i) af[i] = frag_direct(sA, wm * 64 + i * 16 + r, kk * 4 + g);
      if (nz) {  // uniform: the K-tail slab only
        const bool z = kk * 4 + g < nz;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int q = 0; q < 8; ++q) af[i][q] = z ? (short)0 : af[i][q];
      }
    };
    // this wave's MFMAs of one k half; (dma) its DMA instructions of `slab` into the stage at `dst` between them
    auto mma = [&](const bool dma, long da, char* dst, const bool dmab = false, long db = 0L) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (TR) mfma_acc<true>(acc[i][j], bf[j], af[i]);  // D^T: rows = this fragment's 16 columns, columns = its 16 rows
          else if (j < 8) mfma_acc<true>(acc[i][j], af[i], bf[j]); else mfma_acc<false>(acc[i][j], af[i], bf[j]);  // BN 320: 128 + 32
        }
        if (dma) dma_a(cur, i, da, dst);  // uniform: a scalar branch around one instruction (NA == NI pieces, one per fragment row)
        if (dmab) {
          dma_b(cur, i, db, dst);
          if (NB > NI && i == NI - 1) dma_b(cur, NB - 1, db, dst);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // every wave's pieces of slab 0 have landed BEFORE it arrives at the barrier that lets group A read them (first tile; later tiles
    // are drained after the epilogue anyway)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible (see gemm_fast.hip: tile boundary)
    if (grpB) __builtin_amdgcn_s_barrier();
    // The CU's vector-memory path takes ~16 clocks per 1-KiB piece whoever issues it: all 64-72 pieces of a slab in ONE segment made
    // that segment 1000+ clocks longer for both groups.  So they are spread over three: slab s+1's pieces are issued
    //   A: op(A) half in L0(s) [after P1(s)], op(B) half in L1(s) [after P3(s)]      B: op(A) half in M1(s-1) [after P1(s)], op(B) half
    //   in L0(s) [after P2(s)]
    // (stage (s+1)&1 is free from P1(s)); A waits for its pieces at the end of M1(s), B at the end of L1(s), both before P1(s+1).
    for (int s = 0; s < nsl; ++s) {
      char* sA = smem + (s & 1) * STAGE;
      const bool nx = s + 1 < nsl;
      const int nz = s < nfull ? 0 : nzero;
      __builtin_amdgcn_s_barrier();  // A: P1(s)   B: P2(s)
      BIG_TICK(0)
      if (SCHED == 0 && nx) issue_half(cur, s + 1, (s + 1) & 1, grpB ? 1 : 0);
      rd(sA, sA + A_B, 0, nz);
      if (SCHED == 2 && nx) issue_half(cur, s + 1, (s + 1) & 1, grpB ? 1 : 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      BIG_TICK(1)
      __builtin_amdgcn_s_barrier();  // A: P2   B: P3
      BIG_TICK(2)
      if (SCHED == 1) {
        const bool dma = !grpB && nx;  // A: every piece of slab s+1 between the MFMAs of M0(s)
        mma(dma, dma ? dA(s + 1) : 0L, smem + ((s + 1) & 1) * STAGE, dma, dma ? dB(s + 1) : 0L);
      } else {
        mma(false, 0L, sA);
      }
      BIG_TICK(3)
      __builtin_amdgcn_s_barrier();  // A: P3   B: P4
      BIG_TICK(4)
      if (SCHED == 0 && nx && !grpB) issue_half(cur, s + 1, (s + 1) & 1, 1);
      rd(sA, sA + A_B, 1, nz);
      if (SCHED == 2 && nx && !grpB) issue_half(cur, s + 1, (s + 1) & 1, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // B's pieces of slab s+1 before P1(s+1)
      __builtin_amdgcn_sched_barrier(0);
      BIG_TICK(5)
      __builtin_amdgcn_s_barrier();  // A: P4   B: P1(s+1): stage s&1 is free
      BIG_TICK(6)
      {
        const bool dma = grpB && s + 2 < nsl;
        if (SCHED == 1) mma(dma, dma ? dA(s + 2) : 0L, sA, dma, dma ? dB(s + 2) : 0L);  // B: every piece of slab s+2 in M1(s)
        else mma(dma, dma ? dA(s + 2) : 0L, sA);
      }
      if (!grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // A's pieces of slab s+1 before P1(s+1)
      BIG_TICK(7)
    }
    if (!grpB) __builtin_amdgcn_s_barrier();
    __syncthreads();  // both stages idle
#ifdef TFASR_GEMM_TIMING
    const long long bt_main = __builtin_readcyclecounter();
#endif
    const Tile nxt = tile_of(it + 1);
    if (nxt.ok) {
      if (nfull > 0) issue(nxt, 0, 0);
      if (nsl > 1 && grpB && !ALIAS) { issue_half(nxt, 1, 1, 0); if (SCHED == 1) issue_half(nxt, 1, 1, 1); }
    }

#ifdef TFASR_GEMM_TIMING
    long long eph[4] = {0, 0, 0, 0};
    long long etp = __builtin_readcyclecounter();
#define EPI_TICK(k) { const long long t_ = __builtin_readcyclecounter(); eph[k] += t_ - etp; etp = t_; }
#else
#define EPI_TICK(k)
#endif
    // ---- epilogue, transposed accumulators ----
    if constexpr (TR) {
      const int m0 = cur.m0s, mlo = cur.m0, n0 = cur.n0;
      bf16_t* Dt = (bf16_t*)p.D;
      int le = lane;
      asm volatile("" : "+v"(le));  // opaque per tile (see tile_of)
      const int r = le & 15, g = le >> 4;
      const int cb = cur.n0s + wn * WNC;          // first column of this wave
      const int c0l = cb + g * 4;                 // this lane's first column in fragment 0 (fragment j: + 16 j)
      const bool vec_ok = ((p.ldd & 3) == 0) && ((((uintptr_t)p.D) & 7) == 0) && ((p.N & 3) == 0);
      const bool vec16_ok = ((p.ldd & 7) == 0) && ((((uintptr_t)p.D) & 15) == 0) && ((p.N & 7) == 0) && (NJ % 2 == 0);  // (uniform)
      // bias of the tile's 256 columns in LDS (the strip region of the row-oriented epilogue is free here; a column past N carries -inf
      // into the statistics): 32 registers per lane less than holding this lane's 4 x 8 values across the four fragment blocks
      float* sBias = reinterpret_cast<float*>(smem + 2 * STAGE);
      if constexpr (!ALIAS) {
        if (threadIdx.x < BN_) {
          const int col = cur.n0s + (int)threadIdx.x;
          sBias[threadIdx.x] = col < p.N ? (p.bias ? p.bias[col] : 0.f) : (C_LSE ? -INFINITY : 0.f);
        }
        __syncthreads();
      }
      const float* sb = sBias + wn * WNC + g * 4;
      int lab[NI];
      if constexpr (C_LSE) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int rw = m0 + wm * 64 + i * 16 + r;
          lab[i] = rw < p.M ? p.row_label[rw] : -1;
        }
      }
      EPI_TICK(0)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const int row = m0 + wm * 64 + i * 16 + r;
        const bool rok = row >= mlo && row < p.M;
        if constexpr (!C_LSE) {
          // no row statistics: one fragment PAIR at a time (8 values live, not the wave's 32-40 - the 320-column kernel has no register to spare)
          bf16_t* drow = Dt + (long)row * p.ldd;
          const bool odd = (g & 1) != 0;
          // dact = TANH_OUT: times 1 - h^2, h = the activation's output (row-major like D): this lane's 4 columns per fragment, requested ZP
          // fragments ahead of their use (all of them at 256 columns; two at 320, whose 160 accumulators leave no room for ten pairs)
          constexpr int ZP = (EPI & E_DACT) ? (BN_ == 256 ? NJ : 4) : 1;
          uint2 zr[ZP];
          const bf16_t* hz = (EPI & E_DACT) ? (const bf16_t*)p.dact_z + (long)row * p.ldd : nullptr;
          auto zload = [&](int j) {
            const int col = c0l + j * 16;
            uint2 v = make_uint2(0u, 0u);
            if (rok && col + 4 <= p.N) v = *reinterpret_cast<const uint2*>(hz + col);  // (ldd, N multiples of 8: launch condition of TR)
            return v;
          };
          if constexpr ((EPI & E_DACT) != 0) {
#pragma unroll
            for (int j = 0; j < ZP && j < NJ; ++j) zr[j] = zload(j);
          }
#pragma unroll
          for (int jp = 0; jp < NJ; jp += 2) {
            float xp[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int j = jp + q;
              float bj[4];
              if constexpr (!ALIAS) {
                const float4 b4 = *reinterpret_cast<const float4*>(sb + j * 16);
                bj[0] = b4.x; bj[1] = b4.y; bj[2] = b4.z; bj[3] = b4.w;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bj[e] = (p.bias && c0l + j * 16 + e < p.N) ? p.bias[c0l + j * 16 + e] : 0.f;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v;
                if (j < 8) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[i][j][e]));
                else v = acc[i][j][e];
                xp[q][e] = p.alpha * v + bj[e];
              }
              if constexpr ((EPI & E_DACT) != 0) {
                const uint2 zz = zr[j % ZP];
                if (j + ZP < NJ) zr[j % ZP] = zload(j + ZP);  // (compile-time slot: the loops are unrolled)
                const float z0 = __uint_as_float(zz.x << 16), z1 = __uint_as_float(zz.x & 0xffff0000u);
                const float z2 = __uint_as_float(zz.y << 16), z3 = __uint_as_float(zz.y & 0xffff0000u);
                xp[q][0] *= 1.f - z0 * z0; xp[q][1] *= 1.f - z1 * z1; xp[q][2] *= 1.f - z2 * z2; xp[q][3] *= 1.f - z3 * z3;
              }
            }
            uint2 a, b;
            a.x = pack2_bf16(xp[0][0], xp[0][1]); a.y = pack2_bf16(xp[0][2], xp[0][3]);
            b.x = pack2_bf16(xp[1][0], xp[1][1]); b.y = pack2_bf16(xp[1][2], xp[1][3]);
            if (vec16_ok) {
              const uint2 send = odd ? a : b;
              uint2 recv;
              recv.x = xor16_get(send.x, odd);
              recv.y = xor16_get(send.y, odd);
              const uint4 out = odd ? make_uint4(recv.x, recv.y, b.x, b.y) : make_uint4(a.x, a.y, recv.x, recv.y);
              const int col = cb + (odd ? (jp + 1) * 16 + (g - 1) * 4 : jp * 16 + g * 4);
              if (rok && col >= n0 && col < p.N) *reinterpret_cast<uint4*>(drow + col) = out;
            } else if (rok) {
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const int col = c0l + (jp + q) * 16;
                const uint2 v = q ? b : a;
                if (col >= n0) {
                  if (vec_ok) { if (col < p.N) *reinterpret_cast<uint2*>(drow + col) = v; }
                  else {
                    const bf16_t h4[4] = {(bf16_t)(v.x & 0xffffu), (bf16_t)(v.x >> 16), (bf16_t)(v.y & 0xffffu), (bf16_t)(v.y >> 16)};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                      if (col + e < p.N) drow[col + e] = h4[e];
                  }
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          continue;
        }
        float x[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float bj[4];
          if constexpr (!ALIAS) {
            const float4 b4 = *reinterpret_cast<const float4*>(sb + j * 16);  // (16-byte aligned: wave offset + 4 g + 16 j floats)
            bj[0] = b4.x; bj[1] = b4.y; bj[2] = b4.z; bj[3] = b4.w;
          } else {  // BN 320: the stages fill the LDS; its products (the joint's data gradient) carry no bias
#pragma unroll
            for (int e = 0; e < 4; ++e) bj[e] = (p.bias && c0l + j * 16 + e < p.N) ? p.bias[c0l + j * 16 + e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v;
            if (j < 8) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[i][j][e]));
            else v = acc[i][j][e];
            x[j][e] = p.alpha * v + bj[e];
          }
        }
        if constexpr (C_LSE) {
          const float L2E = 1.4426950408889634f;
          // one reference maximum of the row over this wave's 128 columns: in-lane over 32 values, then over the four lanes (r, g = 0..3)
          float m = x[0][0];
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, x[j][e]);
          m = xor32_max(xor16_max(m));
          const float mb = (m == -INFINITY) ? 0.f : m * L2E;
          float ssum[2] = {0.f, 0.f};
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) ssum[j >> 2] += __builtin_amdgcn_exp2f(x[j][e] * L2E - mb);
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            ssum[sl] = xor32_sum(xor16_sum(ssum[sl]));
          }
          if (rok) {
            // lane g = 0 / 1 stores the (max, sum) pair of slice 0 / 1 of its row
            const int slice = (cb >> 6) + g;
            if (g < 2 && slice < p.lse_parts) reinterpret_cast<float2*>(p.lse_part)[(long)row * p.lse_parts + slice] = make_float2(m, g == 0 ? ssum[0] : ssum[1]);
            if (cb == 0 && g == 0) p.pick[2L * row] = x[0][0];  // the blank logit (column 0)
            const int d = lab[i] - c0l;  // the label's column is this lane's element (d >> 4, d & 3) when d & 15 < 4
            if (d >= 0 && d < WNC && (d & 15) < 4) {
              float ve[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                ve[e] = x[0][e];
#pragma unroll
                for (int j = 1; j < NJ; ++j) ve[e] = ((d >> 4) == j) ? x[j][e] : ve[e];
              }
              const int de = d & 3;
              p.pick[2L * row + 1] = de == 0 ? ve[0] : (de == 1 ? ve[1] : (de == 2 ? ve[2] : ve[3]));
            }
          }
        }
        EPI_TICK(1)
        if constexpr (C_LSE) { if (!Dt) continue; }  // statistics-only projection
        if (vec16_ok) {
          // 16-byte stores: the lanes (r, g) and (r, g ^ 1) trade one packed fragment piece (two dwords through the cross-lane network), so
          // that an even g ends up with EIGHT consecutive columns of fragment jp and an odd g with eight of fragment jp + 1: four store
          // instructions per 16-row block, each writing 64 contiguous bytes of 16 rows (8-byte pieces were 32 instructions per block and
          // the vector-memory path's issue time - one instruction per 16 cache lines - was what the first version of this epilogue ran at)
          bf16_t* drow = Dt + (long)row * p.ldd;
          const bool odd = (g & 1) != 0;
#pragma unroll
          for (int jp = 0; jp < NJ; jp += 2) {
            uint2 a, b;
            a.x = pack2_bf16(x[jp][0], x[jp][1]); a.y = pack2_bf16(x[jp][2], x[jp][3]);
            b.x = pack2_bf16(x[jp + 1][0], x[jp + 1][1]); b.y = pack2_bf16(x[jp + 1][2], x[jp + 1][3]);
            const uint2 send = odd ? a : b;
            uint2 recv;
            recv.x = xor16_get(send.x, odd);
            recv.y = xor16_get(send.y, odd);
            const uint4 out = odd ? make_uint4(recv.x, recv.y, b.x, b.y) : make_uint4(a.x, a.y, recv.x, recv.y);
            const int col = cb + (odd ? (jp + 1) * 16 + (g - 1) * 4 : jp * 16 + g * 4);
            if (rok && col >= n0 && col < p.N) *reinterpret_cast<uint4*>(drow + col) = out;
          }
        } else if (rok) {
          bf16_t* drow = Dt + (long)row * p.ldd;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int col = c0l + j * 16;
            if (col >= n0) {  // (columns below n0 belong to the previous tile of a shifted last column tile: never with k-strided B)
              if (vec_ok) {
                if (col < p.N) {
                  uint2 v;
                  v.x = pack2_bf16(x[j][0], x[j][1]);
                  v.y = pack2_bf16(x[j][2], x[j][3]);
                  *reinterpret_cast<uint2*>(drow + col) = v;
                }
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (col + e < p.N) drow[col + e] = f32_to_bf16(x[j][e]);
              }
            }
          }
        }
        EPI_TICK(2)
      }
    } else
    // ---- epilogue ----
    // Per 16-row fragment block: x = alpha * acc + bias in the MFMA C layout (lane (r, g): rows g*4+e, column j*16+r); the log-softmax
    // statistics are taken THERE - a row's 16 columns of one fragment sit in one 16-lane DPP row, so max / sum are register + DPP
    // work (the row-major variant of gemm_fast pays six dependent LDS shuffles per 8-row pass: 2100 clocks per pass measured with
    // one or two waves per SIMD to hide them) - then the block goes through the per-wave LDS strip to row-major bf16 stores.
    {
      const int m0 = cur.m0s, mlo = cur.m0, n0 = cur.n0;  // rows below mlo belong to the previous tile (shifted last tile)
      bf16_t* Dt = (bf16_t*)p.D;
      float* sc = reinterpret_cast<float*>(smem + (ALIAS ? STAGE : 2 * STAGE)) + w * (RPP * SLD);
      int le = lane;
      asm volatile("" : "+v"(le));  // opaque per tile (see tile_of)
      const int r = le & 15, g = le >> 4;
      const int prow = le / LPRW, c8 = (le % LPRW) * 8;
      const bool vec_ok = ((p.ldd & 7) == 0) && ((((uintptr_t)p.D) & 15) == 0);
      const int cb = cur.n0s + wn * WNC;  // first column of this wave (columns below n0 belong to the previous tile: shifted last tile)
      int lab[C_LSE ? NI : 1][C_LSE ? 4 : 1];  // labels of this lane's rows, loaded up front (a load inside the block stalls it for a round trip)
      if constexpr (C_LSE) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int rw = m0 + wm * 64 + i * 16 + g * 4 + e;
            lab[i][e] = rw < p.M ? p.row_label[rw] : -1;  // rw >= 0 always (M >= 256)
          }
      }
      // strip write base of this lane: the writers of pass h are the lanes with (g * 4) / RPP == h, at strip rows (g * 4) % RPP + e
      float* sw = sc + ((g * 4) % RPP) * SLD + r;
      float bc[NJ];
      bool cv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = cb + j * 16 + r;
        cv[j] = col < p.N;
        // statistics: a column past N carries -inf from here on (exp2 -> 0, never the maximum), so the 128 elements of a fragment block
        // need no per-element column test (they were a third of this epilogue's vector instructions)
        bc[j] = cv[j] ? (p.bias ? p.bias[col] : 0.f) : (C_LSE ? -INFINITY : 0.f);
      }
      EPI_TICK(0)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        float x[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          // every element is read out of its accumulation register HERE, by an explicit v_accvgpr_read: left to the compiler the copies of
          // every accumulator into vector registers are placed right behind the main loop and everything else is spilled around them; and
          // re-defining the fragment as an accumulation-register VALUE (asm "+a" on a copy, the first fix) made it a register window -
          // 28 v_accvgpr_mov + 4 v_accvgpr_write per fragment block shifted the remaining accumulators down to it (9 % of the
          // epilogue's vector instructions)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v;
            if (j < 8) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[i][j][e]));
            else v = acc[i][j][e];
            x[j][e] = p.alpha * v + bc[j];
          }
        }
        if constexpr (C_LSE) {
          const float L2E = 1.4426950408889634f;
          float pm[4][2], ps[4][2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // ONE reference maximum per row for both 64-column slices of this wave (any finite reference >= the slice's own maximum is a
            // valid (max, sum) pair for the merge): half the DPP row reductions
            float m = x[0][e];
#pragma unroll
            for (int j = 1; j < NJ; ++j) m = fmaxf(m, x[j][e]);
            m = row16_max(m);
            const float mb = (m == -INFINITY) ? 0.f : m * L2E;  // every column of the wave past N: sums of exp2(-inf) = 0, no inf - inf
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
              float ssum = 0.f;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) ssum += __builtin_amdgcn_exp2f(x[sl * 4 + jj][e] * L2E - mb);
              pm[e][sl] = m;
              ps[e][sl] = row16_sum(ssum);
            }
            const int row = m0 + wm * 64 + i * 16 + g * 4 + e;
            if (row >= mlo && row < p.M) {
              if (cb == 0 && r == 0) p.pick[2L * row] = x[0][e];
              const int d = lab[i][e] - cb - r;  // the label's column is this lane's in fragment d / 16
              if (d >= 0 && d < WNC && (d & 15) == 0) {
                float v = x[0][e];
#pragma unroll
                for (int j = 1; j < NJ; ++j) v = (d == j * 16) ? x[j][e] : v;
                p.pick[2L * row + 1] = v;
              }
            }
          }
          // every lane of a 16-lane row holds the 8 (row, slice) results of its 4 rows: lane r < 8 stores (row r & 3, slice r >> 2), so the
          // block's statistics leave in ONE store instruction (a store per pair cost more issue time than the arithmetic)
          {
            const int es = r & 3, ss = (r >> 2) & 1;
            float om = pm[0][0], os = ps[0][0];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int sl = 0; sl < 2; ++sl) {
                const bool hit = (es == e) && (ss == sl);
                om = hit ? pm[e][sl] : om;
                os = hit ? ps[e][sl] : os;
              }
            const int row = m0 + wm * 64 + i * 16 + g * 4 + es;
            const int slice = (cb >> 6) + ss;
            if (r < 8 && row >= mlo && row < p.M && slice < p.lse_parts)
              reinterpret_cast<float2*>(p.lse_part)[(long)row * p.lse_parts + slice] = make_float2(om, os);
          }
        }
        EPI_TICK(1)
        if constexpr (C_LSE) { if (!Dt) continue; }  // statistics-only projection (D == NULL): no logits leave the chip, no transposition pass
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          const int col0 = cb + ch * CW + c8;
          const bool full = vec_ok && (col0 + 8 <= p.N);
#pragma unroll
          for (int h = 0; h < NPASS; ++h) {
            __builtin_amdgcn_sched_barrier(0);
            if ((g * 4) / RPP == h) {
#pragma unroll
              for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) sw[e * SLD + jj * 16] = x[ch * JC + jj][e];  // compile-time offsets from one per-lane base
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float y[8];
            {
              const float4 t0 = *reinterpret_cast<const float4*>(sc + prow * SLD + c8);
              const float4 t1 = *reinterpret_cast<const float4*>(sc + prow * SLD + c8 + 4);
              y[0] = t0.x; y[1] = t0.y; y[2] = t0.z; y[3] = t0.w; y[4] = t1.x; y[5] = t1.y; y[6] = t1.z; y[7] = t1.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int row = m0 + wm * 64 + i * 16 + h * RPP + prow;
            if (row >= mlo && row < p.M && col0 < p.N && col0 >= n0) {
              const long idx0 = (long)row * p.ldd + col0;
              if constexpr ((EPI & E_DACT) != 0) {  // dact = TANH_OUT: times 1 - h^2, h = the activation's output (row-major like D)
                const bf16_t* hz = (const bf16_t*)p.dact_z;
                float z[8];
                if (full) ld8(hz + idx0, z);
                else
_Pragma("unroll")
                  for (int q = 0; q < 8; ++q) z[q] = (col0 + q < p.N) ? bf16_to_f32(hz[idx0 + q]) : 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] *= 1.f - z[q] * z[q];
              }
              if constexpr ((EPI & E_RGRAD) != 0) {
                // y = the re-computed logits of 8 columns of row `row`: -> d loss / d logit (impl/rnnt.py:233-275; rnnt_grad_kernel's arithmetic)
                const float4 cf = *reinterpret_cast<const float4*>(p.rgrad_coef + 4L * row);
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = __builtin_amdgcn_exp2f(y[q] * 1.4426950408889634f - cf.x) * cf.y;
                if (col0 == 0) y[0] += cf.z;
                const int lb = p.row_label[row] - col0;
                if (lb >= 0 && lb < 8) {
#pragma unroll
                  for (int q = 0; q < 8; ++q) y[q] += (q == lb) ? cf.w : 0.f;
                }
              }
              if (full) st8(Dt + idx0, y);
              else
_Pragma("unroll")
                for (int q = 0; q < 8; ++q) if (col0 + q < p.N) Dt[idx0 + q] = f32_to_bf16(y[q]);
            }
          }
        }
        EPI_TICK(2)
      }
    }
#ifdef TFASR_GEMM_TIMING
    if ((threadIdx.x == 0 || threadIdx.x == 256) && it < 2) {
      const long long t_end = __builtin_readcyclecounter();
      long long* o = g_gemm_timing + 12L * ((bid * 2 + it) * 2 + (threadIdx.x >> 8));
      for (int k = 0; k < 8; ++k) o[k] = bph[k];
      o[8] = bt_main - bt_start; o[9] = t_end - bt_main; o[10] = t_end - bt_start;
      long long* o2 = g_gemm_timing + 16384 + 4L * ((bid * 2 + it) * 2 + (threadIdx.x >> 8));
      for (int k = 0; k < 3; ++k) o2[k] = eph[k];
    }
#endif
    if (!nxt.ok) break;
    if constexpr (ALIAS) {
      __syncthreads();  // every wave is done with its strip
      if (nsl > 1 && grpB) { issue_half(nxt, 1, 1, 0); if (SCHED == 1) issue_half(nxt, 1, 1, 1); }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible (see gemm_fast.hip: tile boundary)
    cur = nxt;
  }
}

int g_gemm_big_mode = -1;  // probes / tests: 0 = never, 1 = default rule; -1 = TFASR_GEMM_BIG from the environment

// eligibility + launch; UNSUPPORTED -> the caller continues with the 128-row tiles
template <bool TB>
int launch_big(const tfasr_gemm_args& a, bool generic, int need, bool tanh_out, hipStream_t stream) {
  static const bool env_off = false;
  const bool off = g_gemm_big_mode < 0 ? env_off : g_gemm_big_mode == 0;
  if (off || generic || need != 0 || a.accumulate || a.split_k > 1 || a.nb1 * a.nb2 != 1 || a.colsum || a.out_f32 || a.K < 2 * BK || (a.K & 7) || a.M < 256 || (TB && (a.N & 7))) return TFASR_STATUS_UNSUPPORTED;
  const bool seg = a.seg_a_off != nullptr;
  if (seg && !(a.seg_k > 0 && (a.seg_k % BK) == 0 && (a.K % a.seg_k) == 0 && a.K / BK <= 64)) return TFASR_STATUS_UNSUPPORTED;
  // BN: 320 for N = 320 / 640 / ... (k-contiguous B only: the joint's data gradient, joint_dim 320 or 640), else 256 when the last
  // column tile is at least 3/4 full
  int bn = 0;
  if (TB && !seg && !a.lse_part && (a.N % 320) == 0) bn = 320;
  else if (a.N >= (TB ? 256 : 192) && ((a.N % 256) == 0 || (a.N % 256) >= 192)) bn = 256;
  if (!bn) return TFASR_STATUS_UNSUPPORTED;
  const int gx = (a.N + bn - 1) / bn, gy = (a.M + 255) / 256;
  const long ntiles = (long)gx * gy;
  static const long min_tiles = 2L * num_cus();
  if (ntiles < min_tiles || ntiles > 0x7fffffffL) return TFASR_STATUS_UNSUPPORTED;
  if (tanh_out && (!TB || seg || a.lse_part)) return TFASR_STATUS_UNSUPPORTED;
  if (a.lse_part && !(a.row_label && a.pick && bn == 256 && a.lse_parts == ((a.N + 127) / 128) * 2)) return TFASR_STATUS_UNSUPPORTED;
  if (a.rgrad_coef && (TB || seg || tanh_out || bn != 256 || !a.row_label)) return TFASR_STATUS_UNSUPPORTED;
  if (!a.D && !a.lse_part) return TFASR_STATUS_UNSUPPORTED;
  const int ncu = num_cus();
  const int G = ntiles >= ncu ? (ncu & ~7) : (int)ntiles;
  auto go = [&](auto kern, int smem) {
    static bool attr_done = false;  // per instantiation (the lambda's static lives in the template instance)
    if (!attr_done) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_done = true; }
    TFASR_KLAUNCH(kern, dim3(G), dim3(512), smem, stream, a, gx, gy, (int)ntiles);
  };
  constexpr int S256 = 2 * (256 * BK * 2 + 256 * BK * 2) + 8 * 8 * 68 * 4;
  constexpr int S320 = 2 * (256 * BK * 2 + 320 * BK * 2);
  // transposed accumulators (the kernel's TR parameter; 16-byte stores straight from the accumulators): every variant whose output rows
  // allow them (ldd % 8 == 0, N % 8 == 0, 16-byte aligned D); TFASR_BIG_TR=0 restores the row-oriented epilogues (A/B)
  static const bool tr_on = !(false);
  const bool tr = tr_on && (a.ldd & 7) == 0 && (a.N & 7) == 0 && (((uintptr_t)a.D) & 15) == 0;
  if (tanh_out) {
    if constexpr (TB) {
      // (320 columns: the 160 accumulators leave no register for the transposed epilogue - seen at compile time, also with the tanh-out
      // operand requested only two fragments ahead: two spill reloads land INSIDE the slab loop, each followed by a vmcnt(0) that drains
      // the DMA prefetch - so the 320-column kernels keep the row-oriented epilogue; tests/test_isa_guard.py watches for scratch traffic)
      if (bn == 320) go(gemm_big_kernel<true, 320, E_DACT, false>, S320);
      else { if (tr) go(gemm_big_kernel<true, 256, E_DACT, false, true>, S256); else go(gemm_big_kernel<true, 256, E_DACT, false>, S256); }
    } else return TFASR_STATUS_UNSUPPORTED;
  } else if (bn == 320) {
    if constexpr (TB) go(gemm_big_kernel<true, 320, 0, false>, S320);
    else return TFASR_STATUS_UNSUPPORTED;
  } else if (a.lse_part) {
    if constexpr (!TB) { if (tr || (tr_on && !a.D)) go(gemm_big_kernel<false, 256, E_LSE, false, true>, S256); else go(gemm_big_kernel<false, 256, E_LSE, false>, S256); }
    else return TFASR_STATUS_UNSUPPORTED;
  } else if (a.rgrad_coef) {
    if constexpr (!TB) go(gemm_big_kernel<false, 256, E_RGRAD, false>, S256);
    else return TFASR_STATUS_UNSUPPORTED;
  } else if (seg) {
    if (tr) go(gemm_big_kernel<TB, 256, 0, true, true>, S256); else go(gemm_big_kernel<TB, 256, 0, true>, S256);
  } else {
    if (tr) go(gemm_big_kernel<TB, 256, 0, false, true>, S256); else go(gemm_big_kernel<TB, 256, 0, false>, S256);
  }
  TFASR_CHECK_LAUNCH();
  return TFASR_STATUS_SUCCESS;
}
