// Dense data gradient + LayerNorm backward in ONE launch for d = 256 (included by gemm_fast.hip, inside its namespace).
//
//   dln = alpha * dy @ W^T                          dy [rows, K] (the gradient of the Dense layer's output), W [d, K] row-major
//   dx  = add + rstd * (dln*g - mean_c(dln*g) - xhat * mean_c(dln*g*xhat)),   xhat = (x - mean) * rstd
//   part[tile] = (sum_rows dln*xhat, sum_rows dln)                             (gamma / beta gradients, folded later)
//   dx_dropped = dropout(dx)                                                   (optional second output, as tfasr_layernorm_bwd_drop)
//
// Sites: the three "LayerNorm -> Dense" heads of a Conformer block whose backward was a K -> 256 product followed by ln_bwd_vec_kernel
// (FFModule d -> 4d, the fused q/k/v projection, the conv module's first pointwise conv: encoders/conformer.py:66-109,
// multihead_attention.py:628-637, convolution.py:159-228).  A workgroup owns 16*MT full rows: the product's output tile IS a set of
// complete LayerNorm rows, so the row reductions need nothing from another workgroup, dln never exists in HBM (one 10 MB write and one
// 10 MB read per site gone) and one launch + one kernel boundary per site leave the chain.
//
// Tile: 16*MT rows x 256 columns, EIGHT waves (two per SIMD: one wave's DMA issue and fragment reads run under the other's MFMAs), wave
// (wm, wn) = row half wm x columns [64 wn, 64 wn + 64): MT/2 x 4 accumulator fragments.  K is walked in 64-deep slabs, both operands
// k-contiguous ("direct" images of gemm_fast.hip), THREE LDS stages (a slab has two iterations to land), LDS-DMA by inline asm with
// hand-counted vmcnt, and the fragments of slab s+1 are read into a second register set between the MFMAs of slab s (one barrier per slab).
// The accumulators are TRANSPOSED (the operand slots of the MFMA swapped, as gemm_big's TR epilogue): lane (r, g) of fragment (i, j)
// holds row i*16 + r, four CONSECUTIVE columns j*16 + g*4 + e.  So the LayerNorm backward runs straight from the accumulators: after one
// lane-pair exchange x / add / dx are 16-byte pieces per lane (eight consecutive columns), a row's sums are in-lane over 16 values +
// two cross-group lane swaps + one LDS exchange between the four column waves, the gamma / beta column sums are DPP sums over the 16 rows
// of a fragment.  No transposition through LDS (a first version with 16-row LDS strips spent 24 k of its 58 k clocks per tile there).
#pragma once

struct DenseLnArgs {
  const bf16_t* dy; long ldy; const bf16_t* W; long ldw; int K;
  const bf16_t* x; const float* gamma; const float* mean; const float* rstd; const bf16_t* add;
  bf16_t* dx; bf16_t* dxd; float drop_p; uint64_t drop_seed; float* part;
  long rows; float alpha; int ntiles;
};

#ifdef TFASR_DLN_TIMING  // probe build (tools/hwprobe/dense_ln_test.hip): shader clocks of wave 0 per phase and workgroup
__device__ long long g_dln_timing[8 * 1024];
#define DLN_TICK(k) { const long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - tp; tp = t_; }
#else
#define DLN_TICK(k)
#endif
#define DLN_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

constexpr int DLN_D = 256, DLN_NS = 3;
constexpr int dln_stage_bytes(int mt) { return mt * 16 * 128 + DLN_D * 128; }
constexpr int dln_smem(int mt) { return DLN_NS * dln_stage_bytes(mt); }

template <int MT>
__global__ __launch_bounds__(512, 2) void dense_ln_bwd_kernel(const DenseLnArgs p) {
  static_assert(MT % 2 == 0 && MT >= 2 && MT <= 6, "row halves of whole fragments");
  constexpr int ROWS = MT * 16, MTW = MT / 2, A_B = ROWS * 128, STAGE = dln_stage_bytes(MT);
  constexpr int NPT = MT * 2 + 32;       // 1-KiB DMA pieces per slab (A rows, then the 256 W rows: the image is linear in the piece index)
  constexpr int CH = (NPT + 7) / 8;      // pieces per wave and slab (the last one only for waves w < NPT - 8 (CH - 1))
  constexpr int NG = 2 * MTW;            // groups of four MFMAs per slab
  constexpr int NR = 2 * (MTW + 4);      // fragment reads per slab
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int r = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  if (tile >= p.ntiles) {  // a partial-sum slot without rows (the fold reads every slot)
    for (int c = threadIdx.x; c < 2 * DLN_D; c += 512) p.part[(long)tile * 2 * DLN_D + c] = 0.f;
    return;
  }
  const long row0 = (long)tile * ROWS;

  // slab-0 source pointers of this lane's DMA pieces q = w + 8 i (a slab further = + 64 elements)
  const bf16_t* src[CH];
  {
    const int pch = lane & 7;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int q = w + 8 * i;
      if (q < MT * 2) {
        const int row = q * 8 + (lane >> 3);
        const long gr = row0 + row < p.rows ? row0 + row : p.rows - 1;
        src[i] = p.dy + gr * p.ldy + ((pch ^ key_d(row)) << 3);
      } else {
        const int n = min((q - MT * 2) * 8 + (lane >> 3), DLN_D - 1);
        src[i] = p.W + (long)n * p.ldw + ((pch ^ key_d(n)) << 3);
      }
    }
  }
  const bool last_piece = w + 8 * (CH - 1) < NPT;  // uniform
  auto issue_piece = [&](int stage, int slab, int i) {
    if (i < CH - 1 || last_piece) glds16(src[i] + slab * 64, smem + stage * STAGE + __builtin_amdgcn_readfirstlane((w + 8 * i) * 1024));
  };
  // leave `left` (0..2) slabs of this wave's pieces in flight
  auto wait_left = [&](int left) {
    if (left == 0) DLN_WAIT(0);
    else if (last_piece) { if (left == 1) DLN_WAIT(CH); else DLN_WAIT(2 * CH); }
    else { if (left == 1) DLN_WAIT(CH - 1); else DLN_WAIT(2 * (CH - 1)); }
  };

  float4_t acc[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int ns = p.K / 64;
#ifdef TFASR_DLN_TIMING
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tp = __builtin_readcyclecounter();
#endif
  {
    const int n0 = ns < DLN_NS ? ns : DLN_NS;
    for (int s = 0; s < n0; ++s)
#pragma unroll
      for (int i = 0; i < CH; ++i) issue_piece(s, s, i);
    wait_left(n0 - 1);
    __builtin_amdgcn_s_barrier();  // slab 0 has landed, for every wave
  }
  short8_t fa[2][2][MTW], fb[2][2][4];  // [register set][kk][fragment]
  auto read_frag = [&](int set, int stage, int ri) {
    const char* sA = smem + stage * STAGE;
    if (ri < 2 * MTW) fa[set][ri / MTW][ri % MTW] = frag_direct(sA, (wm * MTW + ri % MTW) * 16 + r, (ri / MTW) * 4 + g);
    else { const int q = ri - 2 * MTW; fb[set][q / 4][q % 4] = frag_direct(sA + A_B, wn * 64 + (q % 4) * 16 + r, (q / 4) * 4 + g); }
  };
#pragma unroll
  for (int ri = 0; ri < NR; ++ri) read_frag(0, 0, ri);
  DLN_TICK(0)

  // one slab: the MFMAs of slab s on register set `cur`; between them this wave's DMA pieces of slab s+3 and the fragment reads of slab s+1
  auto slab = [&](auto cur_c, int s) {
    constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
    wait_left(s + 2 < ns ? 1 : 0);                       // this wave's pieces of slab s+1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and its fragments of slab s are in registers
    __builtin_amdgcn_s_barrier();                        // both for every wave: stage s % 3 is free, stage (s + 1) % 3 complete
    DLN_TICK(1)
    const bool do_dma = s + DLN_NS < ns, do_read = s + 1 < ns;
    const int st_dma = s % DLN_NS, st_read = (s + 1) % DLN_NS;
    int pi = 0, ri = 0;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int kk = gi / MTW, i = gi % MTW;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur][kk][j], fa[cur][kk][i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (; pi < ((gi + 1) * CH + NG - 1) / NG; ++pi)
        if (do_dma) issue_piece(st_dma, s + DLN_NS, pi);
#pragma unroll
      for (; ri < ((gi + 1) * NR + NG - 1) / NG; ++ri)
        if (do_read) read_frag(nxt, st_read, ri);
      __builtin_amdgcn_sched_barrier(0);
    }
    DLN_TICK(2)
  };
  int s = 0;
  for (; s + 1 < ns; s += 2) {
    slab(std::integral_constant<int, 0>(), s);
    slab(std::integral_constant<int, 1>(), s + 1);
  }
  if (s < ns) slab(std::integral_constant<int, 0>(), s);
  DLN_TICK(3)

  // ---- epilogue: LayerNorm backward straight from the transposed accumulators --------------------------------------------------------
  // lane (r, g), fragment (i, j): row = row0 + (wm*MTW + i)*16 + r, columns wn*64 + j*16 + g*4 .. +3.  The lanes g and g ^ 1 first trade one
  // fragment of every pair (j, j + 1) through the lane-swap network (as gemm_big's transposed epilogue does): an even g then owns EIGHT
  // consecutive columns of fragment j, an odd g eight of fragment j + 1 - x / add are 16-byte loads, dx / dx_dropped 16-byte stores
  // (8-byte accesses run at 0.54-0.70 of the 16-byte rate, and this phase is the memory system's)
  const bool odd = (g & 1) != 0;
  float d8[MTW][2][8];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      float recv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        recv[e] = __uint_as_float(xor16_get(__float_as_uint(odd ? acc[i][2 * jp][e] : acc[i][2 * jp + 1][e]), odd));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d8[i][jp][e] = odd ? recv[e] : acc[i][2 * jp][e];
        d8[i][jp][4 + e] = odd ? acc[i][2 * jp + 1][e] : recv[e];
      }
    }
  // this lane's eight columns of pair jp: cb + jp*32 .. +7
  const int cb = wn * 64 + (odd ? 16 : 0) + (g & 2) * 4;
  uint4 xr[MTW][2], ar[MTW][2];
  float mr[MTW], rr[MTW];
  bool live[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const long row = row0 + (wm * MTW + i) * 16 + r;
    live[i] = row < p.rows;
    const long rc = live[i] ? row : p.rows - 1;
    mr[i] = p.mean[rc];
    rr[i] = p.rstd[rc];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      xr[i][jp] = *reinterpret_cast<const uint4*>(p.x + rc * DLN_D + cb + jp * 32);
      ar[i][jp] = p.add ? *reinterpret_cast<const uint4*>(p.add + rc * DLN_D + cb + jp * 32) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  float gm[2][8];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) ld8(p.gamma + cb + jp * 32, gm[jp]);
  __builtin_amdgcn_s_barrier();  // every wave is past its last fragment read: the stages are dead
  float* red = reinterpret_cast<float*>(smem);  // [4 wn][ROWS][2] row sums, then [2][256] column sums of the upper row half
  float ag[2][8], ab[2][8];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[jp][e] = 0.f; ab[jp][e] = 0.f; }
  float xh[MTW][2][8];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      float xv[8];
      unpack8(xr[i][jp], xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = live[i] ? d8[i][jp][e] * p.alpha : 0.f;
        d8[i][jp][e] = d;
        const float xhat = (xv[e] - mr[i]) * rr[i];
        xh[i][jp][e] = xhat;
        const float dg = d * gm[jp][e];
        s1 += dg;
        s2 += dg * xhat;
        ag[jp][e] += d * xhat;
        ab[jp][e] += d;
      }
    }
    s1 = xor32_sum(xor16_sum(s1));
    s2 = xor32_sum(xor16_sum(s2));
    if (g == 0) *reinterpret_cast<float2*>(red + ((wn * ROWS + (wm * MTW + i) * 16 + r) * 2)) = make_float2(s1, s2);
  }
  __syncthreads();
  const float invC = 1.f / DLN_D;
  const bool drop = p.dxd != nullptr && p.drop_p > 0.f;
  const float drop_inv = 1.f / (1.f - p.drop_p);
  const uint32_t dkey = drop_key(p.drop_seed), dthr = drop_thr(p.drop_p);
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int rl = (wm * MTW + i) * 16 + r;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float2 t = *reinterpret_cast<const float2*>(red + (q * ROWS + rl) * 2); s1 += t.x; s2 += t.y; }
    s1 *= invC; s2 *= invC;
    const long row = row0 + rl;
    if (live[i]) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float o[8];
        unpack8(ar[i][jp], o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rr[i] * (d8[i][jp][e] * gm[jp][e] - s1 - xh[i][jp][e] * s2);
        uint4 v;
        v.x = pack2_bf16(o[0], o[1]); v.y = pack2_bf16(o[2], o[3]); v.z = pack2_bf16(o[4], o[5]); v.w = pack2_bf16(o[6], o[7]);
        *reinterpret_cast<uint4*>(p.dx + row * DLN_D + cb + jp * 32) = v;
        if (drop) {  // dropout of the ROUNDED dx (= tfasr_dropout(dx)): an even / odd element pair shares one hash
          const uint64_t e0 = (uint64_t)(row * DLN_D + cb + jp * 32);
          const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
          uint32_t uu[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t hh = drop_mix(dkey, (uint32_t)((e0 + 2 * q) >> 1), (uint32_t)((e0 + 2 * q) >> 33));
            const float q0 = (hh & 0xffffu) >= dthr ? __uint_as_float(vv[q] << 16) * drop_inv : 0.f;
            const float q1 = (hh >> 16) >= dthr ? __uint_as_float(vv[q] & 0xffff0000u) * drop_inv : 0.f;
            uu[q] = pack2_bf16(q0, q1);
          }
          *reinterpret_cast<uint4*>(p.dxd + row * DLN_D + cb + jp * 32) = make_uint4(uu[0], uu[1], uu[2], uu[3]);
        }
      }
    }
  }
  DLN_TICK(4)
  // gamma / beta partial sums of the tile: over the 16 rows of a fragment by DPP, over the two row halves through LDS, one writer per column
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[jp][e] = row16_sum(ag[jp][e]); ab[jp][e] = row16_sum(ab[jp][e]); }
  __syncthreads();  // (the row sums have been read)
  if (wm == 1 && r == 0) {
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      st8(red + cb + jp * 32, ag[jp]);
      st8(red + DLN_D + cb + jp * 32, ab[jp]);
    }
  }
  __syncthreads();
  if (wm == 0 && r == 0) {
    float* po = p.part + (long)tile * 2 * DLN_D;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      float u[8], v[8];
      ld8(red + cb + jp * 32, u);
      ld8(red + DLN_D + cb + jp * 32, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { u[e] += ag[jp][e]; v[e] += ab[jp][e]; }
      st8(po + cb + jp * 32, u);
      st8(po + DLN_D + cb + jp * 32, v);
    }
  }
#ifdef TFASR_DLN_TIMING
  DLN_TICK(5)
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 8; ++q) g_dln_timing[8 * blockIdx.x + q] = ph[q];
#endif
}

static int launch_dense_ln_bwd(DenseLnArgs a, int nblk, hipStream_t stream) {
  // rows per workgroup: the smallest tile that still fits the partial-sum slots and one round of workgroups (one per CU)
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const int cap = nblk < ncu ? nblk : ncu;
  int mt = 0;
  for (int m = 2; m <= 6; m += 2)
    if ((a.rows + 16 * m - 1) / (16 * m) <= cap) { mt = m; break; }
  if (mt == 0) {
    if ((a.rows + 95) / 96 > nblk) return TFASR_STATUS_UNSUPPORTED;
    mt = 6;  // more tiles than CUs: several rounds of the largest tile
  }
  a.ntiles = (int)((a.rows + 16 * mt - 1) / (16 * mt));
  auto go = [&](auto kern, int m) {
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, dln_smem(m)); attr_done = true; }
    TFASR_KLAUNCH(kern, dim3((unsigned)nblk), dim3(512), dln_smem(m), stream, a);
  };
  if (mt == 2) go(dense_ln_bwd_kernel<2>, 2);
  else if (mt == 4) go(dense_ln_bwd_kernel<4>, 4);
  else go(dense_ln_bwd_kernel<6>, 6);
  return TFASR_STATUS_SUCCESS;
}
