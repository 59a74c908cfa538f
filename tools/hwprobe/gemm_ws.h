// Weights-stationary bf16 GEMM for the Conformer block's Dense layers (included by gemm_fast.hip).
//
//     D[M, N] = alpha * A[M, K] @ Wt[N, K]^T  [+ bias] [epilogue terms]          A = activations (many rows), Wt = a weight, k contiguous
//
// Every Dense / pointwise-conv product of a Conformer block has a LONG M (B*T' = 15-24 k rows) and a SMALL weight (K, N <= 1024: 128 KiB
// to 512 KiB).  The tiled kernels above stream BOTH operands slab by slab through a two-stage LDS ring: a 128 x 64 tile with K = 256 is four
// dependent DMA round trips (~2 400 clocks each) plus a 4 300-clock epilogue for 512 clocks of MFMA per slab - the products run at the DMA
// round trip, 13-18 us for 20 MB of traffic.  Here the weight does not move:
//   * a workgroup loads ONE column group of the weight (CG columns x K, <= 128 KiB) into LDS once and keeps it for its whole life;
//   * it then walks over row panels (64 * MT rows; 16 * MT per wave).  A wave's activation rows are the MFMA B operand read STRAIGHT from
//     global memory into registers (16 contiguous bytes per lane: row-major activations are already k-contiguous) - no LDS, no barrier, the
//     next panel's rows are in flight while the current panel is multiplied;
//   * transposed orientation: D^T = Wt A^T, i.e. the weight fragment is the MFMA A operand and the C layout gives every lane FOUR
//     CONSECUTIVE output columns of one row: the epilogue is register-direct (8-byte loads of the multiplier / residual, 8-byte stores),
//     no LDS transposition pass;
//   * no barrier after the weight has landed: the four waves drift apart and overlap each other's loads, MFMAs and stores by themselves.
// XCD placement: panels are dealt to XCDs (panel p on XCD p % 8) and the column groups of a panel run on the SAME XCD, so an activation
// panel enters one L2 once; the whole weight (<= 512 KiB) sits in every XCD's 4 MiB L2.
//
// LDS image of the column group: row n (a weight column) = K/8 chunks of 16 B, chunk q stored at position q ^ (n & 15): the 16 lanes of a
// ds_read_b128 service group read 16 different rows at logical chunk 4 kk + g - distinct positions mod 16, conflict-free.
#pragma once

namespace ws {

typedef short8_t bf8_t;  // eight bf16 (the MFMA builtin's operand type in this tool chain)

struct Args {
  const bf16_t* A; long lda;      // [M, K]
  const bf16_t* Wt; long ldw;     // [N, K] (k contiguous)
  bf16_t* D; long ldd;            // [M, N]
  const float* bias;              // [N] or null
  const bf16_t* mul;              // [M, N] (ldd) or null: D *= mul   (stored gradient factor: tfasr_gemm_args.dact = TFASR_ACT_MUL)
  const bf16_t* res; float beta;  // [M, N] (ldd) or null: D = res + beta * D
  float alpha;
  int M, N;
  int npanels;                    // ceil(M / (64 * MT))
  int ngroups;                    // ceil(N / CG)
  int spx;                        // workgroups per (XCD, column group): gridDim.x = 8 * ngroups * spx
};

template <int K, int CG, int MT, bool MUL, bool RES>
__global__ __launch_bounds__(256, 1) void gemm_ws_nt_kernel(const Args p) {
  constexpr int CPR = K / 8;          // 16-byte chunks per weight row
  constexpr int NT = CG / 16;         // n-tiles per wave (every wave owns ALL columns of the group for its own rows)
  constexpr int KS = K / 32;          // k-steps
  constexpr int RP = 64 * MT;         // rows per panel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int grp = j % p.ngroups, slot = j / p.ngroups;
  const int n0 = grp * CG;

  // ---- the column group -> LDS (once) ----
  {
    constexpr int NCH = CG * CPR;
    for (int c = threadIdx.x; c < NCH; c += 256) {
      const int n = c / CPR, q = c % CPR;
      const int nn = min(n0 + n, p.N - 1);
      const uint4 v = *reinterpret_cast<const uint4*>(p.Wt + (long)nn * p.ldw + q * 8);
      *reinterpret_cast<uint4*>(smem + ((size_t)(n * CPR + (q ^ (n & 15))) << 4)) = v;
    }
  }
  float bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = n0 + nt * 16 + g * 4 + e;
      bv[nt][e] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
    }

  // panels of this workgroup: p = xcd + 8 * (slot + t * spx)
  auto panel_of = [&](int t) { return xcd + 8 * (slot + t * p.spx); };
  auto load_x = [&](int pn, bf8_t (&x)[MT][KS]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long row = min((long)pn * RP + (w * MT + mt) * 16 + r, (long)p.M - 1);
      const bf16_t* src = p.A + row * p.lda + g * 8;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) x[mt][kk] = *reinterpret_cast<const bf8_t*>(src + kk * 32);
    }
  };
  bf8_t xa[MT][KS], xb[MT][KS];
  int t = 0;
  int pn = panel_of(0);
  if (pn < p.npanels) load_x(pn, xa);
  __syncthreads();  // the weight image is complete

  const char* wbase = smem + ((size_t)(r * CPR) << 4);
  auto compute = [&](int pcur, bf8_t (&x)[MT][KS]) {
    float4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf8_t a = *reinterpret_cast<const bf8_t*>(wbase + ((size_t)(nt * 16 * CPR + ((kk * 4 + g) ^ r)) << 4));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, x[mt][kk], acc[mt][nt], 0, 0, 0);
      }
    // ---- epilogue: a lane holds D[row = .. + r][n0 + nt*16 + g*4 .. +4] ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long row = (long)pcur * RP + (w * MT + mt) * 16 + r;
      if (row < p.M) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = n0 + nt * 16 + g * 4;
          if (col < p.N) {  // (N is a multiple of 4: a lane's four columns exist together)
            const long idx = row * p.ldd + col;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = p.alpha * acc[mt][nt][e] + bv[nt][e];
            if constexpr (MUL) {
              const uint2 m = *reinterpret_cast<const uint2*>(p.mul + idx);
              v[0] *= __uint_as_float(m.x << 16); v[1] *= __uint_as_float(m.x & 0xffff0000u);
              v[2] *= __uint_as_float(m.y << 16); v[3] *= __uint_as_float(m.y & 0xffff0000u);
            }
            if constexpr (RES) {
              const uint2 m = *reinterpret_cast<const uint2*>(p.res + idx);
              v[0] = __uint_as_float(m.x << 16) + p.beta * v[0]; v[1] = __uint_as_float(m.x & 0xffff0000u) + p.beta * v[1];
              v[2] = __uint_as_float(m.y << 16) + p.beta * v[2]; v[3] = __uint_as_float(m.y & 0xffff0000u) + p.beta * v[3];
            }
            uint2 o;
            o.x = pack2_bf16(v[0], v[1]);
            o.y = pack2_bf16(v[2], v[3]);
            *reinterpret_cast<uint2*>(p.D + idx) = o;
          }
        }
      }
    }
  };

  while (pn < p.npanels) {
    const int pnext = panel_of(t + 1);
    if (pnext < p.npanels) load_x(pnext, xb);
    compute(pn, xa);
    pn = pnext; ++t;
    if (pn >= p.npanels) break;
    const int pnext2 = panel_of(t + 1);
    if (pnext2 < p.npanels) load_x(pnext2, xa);
    compute(pn, xb);
    pn = pnext2; ++t;
  }
}

}  // namespace ws
