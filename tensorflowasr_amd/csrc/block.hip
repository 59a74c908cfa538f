// Native executor of one Conformer block (host code only): queues every kernel of ConformerBlock.call
// (encoders/conformer.py:430-520: FFModule -> MHSAModule -> ConvModule -> FFModule -> LayerNorm) and of its backward on
// one HIP stream, carving all intermediates out of two caller-owned arenas.  One ctypes call replaces ~20 (forward) /
// ~50 (backward) per-kernel calls from the Python host, whose per-launch overhead (~25 us) otherwise rivals the GPU time
// of a Conformer-M step.  The kernels are the same C-ABI entry points the Python host uses, so results are identical.
//
// Phases: the synchronised BatchNorm of the conv module needs an all-reduce of its batch statistics between
// tfasr_bn_stats and tfasr_bn_finalize (forward) and between tfasr_bn_bwd_stats and tfasr_bn_apply_bwd (backward);
// that communicator belongs to the caller, so each direction can be run as phase A | all-reduce | phase B.
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <vector>

extern int g_tfasr_group_beside;  // gemm_fast.hip: 1 around a grouped launch that shares the chip with another stream's chain

bool tfasr_bn_rows_kernel_ok(int C);  // norm.hip
namespace {

struct Arena {
  char* base;
  size_t off, cap;
  bool ok;
  size_t peak;
  void* get(size_t n) {
    const size_t a = (off + 255) & ~(size_t)255;
    if (base && a + n > cap) { ok = false; return nullptr; }
    off = a + n;
    if (off > peak) peak = off;
    return base ? base + a : (void*)(uintptr_t)256;  // dry run (sizing): non-null dummy, never dereferenced
  }
};

// saved-for-backward pointers + phase carry (lives in the caller's opaque ctx buffer)
struct Ctx {
  // feed-forward modules
  void *ff_x[2], *ff_ln[2], *ff_z[2], *ff_h[2];
  float *ff_mean[2], *ff_rstd[2];
  int ff_zfactor[2];  // 1: ff_z holds the data gradient's factor swish'(z) * mask1 / (1 - p) (fused forward), 0: the pre-activation z
  // attention
  void *at_x, *at_ln, *at_qkv, *at_pext, *at_att, *at_qu, *at_qv, *at_probs;
  float *at_mean, *at_rstd, *at_lse;
  // convolution module
  void *cv_x, *cv_ln, *cv_a, *cv_g, *cv_cv, *cv_sw, *cv_y;
  float *cv_mean, *cv_rstd, *cv_fin, *cv_nmean, *cv_nrstd;
  // block layer norm
  void* ln_x;
  float *ln_mean, *ln_rstd;
  // phase carry
  size_t stash_off, scratch_off;
  void *bw_dx, *bw_dsw, *bw_cur, *bw_nxt;
  void *bw_curd, *bw_nxtd;  // dropout(bw_cur / bw_nxt) for the consumer's dropped branch, written by the producing LayerNorm backward
  long drop_epoch;
  int fused;
  // LayerNorm gamma / beta gradients as per-block partial sums, folded by ONE launch at the end of the block's backward
  float* ln_part;
  int ln_nblk, ln_nsets;
  bool ln_ext;  // ln_part is the caller's buffer (tfasr_block_io.ln_part_ext): the fold is tfasr_block_ln_fold_all's
  float *ln_dg[8], *ln_db[8];
  // what the last backward call LEFT TO THE CALLER (tfasr_block_bwd_left): bit 0 = the table gradient from dS (io->ds_keep / qv_keep are
  // filled, tfasr_relattn_dpext not launched), bit 1 = the depthwise weight gradient (io->dcv_keep), bit 2 = the positional-projection
  // gradients (defer_pos_grad honoured), bit 3 = the LayerNorm fold (ln_part_ext)
  int left;
};

// one GEMM launch description (defaults = plain product)
struct G {
  const void* A = nullptr; int lda = 0; int ta = 0; const void* B = nullptr; int ldb = 0; int tb = 0; void* D = nullptr; int ldd = 0; int M = 0, N = 0, K = 0;
  const float* bias = nullptr; const void* res = nullptr; const void* dact_z = nullptr; void* prez = nullptr;
  float alpha = 1.f, beta = 1.f; int act = 0, dact = 0, out_f32 = 0, accumulate = 0, split_k = 1; float drop_p = 0.f; long drop_seed = 0;
  float* colsum = nullptr;
  const void* bns_x = nullptr; const float* bns_fin = nullptr; float* bns_out = nullptr; int bns_copies = 0;  // tfasr_gemm_args.bns_*
  int side = 0;  // 1 = a weight gradient: collected into the block's grouped launch when the executor defers them
  int nb1 = 1, nb2 = 1; long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0, sD1 = 0, sD2 = 0;
};

// Per-device stream state of the executor: the low-priority stream the grouped weight gradients of a block run on, beside the next
// block's backward chain (a per-pair fork / join of single weight gradients measured slower - 37.9 vs 36.6 ms per step in round 2 - and is gone).
struct Side {
  // asynchronous grouped weight gradients (tfasr_block_io.wgrad_slot): a stream of their own, per slot the event behind the last
  // launch queued under it
  hipStream_t sw = nullptr;
  hipEvent_t wfork = nullptr, wdone[2] = {nullptr, nullptr};
  bool wpending[2] = {false, false};
  bool wok = false, wtried = false;
  // probe (tfasr_block_wgrad_probe): HIP events around every grouped weight-gradient launch ON THE STREAM IT RUNS ON, so that bench.py can
  // report the group's duration inside the step (beside the next block's chain), not only in an isolated loop
  bool probe = false;
  std::vector<hipEvent_t> pev;  // pairs (before, after)
  size_t pused = 0;
};
Side& side_for_device() {
  static Side sides[64];
  static bool tried[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  Side& sd = sides[dev];
  if (!tried[dev]) {
    tried[dev] = true;
  }
  return sd;
}
// the weight-gradient stream, created on first use
bool wgrad_stream_ready(Side& sd) {
  if (!sd.wtried) {
    sd.wtried = true;
    int least = 0, greatest = 0;  // lowest priority: the group only fills what the main chain leaves idle
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    sd.wok = hipStreamCreateWithPriority(&sd.sw, hipStreamNonBlocking, least) == hipSuccess &&
             hipEventCreateWithFlags(&sd.wfork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sd.wdone[0], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sd.wdone[1], hipEventDisableTiming) == hipSuccess;
  }
  return sd.wok;
}

int wgrad_wait(Side& sd, int slot_mask, hipStream_t s) {
  for (int k = 0; k < 2; ++k)
    if ((slot_mask >> k & 1) && sd.wpending[k]) {
      if (hipStreamWaitEvent(s, sd.wdone[k], 0) != hipSuccess) return TFASR_STATUS_EXECUTION_FAILED;
      sd.wpending[k] = false;
    }
  return TFASR_STATUS_SUCCESS;
}

struct Ex {
  const tfasr_block_cfg* c;
  const tfasr_block_params* P;
  const tfasr_block_io* io;
  Ctx* k;
  Arena stash, scratch;
  hipStream_t s;
  Side* side = nullptr;
  bool dry;
  int st;
  long rows;
  int esz;
  // Deferred weight gradients (backward only): the Dense-layer gradients of the block are collected and queued as ONE grouped
  // launch at the end of the phase (tfasr_gemm_group).  Their operands must then outlive the module that produced them, so
  // in this mode the backward never rewinds its scratch arena and every module writes into fresh gradient buffers.
  bool defer = false;
  std::vector<tfasr_gemm_args> pending;

  const float* fp(int i) const { return P->flat + P->off[i]; }
  const void* wp(int i) const { return (const char*)P->shadow + P->off[i] * (long)esz; }
  float* gp(int i) const { return P->grad + P->off[i]; }
  void chk(int status) { if (status != TFASR_STATUS_SUCCESS && st == TFASR_STATUS_SUCCESS) st = status; }
  void* act(Arena& a, long elems) { return a.get((size_t)elems * esz); }
  float* f32(Arena& a, long elems) { return (float*)a.get((size_t)elems * 4); }
  void zero(void* p, size_t bytes) { if (!dry) { if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess) chk(TFASR_STATUS_EXECUTION_FAILED); } }

  float drop_p() const { return c->training ? c->drop_p : 0.f; }
  long seed(int site) const { return drop_p() > 0.f ? ((k->drop_epoch * 8192 + c->site0 + site) & 0x7FFFFFFFFFFFL) : 0; }

  void gemm(const G& g) {
    // split-K weight gradients reduce through scratch (released right after the call: same-stream ordering keeps it safe)
    float* ws = nullptr;
    long ws_elems = 0;
    const size_t mark = scratch.off;
    scratch.off = mark;
    if (dry) return;
    tfasr_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = g.A; a.B = g.B; a.D = g.D; a.bias = g.bias; a.res = g.res; a.dact_z = g.dact_z; a.prez = g.prez;
    a.M = g.M; a.N = g.N; a.K = g.K; a.lda = g.lda; a.ldb = g.ldb; a.ldd = g.ldd; a.trans_a = g.ta; a.trans_b = g.tb;
    a.nb1 = g.nb1; a.nb2 = g.nb2; a.sA1 = g.sA1; a.sA2 = g.sA2; a.sB1 = g.sB1; a.sB2 = g.sB2; a.sD1 = g.sD1; a.sD2 = g.sD2;
    a.alpha = g.alpha; a.beta = g.beta; a.act = g.act; a.dact = g.dact; a.dtype = c->dtype; a.out_f32 = g.out_f32;
    a.accumulate = g.accumulate; a.split_k = g.split_k; a.drop_p = g.drop_p; a.drop_seed = g.drop_seed;
    a.ws = ws; a.ws_elems = ws ? ws_elems : 0;
    a.colsum = g.colsum;
    a.bns_x = g.bns_x; a.bns_fin = g.bns_fin; a.bns_out = g.bns_out; a.bns_copies = g.bns_copies;
    if (defer && g.side && !ws) { pending.push_back(a); return; }
    int st = tfasr_gemm(&a, s);
    if (st == TFASR_STATUS_UNSUPPORTED && a.bns_out) {  // the statistics epilogue does not take this product: plain product, the caller's second pass
      a.bns_x = nullptr; a.bns_fin = nullptr; a.bns_out = nullptr; a.bns_copies = 0;
      st = tfasr_gemm(&a, s);
      bns_done = false;
    }
    chk(st);
  }
  bool bns_done = false;  // set by the caller before a product with bns_out, cleared when the epilogue was not available
  void probe_mark(hipStream_t st_) {
    if (!side || !side->probe) return;
    if (side->pused == side->pev.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) { side->probe = false; return; }
      side->pev.push_back(e);
    }
    if (hipEventRecord(side->pev[side->pused++], st_) != hipSuccess) chk(TFASR_STATUS_EXECUTION_FAILED);
  }
  void flush_wgrads() {
    if (pending.empty()) return;
    const int slot = io->wgrad_slot;
    if ((slot == 1 || slot == 2) && side && wgrad_stream_ready(*side)) {
      // beside the next block's backward: everything queued on `s` so far (the operands) precedes the launch, nothing on `s` waits
      // for it until the slot's arena is reused or the caller joins
      if (hipEventRecord(side->wfork, s) != hipSuccess || hipStreamWaitEvent(side->sw, side->wfork, 0) != hipSuccess) chk(TFASR_STATUS_EXECUTION_FAILED);
      g_tfasr_group_beside = 1;  // (one workgroup per CU: the group shares the chip with the next block's chain)
      probe_mark(side->sw);
      chk(tfasr_gemm_group(pending.data(), (int)pending.size(), side->sw));
      probe_mark(side->sw);
      g_tfasr_group_beside = 0;
      if (hipEventRecord(side->wdone[slot - 1], side->sw) != hipSuccess) chk(TFASR_STATUS_EXECUTION_FAILED);
      side->wpending[slot - 1] = true;
    } else {
      probe_mark(s);
      chk(tfasr_gemm_group(pending.data(), (int)pending.size(), s));
      probe_mark(s);
    }
    pending.clear();
  }
  void rewind(size_t mark) { if (!defer) scratch.off = mark; }
  // Split-K factor of a block's weight gradient (few 128x128 output tiles, K = B*T rows).  The f32 accumulate costs a flat
  // ~3.1 ns per 1000 atomics (320 G atomics/s, independent of contention or XCD placement: tools/hwprobe/atomic_bench.hip),
  // i.e. every extra k-slice adds M*N atomics, so the sweet spot is ONE workgroup per CU, not two: on [256,1024,12096]
  // split 16 = 24.0 us, 11 = 26.6, 24 = 28.4, 32 = 33.3 (tools/hwprobe/gemm_timing.hip).
  static int split_k(int M, int N, long K) {
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles >= 256 || K <= 2048) return 1;
    const long target = 256;
    long v = target / tiles;
    if (K / 512 < v) v = K / 512;
    if (v >= 8) v = v / 8 * 8;  // whole k-slices per XCD (gemm_fast.hip split-K mapping needs split % 8 == 0)
    return (int)(v < 1 ? 1 : v);
  }
  // y = x @ W + b with the fused epilogue terms (W stored [din, dout] in the compute-dtype shadow)
  void dense(const void* x, int wi, int bi, void* y, int din, int dout, G o = G()) {
    o.A = x; o.lda = din; o.ta = 0; o.B = wp(wi); o.ldb = dout; o.tb = 0; o.D = y; o.ldd = dout; o.M = (int)rows; o.N = dout; o.K = din;
    o.bias = fp(bi);
    gemm(o);
  }
  // y = LN(x) W + b with ln / mean / rstd saved: one launch where the shape allows it (tfasr_ln_dense_fwd), else LayerNorm + Dense
  void ln_dense(const void* x, int gi, int bi_ln, void* ln, float* mean, float* rstd, int wi, int bi, void* y, int dout) {
    if (!dry && ln_dense_on()) {
      const int st = tfasr_ln_dense_fwd(x, fp(gi), fp(bi_ln), wp(wi), fp(bi), y, ln, mean, rstd, rows, c->d, dout, c->ln_eps, c->dtype, s);
      if (st != TFASR_STATUS_UNSUPPORTED) { chk(st); return; }
    }
    ln_fwd(x, gi, bi_ln, ln, mean, rstd);
    dense(ln, wi, bi, y, c->d, dout);
  }
  static bool ln_dense_on() { return true; }  // (22.4 us against 7.5 + 14..17 for the two launches; step -0.11 ms, 4 pairs on one box)
  // gW += alpha x^T dy ; gb += alpha colsum(dy) ; dx = alpha (dy @ W^T) [* act'(z)] [* dropmask]
  void dense_bwd(const void* dy, const void* x, int wi, int bi, int din, int dout, void* dx, float alpha = 1.f, const void* dact_z = nullptr,
                 int dact = 0, float dp = 0.f, long dseed = 0, const void* bns_x = nullptr, const float* bns_fin = nullptr, float* bns_out = nullptr,
                 int bns_copies = 0) {
    G w; w.A = x; w.lda = din; w.ta = 1; w.B = dy; w.ldb = dout; w.tb = 0; w.D = gp(wi); w.ldd = dout; w.M = din; w.N = dout; w.K = (int)rows;
    w.alpha = alpha; w.out_f32 = 1; w.accumulate = 1; w.split_k = split_k(din, dout, rows);
    w.colsum = gp(bi);  // bias gradient in the same launch
    w.side = 1;
    gemm(w);
    if (!dx) return;
    G d; d.A = dy; d.lda = dout; d.ta = 0; d.B = wp(wi); d.ldb = dout; d.tb = 1; d.D = dx; d.ldd = din; d.M = (int)rows; d.N = din; d.K = dout;
    d.alpha = alpha; d.dact_z = dact_z; d.dact = dact; d.drop_p = dp; d.drop_seed = dseed;
    d.bns_x = bns_x; d.bns_fin = bns_fin; d.bns_out = bns_out; d.bns_copies = bns_copies;
    gemm(d);
  }
  const void* mask_grad(const void* dy, long elems, int site) {
    if (drop_p() <= 0.f) return dy;
    void* t = act(scratch, elems);
    if (!dry) chk(tfasr_dropout(dy, t, elems, drop_p(), seed(site), c->dtype, s));
    return t;
  }
  void ln_fwd(const void* x, int gi, int bi, void* y, float* mean, float* rstd) {
    if (!dry) chk(tfasr_layernorm_fwd(x, fp(gi), fp(bi), y, mean, rstd, rows, c->d, c->ln_eps, c->dtype, s));
  }
  // dxd != nullptr: the same kernel also writes dropout(dx) with the NEXT module's (backward order) output-dropout mask, which
  // that module would otherwise produce with a separate pass over dx (mask_grad): 4-5 launches and 2 x rows*d of traffic per block
  void ln_bwd(const void* dy, const void* x, int gi, int bi, const float* mean, const float* rstd, const void* add, void* dx,
              void* dxd = nullptr, int next_site = -1) {
    if (dry) return;
    const bool with_drop = dxd && drop_p() > 0.f && next_site >= 0;
    if (k->ln_part && k->ln_nsets < 8) {
      float* part = k->ln_part + (size_t)k->ln_nsets * k->ln_nblk * 2 * c->d;
      chk(tfasr_layernorm_bwd_part_n(dy, x, fp(gi), mean, rstd, add, dx, part, k->ln_nblk, with_drop ? dxd : nullptr, with_drop ? drop_p() : 0.f,
                                     with_drop ? seed(next_site) : 0, rows, c->d, c->dtype, s));
      k->ln_dg[k->ln_nsets] = gp(gi);
      k->ln_db[k->ln_nsets] = gp(bi);
      ++k->ln_nsets;
      return;
    }
    if (with_drop)
      chk(tfasr_layernorm_bwd_drop(dy, x, fp(gi), mean, rstd, add, dx, gp(gi), gp(bi), dxd, drop_p(), seed(next_site), rows, c->d, c->dtype, s));
    else
      chk(tfasr_layernorm_bwd(dy, x, fp(gi), mean, rstd, add, dx, gp(gi), gp(bi), rows, c->d, c->dtype, s));
  }
  // Dense data gradient + the LayerNorm backward in front of the layer in ONE launch (tfasr_dense_ln_bwd) where its shape range allows;
  // the weight / bias gradients are queued as by dense_bwd.  false = nothing was launched for the data side: the caller takes the
  // two-launch route.  (dry runs take that route too: it allocates a superset.)
  bool dense_ln_bwd(const void* dy, const void* xw, int wi, int bi, int dout, const void* x, int gi, int bi_ln, const float* mean, const float* rstd,
                    const void* add, void* dx, void* dxd, int next_site) {
    if (dry || !k->ln_part || k->ln_nsets >= 8 || c->dtype != TFASR_BF16 || c->d != 256 || (dout % 64) != 0 || !fuse_dense_ln()) return false;
    if ((rows + 95) / 96 > k->ln_nblk) return false;
    const bool with_drop = dxd && drop_p() > 0.f && next_site >= 0;
    float* part = k->ln_part + (size_t)k->ln_nsets * k->ln_nblk * 2 * c->d;
    const int fst = tfasr_dense_ln_bwd(dy, wp(wi), dout, x, fp(gi), mean, rstd, add, dx, part, k->ln_nblk, with_drop ? dxd : nullptr,
                                       with_drop ? drop_p() : 0.f, with_drop ? seed(next_site) : 0, rows, c->d, 1.f, c->dtype, s);
    if (fst == TFASR_STATUS_UNSUPPORTED) return false;
    chk(fst);
    k->ln_dg[k->ln_nsets] = gp(gi);
    k->ln_db[k->ln_nsets] = gp(bi_ln);
    ++k->ln_nsets;
    dense_bwd(dy, xw, wi, bi, c->d, dout, nullptr);  // weight / bias gradient only
    return true;
  }
  static bool fuse_dense_ln() {
    static const bool on = !(getenv("TFASR_DENSE_LN") && getenv("TFASR_DENSE_LN")[0] == '0');  // A/B switch of the fused launch
    return on;
  }
  const void* masked(const void* dy, const void* pre, long elems, int site) {
    if (drop_p() <= 0.f) return dy;
    return pre ? pre : mask_grad(dy, elems, site);
  }

  // ------------------------------------------------------------------------------------------ FFModule
  static bool bn_in_conv() { return true; }   // (32.9 us against 20.0 + 19.6 for the two launches; step -0.03 ms, 4 pairs on one box)
  static bool glu_in_conv() { return true; }  // (18.1 us against 6.4 + 13.2 for the two launches; step -0.04 ms, 4 pairs on one box)
  static bool ffn_factor() { return true; }  // (same-box A/B against storing z: -0.16 ms per step, profiles/r06_ab/ffn_backward_factor.txt)
  void ffm_fwd(int m, const void* x, void* y, int site) {
    const int b0 = m == 0 ? TFASR_BP_FF1_LN_G : TFASR_BP_FF2_LN_G;
    const int d = c->d, F = c->dff;
    k->ff_x[m] = (void*)x;
    k->ff_ln[m] = act(stash, rows * d);
    k->ff_mean[m] = f32(stash, rows);
    k->ff_rstd[m] = f32(stash, rows);
    k->ff_z[m] = c->save ? act(stash, rows * F) : nullptr;
    k->ff_h[m] = act(stash, rows * F);
    if (!dry) {  // one launch where the shape allows it (ffn_fused.h); UNSUPPORTED -> the three launches below
      // the fused launch stores the data gradient's FACTOR swish'(z) * mask1 / (1 - p) where the three-launch route stores z (k->ff_zfactor)
      const int fst = tfasr_ffn_fused_fwd2(x, fp(b0), fp(b0 + 1), wp(b0 + 2), fp(b0 + 3), wp(b0 + 4), fp(b0 + 5), y, k->ff_ln[m], k->ff_mean[m], k->ff_rstd[m],
                                           k->ff_z[m], ffn_factor() ? 1 : 0, k->ff_h[m], rows, d, F, c->ln_eps, c->ffm_res, drop_p(), seed(site), seed(site + 1), c->dtype, s);
      if (fst != TFASR_STATUS_UNSUPPORTED) { chk(fst); k->ff_zfactor[m] = ffn_factor() ? 1 : 0; return; }
    } else if (c->dtype == TFASR_BF16 && d == 256 && (F % 64) == 0) {
      return;  // (sizing run: the fused launch needs no scratch)
    }
    k->ff_zfactor[m] = 0;
    ln_fwd(x, b0, b0 + 1, k->ff_ln[m], k->ff_mean[m], k->ff_rstd[m]);
    G a; a.act = TFASR_ACT_SWISH; a.prez = k->ff_z[m]; a.drop_p = drop_p(); a.drop_seed = seed(site);
    dense(k->ff_ln[m], b0 + 2, b0 + 3, k->ff_h[m], d, F, a);
    G b; b.res = x; b.beta = c->ffm_res; b.drop_p = drop_p(); b.drop_seed = seed(site + 1);
    b.A = k->ff_h[m]; b.lda = F; b.ta = 0; b.B = wp(b0 + 4); b.ldb = d; b.tb = 0; b.D = y; b.ldd = d; b.M = (int)rows; b.N = d; b.K = F;
    b.bias = fp(b0 + 5);
    gemm(b);
  }
  void ffm_bwd(int m, const void* dy, const void* dy_dropped, void* dx, void* dxd, int site, int next_site) {
    const int b0 = m == 0 ? TFASR_BP_FF1_LN_G : TFASR_BP_FF2_LN_G;
    const int d = c->d, F = c->dff;
    const size_t mark = scratch.off;
    const void* dyd = masked(dy, dy_dropped, rows * d, site + 1);
    void* dz = act(scratch, rows * F);
    // (x = h, W = d2) -> dz [rows, F] = f * (dyd @ W2^T) * swish'(z) * mask1
    {
      G w; w.A = k->ff_h[m]; w.lda = F; w.ta = 1; w.B = dyd; w.ldb = d; w.tb = 0; w.D = gp(b0 + 4); w.ldd = d; w.M = F; w.N = d; w.K = (int)rows;
      w.alpha = c->ffm_res; w.out_f32 = 1; w.accumulate = 1; w.split_k = split_k(F, d, rows);
      w.colsum = gp(b0 + 5);
      w.side = 1;
      gemm(w);
      G g; g.A = dyd; g.lda = d; g.ta = 0; g.B = wp(b0 + 4); g.ldb = d; g.tb = 1; g.D = dz; g.ldd = F; g.M = (int)rows; g.N = F; g.K = d;
      g.alpha = c->ffm_res; g.dact_z = k->ff_z[m];
      if (k->ff_zfactor[m]) g.dact = TFASR_ACT_FACTOR;  // the forward stored swish'(z) * mask1 / (1 - p)
      else { g.dact = TFASR_ACT_SWISH; g.drop_p = drop_p(); g.drop_seed = seed(site); }
      gemm(g);
    }
    void* dln = act(scratch, rows * d);
    if (!dense_ln_bwd(dz, k->ff_ln[m], b0 + 2, b0 + 3, F, k->ff_x[m], b0, b0 + 1, k->ff_mean[m], k->ff_rstd[m], dy, dx, dxd, next_site)) {
      dense_bwd(dz, k->ff_ln[m], b0 + 2, b0 + 3, d, F, dln);
      ln_bwd(dln, k->ff_x[m], b0, b0 + 1, k->ff_mean[m], k->ff_rstd[m], dy, dx, dxd, next_site);
    }
    rewind(mark);
  }
  // ------------------------------------------------------------------------------------------ MHSAModule
  void mhsa_fwd(const void* x, void* y, int site) {
    const int d = c->d, H = c->H, dh = c->dh, HD = H * dh, T = c->T, B = c->B, R1 = 2 * T;
    const float scale = 1.f / sqrtf((float)(c->dh_logical > 0 ? c->dh_logical : dh));
    k->at_x = (void*)x;
    k->at_ln = act(stash, rows * d);
    k->at_mean = f32(stash, rows);
    k->at_rstd = f32(stash, rows);
    k->at_qkv = act(stash, rows * 3 * HD);
    k->at_pext = act(stash, (long)R1 * HD);
    k->at_att = act(stash, rows * HD);
    ln_dense(x, TFASR_BP_AT_LN_G, TFASR_BP_AT_LN_B, k->at_ln, k->at_mean, k->at_rstd, TFASR_BP_AT_QKV_W, TFASR_BP_AT_QKV_B, k->at_qkv, 3 * HD);
    if (!dry && io->pext_pre) {
      // the projected position table does not depend on the activations: the caller has computed it ahead of the chain (io->pext_pre);
      // (the stash slot stays allocated so that the arena layout does not depend on the option)
      k->at_pext = (void*)io->pext_pre;
    } else {
      G p; p.A = P->pe; p.lda = d; p.ta = 0; p.B = wp(TFASR_BP_AT_POS_W); p.ldb = HD; p.tb = 0; p.D = k->at_pext; p.ldd = HD; p.M = R1; p.N = HD; p.K = d;
      p.bias = fp(TFASR_BP_AT_POS_B);
      gemm(p);
    }
    if (k->fused) {
      k->at_lse = f32(stash, (long)B * H * T);
      if (!dry)
        chk(tfasr_relattn_fused_fwd(k->at_qkv, fp(TFASR_BP_AT_U), fp(TFASR_BP_AT_V), k->at_pext, io->lengths, k->at_att, k->at_lse, B, H, T, dh,
                                    scale, c->use_mask, c->chunk_size, c->history_size, c->dtype, s));
    } else {
      const int Tp = (T + 7) / 8 * 8, R1p = (R1 + 7) / 8 * 8;
      k->at_qu = act(stash, rows * HD);
      k->at_qv = act(stash, rows * HD);
      k->at_probs = act(stash, (long)B * H * T * Tp);
      const size_t mark = scratch.off;
      void* pos = act(scratch, (long)B * H * T * R1p);
      if (!dry) chk(tfasr_bias2_fwd(k->at_qkv, 3 * HD, fp(TFASR_BP_AT_U), fp(TFASR_BP_AT_V), k->at_qu, k->at_qv, rows, HD, c->dtype, s));
      const char* kk = (const char*)k->at_qkv + (size_t)HD * esz;
      const char* vv = (const char*)k->at_qkv + (size_t)2 * HD * esz;
      G cg; cg.A = k->at_qu; cg.lda = HD; cg.ta = 0; cg.B = kk; cg.ldb = 3 * HD; cg.tb = 1; cg.D = k->at_probs; cg.ldd = Tp; cg.M = T; cg.N = T; cg.K = dh;
      cg.nb1 = B; cg.nb2 = H; cg.sA1 = (long)T * HD; cg.sA2 = dh; cg.sB1 = (long)T * 3 * HD; cg.sB2 = dh; cg.sD1 = (long)H * T * Tp; cg.sD2 = (long)T * Tp;
      cg.alpha = scale;
      gemm(cg);
      G pg; pg.A = k->at_qv; pg.lda = HD; pg.ta = 0; pg.B = k->at_pext; pg.ldb = HD; pg.tb = 1; pg.D = pos; pg.ldd = R1p; pg.M = T; pg.N = R1; pg.K = dh;
      pg.nb1 = B; pg.nb2 = H; pg.sA1 = (long)T * HD; pg.sA2 = dh; pg.sB1 = 0; pg.sB2 = dh; pg.sD1 = (long)H * T * R1p; pg.sD2 = (long)T * R1p;
      pg.alpha = scale;
      gemm(pg);
      if (!dry) chk(tfasr_relattn_softmax_fwd_streaming(k->at_probs, pos, io->lengths, k->at_probs, B, H, T, Tp, R1p, c->use_mask, c->chunk_size,
                                                          c->history_size, c->dtype, s));
      G ag; ag.A = k->at_probs; ag.lda = Tp; ag.ta = 0; ag.B = vv; ag.ldb = 3 * HD; ag.tb = 0; ag.D = k->at_att; ag.ldd = HD; ag.M = T; ag.N = dh; ag.K = T;
      ag.nb1 = B; ag.nb2 = H; ag.sA1 = (long)H * T * Tp; ag.sA2 = (long)T * Tp; ag.sB1 = (long)T * 3 * HD; ag.sB2 = dh; ag.sD1 = (long)T * HD; ag.sD2 = dh;
      gemm(ag);
      scratch.off = mark;
    }
    G o; o.res = x; o.beta = c->mhsa_res; o.drop_p = drop_p(); o.drop_seed = seed(site);
    o.A = k->at_att; o.lda = HD; o.ta = 0; o.B = wp(TFASR_BP_AT_O_W); o.ldb = d; o.tb = 0; o.D = y; o.ldd = d; o.M = (int)rows; o.N = d; o.K = HD;
    o.bias = fp(TFASR_BP_AT_O_B);
    gemm(o);
  }
  void mhsa_bwd(const void* dy, const void* dy_dropped, void* dx, void* dxd, int site, int next_site) {
    const int d = c->d, H = c->H, dh = c->dh, HD = H * dh, T = c->T, B = c->B, R1 = 2 * T;
    const int Tp = (T + 7) / 8 * 8, R1p = (R1 + 7) / 8 * 8;
    const float scale = 1.f / sqrtf((float)(c->dh_logical > 0 ? c->dh_logical : dh));
    const size_t mark = scratch.off;
    const void* dyd = masked(dy, dy_dropped, rows * d, site);
    void* datt = act(scratch, rows * HD);
    // output projection (din = HD, dout = d)
    {
      G w; w.A = k->at_att; w.lda = HD; w.ta = 1; w.B = dyd; w.ldb = d; w.tb = 0; w.D = gp(TFASR_BP_AT_O_W); w.ldd = d; w.M = HD; w.N = d; w.K = (int)rows;
      w.alpha = c->mhsa_res; w.out_f32 = 1; w.accumulate = 1; w.split_k = split_k(HD, d, rows);
      w.colsum = gp(TFASR_BP_AT_O_B);
      w.side = 1;
      gemm(w);
      G g; g.A = dyd; g.lda = d; g.ta = 0; g.B = wp(TFASR_BP_AT_O_W); g.ldb = d; g.tb = 1; g.D = datt; g.ldd = HD; g.M = (int)rows; g.N = HD; g.K = d;
      g.alpha = c->mhsa_res;
      gemm(g);
    }
    void* dqkv = act(scratch, rows * 3 * HD);
    bool dq_done = false;
    void* dqu = act(scratch, rows * HD);
    void* dqv = act(scratch, rows * HD);
    // fused path: the skewed score gradient never exists in HBM - the query-side kernel forms dq = dqu + dqv and the u / v bias gradients
    // itself, stores the UNSKEWED dS, and tfasr_relattn_dpext accumulates the table gradient from it
    const bool v2 = k->fused;
    void* dpos = act(scratch, v2 ? (long)B * H * T * Tp : (long)B * H * T * R1p);  // v2: the unskewed dS [B,H,T,Tp]
    // caller-owned dS / q + v buffers: they outlive this call and the caller runs tfasr_relattn_dpext itself (on another stream, beside
    // the next block's backward): the table gradient is only needed by the deferred positional-projection gradients
    const bool dpext_deferred = !dry && v2 && io->defer_pos_grad && io->dpext_zero && io->ds_keep && io->qv_keep;
    if (dpext_deferred) dpos = io->ds_keep;
    if (!dry && dpext_deferred) k->left |= 1;
    const void* qv;
    float tail_scale;
    float* dpext = io->dpext_zero;
    if (v2) {
      if (!dpext) {
        dpext = f32(scratch, (long)R1 * HD);
        zero(dpext, (size_t)R1 * HD * 4);
      }
      float* dvec = f32(scratch, 2L * B * H * T);  // rowsum(dO * o) and the bias-row score of every query (two planes)
      void* qu = act(scratch, rows * HD);
      void* qvb = act(scratch, rows * HD);
      if (dpext_deferred) qvb = io->qv_keep;
      if (!dry) {
        // (also writes qu / qvb = q + u / q + v for the two kernels below: no tfasr_bias2_fwd launch)
        chk(tfasr_relattn_fused_bwd_q3(k->at_qkv, fp(TFASR_BP_AT_U), fp(TFASR_BP_AT_V), k->at_pext, io->lengths, k->at_att, datt, k->at_lse, dqkv,
                                       3L * HD, gp(TFASR_BP_AT_U), gp(TFASR_BP_AT_V), dpos, dvec, dpext, qu, qvb, B, H, T, dh, Tp, scale, c->use_mask,
                                       c->chunk_size, c->history_size, c->dtype, s));
        dq_done = true;
        chk(tfasr_relattn_fused_bwd_k(k->at_qkv, qu, qvb, k->at_pext, io->lengths, datt, k->at_lse, dvec, dqkv, B, H, T, dh, scale, c->use_mask,
                                      c->chunk_size, c->history_size, c->dtype, s));
        if (!dpext_deferred) chk(tfasr_relattn_dpext(dpos, qvb, io->lengths, dpext, B, H, T, dh, Tp, c->use_mask, c->dtype, s));
      }
      qv = qvb;
      tail_scale = 1.f;
    } else {
      const char* kk = (const char*)k->at_qkv + (size_t)HD * esz;
      const char* vv = (const char*)k->at_qkv + (size_t)2 * HD * esz;
      char* dk = (char*)dqkv + (size_t)HD * esz;
      char* dv = (char*)dqkv + (size_t)2 * HD * esz;
      void* dprobs = act(scratch, (long)B * H * T * Tp);
      G a; a.A = datt; a.lda = HD; a.ta = 0; a.B = vv; a.ldb = 3 * HD; a.tb = 1; a.D = dprobs; a.ldd = Tp; a.M = T; a.N = T; a.K = dh;
      a.nb1 = B; a.nb2 = H; a.sA1 = (long)T * HD; a.sA2 = dh; a.sB1 = (long)T * 3 * HD; a.sB2 = dh; a.sD1 = (long)H * T * Tp; a.sD2 = (long)T * Tp;
      gemm(a);
      G b; b.A = k->at_probs; b.lda = Tp; b.ta = 1; b.B = datt; b.ldb = HD; b.tb = 0; b.D = dv; b.ldd = 3 * HD; b.M = T; b.N = dh; b.K = T;
      b.nb1 = B; b.nb2 = H; b.sA1 = (long)H * T * Tp; b.sA2 = (long)T * Tp; b.sB1 = (long)T * HD; b.sB2 = dh; b.sD1 = (long)T * 3 * HD; b.sD2 = dh;
      gemm(b);
      if (!dry) chk(tfasr_relattn_softmax_bwd(k->at_probs, dprobs, io->lengths, dprobs, dpos, B, H, T, Tp, R1p, c->use_mask, c->dtype, s));
      G q; q.A = dprobs; q.lda = Tp; q.ta = 0; q.B = kk; q.ldb = 3 * HD; q.tb = 0; q.D = dqu; q.ldd = HD; q.M = T; q.N = dh; q.K = T;
      q.nb1 = B; q.nb2 = H; q.sA1 = (long)H * T * Tp; q.sA2 = (long)T * Tp; q.sB1 = (long)T * 3 * HD; q.sB2 = dh; q.sD1 = (long)T * HD; q.sD2 = dh;
      q.alpha = scale;
      gemm(q);
      G kq; kq.A = dprobs; kq.lda = Tp; kq.ta = 1; kq.B = k->at_qu; kq.ldb = HD; kq.tb = 0; kq.D = dk; kq.ldd = 3 * HD; kq.M = T; kq.N = dh; kq.K = T;
      kq.nb1 = B; kq.nb2 = H; kq.sA1 = (long)H * T * Tp; kq.sA2 = (long)T * Tp; kq.sB1 = (long)T * HD; kq.sB2 = dh; kq.sD1 = (long)T * 3 * HD; kq.sD2 = dh;
      kq.alpha = scale;
      gemm(kq);
      qv = k->at_qv;
      tail_scale = scale;
    }
    if (!v2) {
      // dqv = s * dpos @ pext ; dpext (f32) = s * sum_b dpos^T @ qv
      {
        G a; a.A = dpos; a.lda = R1p; a.ta = 0; a.B = k->at_pext; a.ldb = HD; a.tb = 0; a.D = dqv; a.ldd = HD; a.M = T; a.N = dh; a.K = R1;
        a.nb1 = B; a.nb2 = H; a.sA1 = (long)H * T * R1p; a.sA2 = (long)T * R1p; a.sB1 = 0; a.sB2 = dh; a.sD1 = (long)T * HD; a.sD2 = dh;
        a.alpha = tail_scale;
        gemm(a);
      }
      if (!dpext) {
        dpext = f32(scratch, (long)R1 * HD);
        zero(dpext, (size_t)R1 * HD * 4);
      }
      {
        G a; a.A = dpos; a.lda = R1p; a.ta = 1; a.B = qv; a.ldb = HD; a.tb = 0; a.D = dpext; a.ldd = HD; a.M = R1; a.N = dh; a.K = T;
        a.nb1 = B; a.nb2 = H; a.sA1 = (long)H * T * R1p; a.sA2 = (long)T * R1p; a.sB1 = (long)T * HD; a.sB2 = dh; a.sD1 = 0; a.sD2 = dh;
        a.alpha = tail_scale; a.out_f32 = 1; a.accumulate = 1;
        gemm(a);
      }
    }
    if (!dry && !dq_done) chk(tfasr_bias2_bwd(dqu, dqv, dqkv, 3 * HD, gp(TFASR_BP_AT_U), gp(TFASR_BP_AT_V), rows, HD, c->dtype, s));
    // positional projection: gWpos += pe^T dpext ; gbpos += colsum(dpext) - unless the caller takes them for all blocks at once
    // (io->defer_pos_grad: dpext stays in io->dpext_zero; three launches per block leave the chain)
    const bool pos_deferred = io->defer_pos_grad && io->dpext_zero && dpext == io->dpext_zero;
    if (!dry && pos_deferred) k->left |= 4;
    const void* dpext_t = dpext;
    if (c->dtype != TFASR_F32) {
      void* t = act(scratch, (long)R1 * HD);  // (allocated either way: the arena layout does not depend on the option)
      if (!dry && !pos_deferred) chk(tfasr_cast(dpext, t, (long)R1 * HD, TFASR_F32, c->dtype, s));
      dpext_t = t;
    }
    if (!pos_deferred) {
      G a; a.A = P->pe; a.lda = d; a.ta = 1; a.B = dpext_t; a.ldb = HD; a.tb = 0; a.D = gp(TFASR_BP_AT_POS_W); a.ldd = HD; a.M = d; a.N = HD; a.K = R1;
      a.out_f32 = 1; a.accumulate = 1;
      a.side = 1;
      gemm(a);
      if (!dry) chk(tfasr_colsum(dpext, HD, gp(TFASR_BP_AT_POS_B), R1, HD, 1.f, TFASR_F32, s));
    }
    void* dln = act(scratch, rows * d);
    if (!dense_ln_bwd(dqkv, k->at_ln, TFASR_BP_AT_QKV_W, TFASR_BP_AT_QKV_B, 3 * HD, k->at_x, TFASR_BP_AT_LN_G, TFASR_BP_AT_LN_B, k->at_mean, k->at_rstd,
                      dy, dx, dxd, next_site)) {
      dense_bwd(dqkv, k->at_ln, TFASR_BP_AT_QKV_W, TFASR_BP_AT_QKV_B, d, 3 * HD, dln);
      ln_bwd(dln, k->at_x, TFASR_BP_AT_LN_G, TFASR_BP_AT_LN_B, k->at_mean, k->at_rstd, dy, dx, dxd, next_site);
    }
    rewind(mark);
  }

  // ------------------------------------------------------------------------------------------ ConvModule
  void conv_fwd_a(const void* x) {
    const int d = c->d;
    k->cv_x = (void*)x;
    k->cv_ln = act(stash, rows * d);
    k->cv_mean = f32(stash, rows);
    k->cv_rstd = f32(stash, rows);
    k->cv_a = act(stash, rows * 2 * d);
    k->cv_g = act(stash, rows * d);
    k->cv_cv = act(stash, rows * d);
    k->cv_fin = f32(stash, 4 * d);
    k->cv_sw = act(stash, rows * d);
    if (c->dw_norm_layer) {  // LayerNormalization variant: its output (the swish input) and row statistics are kept for the backward
      k->cv_y = act(stash, rows * d);
      k->cv_nmean = f32(stash, rows);
      k->cv_nrstd = f32(stash, rows);
    }
    ln_dense(x, TFASR_BP_CV_LN_G, TFASR_BP_CV_LN_B, k->cv_ln, k->cv_mean, k->cv_rstd, TFASR_BP_CV_PW1_W, TFASR_BP_CV_PW1_B, k->cv_a, 2 * d);
    if (!dry) {
      // GLU, depthwise conv and BatchNorm statistics as ONE launch when the statistics have copies (else GLU first, below)
      int glu_fused = TFASR_STATUS_UNSUPPORTED;
      if (c->training && !c->dw_norm_layer && bn_copies() > 1 && glu_in_conv()) {
        if (!(io->prezeroed & 1)) zero(io->bn_stats, ((size_t)(io->bn_stats_copies > 1 ? io->bn_stats_copies : 1) * 2 * d + 1) * 4);
        glu_fused = tfasr_glu_dwconv_fwd_stats(k->cv_a, k->cv_g, fp(TFASR_BP_CV_DW_W), fp(TFASR_BP_CV_DW_B), k->cv_cv, io->bn_stats, bn_copies(), c->B, c->T, d,
                                               c->ksize, c->dtype, s);
        if (glu_fused != TFASR_STATUS_UNSUPPORTED) chk(glu_fused);
      }
      if (glu_fused == TFASR_STATUS_UNSUPPORTED) {
      chk(tfasr_glu_fwd(k->cv_a, k->cv_g, rows, d, c->dtype, s));
      // BatchNorm statistics inside the conv kernel when the caller gave the statistics buffer several copies (io->bn_stats_copies: the
      // workgroups spread their atomics over them; with ONE copy 384 workgroups adding into the same 512 addresses were a serial chain of
      // ~30 ns links - 27.8 us against 15.8 + 11.2 us for the two launches - so a single copy keeps the two launches)
      const int ncp = bn_copies();
      int fused = TFASR_STATUS_UNSUPPORTED;
      if (c->training && !c->dw_norm_layer) {
        if (!(io->prezeroed & 1)) zero(io->bn_stats, ((size_t)(io->bn_stats_copies > 1 ? io->bn_stats_copies : 1) * 2 * d + 1) * 4);
        if (ncp > 1) fused = tfasr_dwconv_fwd_stats(k->cv_g, fp(TFASR_BP_CV_DW_W), fp(TFASR_BP_CV_DW_B), k->cv_cv, io->bn_stats, ncp, c->B, c->T, d, c->ksize, c->dtype, s);
        if (fused != TFASR_STATUS_UNSUPPORTED) chk(fused);
      }
      if (fused == TFASR_STATUS_UNSUPPORTED) {
        chk(tfasr_dwconv_fwd(k->cv_g, fp(TFASR_BP_CV_DW_W), fp(TFASR_BP_CV_DW_B), k->cv_cv, c->B, c->T, d, c->ksize, c->dtype, s));
        if (c->training && !c->dw_norm_layer) chk(tfasr_bn_stats(k->cv_cv, io->bn_stats, rows, d, c->dtype, s));  // (into copy 0; the others stay zero)
      }
      }
    }
  }
  // copies of the BatchNorm statistics in use: the caller's, when the finalize + apply row kernel (the only reader that adds copies up) takes
  // the channel count and the conv kernel the dtype; else 1 (everything goes through copy 0, the other copies stay zero)
  int bn_copies() const {
    return (io->bn_stats_copies > 1 && c->dtype == TFASR_BF16 && tfasr_bn_rows_kernel_ok(c->d)) ? io->bn_stats_copies : 1;
  }
  void conv_fwd_b(void* y, int site) {
    const int d = c->d;
    if (!dry && c->dw_norm_layer) {
      chk(tfasr_layernorm_fwd(k->cv_cv, fp(TFASR_BP_CV_BN_G), fp(TFASR_BP_CV_BN_B), k->cv_y, k->cv_nmean, k->cv_nrstd, rows, d, c->ln_eps, c->dtype, s));
      chk(tfasr_add_act_fwd(k->cv_y, nullptr, k->cv_sw, rows * d, TFASR_ACT_SWISH, c->dtype, s));
    } else if (!dry) {
      // statistics -> coefficients (+ moving statistics) -> normalise + swish in one launch; UNSUPPORTED (channel counts outside the row
      // kernel) -> the two launches
      const int ncp = bn_copies();
      const int fst = tfasr_bn_finalize_apply_fwd_copies(k->cv_cv, c->training ? io->bn_stats : nullptr, ncp, c->training ? (float)(rows * c->world) : 1.f,
                                                  fp(TFASR_BP_CV_BN_G), fp(TFASR_BP_CV_BN_B), k->cv_fin, P->bn_mm, P->bn_mv, c->bn_momentum, c->bn_eps,
                                                  k->cv_sw, rows, d, TFASR_ACT_SWISH, c->training ? 1 : 0, c->dtype, s);
      if (fst == TFASR_STATUS_UNSUPPORTED) {
        if (c->training)
          chk(tfasr_bn_finalize(io->bn_stats, (float)(rows * c->world), fp(TFASR_BP_CV_BN_G), fp(TFASR_BP_CV_BN_B), k->cv_fin, P->bn_mm, P->bn_mv,
                                c->bn_momentum, c->bn_eps, d, 1, s));
        else
          chk(tfasr_bn_finalize(nullptr, 1.f, fp(TFASR_BP_CV_BN_G), fp(TFASR_BP_CV_BN_B), k->cv_fin, P->bn_mm, P->bn_mv, c->bn_momentum, c->bn_eps,
                                d, 0, s));
        chk(tfasr_bn_apply_fwd(k->cv_cv, k->cv_fin, k->cv_sw, rows, d, TFASR_ACT_SWISH, c->dtype, s));
      } else chk(fst);
    }
    G o; o.res = k->cv_x; o.beta = c->conv_res; o.drop_p = drop_p(); o.drop_seed = seed(site);
    o.A = k->cv_sw; o.lda = d; o.ta = 0; o.B = wp(TFASR_BP_CV_PW2_W); o.ldb = d; o.tb = 0; o.D = y; o.ldd = d; o.M = (int)rows; o.N = d; o.K = d;
    o.bias = fp(TFASR_BP_CV_PW2_B);
    gemm(o);
  }
  void conv_bwd_a(const void* dy, const void* dy_dropped, int site) {
    const int d = c->d;
    const void* dyd = masked(dy, dy_dropped, rows * d, site);
    void* dsw = act(scratch, rows * d);
    // BatchNorm backward sums (sum dz, sum dz xhat) in the epilogue of the pointwise conv's data gradient when the caller gave the buffer
    // several copies (the second pass over x and dsw - tfasr_bn_bwd_stats, 20 us - otherwise)
    const int ncp = bn_copies();
    const bool fuse = !c->dw_norm_layer && ncp > 1;
    if (!c->dw_norm_layer && !(io->prezeroed & 2)) zero(io->bn_bstats, (size_t)(io->bn_stats_copies > 1 ? io->bn_stats_copies : 1) * 2 * d * 4);
    bns_done = fuse;
    if (fuse) dense_bwd(dyd, k->cv_sw, TFASR_BP_CV_PW2_W, TFASR_BP_CV_PW2_B, d, d, dsw, c->conv_res, nullptr, 0, 0.f, 0, k->cv_cv, k->cv_fin, io->bn_bstats, ncp);
    else dense_bwd(dyd, k->cv_sw, TFASR_BP_CV_PW2_W, TFASR_BP_CV_PW2_B, d, d, dsw, c->conv_res);
    if (!c->dw_norm_layer && !(bns_done && fuse)) {
      if (!dry) chk(tfasr_bn_bwd_stats(k->cv_cv, dsw, k->cv_fin, io->bn_bstats, rows, d, TFASR_ACT_SWISH, c->dtype, s));
    }
    k->bw_dsw = dsw;
  }
  void conv_bwd_b(const void* dy, void* dx, void* dxd, int next_site) {
    const int d = c->d;
    size_t dwws_bytes = 0;
    tfasr_dwconv_bwd_weight_workspace_size(c->B, c->T, d, c->ksize, &dwws_bytes);
    void* dwws = scratch.get(dwws_bytes);  // per-block partial sums of the depthwise weight gradient
    void* dcv = act(scratch, rows * d);
    // caller-owned buffer for the depthwise conv's output gradient: it outlives this call, and the depthwise WEIGHT gradient (two launches
    // that nothing on the chain waits for) is left to tfasr_block_dwconv_wgrad_all - one launch pair for all blocks of the step
    const bool dw_deferred = !dry && io->dcv_keep && !c->dw_norm_layer && c->dtype == TFASR_BF16;
    if (dw_deferred) { dcv = io->dcv_keep; k->left |= 2; }
    void* dg = act(scratch, rows * d);
    void* da = act(scratch, rows * 2 * d);
    void* dln = act(scratch, rows * d);
    void* dyn = c->dw_norm_layer ? act(scratch, rows * d) : nullptr;
    int bn_fused = TFASR_STATUS_UNSUPPORTED;
    if (!dry) {
      if (c->dw_norm_layer) {
        chk(tfasr_add_act_bwd(k->cv_y, nullptr, k->bw_dsw, dyn, rows * d, TFASR_ACT_SWISH, c->dtype, s));
        chk(tfasr_layernorm_bwd(dyn, k->cv_cv, fp(TFASR_BP_CV_BN_G), k->cv_nmean, k->cv_nrstd, nullptr, dcv, gp(TFASR_BP_CV_BN_G), gp(TFASR_BP_CV_BN_B), rows, d,
                                c->dtype, s));
      } else if (bn_in_conv() && c->dtype == TFASR_BF16 &&
                 (bn_fused = tfasr_bn_dwconv_bwd_data_glu(k->cv_cv, k->bw_dsw, k->cv_fin, io->bn_bstats, bn_copies(), (float)(rows * c->world), gp(TFASR_BP_CV_BN_G),
                                                         gp(TFASR_BP_CV_BN_B), 1.f / (float)c->world, dcv, fp(TFASR_BP_CV_DW_W), k->cv_a, da, c->B, c->T, d, c->ksize,
                                                         c->dtype, s)) != TFASR_STATUS_UNSUPPORTED) {
        chk(bn_fused);  // BatchNorm apply pass + depthwise data gradient + GLU backward as one launch (dcv written for the weight gradient)
      } else {
        // bstats = (sum dz, sum dz xhat) over the GLOBAL batch = the beta / gamma gradients; the flat-gradient all-reduce sums over ranks again
        chk(tfasr_bn_apply_bwd_grads_copies(k->cv_cv, k->bw_dsw, k->cv_fin, io->bn_bstats, bn_copies(), (float)(rows * c->world), dcv, rows, d, TFASR_ACT_SWISH,
                                     gp(TFASR_BP_CV_BN_G), gp(TFASR_BP_CV_BN_B), 1.f / (float)c->world, c->dtype, s));
      }
      if (!dw_deferred)
        chk(tfasr_dwconv_bwd_weight_ws(k->cv_g, dcv, gp(TFASR_BP_CV_DW_W), gp(TFASR_BP_CV_DW_B), c->B, c->T, d, c->ksize, c->dtype, dwws, dwws_bytes, s));
      // depthwise data gradient + GLU backward in one launch when the channel-pair kernel applies
      const int fst = bn_fused == TFASR_STATUS_SUCCESS ? TFASR_STATUS_SUCCESS
                                                        : tfasr_dwconv_bwd_data_glu(dcv, fp(TFASR_BP_CV_DW_W), k->cv_a, da, c->B, c->T, d, c->ksize, c->dtype, s);
      if (fst == TFASR_STATUS_UNSUPPORTED) {
        chk(tfasr_dwconv_bwd_data(dcv, fp(TFASR_BP_CV_DW_W), dg, c->B, c->T, d, c->ksize, c->dtype, s));
        chk(tfasr_glu_bwd(k->cv_a, dg, da, rows, d, c->dtype, s));
      } else chk(fst);
    }
    if (!dense_ln_bwd(da, k->cv_ln, TFASR_BP_CV_PW1_W, TFASR_BP_CV_PW1_B, 2 * d, k->cv_x, TFASR_BP_CV_LN_G, TFASR_BP_CV_LN_B, k->cv_mean, k->cv_rstd, dy,
                      dx, dxd, next_site)) {
      dense_bwd(da, k->cv_ln, TFASR_BP_CV_PW1_W, TFASR_BP_CV_PW1_B, d, 2 * d, dln);
      ln_bwd(dln, k->cv_x, TFASR_BP_CV_LN_G, TFASR_BP_CV_LN_B, k->cv_mean, k->cv_rstd, dy, dx, dxd, next_site);
    }
  }

  // ------------------------------------------------------------------------------------------ block
  // dropout sites inside a block (conformer.py site numbering): ff1 = 0,1; mhsa = 2; conv = 3; ff2 = 4,5
  void forward(int phase) {
    const int d = c->d;
    if (phase & TFASR_PHASE_A) {
      void* x1 = act(stash, rows * d);
      void* x2 = act(stash, rows * d);
      ffm_fwd(0, io->x_in, x1, 0);
      mhsa_fwd(x1, x2, 2);
      conv_fwd_a(x2);
      k->stash_off = stash.off;
    }
    if (phase & TFASR_PHASE_B) {
      stash.off = k->stash_off;
      void* x3 = act(stash, rows * d);
      void* x4 = act(stash, rows * d);
      conv_fwd_b(x3, 3);
      ffm_fwd(1, x3, x4, 4);
      k->ln_x = x4;
      k->ln_mean = f32(stash, rows);
      k->ln_rstd = f32(stash, rows);
      ln_fwd(x4, TFASR_BP_LN_G, TFASR_BP_LN_B, io->x_out, k->ln_mean, k->ln_rstd);
      k->stash_off = stash.off;
    }
  }
  // gradient buffer pair for the next module: rotating (two buffers) normally, fresh ones when weight gradients are deferred
  void next_bufs(bool dr) {
    const int d = c->d;
    if (defer) {
      k->bw_cur = k->bw_nxt; k->bw_curd = k->bw_nxtd;
      k->bw_nxt = act(scratch, rows * d);
      k->bw_nxtd = dr ? act(scratch, rows * d) : nullptr;
    } else {
      void* t = k->bw_cur; k->bw_cur = k->bw_nxt; k->bw_nxt = t; t = k->bw_curd; k->bw_curd = k->bw_nxtd; k->bw_nxtd = t;
    }
  }
  void backward(int phase) {
    const int d = c->d;
    const bool dr = drop_p() > 0.f;
    if (phase & TFASR_PHASE_A) {
      // the arena of this slot may still be read by the weight gradients an earlier block queued on the second stream
      if (!dry && side && (io->wgrad_slot == 1 || io->wgrad_slot == 2)) chk(wgrad_wait(*side, 1 << (io->wgrad_slot - 1), s));
      k->bw_cur = act(scratch, rows * d);
      k->bw_nxt = act(scratch, rows * d);
      k->bw_curd = dr ? act(scratch, rows * d) : nullptr;
      k->bw_nxtd = dr ? act(scratch, rows * d) : nullptr;
      {  // partial-sum buffers of the block's five LayerNorm backward passes (below every module's rewind mark: they live until the fold)
        int nblk = tfasr_layernorm_bwd_part_blocks(rows, d, c->dtype);
        k->ln_nblk = nblk;
        k->ln_nsets = 0;
        k->ln_part = nblk > 0 ? f32(scratch, (long)8 * nblk * 2 * d) : nullptr;  // up to 8 sets (5 LayerNorms + the LayerNorm variant of the depthwise norm)
        // caller-owned buffer: the sums outlive this call and tfasr_block_ln_fold_all folds every block of the step in one launch
        k->ln_ext = !dry && k->ln_part && io->ln_part_ext && io->ln_part_ext_floats >= (size_t)8 * nblk * 2 * d;
        if (k->ln_ext) k->ln_part = io->ln_part_ext;
        if (!dry) k->left = k->ln_ext ? 8 : 0;  // (first statement of a backward that touches it: the other bits are set by the modules)
      }
      ln_bwd(io->dy, k->ln_x, TFASR_BP_LN_G, TFASR_BP_LN_B, k->ln_mean, k->ln_rstd, nullptr, k->bw_cur, k->bw_curd, 5);
      ffm_bwd(1, k->bw_cur, k->bw_curd, k->bw_nxt, k->bw_nxtd, 4, 3);
      next_bufs(dr);
      conv_bwd_a(k->bw_cur, k->bw_curd, 3);
      k->scratch_off = scratch.off;
    }
    if (phase & TFASR_PHASE_B) {
      // deferred weight gradients of phase A are flushed with phase B's when both run in one call; in a split call they were
      // flushed at the end of phase A, and the operands of phase A lie below scratch_off, untouched
      scratch.off = k->scratch_off;
      const size_t mark = scratch.off;
      conv_bwd_b(k->bw_cur, k->bw_nxt, k->bw_nxtd, 2);
      rewind(mark);
      next_bufs(dr);
      mhsa_bwd(k->bw_cur, k->bw_curd, k->bw_nxt, k->bw_nxtd, 2, 1);
      ffm_bwd(0, k->bw_nxt, k->bw_nxtd, io->dx, nullptr, 0, -1);
      if (!dry && k->ln_part && k->ln_nsets > 0 && !k->ln_ext) {
        chk(tfasr_layernorm_bwd_fold(k->ln_part, k->ln_nsets, k->ln_nblk, d, k->ln_dg, k->ln_db, s));
        k->ln_nsets = 0;
      }
    }
    flush_wgrads();
  }
};

int check_args(const tfasr_block_cfg* c, const tfasr_block_params* P, const tfasr_block_io* io, const void* ctx) {
  if (!c || !P || !io || !ctx) return TFASR_STATUS_INVALID_VALUE;
  if (c->B <= 0 || c->T <= 0 || c->d <= 0 || c->H <= 0 || c->dh <= 0 || c->dff <= 0 || c->ksize <= 0 || c->world <= 0) return TFASR_STATUS_INVALID_VALUE;
  if (c->dtype != TFASR_F32 && c->dtype != TFASR_BF16) return TFASR_STATUS_INVALID_VALUE;
  if (!P->flat || !P->shadow || !P->grad || !P->pe) return TFASR_STATUS_INVALID_VALUE;
  if (!c->dw_norm_layer && (!P->bn_mm || !P->bn_mv)) return TFASR_STATUS_INVALID_VALUE;  // (a LayerNormalization depthwise norm has no moving statistics)
  return TFASR_STATUS_SUCCESS;
}

void setup(Ex& e, const tfasr_block_cfg* c, const tfasr_block_params* P, const tfasr_block_io* io, void* ctx, void* stream, bool dry) {
  e.c = c; e.P = P; e.io = io; e.k = (Ctx*)ctx; e.s = (hipStream_t)stream; e.dry = dry; e.st = TFASR_STATUS_SUCCESS;
  e.side = dry ? nullptr : &side_for_device();
  e.defer = c->dtype == TFASR_BF16;  // the block's Dense weight gradients as ONE grouped launch
  e.rows = (long)c->B * c->T;
  e.esz = c->dtype == TFASR_F32 ? 4 : 2;
  e.stash = Arena{dry ? nullptr : (char*)io->stash, 0, dry ? 0 : io->stash_bytes, true, 0};
  e.scratch = Arena{dry ? nullptr : (char*)io->scratch, 0, dry ? 0 : io->scratch_bytes, true, 0};
}

bool use_fused(const tfasr_block_cfg* c) { return c->dtype == TFASR_BF16 && c->dh == 64 && !c->force_unfused; }

}  // namespace

extern "C" size_t tfasr_block_ctx_bytes(void) { return sizeof(Ctx); }

extern "C" int tfasr_block_workspace_sizes(const tfasr_block_cfg* c, size_t* stash_bytes, size_t* fwd_scratch_bytes, size_t* bwd_scratch_bytes) {
  if (!c || !stash_bytes || !fwd_scratch_bytes || !bwd_scratch_bytes) return TFASR_STATUS_INVALID_VALUE;
  // sizing = a dry run of the very same allocation sequence (no launches, null arenas)
  tfasr_block_params P;
  memset(&P, 0, sizeof(P));
  tfasr_block_io io;
  memset(&io, 0, sizeof(io));
  Ctx k;
  memset(&k, 0, sizeof(k));
  k.fused = use_fused(c);
  Ex e;
  setup(e, c, &P, &io, &k, nullptr, true);
  e.forward(TFASR_PHASE_A | TFASR_PHASE_B);
  *stash_bytes = e.stash.peak + 256;
  *fwd_scratch_bytes = e.scratch.peak + 256;
  Ex b;
  setup(b, c, &P, &io, &k, nullptr, true);
  b.backward(TFASR_PHASE_A | TFASR_PHASE_B);
  *bwd_scratch_bytes = b.scratch.peak + 256;
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_block_fwd(const tfasr_block_cfg* c, const tfasr_block_params* P, const tfasr_block_io* io, void* ctx, int phase,
                               void* stream) {
  int st = check_args(c, P, io, ctx);
  if (st != TFASR_STATUS_SUCCESS) return st;
  if (!io->x_in || !io->x_out || !io->stash || (c->training && !c->dw_norm_layer && !io->bn_stats) || !(phase & (TFASR_PHASE_A | TFASR_PHASE_B)))
    return TFASR_STATUS_INVALID_VALUE;
  Ex e;
  setup(e, c, P, io, ctx, stream, false);
  if (phase & TFASR_PHASE_A) {
    memset(e.k, 0, sizeof(Ctx));
    e.k->fused = use_fused(c);
    e.k->drop_epoch = c->drop_epoch;
  }
  e.forward(phase);
  if (!e.stash.ok || !e.scratch.ok) return TFASR_STATUS_INVALID_VALUE;  // arena too small
  return e.st;
}

extern "C" int tfasr_block_side_stream(void** stream) {
  if (!stream) return TFASR_STATUS_INVALID_VALUE;
  Side& sd = side_for_device();
  if (!wgrad_stream_ready(sd)) return TFASR_STATUS_EXECUTION_FAILED;
  *stream = (void*)sd.sw;
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_block_wgrad_probe(int enable) {
  Side& sd = side_for_device();
  sd.probe = enable != 0;
  sd.pused = 0;
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_block_wgrad_probe_read(float* total_ms, int* launches) {
  if (!total_ms || !launches) return TFASR_STATUS_INVALID_VALUE;
  Side& sd = side_for_device();
  double tot = 0.0;
  int n = 0;
  for (size_t i = 0; i + 1 < sd.pused; i += 2) {
    float ms = 0.f;
    if (hipEventSynchronize(sd.pev[i + 1]) != hipSuccess || hipEventElapsedTime(&ms, sd.pev[i], sd.pev[i + 1]) != hipSuccess) return TFASR_STATUS_EXECUTION_FAILED;
    tot += ms;
    ++n;
  }
  *total_ms = (float)tot;
  *launches = n;
  sd.pused = 0;
  return TFASR_STATUS_SUCCESS;
}

extern "C" int tfasr_block_bwd_left(const void* ctx) { return ctx ? ((const Ctx*)ctx)->left : 0; }

extern "C" int tfasr_block_wgrad_join(int slot_mask, void* stream) {
  return wgrad_wait(side_for_device(), slot_mask, (hipStream_t)stream);
}

extern "C" int tfasr_block_bwd(const tfasr_block_cfg* c, const tfasr_block_params* P, const tfasr_block_io* io, void* ctx, int phase,
                               void* stream) {
  int st = check_args(c, P, io, ctx);
  if (st != TFASR_STATUS_SUCCESS) return st;
  if (!io->dy || !io->dx || !io->scratch || (!c->dw_norm_layer && !io->bn_bstats) || !(phase & (TFASR_PHASE_A | TFASR_PHASE_B))) return TFASR_STATUS_INVALID_VALUE;
  Ex e;
  setup(e, c, P, io, ctx, stream, false);
  e.backward(phase);
  if (!e.scratch.ok) return TFASR_STATUS_INVALID_VALUE;
  return e.st;
}

// One fold launch for every block whose backward left its LayerNorm partial sums in a caller-owned buffer (tfasr_block_io.ln_part_ext).
extern "C" int tfasr_block_ln_fold_all(void* const* ctx, int n, int d, void* stream) {
  if (!ctx || n <= 0 || d <= 0) return TFASR_STATUS_INVALID_VALUE;
  const float* part[128];
  float* dg[128];
  float* db[128];
  int ns = 0, nblk = 0;
  for (int i = 0; i < n; ++i) {
    Ctx* k = (Ctx*)ctx[i];
    if (!k) return TFASR_STATUS_INVALID_VALUE;
    if (!k->ln_ext || !k->ln_part || k->ln_nsets <= 0) continue;
    if (nblk == 0) nblk = k->ln_nblk;
    if (k->ln_nblk != nblk) return TFASR_STATUS_INVALID_VALUE;  // (one batch shape per step: every block has the same slot count)
    for (int q = 0; q < k->ln_nsets; ++q) {
      if (ns == 128) {
        const int st = tfasr_layernorm_bwd_fold_sets(part, ns, nblk, d, dg, db, stream);
        if (st != TFASR_STATUS_SUCCESS) return st;
        ns = 0;
      }
      part[ns] = k->ln_part + (size_t)q * nblk * 2 * d;
      dg[ns] = k->ln_dg[q];
      db[ns] = k->ln_db[q];
      ++ns;
    }
    k->ln_nsets = 0;
  }
  if (ns > 0) return tfasr_layernorm_bwd_fold_sets(part, ns, nblk, d, dg, db, stream);
  return TFASR_STATUS_SUCCESS;
}

int tfasr_dwconv_wgrad_many_try(const void* const* x, const void* const* dy, float* const* dw, float* const* dbias, int n, int B, int T, int C, int K,
                                float* ws, size_t ws_bytes, hipStream_t s);  // dwconv.hip

// The depthwise-conv weight / bias gradients of `n` blocks whose backward ran with io->dcv_keep: x = the block's saved GLU output (ctx),
// dy = dcv[i] (the caller's buffers), one tile launch + one reduce launch for all of them (fallback: one pair per block).
extern "C" int tfasr_block_dwconv_wgrad_all(const tfasr_block_cfg* c, const tfasr_block_params* const* params, void* const* ctx, const void* const* dcv,
                                            int n, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !params || !ctx || !dcv || n <= 0 || n > 32) return TFASR_STATUS_INVALID_VALUE;
  const void* x[32];
  float* dw[32];
  float* db[32];
  for (int i = 0; i < n; ++i) {
    if (!params[i] || !ctx[i] || !dcv[i] || !params[i]->grad) return TFASR_STATUS_INVALID_VALUE;
    x[i] = ((const Ctx*)ctx[i])->cv_g;
    dw[i] = params[i]->grad + params[i]->off[TFASR_BP_CV_DW_W];
    db[i] = params[i]->grad + params[i]->off[TFASR_BP_CV_DW_B];
  }
  if (c->dtype == TFASR_BF16) {
    const int st = tfasr_dwconv_wgrad_many_try(x, dcv, dw, db, n, c->B, c->T, c->d, c->ksize, (float*)workspace, workspace_bytes, (hipStream_t)stream);
    if (st != TFASR_STATUS_UNSUPPORTED) return st;
  }
  for (int i = 0; i < n; ++i) {
    const int st = tfasr_dwconv_bwd_weight(x[i], dcv[i], dw[i], db[i], c->B, c->T, c->d, c->ksize, c->dtype, stream);
    if (st != TFASR_STATUS_SUCCESS) return st;
  }
  return TFASR_STATUS_SUCCESS;
}
