"""Conformer-Transducer on MI355X: explicit forward / backward over the HIP kernels of libtfasr_hip.so.

Mirrors the reference's model surface for this path:
    tensorflow_asr.models.transducer.conformer.Conformer        (models/transducer/conformer.py:22-143)
    Transducer.call / call_next / recognize*                    (models/transducer/base_transducer.py:427-712)
    BaseModel.train_step / _train_step / _apply_gradients       (models/base_model.py:149-198)
but is NOT a Keras graph: every module has a hand-written backward that reuses the forward's saved tensors, the
residual additions / biases / activations are fused into GEMM epilogues, and the whole step is a fixed sequence of
C-ABI kernel launches on one HIP stream (prediction network on a second stream, overlapped with the encoder).
PyTorch only owns device memory and streams here.  There is no CPU / eager fallback.
"""
import math
import os

import numpy as np
import torch

from . import kernels as K
from .base_model import BaseModel
from .kernels import ACT_NONE, ACT_SWISH, ACT_TANH_OUT
from .params import ParamStore
from .schemas import PredictInput, PredictOutput, TrainData, TrainInput, TrainOutput


class SingleProcess:
    """Collective hooks for one GPU (tensorflowasr_amd.dp.DataParallel implements them over RCCL)."""

    world = 1
    rank = 0

    def allreduce_stats_(self, t):  # sync-BN statistics (sum over replicas)
        return t

    def set_reduce(self, on):  # gradient accumulation: only the apply micro-step exchanges gradients
        pass

    def grads_ready(self, lo, hi):  # gradient slice [lo, hi) of the flat buffer is final
        pass

    def finish_grads(self):
        pass


def _split_k(M, N, Kd):
    tiles = -(-M // 128) * -(-N // 128)
    if tiles >= 512 or Kd <= 2048:
        return 1
    v = int(max(1, min(-(-1024 // tiles), Kd // 1024, 64)))
    return (v + 4) // 8 * 8 if v >= 12 else v  # whole k-slices per XCD (gemm_fast.hip split-K mapping needs split % 8 == 0)


class ConformerTransducer(BaseModel):
    def __init__(self, cfg, device=None, dtype=torch.bfloat16, seed=0, dp=None):
        if not torch.cuda.is_available():
            raise K._lib.TfasrError("ConformerTransducer needs an MI355X (HIP) device; there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dtype = dtype
        # bf16 models store attention heads narrower than 64 zero-padded to 64 (ParamStore.__init__): the reference's shipped head sizes
        # (36, 44) then run on the fused LDS-staged attention kernels, results unchanged
        pad = (dtype == torch.bfloat16 and cfg.head_size < 64 and getattr(cfg, "encoder", "conformer") == "conformer"
               and os.environ.get("TFASR_HEAD_PAD", "1") != "0")
        # ... and the convolutional subsampling's channels padded to a multiple of 64 (144 -> 192): conv2 then runs as K-segmented GEMMs
        # over the space-to-depth layout with conv1 + BatchNorm recomputed, like the 256-filter model (Conformer-S: 14.3 -> 13.6 ms/step)
        # (a LayerNormalization in the subsampling - `norms: layer`, small-streaming.yml.j2 - normalises over the channel axis: no padding)
        cpad = (dtype == torch.bfloat16 and cfg.filters % 64 != 0 and getattr(cfg, "encoder", "conformer") == "conformer"
                and getattr(cfg, "sub_norm", "batch") == "batch" and os.environ.get("TFASR_FILTER_PAD", "1") != "0")
        self.ps = ParamStore(cfg, self.device, dtype, seed, head_phys=64 if pad else None, filt_phys=-(-cfg.filters // 64) * 64 if cpad else None)
        self.dp = dp or SingleProcess()
        self.blank = cfg.blank
        self.time_reduction_factor = cfg.time_reduction_factor
        self.step = 0
        self._consts = {}
        self._bn1_sums_in_gemm = True  # subsampling BatchNorm1 backward sums in the linear layer's data gradient
        self._bn_stats_copies = 8  # ConvModule BatchNorm statistics inside the depthwise conv, atomics spread over this many copies; 1: the two launches (tests)
        self._conv1_gram = True  # conv1 / BatchNorm0 sums through the patch Gram matrix (one backward pass); False: the two-pass kernels (tests)
        # SpecAugment draws and dropout masks are per replica (MirroredStrategy draws independent randomness on every
        # replica); the parameter initialisation seed above is shared by all ranks
        self._rng = np.random.default_rng([seed + 1000, int(self.dp.rank)])
        # Streams = hardware queues, and on this chip the step time depends on how many are live (with eight queues a
        # data-parallel rank ran 38 instead of 23 ms).  HIP never lets streams of DIFFERENT priority share a queue, so the three streams of the
        # step sit on three priority levels - the encoder chain on the default stream, the prediction network (hundreds of tiny launches the
        # joint network waits for) on a HIGH-priority stream, everything nothing on the chain waits for on the executor's LOW-priority stream -
        # which keeps them on separate queues whatever GPU_MAX_HW_QUEUES is and whatever streams RCCL adds.
        prio = -1
        self.pred_stream = torch.cuda.Stream(device=self.device, priority=prio)
        self.use_pred_stream = os.environ.get("TFASR_NO_PRED_STREAM", "0") != "1"
        # probe (bench.py --dp-hooks): take the world > 1 route of the block executor - two phases per block and direction around the sync-BN
        # all-reduce - with a one-rank group, so everything of a multi-GPU step except the wire time can be measured on one GPU
        self._dp_force_split = os.environ.get("TFASR_DP_FORCE_SPLIT", "0") == "1" and not isinstance(self.dp, SingleProcess)
        # prediction-network recurrence: the persistent one-launch kernels are faster in isolation but hold P/16 CUs for ~2 ms per step;
        # beside the encoder on the other stream the per-step kernels make the whole step 0.2 ms faster (measured on M and S), so:
        # persistent only when the prediction network has the device to itself.  TFASR_LSTM_PERSIST=0/1 in the environment overrides.
        self._lstm_persist_auto = os.environ.get("TFASR_LSTM_PERSIST") is None
        self.optimizer = dict(beta1=0.9, beta2=0.98, eps=1e-9, weight_decay=1e-6, schedule=dict(
            dmodel=cfg.dmodel, warmup_steps=10000, scale=2.0, max_lr=0.05 / math.sqrt(cfg.dmodel)))
        self.ga_steps = 1
        self.prefetched_inputs = False  # True: every batch handed to train_step is complete in HBM (see _forward: early front end)
        self._ga_count = 0
        self._drop_epoch = 0  # bumped once per forward pass so every step draws fresh dropout masks
        # one native call per Conformer block (csrc/block.hip) instead of ~70 per-kernel calls from Python
        self.native_blocks = os.environ.get("TFASR_NATIVE_BLOCK", "1") != "0"
        # grouped weight gradients of a block on the executor's second stream (tfasr_block_io.wgrad_slot), beside the next block's
        # backward.  Round 2 measured it SLOWER (29.06 vs 28.45 ms/step: the 512-workgroup group launch took CUs from the dependent chain);
        # with the round-4 chain (fused FFModule forward, one-tile GEMMs, hoisted launches) it is FASTER: 22.18 vs 22.45 ms/step, same box,
        # three interleaved pairs - the chain's kernels now leave more of the chip idle than the group takes.  TFASR_WGRAD_STREAM=0: in line.
        # Round 4 kept this to ONE GPU: with a process group the second stream TOGETHER with the auxiliary stream of the hoists below made the
        # step 38 instead of 22.7 ms.  Round 5 found the cause - the number of live hardware queues (GPU_MAX_HW_QUEUES 2 / 3 / 6: 23.1 ms,
        # 8 / 12: 39 ms, profiles/r05_dp_queue_sweep.txt) - and merged the two into one low-priority stream: a data-parallel rank runs it too.
        self.wgrad_stream = {"0": False, "1": True}.get(os.environ.get("TFASR_WGRAD_STREAM"), None)
        self._wgrad_keep = []
        self._blk_params, self._blk_sizes = {}, {}
        self._zero_pool = {}
        # Launches that do not belong to the blocks' dependent chain leave it (a kernel boundary costs 2.65 us on this chip, tools/hwprobe/
        # anyorder_test, and hipExtAnyOrderLaunch is a no-op on gfx9): every block's positional projection pe @ Wpos + bpos is computed on a
        # third stream while the subsampling runs (tfasr_block_io.pext_pre), and the projections' gradients and the LayerNorm gamma / beta
        # folds of all blocks run once after the last block's backward (defer_pos_grad, ln_part_ext).  With a data-parallel group a block's
        # gradient bucket is released right behind the block: the deferred variables therefore live in a region of their own behind the last
        # block (ParamStore.defer_lo / defer_hi), released after the deferred launches.  TFASR_BLOCK_HOIST=0: off.
        self.block_hoist = {"0": False, "1": True}.get(os.environ.get("TFASR_BLOCK_HOIST"), None)  # None: on
        # ... and with the gradients deferred, the kernel that accumulates a block's table gradient from dS (tfasr_relattn_dpext, 37 us per
        # block, only the deferred products wait for it) runs on the auxiliary stream beside the next block's backward (attribute False: in line).
        self.dpext_aux = True
        self.joint_wgrad_aux = False  # (measured no gain: the vocabulary weight gradient on the auxiliary stream)
        self._aux_pending = False
        # the auxiliary stream IS the block executor's second stream (one queue for the grouped weight gradients, the positional tables ahead
        # of the chain and the table gradients beside the next block; a fourth stream of its own measured slower, profiles/r05_dp_queue_sweep.txt)
        if self.device.type != "cuda":
            self.aux_stream = None
        else:
            with torch.cuda.device(self.device):
                self.aux_stream = torch.cuda.ExternalStream(K.block_side_stream(), device=self.device)
        self._hoisted = {}
        self.fuse_joint_stats = True
        # Joint + loss WITHOUT materialised lattice logits (SURVEY section 7 step 8 / 8(d) "report both"): the projection emits only the
        # log-softmax statistics, the gradient pass re-computes the logit tile and turns it into the loss gradient in its epilogue
        # (tfasr_gemm_args.rgrad_coef).  Trades two passes over the [cells, V] tensor for one more vocabulary product: measured SLOWER
        # than the materialised route on MI355X (bench.py reports both), so opt-in: TFASR_JOINT_RECOMPUTE=1.
        self.joint_recompute = os.environ.get("TFASR_JOINT_RECOMPUTE", "0") == "1"
        self._after_encoder = "pred/emb" if cfg.head == "transducer" else "dec/logits/w"  # first regularised variable after the encoder
        self.timers = None  # optional dict name -> list[(start_event, end_event)] filled by bench.py
        self.timer_work = {}
        self.time_sections = False  # per-module section timers (bench.py TFASR_BENCH_SECTIONS) need the per-kernel Python path
        # Greedy search: the reference's tokens come from its f32 CPU path and the north star asks for bit-exact token indices, which a
        # bf16 encoder cannot promise (near-tied arg-max decisions flip).  Inference therefore runs on the f32 master weights with the
        # exact-f32 MFMA kernels by default, whatever the training storage type ("bf16" = the training kernels, faster, not token-exact)
        self.decode_precision = os.environ.get("TFASR_DECODE_PRECISION", "f32")
        self.decode_fused = True  # the fused search step (4 launches per iteration, queued from C); False: the per-kernel loop (tests)
        self._twin = None

    # =================================================================================== constants
    def _frontend_consts(self):
        if "fe" not in self._consts:
            c = self.cfg
            n = c.frame_length
            window = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)  # periodic Hann
            melw = _mel_weight_matrix(c.num_feature_bins, c.nfft // 2 + 1, c.sample_rate, c.lower_edge_hertz, c.upper_edge_hertz)
            band = np.zeros((melw.shape[1], 2), np.int32)
            for m in range(melw.shape[1]):
                nz = np.nonzero(melw[:, m])[0]
                band[m] = (nz[0], nz[-1]) if len(nz) else (0, -1)
            dev = self.device
            self._consts["fe"] = (torch.from_numpy(window).to(dev), torch.from_numpy(melw).to(dev), torch.from_numpy(band).to(dev))
        return self._consts["fe"]

    def _pe_ext(self, T):
        """Relative sinusoid table for length T: rows r=0..2T-2 <-> positions T-1..-(T-1) (positional_encoding.py:119-121,
        31-52 interleaved), plus one zero row (the projection of a masked-out encoding row = the bias)."""
        key = ("pe", T)
        if key not in self._consts:
            d = self.cfg.dmodel
            pos = np.concatenate([np.arange(T - 1, 0, -1), np.arange(0, -T, -1)]).astype(np.float32)
            ts = np.power(np.float32(1.0 / 10000.0), (2 * (np.arange(d, dtype=np.float32) // 2)) / np.float32(d)).astype(np.float32)
            ang = pos[:, None] * ts[None, :]
            pe = np.where((np.arange(d) % 2 == 1)[None, :], np.cos(ang), np.sin(ang)).astype(np.float32)
            pe = np.concatenate([pe, np.zeros((1, d), np.float32)], 0)
            self._consts[key] = torch.from_numpy(pe).to(self.device).to(self.dtype).contiguous()
        return self._consts[key]

    # =================================================================================== dropout bookkeeping
    def _drop(self, site, training):
        """(rate, seed) of dropout site `site` for the current step; masks are a pure function of (seed, element index), so
        the backward regenerates them instead of storing them (keras Dropout sites: conformer.py:80-87,197,358,681)."""
        p = float(self.cfg.dropout) if training else 0.0
        if p <= 0.0:
            return 0.0, 0
        return p, (self._drop_epoch_eff() * 8192 + site) & 0x7FFFFFFFFFFF

    def _drop_epoch_eff(self):
        """Mask epoch with the data-parallel rank folded into bits 40..46 of the seed (epoch * 8192 + site stays below 2^40
        for 2^27 forward passes): replicas draw different dropout masks; the native block executor forms the same seed."""
        return self._drop_epoch + ((int(self.dp.rank) & 0x7F) << 27)

    def _mask_grad(self, dy, drop):
        return K.dropout(dy, drop[0], drop[1]) if drop[0] > 0.0 else dy

    # =================================================================================== dense helpers
    def _dense_bwd(self, dy, x, wname, bname, alpha=1.0, dact_z=None, dact=ACT_NONE, need_dx=True, drop=(0.0, 0)):
        """dx = alpha * (dy @ W^T) [* dact'(z)];  gW += alpha * x^T dy;  gb += alpha * colsum(dy)."""
        ps = self.ps
        W = ps.w2d(wname)
        rows = dy.shape[0]
        din, dout = W.shape
        # weight gradient; the bias gradient (column sums of dy) rides in the same launch
        K.gemm(x, dy, ps.g2d(wname), din, dout, rows, x.stride(0), dy.stride(0), dout, trans_a=True, accumulate=True,
               split_k=_split_k(din, dout, rows), alpha=alpha, colsum=ps.g(bname) if bname is not None else None)
        if not need_dx:
            return None
        return K.matmul(dy, W, trans_b=True, alpha=alpha, dact_z=dact_z, dact=dact, drop_p=drop[0], drop_seed=drop[1])

    def _h2d(self, x, dtype=torch.int32):
        """Small host array -> device through PINNED staging: a pageable hipMemcpyAsync makes the host wait until the stream
        has drained up to the copy, which stops the host from queueing kernels ahead of the GPU several times per step."""
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        t = t.to(dtype)
        if t.device.type == "cpu":
            if self.device.type == "cuda":
                t = t.pin_memory()
            t = t.to(self.device, non_blocking=True)
        return t

    # =================================================================================== frontend
    def frontend(self, signals, signals_length, training=False, masks=None):
        """FeatureExtraction.call (feature_extraction.py:255-303) -> features [B,T0,F] (compute dtype), feature lengths (host)."""
        c = self.cfg
        window, melw, band = self._frontend_consts()
        feats = K.logmel(signals, window, melw, band, c.frame_step, c.nfft, c.preemphasis, c.epsilon, self.dtype)
        flen = [-(-int(n) // c.frame_step) for n in signals_length]
        if training:
            if masks is None:
                masks = self.draw_specaugment(flen)
            fm, tm = masks
            if fm is not None or tm is not None:
                K.specaugment(feats, None if fm is None else self._h2d(fm), None if tm is None else self._h2d(tm), 0.0)
        return feats, flen

    def draw_specaugment(self, flen):
        """Random draws of FreqMasking/TimeMasking.augment (specaugment.py:72-77,122-131; freq before time:
        augmentation.py:95; TimeMasking ignores mask_factor)."""
        c = self.cfg
        fcfg, tcfg = c.freq_masking, c.time_masking
        B = len(flen)
        rng = self._rng
        nf = fcfg.get("num_masks", 0) if fcfg else 0
        nt = tcfg.get("num_masks", 0) if tcfg else 0
        fm = np.zeros((B, nf, 2), np.int32) if nf > 0 else None
        tm = np.zeros((B, nt, 2), np.int32) if nt > 0 else None
        nb = c.num_feature_bins
        # the reference's consumption order: utterance by utterance (tf.map_fn, augmentation.py:78-90), frequency masks before time
        # masks, and per mask (prob, width, start) are drawn unconditionally and multiplied by do_apply afterwards
        # (hot host loop, ~1 000 generator calls per batch: lookups hoisted; rng.random() is uniform(0, 1) - same stream, same values)
        rnd, rint = rng.random, rng.integers
        fprob = fcfg.get("prob", 1.0) if fcfg else 1.0
        fmax = max(1, int(fcfg["mask_factor"])) if nf > 0 else 1
        tprob = tcfg.get("prob", 1.0) if tcfg else 1.0
        tup = np.float32(tcfg.get("p_upperbound", 1.0)) if tcfg else np.float32(1.0)
        for b in range(B):
            ln = int(flen[b])
            for k in range(nf):
                do = 1 if rnd() <= fprob else 0
                f = do * min(int(rint(0, fmax)), nb)
                fm[b, k] = (do * int(rint(0, max(1, nb - f))), f)
            if nt > 0:
                Tb = max(1, int(math.floor(np.float32(ln) * tup)))
            for k in range(nt):
                do = 1 if rnd() <= tprob else 0
                # tf.random.uniform(maxval=0) is an InvalidArgumentError in the reference (fewer than 20 frames at 0.05):
                # a zero-width mask here
                t = do * min(int(rint(0, Tb)), ln)
                tm[b, k] = (do * int(rint(0, max(1, ln - t))), t)
        return (None if fm is None else torch.from_numpy(fm)), (None if tm is None else torch.from_numpy(tm))

    # =================================================================================== batch norm
    def _zeros_f32(self, n):
        """n zeroed f32 accumulators for one launch (BN sums): slices of an arena cleared by ONE fill per ~1 MB instead of one fill
        launch per use (the ContextNet step had 300 of them).  An exhausted arena is replaced, never re-cleared, so slices that are
        still referenced stay valid."""
        n = (n + 63) // 64 * 64
        size = max(1 << 18, n)
        arena, cur = self._consts.get("zero_arena", (None, 0))
        if arena is None or cur + n > arena.numel():
            arena, cur = torch.zeros(size, dtype=torch.float32, device=self.device), 0
        self._consts["zero_arena"] = (arena, cur + n)
        return arena[cur:cur + n]

    def _bn_fwd(self, x2d, name, training, act, rows=None, y=None):
        """rows: number of REAL rows when x2d carries exactly-zero padding rows (haloed layouts): zeros change neither sum."""
        ps = self.ps
        C = x2d.shape[1]
        fin = torch.empty(4 * C, dtype=torch.float32, device=self.device)
        if training:
            stats = self._zeros_f32(2 * C + 1)[:2 * C + 1]
            K.bn_stats(x2d, stats)
            count = (x2d.shape[0] if rows is None else rows) * self.dp.world
            self.dp.allreduce_stats_(stats[:2 * C])
        else:
            stats, count = None, x2d.shape[0]
        # statistics -> coefficients (+ moving statistics) -> normalise + activation in ONE launch where the row kernel applies
        got = K.bn_finalize_apply_fwd(x2d, stats, count if training else 1, ps.p(name + "/g"), ps.p(name + "/b"), fin, ps.state[name + "/mm"],
                                      ps.state[name + "/mv"], act, y=y, training=training)
        if got is None:
            K.bn_finalize(stats, count if training else 1, ps.p(name + "/g"), ps.p(name + "/b"), fin, ps.state[name + "/mm"], ps.state[name + "/mv"],
                          0.99, 1e-3, training)
            got = K.bn_apply_fwd(x2d, fin, act, y=y)
        return got, (fin, count)

    def _bn_bwd(self, x2d, dy2d, name, saved, act, dx=None):
        fin, count = saved
        ps = self.ps
        C = x2d.shape[1]
        bstats = self._zeros_f32(2 * C)[:2 * C]
        K.bn_bwd_stats(x2d, dy2d, fin, bstats, act)
        self.dp.allreduce_stats_(bstats)
        # bstats = (sum dz, sum dz*xhat) over the GLOBAL batch = the beta / gamma gradients; the flat-gradient all-reduce sums over ranks again
        return K.bn_apply_bwd(x2d, dy2d, fin, bstats, count, act, dx=dx, dgamma=ps.g(name + "/g"), dbeta=ps.g(name + "/b"), grad_scale=1.0 / self.dp.world)

    # =================================================================================== subsampling
    # ---- conv2 without a patch matrix: haloed space-to-depth layout (csrc/conv2d.hip, include/tfasr_hip.h) -----------------
    _SEG = [(kh, kw) for kh in range(3) for kw in range(3)]

    def _s2d_enabled(self):
        if os.environ.get("TFASR_CONV2_IM2COL", "0") == "1" or getattr(self.cfg, "sub_norm", "batch") == "layer":
            return False  # the recomputed conv1 + BatchNorm kernels of the space-to-depth route have no LayerNormalization variant
        # bf16: the K-segmented MFMA GEMM needs whole 64-wide slabs per tap; f32 (parity mode) issues one product per tap
        return self.dtype == torch.float32 or self.ps.filt_phys % 64 == 0

    def _salloc(self, rows, width, slack, tag=None):
        """[rows, width] buffer with `slack` zeroed rows before and after it (tap shifts reach outside the first / last sample).
        With a `tag` the buffer is kept for the next step (one live forward / backward pair per model: the kernels only ever READ the
        slack rows, so they are cleared once, not by two fill launches per buffer and step)."""
        key = ("salloc", tag, rows, width, slack)
        if tag is not None and slack > 0 and key in self._consts:
            full = self._consts[key]
        else:
            full = torch.empty((rows + 2 * slack) * width, dtype=self.dtype, device=self.device)
            if slack > 0:
                full[:slack * width].zero_()
                full[(slack + rows) * width:].zero_()
                if tag is not None:
                    # a few batch shapes are kept per tag (bucketed batches alternate between shapes: with one buffer per tag every step
                    # re-allocated and re-cleared all of them); the oldest goes when a fifth shape turns up
                    old = [k for k in self._consts if isinstance(k, tuple) and k[:2] == ("salloc", tag)]
                    for k in old[:max(0, len(old) - 3)]:
                        del self._consts[k]
                    self._consts[key] = full
        return full, full[slack * width:(slack + rows) * width].view(rows, width)

    def _seg_tables(self, F2, C):
        """per tap (kh, kw): (row shift, parity block) of its input in the S layout + device offset tables for the segmented GEMMs."""
        key = ("seg", F2, C)
        if key not in self._consts:
            shift, blk = [], []
            for kh, kw in self._SEG:
                dt, pt = (-1, kh) if kh < 2 else (0, 0)
                df, pf = (-1, kw) if kw < 2 else (0, 0)
                shift.append(dt * (F2 + 1) + df)
                blk.append(pt * 2 + pf)
            dev = self.device
            # K order of the segmented GEMMs: 64-channel segments, channel chunk outermost and the taps of one parity block next to each
            # other, so that consecutive slabs read the SAME rows of the S tensor shifted by one slot (taps of a parity block differ by a
            # row shift only) while they are still in L2 - in tap-major order (a whole 256-channel tap at a time) every tap re-fetched
            # its rows: 1.48 GB fetched for a 0.51 GB input (rocprofv3 FETCH_SIZE), L2 hit rate 0.49
            nck = C // 64 if C % 64 == 0 else 1
            ck = C // nck
            order = sorted(range(9), key=lambda i: (blk[i], i))
            fwd_a = torch.tensor([shift[i] * 4 * C + blk[i] * C + cc * ck for cc in range(nck) for i in order], dtype=torch.int64, device=dev)
            fwd_b = torch.tensor([(i * C + cc * ck) * C for cc in range(nck) for i in order], dtype=torch.int64, device=dev)
            dgrad = {}
            for b in range(4):
                segs = [i for i in range(9) if blk[i] == b]
                dgrad[b] = (segs, torch.tensor([-shift[i] * C + cc * ck for cc in range(nck) for i in segs], dtype=torch.int64, device=dev),
                            torch.tensor([i * C * C + cc * ck for cc in range(nck) for i in segs], dtype=torch.int64, device=dev))
            self._consts[key] = (shift, blk, (fwd_a, fwd_b, ck), dgrad)
        return self._consts[key]

    def _subsampling_fwd_s2d(self, feats, flen, training, ctx):
        ps, c = self.ps, self.cfg
        B, T0, F0 = feats.shape
        C = ps.filt_phys
        T1, F1 = (T0 + 1) // 2, (F0 + 1) // 2
        T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
        rows, slack = B * (T2 + 1) * (F2 + 1), F2 + 2
        shift, blk, fwd_a, _ = self._seg_tables(F2, C)
        # conv1 + BatchNorm + swish straight into the S layout; conv1's own output is never stored (recomputed from the feature map
        # by the statistics / apply / backward kernels: csrc/conv2d.hip conv1_bn_kernel)
        w0, b0 = ps.p("enc/sub/conv0/w"), ps.p("enc/sub/conv0/b")
        fin0 = torch.empty(4 * C, dtype=torch.float32, device=self.device)
        nm = "enc/sub/bn0"
        gram = None
        if training:
            stats = self._zeros_f32(2 * C + 1)[:2 * C + 1]
            if self._conv1_gram:
                # conv1 has one input channel: its BatchNorm's sums over positions are functions of the 3x3 patches' Gram matrix (91
                # numbers from the feature map; csrc/conv2d.hip) - no pass over C channels x 9 taps per position, and the backward
                # needs ONE pass over the 1 GB gradient instead of two
                gram = K.conv1_gram(feats, torch.empty(K.CONV1_GRAM_DOUBLES, dtype=torch.float64, device=self.device))
                K.conv1_stats_from_gram(gram, w0, b0, stats)
            else:
                K.conv1_stats(feats, w0, b0, stats)
            count0 = B * T1 * F1 * self.dp.world
            self.dp.allreduce_stats_(stats[:2 * C])
            K.bn_finalize(stats, count0, ps.p(nm + "/g"), ps.p(nm + "/b"), fin0, ps.state[nm + "/mm"], ps.state[nm + "/mv"], 0.99, 1e-3, True)
        else:
            count0 = B * T1 * F1
            K.bn_finalize(None, 1, ps.p(nm + "/g"), ps.p(nm + "/b"), fin0, ps.state[nm + "/mm"], ps.state[nm + "/mv"], 0.99, 1e-3, False)
        a1_full, a1 = self._salloc(rows, 4 * C, slack, tag="a1")
        K.conv1_bn_apply_s2d(feats, w0, b0, fin0, a1)
        K.halo_zero(a1, B, T2, F2, 4 * C)
        K.s2d_edge_zero(a1, B, T1, F1, C)
        bn0 = (fin0, count0, gram)
        # conv2: every tap reads the same rows shifted by a constant -> one GEMM over 9 K-segments (bf16) / 9 products (f32)
        W = ps.w2d("enc/sub/conv1/w")  # [9C, C]
        _, o = self._salloc(rows, C, slack, tag="o")
        if self.dtype == torch.float32:
            flat, base = a1_full, slack * 4 * C
            for i in range(9):
                K.gemm(flat[base + shift[i] * 4 * C + blk[i] * C:], W[i * C:(i + 1) * C], o, rows, C, C, 4 * C, C, C,
                       bias=ps.p("enc/sub/conv1/b") if i == 0 else None, accumulate=i > 0)
        else:
            K.gemm(a1, W, o, rows, C, 9 * C, 4 * C, C, C, bias=ps.p("enc/sub/conv1/b"), seg=fwd_a)
        K.halo_zero(o, B, T2, F2, C)
        _, a2 = self._salloc(rows, C, 0)
        _, bn1 = self._bn_fwd(o, "enc/sub/bn1", training, ACT_SWISH, rows=B * T2 * F2, y=a2)
        K.halo_zero(a2, B, T2, F2, C)  # finite halos: the linear layer's weight gradient runs over all (b, tt) rows
        # linear over merge_two_last_dims: rows (b, tt >= 1), columns (ff >= 1, c) = one strided batch view of a2
        drop = self._drop(0, training)
        d = c.dmodel
        x0 = torch.empty(B * T2, d, dtype=self.dtype, device=self.device)
        a2flat = a2.view(-1)
        K.gemm(a2flat[((F2 + 1) + 1) * C:], ps.w2d("enc/linear/w"), x0, T2, d, F2 * C, (F2 + 1) * C, d, d, bias=ps.p("enc/linear/b"),
               nb1=B, sA=((T2 + 1) * (F2 + 1) * C, 0), sD=(T2 * d, 0), drop_p=drop[0], drop_seed=drop[1])
        elen = [-(-(-(-n // 2)) // 2) for n in flen]
        if ctx is not None:
            ctx["sub"] = dict(s2d=True, feats=feats, bn0=bn0, a1=a1, a1_full=a1_full, o=o, bn1=bn1, a2=a2,
                              dims=(B, T0, F0, T1, F1, T2, F2), drop=drop)
        return x0, T2, elen

    def _subsampling_bwd_s2d(self, dx0, ctx):
        ps, c = self.ps, self.cfg
        s = ctx["sub"]
        B, T0, F0, T1, F1, T2, F2 = s["dims"]
        C, d = ps.filt_phys, c.dmodel
        rows, slack = B * (T2 + 1) * (F2 + 1), F2 + 2
        shift, blk, _, dgrad = self._seg_tables(F2, C)
        dx0m = self._mask_grad(dx0, s["drop"])
        # gradient rows in the (b, tt) indexing of the S layout: one zero row per sample in front (plain device copy)
        dx0h = torch.zeros(B, T2 + 1, d, dtype=self.dtype, device=self.device)
        dx0h[:, 1:].copy_(dx0m.view(B, T2, d))
        dx0h = dx0h.view(B * (T2 + 1), d)
        a2flat = s["a2"].view(-1)
        K.gemm(a2flat[C:], dx0h, ps.g2d("enc/linear/w"), F2 * C, d, B * (T2 + 1), (F2 + 1) * C, d, d, trans_a=True, accumulate=True,
               split_k=_split_k(F2 * C, d, B * (T2 + 1)), colsum=ps.g("enc/linear/b"))
        _, da2 = self._salloc(rows, C, 0)
        # the BatchNorm1 backward sums (sum dz, sum dz xhat per channel) in the epilogue of this product - its columns are (frequency,
        # channel) pairs - instead of a pass of their own over o and da2 (2 x 196 MB at the bench shape)
        bst, ncp = None, 8
        if self._bn1_sums_in_gemm and self.dtype == torch.bfloat16 and C % 8 == 0 and 256 % (C // 8) == 0 and 64 <= C <= 2048:
            bst = self._zeros_f32(ncp * 2 * C)[:ncp * 2 * C].view(ncp, 2 * C)
            try:
                K.gemm(dx0h, ps.w2d("enc/linear/w"), da2.view(-1)[C:], B * (T2 + 1), F2 * C, d, d, d, (F2 + 1) * C, trans_b=True,
                       bns=(s["o"].view(-1)[C:], s["bn1"][0], bst, C))
            except K._lib.TfasrUnsupported:
                bst = None
        if bst is None:
            K.gemm(dx0h, ps.w2d("enc/linear/w"), da2.view(-1)[C:], B * (T2 + 1), F2 * C, d, d, d, (F2 + 1) * C, trans_b=True)
        K.halo_zero(da2, B, T2, F2, C)
        do_full, do = self._salloc(rows, C, slack, tag="do")
        if bst is None:
            self._bn_bwd(s["o"], da2, "enc/sub/bn1", s["bn1"], ACT_SWISH, dx=do)
        else:
            fin1, count1 = s["bn1"]
            self.dp.allreduce_stats_(bst.view(-1))
            K.bn_apply_bwd(s["o"], da2, fin1, bst, count1, ACT_SWISH, dx=do, dgamma=ps.g("enc/sub/bn1/g"), dbeta=ps.g("enc/sub/bn1/b"),
                           grad_scale=1.0 / self.dp.world, copies=ncp)
        K.halo_zero(do, B, T2, F2, C)
        # conv2 weight gradient: 9 products gW[tap] += a1[rows shifted by the tap]^T @ do; bias gradient = column sums of do
        a1_full, base = s["a1_full"], slack * 4 * C
        gW = ps.g2d("enc/sub/conv1/w")
        calls = []
        for i in range(9):
            calls.append(dict(A=a1_full[base + shift[i] * 4 * C + blk[i] * C:], B=do, out=gW[i * C:(i + 1) * C], M=C, N=C, K=rows, lda=4 * C, ldb=C,
                              ldd=C, trans_a=True, accumulate=True, split_k=_split_k(C, C, rows), colsum=ps.g("enc/sub/conv1/b") if i == 0 else None))
        K.gemm_group(calls)
        # conv2 data gradient, one product per parity block of the S layout: da1[:, block] = sum over its taps of do[rows shifted back] @ W[tap]^T
        W = ps.w2d("enc/sub/conv1/w")
        _, da1 = self._salloc(rows, 4 * C, 0)
        dobase = slack * C
        for b4 in range(4):
            segs, a_off, b_off = dgrad[b4]
            out = da1.view(-1)[b4 * C:]
            if self.dtype == torch.float32:
                for j, i in enumerate(segs):
                    K.gemm(do_full[dobase - shift[i] * C:], W[i * C:(i + 1) * C], out, rows, C, C, C, C, 4 * C, trans_b=True, accumulate=j > 0)
            else:
                K.gemm(do, W, out, rows, C, len(segs) * C, C, C, 4 * C, trans_b=True, seg=(a_off, b_off, C // (a_off.numel() // len(segs))))
        # BatchNorm0 + conv1 backward from the feature map (only the valid slots of da1 are read: its halos need no clearing)
        fin0, count0, gram = s["bn0"]
        w0, b0 = ps.p("enc/sub/conv0/w"), ps.p("enc/sub/conv0/b")
        if gram is not None:
            zz = self._zeros_f32(12 * C)
            bstats, pbuf = zz[:2 * C], zz[2 * C:12 * C]
            K.conv1_bn_bwd_onepass_s2d(s["feats"], w0, b0, fin0, da1, bstats, pbuf)
            self.dp.allreduce_stats_(bstats)
            K.conv1_bn_bwd_finalize(gram, w0, b0, fin0, bstats, count0, pbuf, ps.g("enc/sub/conv0/w"), ps.g("enc/sub/conv0/b"))
        else:
            bstats = torch.zeros(2 * C, dtype=torch.float32, device=self.device)
            K.conv1_bn_bwd_stats_s2d(s["feats"], w0, b0, fin0, da1, bstats)
            self.dp.allreduce_stats_(bstats)
            K.conv1_bn_bwd_apply_s2d(s["feats"], w0, b0, fin0, bstats, count0, da1, ps.g("enc/sub/conv0/w"), ps.g("enc/sub/conv0/b"))
        inv = 1.0 / self.dp.world
        K.axpy(ps.g("enc/sub/bn0/b"), bstats[:C].contiguous(), inv)
        K.axpy(ps.g("enc/sub/bn0/g"), bstats[C:].contiguous(), inv)

    def _sub_norm_fwd(self, x2d, name, training):
        """the subsampling's norm + swish (subsampling.py:197-214): BatchNormalization, or - `norms: layer` - a LayerNormalization over
        the channel axis of the [B, T, F, C] conv output (gamma / beta in the bn{i}/g, bn{i}/b slots)."""
        if self.cfg.sub_norm == "layer":
            yn, mean, rstd = K.layernorm_fwd(x2d, self.ps.p(name + "/g"), self.ps.p(name + "/b"))
            return K.add_act_fwd(yn, None, ACT_SWISH), (yn, mean, rstd)
        return self._bn_fwd(x2d, name, training, ACT_SWISH)

    def _sub_norm_bwd(self, x2d, dy2d, name, saved):
        if self.cfg.sub_norm == "layer":
            yn, mean, rstd = saved
            dyn = K.add_act_bwd(yn, None, dy2d, ACT_SWISH)
            return K.layernorm_bwd(dyn, x2d, self.ps.p(name + "/g"), mean, rstd, self.ps.g(name + "/g"), self.ps.g(name + "/b"))
        return self._bn_bwd(x2d, dy2d, name, saved, ACT_SWISH)

    def _subsampling_fwd(self, feats, flen, training, ctx):
        if self._s2d_enabled():
            return self._subsampling_fwd_s2d(feats, flen, training, ctx)
        ps, c = self.ps, self.cfg
        B, T0, F0 = feats.shape
        C = ps.filt_phys
        c1 = K.conv1_fwd(feats, ps.p("enc/sub/conv0/w"), ps.p("enc/sub/conv0/b"))  # [B,T1,F1,C]
        T1, F1 = c1.shape[1], c1.shape[2]
        a1, bn0 = self._sub_norm_fwd(c1.view(-1, C), "enc/sub/bn0", training)
        a1 = a1.view(B, T1, F1, C)
        col = K.im2col_3x3s2(a1)  # [B*T2*F2, 9C]
        T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
        c2 = K.matmul(col, ps.w2d("enc/sub/conv1/w"), bias=ps.p("enc/sub/conv1/b"))  # [B*T2*F2, C]
        a2, bn1 = self._sub_norm_fwd(c2, "enc/sub/bn1", training)
        merged = a2.view(B * T2, F2 * C)  # math_util.merge_two_last_dims
        drop = self._drop(0, training)
        x0 = K.matmul(merged, ps.w2d("enc/linear/w"), bias=ps.p("enc/linear/b"), drop_p=drop[0], drop_seed=drop[1])  # [B*T2, d]
        elen = [-(-(-(-n // 2)) // 2) for n in flen]
        if ctx is not None:
            ctx["sub"] = dict(feats=feats, c1=c1, bn0=bn0, col=col, c2=c2, bn1=bn1, merged=merged, dims=(B, T0, F0, T1, F1, T2, F2), drop=drop)
        return x0, T2, elen

    def _subsampling_bwd(self, dx0, ctx):
        if ctx["sub"].get("s2d"):
            return self._subsampling_bwd_s2d(dx0, ctx)
        ps, c = self.ps, self.cfg
        s = ctx["sub"]
        B, T0, F0, T1, F1, T2, F2 = s["dims"]
        C = ps.filt_phys
        dmerged = self._dense_bwd(self._mask_grad(dx0, s["drop"]), s["merged"], "enc/linear/w", "enc/linear/b")
        dc2 = self._sub_norm_bwd(s["c2"], dmerged.view(-1, C), "enc/sub/bn1", s["bn1"])
        dcol = self._dense_bwd(dc2, s["col"], "enc/sub/conv1/w", "enc/sub/conv1/b")
        da1 = K.col2im_3x3s2(dcol, B, T1, F1, C)
        dc1 = self._sub_norm_bwd(s["c1"].view(-1, C), da1.view(-1, C), "enc/sub/bn0", s["bn0"])
        K.conv1_bwd_weight(s["feats"], dc1.view(B, T1, F1, C), ps.g("enc/sub/conv0/w"), ps.g("enc/sub/conv0/b"))

    # =================================================================================== conformer block
    def _ffm_fwd(self, x, pfx, ctx, site, training):
        ps, f = self.ps, self.cfg.ffm_residual
        d1, d2 = self._drop(site, training), self._drop(site + 1, training)
        if self.dtype == torch.bfloat16:  # one launch (csrc/ffn_fused.h) where the shape allows it; None = three-launch route below
            got = K.ffn_fused_fwd(x, ps.p(pfx + "ln/g"), ps.p(pfx + "ln/b"), ps.w2d(pfx + "d1/w"), ps.p(pfx + "d1/b"), ps.w2d(pfx + "d2/w"),
                                  ps.p(pfx + "d2/b"), f, d1[0], d1[1], d2[1], save_z=ctx is not None)
            if got is not None:
                y, ln, mean, rstd, z, h = got
                if ctx is not None:
                    ctx[pfx] = dict(x=x, ln=ln, mean=mean, rstd=rstd, z=z, h=h, d1=d1, d2=d2)
                return y
        ln, mean, rstd = K.layernorm_fwd(x, ps.p(pfx + "ln/g"), ps.p(pfx + "ln/b"))
        z = torch.empty(x.shape[0], ps.shapes[pfx + "d1/w"][1], dtype=self.dtype, device=self.device) if ctx is not None else None
        h = K.matmul(ln, ps.w2d(pfx + "d1/w"), bias=ps.p(pfx + "d1/b"), act=ACT_SWISH, prez=z, drop_p=d1[0], drop_seed=d1[1])
        y = K.matmul(h, ps.w2d(pfx + "d2/w"), bias=ps.p(pfx + "d2/b"), res=x, beta=f, drop_p=d2[0], drop_seed=d2[1])
        if ctx is not None:
            ctx[pfx] = dict(x=x, ln=ln, mean=mean, rstd=rstd, z=z, h=h, d1=d1, d2=d2)
        return y

    def _ffm_bwd(self, dy, pfx, ctx):
        ps, f = self.ps, self.cfg.ffm_residual
        s = ctx.pop(pfx)
        dz = self._dense_bwd(self._mask_grad(dy, s["d2"]), s["h"], pfx + "d2/w", pfx + "d2/b", alpha=f, dact_z=s["z"], dact=ACT_SWISH,
                             drop=s["d1"])
        dln = self._dense_bwd(dz, s["ln"], pfx + "d1/w", pfx + "d1/b")
        return K.layernorm_bwd(dln, s["x"], ps.p(pfx + "ln/g"), s["mean"], s["rstd"], ps.g(pfx + "ln/g"), ps.g(pfx + "ln/b"), add=dy)

    def _mhsa_fwd(self, x, pfx, B, T, elen_dev, ctx, site, training):
        ps, c = self.ps, self.cfg
        H, dh = c.num_heads, ps.head_phys
        HD = H * dh
        R1 = 2 * T
        scale = 1.0 / math.sqrt(c.head_size)
        ln, mean, rstd = K.layernorm_fwd(x, ps.p(pfx + "ln/g"), ps.p(pfx + "ln/b"))
        qkv = K.matmul(ln, ps.w2d(pfx + "qkv/w"), bias=ps.p(pfx + "qkv/b"))  # [B*T, 3HD]
        pe = self._pe_ext(T)
        pext = K.matmul(pe, ps.w2d(pfx + "pos/w"), bias=ps.p(pfx + "pos/b"))  # [2T, HD]
        drop = self._drop(site, training)
        att, saved = self.attention_core(qkv, pext, B, T, elen_dev, pfx)
        y = K.matmul(att, ps.w2d(pfx + "o/w"), bias=ps.p(pfx + "o/b"), res=x, beta=c.mhsam_residual, drop_p=drop[0], drop_seed=drop[1])
        if ctx is not None:
            ctx[pfx] = dict(x=x, ln=ln, mean=mean, rstd=rstd, qkv=qkv, pext=pext, att=att, drop=drop, **saved)
        return y

    def _uv(self, pfx, grad=False):
        """content / positional attention biases of the attention layer `pfx` (shared pair or per-layer, see configs)."""
        get = self.ps.g if grad else self.ps.p
        if self.cfg.mhsam_use_attention_bias:
            return get(pfx + "u"), get(pfx + "v")
        return get("enc/u"), get("enc/v")

    def attention_core(self, qkv, pext, B, T, elen_dev, pfx=None):
        """MultiHeadRelativeAttention._compute_attention (multihead_attention.py:543-582) on the fused projection output
        qkv [B*T, 3*H*dh] and the projected position table pext [2T, H*dh] (rows 0..2T-2 = positions T-1..-(T-1), row 2T-1 = the
        projection of a zeroed encoding row).  Returns (context [B*T, H*dh], tensors the backward needs)."""
        ps, c = self.ps, self.cfg
        H, dh = c.num_heads, ps.head_phys
        HD = H * dh
        R1 = 2 * T
        scale = 1.0 / math.sqrt(c.head_size)
        ub, vb = self._uv(pfx) if pfx is not None else (ps.p("enc/u"), ps.p("enc/v"))
        if self._fused_attention():
            # flash-style kernel: scores, shift, mask, softmax and P@V never leave the CU (csrc/attn_fused.hip)
            att, lse = K.relattn_fused_fwd(qkv, ub, vb, pext, elen_dev, B, H, T, dh, scale,
                                           use_mask=c.use_attention_auto_mask, chunk_size=c.chunk_size, history_size=c.history_size)
            return att, dict(lse=lse)
        qu, qv = K.bias2_fwd(qkv, 3 * HD, ub, vb, B * T, HD)
        kk = qkv[:, HD:]
        vv = qkv[:, 2 * HD:]
        Tp, R1p = -(-T // 8) * 8, -(-R1 // 8) * 8  # row strides padded to 16 B so the score matrices can be LDS-DMA'd
        content = torch.empty(B, H, T, Tp, dtype=self.dtype, device=self.device)
        K.gemm(qu, kk, content, T, T, dh, HD, 3 * HD, Tp, trans_b=True, nb1=B, nb2=H, sA=(T * HD, dh), sB=(T * 3 * HD, dh),
               sD=(H * T * Tp, T * Tp), alpha=scale)
        pos = torch.empty(B, H, T, R1p, dtype=self.dtype, device=self.device)
        K.gemm(qv, pext, pos, T, R1, dh, HD, HD, R1p, trans_b=True, nb1=B, nb2=H, sA=(T * HD, dh), sB=(0, dh),
               sD=(H * T * R1p, T * R1p), alpha=scale)
        probs = K.relattn_softmax_fwd(content, pos, elen_dev, T, use_mask=c.use_attention_auto_mask, probs=content,
                                      chunk_size=c.chunk_size, history_size=c.history_size)
        att = torch.empty(B * T, HD, dtype=self.dtype, device=self.device)
        K.gemm(probs, vv, att, T, dh, T, Tp, 3 * HD, HD, nb1=B, nb2=H, sA=(H * T * Tp, T * Tp), sB=(T * 3 * HD, dh), sD=(T * HD, dh))
        return att, dict(qu=qu, qv=qv, probs=probs)

    def _mhsa_bwd(self, dy, pfx, B, T, elen_dev, ctx):
        ps, c = self.ps, self.cfg
        H, dh = c.num_heads, ps.head_phys
        HD = H * dh
        R1 = 2 * T
        scale = 1.0 / math.sqrt(c.head_size)
        s = ctx.pop(pfx)
        qkv = s["qkv"]
        ub, vb = self._uv(pfx)
        datt = self._dense_bwd(self._mask_grad(dy, s["drop"]), s["att"], pfx + "o/w", pfx + "o/b", alpha=c.mhsam_residual)
        dqkv = torch.empty_like(qkv)
        if "lse" in s:
            # fused kernels: no skewed score gradient in HBM - the query-side kernel finishes dq = dqu + dqv and the u / v bias gradients,
            # the key-side kernel writes dk / dv, dpext is accumulated from the unskewed dS by tfasr_relattn_dpext
            R1p = -(-R1 // 8) * 8
            um = c.use_attention_auto_mask
            dpext = torch.zeros(R1, HD, dtype=torch.float32, device=self.device)
            win = dict(chunk_size=c.chunk_size, history_size=c.history_size)
            gu, gv = self._uv(pfx, grad=True)
            ds, dvec, qu, qv = K.relattn_fused_bwd_q3(qkv, ub, vb, s["pext"], elen_dev, s["att"], datt, s["lse"], dqkv, gu, gv, dpext, B, H, T, dh, scale,
                                                      use_mask=um, **win)
            K.relattn_fused_bwd_k(qkv, qu, qv, s["pext"], elen_dev, datt, s["lse"], dvec, dqkv, B, H, T, dh, scale, use_mask=um, **win)
            K.relattn_dpext(ds, qv, elen_dev, dpext, B, H, T, dh, use_mask=um)
            return self._mhsa_bwd_tail(dy, pfx, B, T, s, dqkv, None, None, qv, R1p, 1.0, dqv=None, dpext=dpext, dq_done=True)
        probs = s["probs"]
        kk, vv = qkv[:, HD:], qkv[:, 2 * HD:]
        # dprobs = datt @ v^T
        Tp, R1p = -(-T // 8) * 8, -(-R1 // 8) * 8
        dprobs = torch.empty(B, H, T, Tp, dtype=self.dtype, device=self.device)
        K.gemm(datt, vv, dprobs, T, T, dh, HD, 3 * HD, Tp, trans_b=True, nb1=B, nb2=H, sA=(T * HD, dh), sB=(T * 3 * HD, dh),
               sD=(H * T * Tp, T * Tp))
        # dv = probs^T @ datt -> v slice of dqkv
        K.gemm(probs, datt, dqkv[:, 2 * HD:], T, dh, T, Tp, HD, 3 * HD, trans_a=True, nb1=B, nb2=H, sA=(H * T * Tp, T * Tp),
               sB=(T * HD, dh), sD=(T * 3 * HD, dh))
        dcontent, dpos = K.relattn_softmax_bwd(probs, dprobs, elen_dev, T, R1p, use_mask=c.use_attention_auto_mask, dcontent=dprobs)
        # dqu = scale * dcontent @ k ; dk = scale * dcontent^T @ qu
        dqu = torch.empty(B * T, HD, dtype=self.dtype, device=self.device)
        K.gemm(dcontent, kk, dqu, T, dh, T, Tp, 3 * HD, HD, nb1=B, nb2=H, sA=(H * T * Tp, T * Tp), sB=(T * 3 * HD, dh),
               sD=(T * HD, dh), alpha=scale)
        K.gemm(dcontent, s["qu"], dqkv[:, HD:], T, dh, T, Tp, HD, 3 * HD, trans_a=True, nb1=B, nb2=H, sA=(H * T * Tp, T * Tp),
               sB=(T * HD, dh), sD=(T * 3 * HD, dh), alpha=scale)
        return self._mhsa_bwd_tail(dy, pfx, B, T, s, dqkv, dqu, dpos, s["qv"], R1p, scale)

    def _fused_attention(self):
        # bf16 with a physical head dimension of 64 (narrower reference heads are stored zero-padded); the streaming (chunked) mask is
        # handled inside the fused kernels
        return (self.dtype == torch.bfloat16 and self.ps.head_phys == 64 and os.environ.get("TFASR_ATTN_UNFUSED", "0") != "1")

    def _mhsa_bwd_tail(self, dy, pfx, B, T, s, dqkv, dqu, dpos, qv, R1p, scale, dqv=None, dpext=None, dq_done=False):
        """dpos [B,H,T,R1p] (gradient of the un-shifted position scores) -> dqv, dpext, bias and projection gradients
        (dq_done: the fused kernels have already written dq and the bias gradients and accumulated dpext)."""
        ps, c = self.ps, self.cfg
        H, dh = c.num_heads, ps.head_phys
        HD = H * dh
        R1 = 2 * T
        if dqv is None and not dq_done:
            # dqv = scale * dpos @ pext ; dpext += scale * sum_b dpos^T @ qv
            dqv = torch.empty(B * T, HD, dtype=self.dtype, device=self.device)
            K.gemm(dpos, s["pext"], dqv, T, dh, R1, R1p, HD, HD, nb1=B, nb2=H, sA=(H * T * R1p, T * R1p), sB=(0, dh), sD=(T * HD, dh),
                   alpha=scale)
            dpext = torch.zeros(R1, HD, dtype=torch.float32, device=self.device)
            K.gemm(dpos, qv, dpext, R1, dh, T, R1p, HD, HD, trans_a=True, nb1=B, nb2=H, sA=(H * T * R1p, T * R1p), sB=(T * HD, dh),
                   sD=(0, dh), alpha=scale, accumulate=True)
        if not dq_done:
            gu, gv = self._uv(pfx, grad=True)
            K.bias2_bwd(dqu, dqv, dqkv, 3 * HD, gu, gv, B * T, HD)
        # positional projection: gWpos += pe^T dpext ; gbpos += colsum(dpext)
        dpext_t = dpext if self.dtype == torch.float32 else K.cast(dpext, torch.empty(R1, HD, dtype=self.dtype, device=self.device))
        pe = self._pe_ext(T)
        d = c.dmodel
        K.gemm(pe, dpext_t, ps.g2d(pfx + "pos/w"), d, HD, R1, d, HD, HD, trans_a=True, accumulate=True)
        K.colsum(dpext, ps.g(pfx + "pos/b"))
        dln = self._dense_bwd(dqkv, s["ln"], pfx + "qkv/w", pfx + "qkv/b")
        return K.layernorm_bwd(dln, s["x"], ps.p(pfx + "ln/g"), s["mean"], s["rstd"], ps.g(pfx + "ln/g"), ps.g(pfx + "ln/b"), add=dy)

    def _convm_fwd(self, x, pfx, B, T, training, ctx, site):
        ps, c = self.ps, self.cfg
        d = c.dmodel
        ln, mean, rstd = K.layernorm_fwd(x, ps.p(pfx + "ln/g"), ps.p(pfx + "ln/b"))
        a = K.matmul(ln, ps.w2d(pfx + "pw1/w"), bias=ps.p(pfx + "pw1/b"))  # [B*T, 2d]
        g = K.glu_fwd(a)  # [B*T, d]
        cv = K.dwconv_fwd(g.view(B, T, d), ps.p(pfx + "dw/w"), ps.p(pfx + "dw/b")).view(B * T, d)
        if c.convm_dw_norm == "layer":  # encoder_convm_dw_norm_type: layer (encoders/conformer.py:334-340), in the bn/g, bn/b slots
            yn, nmean, nrstd = K.layernorm_fwd(cv, ps.p(pfx + "bn/g"), ps.p(pfx + "bn/b"))
            sw, bn = K.add_act_fwd(yn, None, ACT_SWISH), (yn, nmean, nrstd)
        else:
            sw, bn = self._bn_fwd(cv, pfx + "bn", training, ACT_SWISH)
        drop = self._drop(site, training)
        y = K.matmul(sw, ps.w2d(pfx + "pw2/w"), bias=ps.p(pfx + "pw2/b"), res=x, beta=c.convm_residual, drop_p=drop[0], drop_seed=drop[1])
        if ctx is not None:
            ctx[pfx] = dict(x=x, ln=ln, mean=mean, rstd=rstd, a=a, g=g, cv=cv, bn=bn, sw=sw, drop=drop)
        return y

    def _convm_bwd(self, dy, pfx, B, T, ctx):
        ps, c = self.ps, self.cfg
        d = c.dmodel
        s = ctx.pop(pfx)
        dsw = self._dense_bwd(self._mask_grad(dy, s["drop"]), s["sw"], pfx + "pw2/w", pfx + "pw2/b", alpha=c.convm_residual)
        if c.convm_dw_norm == "layer":
            yn, nmean, nrstd = s["bn"]
            dyn = K.add_act_bwd(yn, None, dsw, ACT_SWISH)
            dcv = K.layernorm_bwd(dyn, s["cv"], ps.p(pfx + "bn/g"), nmean, nrstd, ps.g(pfx + "bn/g"), ps.g(pfx + "bn/b"))
        else:
            dcv = self._bn_bwd(s["cv"], dsw, pfx + "bn", s["bn"], ACT_SWISH)
        dcv3 = dcv.view(B, T, d)
        K.dwconv_bwd_weight(s["g"].view(B, T, d), dcv3, ps.g(pfx + "dw/w"), ps.g(pfx + "dw/b"))
        dg = K.dwconv_bwd_data(dcv3, ps.p(pfx + "dw/w")).view(B * T, d)
        da = K.glu_bwd(s["a"], dg)
        dln = self._dense_bwd(da, s["ln"], pfx + "pw1/w", pfx + "pw1/b")
        return K.layernorm_bwd(dln, s["x"], ps.p(pfx + "ln/g"), s["mean"], s["rstd"], ps.g(pfx + "ln/g"), ps.g(pfx + "ln/b"), add=dy)

    def _block_fwd(self, x, i, B, T, elen_dev, training, ctx):
        p = f"enc/block{i}/"
        ps = self.ps
        site = 16 + i * 8
        t0 = self._tick("ffm_fwd")
        x = self._ffm_fwd(x, p + "ff1/", ctx, site, training)
        self._tock("ffm_fwd", t0)
        t0 = self._tick("mhsa_fwd")
        x = self._mhsa_fwd(x, p + "mhsa/", B, T, elen_dev, ctx, site + 2, training)
        self._tock("mhsa_fwd", t0)
        t0 = self._tick("convm_fwd")
        x = self._convm_fwd(x, p + "conv/", B, T, training, ctx, site + 3)
        self._tock("convm_fwd", t0)
        t0 = self._tick("ffm_fwd")
        x = self._ffm_fwd(x, p + "ff2/", ctx, site + 4, training)
        self._tock("ffm_fwd", t0)
        y, mean, rstd = K.layernorm_fwd(x, ps.p(p + "ln/g"), ps.p(p + "ln/b"))
        if ctx is not None:
            ctx[p + "ln"] = dict(x=x, mean=mean, rstd=rstd)
        return y

    def _block_bwd(self, dy, i, B, T, elen_dev, ctx):
        p = f"enc/block{i}/"
        ps = self.ps
        s = ctx.pop(p + "ln")
        dx = K.layernorm_bwd(dy, s["x"], ps.p(p + "ln/g"), s["mean"], s["rstd"], ps.g(p + "ln/g"), ps.g(p + "ln/b"))
        t0 = self._tick("ffm_bwd")
        dx = self._ffm_bwd(dx, p + "ff2/", ctx)
        self._tock("ffm_bwd", t0)
        t0 = self._tick("convm_bwd")
        dx = self._convm_bwd(dx, p + "conv/", B, T, ctx)
        self._tock("convm_bwd", t0)
        t0 = self._tick("mhsa_bwd")
        dx = self._mhsa_bwd(dx, p + "mhsa/", B, T, elen_dev, ctx)
        self._tock("mhsa_bwd", t0)
        t0 = self._tick("ffm_bwd")
        dx = self._ffm_bwd(dx, p + "ff1/", ctx)
        self._tock("ffm_bwd", t0)
        return dx

    # =================================================================================== native block executor
    def _native_cfg(self, i, B, T, training, save):
        c = self.cfg
        k = K._lib.BlockCfg()
        k.B, k.T, k.d, k.H, k.dh, k.dff, k.ksize = B, T, c.dmodel, c.num_heads, self.ps.head_phys, self.ps.shapes["enc/block0/ff1/d1/w"][1], c.kernel_size
        k.dh_logical = c.head_size  # softmax scale 1 / sqrt(head_size); dh is the stored (possibly zero-padded) head dimension
        k.dtype = K._dt(self.ps.shadow)
        k.training, k.save, k.use_mask = int(training), int(save), int(c.use_attention_auto_mask)
        k.force_unfused = int(not self._fused_attention())
        k.world = self.dp.world
        k.site0 = 16 + i * 8
        k.drop_epoch = self._drop_epoch_eff()
        k.drop_p = float(c.dropout)
        k.ffm_res, k.mhsa_res, k.conv_res = c.ffm_residual, c.mhsam_residual, c.convm_residual
        k.ln_eps, k.bn_eps, k.bn_momentum = 1e-3, 1e-3, 0.99
        k.chunk_size = int(c.chunk_size) if c.chunk_size else 0
        k.history_size = int(c.history_size) if c.history_size is not None else -1
        k.dw_norm_layer = int(c.convm_dw_norm == "layer")
        return k

    def _native_params(self, i, T):
        ps = self.ps
        P = self._blk_params.get((i, T))  # one struct per (block, T'): a saved forward keeps ITS positional table
        if P is None:
            P = K._lib.BlockParams()
            P.flat, P.shadow, P.grad = ps.flat.data_ptr(), ps.shadow.data_ptr(), ps.grad.data_ptr()
            if self.cfg.convm_dw_norm != "layer":  # (the LayerNormalization variant has no moving statistics; the executor ignores these)
                P.bn_mm, P.bn_mv = ps.state[f"enc/block{i}/conv/bn/mm"].data_ptr(), ps.state[f"enc/block{i}/conv/bn/mv"].data_ptr()
            for j, nm in enumerate(K._lib.BLOCK_PARAM_NAMES):
                if nm.startswith("/") and self.cfg.mhsam_use_attention_bias:  # "/enc/u" -> this layer's own bias
                    nm = "mhsa/" + nm.rsplit("/", 1)[1]
                P.off[j] = ps.offsets[nm[1:] if nm.startswith("/") else f"enc/block{i}/{nm}"]
            P.pe = self._pe_ext(T).data_ptr()
            self._blk_params[(i, T)] = P
        return P

    def _native_sizes(self, cfgk):
        key = (cfgk.B, cfgk.T, cfgk.training, cfgk.save, cfgk.force_unfused, cfgk.drop_p > 0)
        v = self._blk_sizes.get(key)
        if v is None:
            v = K.block_workspace_sizes(cfgk)
            self._blk_sizes[key] = v
        return v

    def _block_fwd_native(self, x, i, B, T, elen_dev, training, ctx):
        d = self.cfg.dmodel
        save = ctx is not None
        cfgk = self._native_cfg(i, B, T, training, save)
        P = self._native_params(i, T)
        stash_b, fscr_b, _ = self._native_sizes(cfgk)
        y = torch.empty(B * T, d, dtype=self.dtype, device=self.device)
        stash = torch.empty(stash_b, dtype=torch.uint8, device=self.device)
        pool = self._zero_pool.get("fwd") if training else None
        ncp = self._bn_stats_copies if (training and self.dtype == torch.bfloat16) else 1
        stats = pool[i, :ncp * 2 * d + 1] if pool is not None else torch.empty(ncp * 2 * d + 1, dtype=torch.float32, device=self.device)
        scratch = K.workspace(fscr_b, self.device, "blk_fwd")
        io = K._lib.BlockIO()
        io.x_in, io.x_out, io.lengths = x.data_ptr(), y.data_ptr(), elen_dev.data_ptr()
        io.bn_stats = stats.data_ptr()
        io.bn_stats_copies = ncp
        io.prezeroed = 1 if pool is not None else 0
        io.stash, io.stash_bytes, io.scratch, io.scratch_bytes = stash.data_ptr(), stash_b, scratch.data_ptr(), scratch.numel()
        pext_all = self._hoisted.get("pext")
        if pext_all is not None:
            io.pext_pre = pext_all[i].data_ptr()
        cbuf = K.block_ctx()
        if training and (self.dp.world > 1 or self._dp_force_split) and not cfgk.dw_norm_layer:
            K.block_fwd(cfgk, P, io, cbuf, K._lib.PHASE_A)
            self.dp.allreduce_stats_(stats[:ncp * 2 * d])
            K.block_fwd(cfgk, P, io, cbuf, K._lib.PHASE_B)
        else:
            K.block_fwd(cfgk, P, io, cbuf, K._lib.PHASE_A | K._lib.PHASE_B)
        if save:
            ctx[f"enc/block{i}/native"] = dict(cfg=cfgk, P=P, io=io, cbuf=cbuf, keep=(x, y, stash, stats, elen_dev, pext_all))
        return y

    def _block_bwd_native(self, dy, i, ctx):
        s = ctx.pop(f"enc/block{i}/native")
        cfgk, P, io, cbuf = s["cfg"], s["P"], s["io"], s["cbuf"]
        d = self.cfg.dmodel
        _, _, bscr_b = self._native_sizes(cfgk)
        dx = torch.empty(cfgk.B * cfgk.T, d, dtype=self.dtype, device=self.device)
        pool = self._zero_pool.get("bwd")
        n_dpext = 2 * cfgk.T * cfgk.H * cfgk.dh
        ncp = max(int(io.bn_stats_copies), 1)  # (the forward's choice: both statistics buffers of a block have that many copies)
        off = self._bwd_pool_off()
        if pool is not None and pool.shape[1] >= off + n_dpext:
            bstats = pool[i, :ncp * 2 * d]
            io.prezeroed |= 2
            io.dpext_zero = pool[i, off:off + n_dpext].data_ptr()
        else:
            bstats = torch.empty(ncp * 2 * d, dtype=torch.float32, device=self.device)
            io.prezeroed &= ~2
            io.dpext_zero = None
        io.defer_pos_grad, io.ln_part_ext, io.ln_part_ext_floats, io.dcv_keep, io.ds_keep, io.qv_keep = 0, None, 0, None, None, None
        hb = self._hoisted.get("bwd")
        aux_dpext = None
        dcv = None
        if hb is not None:
            # REQUESTS: whether the executor honours them depends on its own route (attention kernels, storage type, TFASR_BLOCK_FUSE /
            # TFASR_ATTN_DPOS): what it actually left to us is read back from tfasr_block_bwd_left after the call (the two sides
            # used to decide separately, and a disagreement ran tfasr_relattn_dpext on uninitialised buffers)
            if io.dpext_zero:
                io.defer_pos_grad = 1
                if self.dpext_aux and cfgk.dh == 64:
                    Tp = -(-cfgk.T // 8) * 8
                    ds = torch.empty(cfgk.B, cfgk.H, cfgk.T, Tp, dtype=self.dtype, device=self.device)
                    qv = torch.empty(cfgk.B * cfgk.T, cfgk.H * cfgk.dh, dtype=self.dtype, device=self.device)
                    io.ds_keep, io.qv_keep = ds.data_ptr(), qv.data_ptr()
                    aux_dpext = (ds, qv, pool[i, off:off + n_dpext], s["keep"][4])
            if hb["ln_part"] is not None:
                io.ln_part_ext = hb["ln_part"][i].data_ptr()
                io.ln_part_ext_floats = hb["ln_part"].shape[1]
            if not cfgk.dw_norm_layer:
                dcv = torch.empty(cfgk.B * cfgk.T, d, dtype=self.dtype, device=self.device)
                io.dcv_keep = dcv.data_ptr()
        # grouped weight gradients of this block on the executor's second stream, beside the next block's backward: two arenas, alternating
        slot = 0
        if self._auto(self.wgrad_stream) and self.dtype == torch.bfloat16:
            self._wgrad_flip = 3 - getattr(self, "_wgrad_flip", 2)
            slot = self._wgrad_flip
        if slot:
            # the group on the second stream reads the block's forward stash (and its input / output rows): alive until the join two blocks on
            self._wgrad_keep.append((s["keep"], dy))
            del self._wgrad_keep[:-3]
        scratch = K.workspace(bscr_b, self.device, f"blk_bwd{slot}")
        io.dy, io.dx, io.bn_bstats = dy.data_ptr(), dx.data_ptr(), bstats.data_ptr()
        io.scratch, io.scratch_bytes = scratch.data_ptr(), scratch.numel()
        io.wgrad_slot = slot
        self._last_wgrad_slot = slot
        if (self.dp.world > 1 or self._dp_force_split) and not cfgk.dw_norm_layer:
            K.block_bwd(cfgk, P, io, cbuf, K._lib.PHASE_A)
            self.dp.allreduce_stats_(bstats)
            K.block_bwd(cfgk, P, io, cbuf, K._lib.PHASE_B)
        else:
            K.block_bwd(cfgk, P, io, cbuf, K._lib.PHASE_A | K._lib.PHASE_B)
        left = K.block_bwd_left(cbuf) if hb is not None else 0
        if hb is not None:
            if left & 4:
                hb["pos"].append(i)
            if left & 8:
                hb["ctx"].append(cbuf)
            if left & 2:
                hb["dw"].append((cfgk, P, cbuf, dcv, s["keep"]))  # (the stash holds the other operand: alive until the batched launch)
        if aux_dpext is not None and (left & 1):
            # the positional table's gradient of this block on the auxiliary stream, beside the next block's backward
            ds, qv, dpext, elen_dev = aux_dpext
            main = torch.cuda.current_stream()
            self.aux_stream.wait_stream(main)
            for t in (ds, qv):
                t.record_stream(self.aux_stream)
            with torch.cuda.stream(self.aux_stream):
                K.relattn_dpext(ds, qv, elen_dev, dpext, cfgk.B, cfgk.H, cfgk.T, cfgk.dh, use_mask=bool(cfgk.use_mask))
            hb["aux_used"] = True
        return dx

    # =================================================================================== encoder
    def encoder_fwd(self, feats, flen, training, ctx):
        """ConformerEncoder.call (conformer.py:672-701): subsample -> linear -> relpe -> blocks.  -> [B*T', d], T', lengths."""
        native = self.native_blocks and not self.time_sections
        self._hoisted = {}
        hoist = native and self._auto(self.block_hoist) and self.dtype == torch.bfloat16 and self._fused_attention() and self.aux_stream is not None
        if hoist:
            self._pext_ahead((((feats.shape[1] + 1) // 2) + 1) // 2)
        t0 = self._tick("subsampling_fwd")
        x, T, elen = self._subsampling_fwd(feats, flen, training, ctx)
        self._tock("subsampling_fwd", t0)
        B = feats.shape[0]
        elen_dev = self._h2d(elen)
        if hoist:
            pext_all = self._hoisted["pext"]
            assert pext_all.shape[1] == 2 * T
            torch.cuda.current_stream().wait_stream(self.aux_stream)
            pext_all.record_stream(torch.cuda.current_stream())
        # one memset clears every block's BatchNorm accumulators (forward) / BN + positional-gradient accumulators (backward)
        # instead of three small in-stream memsets per block (tfasr_block_io.prezeroed)
        self._zero_pool = {}
        if native and training:
            c = self.cfg
            self._zero_pool["fwd"] = torch.zeros(c.num_blocks, -(-(self._bn_stats_copies * 2 * c.dmodel + 1) // 64) * 64, dtype=torch.float32, device=self.device)
            if ctx is not None:
                self._zero_pool["bwd_shape"] = (c.num_blocks, self._bwd_pool_off() + -(-2 * T * c.num_heads * self.ps.head_phys // 64) * 64)
        for i in range(self.cfg.num_blocks):
            x = self._block_fwd_native(x, i, B, T, elen_dev, training, ctx) if native else self._block_fwd(x, i, B, T, elen_dev, training, ctx)
            self._side_tick()  # (a slice of the prediction network on its own stream, if one is pending)
        if ctx is not None:
            ctx["enc"] = dict(B=B, T=T, elen_dev=elen_dev, hoist=hoist)
        return x, T, elen, elen_dev

    def _auto(self, switch):
        """a tri-state option (None = automatic).  Rounds 3-4 switched the hoists and the weight-gradient stream OFF inside a process group;
        since round 5 a data-parallel rank runs the same step as one GPU (deferred gradients in a slice of their own, ParamStore.defer_lo /
        defer_hi; three streams on three priority levels), so automatic = on."""
        return True if switch is None else bool(switch)

    def _pext_ahead(self, T):
        """Every block's projected relative-position table pe @ Wpos + bpos [2T', H*dh] on the auxiliary stream, beside the subsampling:
        none of them depends on the activations (MultiHeadRelativeAttention projects the same encoding table in every layer,
        multihead_attention.py:543-558), so 16 launches leave the blocks' dependent chain."""
        ps, c = self.ps, self.cfg
        HD = c.num_heads * ps.head_phys
        pe = self._pe_ext(T)
        main = torch.cuda.current_stream()
        pext_all = torch.empty(c.num_blocks, 2 * T, HD, dtype=self.dtype, device=self.device)
        self.aux_stream.wait_stream(main)  # (the shadow weights were written on `main` by the previous step's optimizer)
        pext_all.record_stream(self.aux_stream)
        with torch.cuda.stream(self.aux_stream):
            for i in range(c.num_blocks):
                pfx = f"enc/block{i}/mhsa/"
                K.matmul(pe, ps.w2d(pfx + "pos/w"), bias=ps.p(pfx + "pos/b"), out=pext_all[i])
        self._hoisted["pext"] = pext_all

    def encoder_bwd(self, dx, ctx):
        e = ctx["enc"]
        if "bwd_shape" in self._zero_pool:
            self._zero_pool["bwd"] = torch.zeros(*self._zero_pool.pop("bwd_shape"), dtype=torch.float32, device=self.device)
        # positional-projection gradients, LayerNorm folds and depthwise weight gradients of all blocks after the loop (see __init__)
        self._hoisted["bwd"] = None
        if e.get("hoist") and self._zero_pool.get("bwd") is not None:
            c = self.cfg
            nblk = K.layernorm_bwd_part_blocks(e["B"] * e["T"], c.dmodel, self.dtype)
            ln_part = torch.empty(c.num_blocks, 8 * nblk * 2 * c.dmodel, dtype=torch.float32, device=self.device) if nblk > 0 else None
            self._hoisted["bwd"] = dict(pos=[], ctx=[], ln_part=ln_part, dw=[])
        # a block's gradients are complete once its weight-gradient group on the second stream is: its bucket is released one block
        # later, after this stream has been made to wait for that group (the wait the next user of the slot's arena needs anyway)
        prev = None
        for i in reversed(range(self.cfg.num_blocks)):
            self._last_wgrad_slot = 0
            if f"enc/block{i}/native" in ctx:
                dx = self._block_bwd_native(dx, i, ctx)
            else:
                dx = self._block_bwd(dx, i, e["B"], e["T"], e["elen_dev"], ctx)
            if prev is not None:
                if prev[1]:
                    K.block_wgrad_join(1 << (prev[1] - 1))
                self._bucket_after_block(prev[0])
            prev = (i, self._last_wgrad_slot)
            self._side_tick()  # (a slice of the prediction network's backward on its own stream, if one is pending)
        if prev is not None:
            if prev[1]:
                K.block_wgrad_join(3)
            self._bucket_after_block(prev[0])
        self._wgrad_keep = []
        hb = self._hoisted.pop("bwd", None)
        # The blocks' LayerNorm / positional-projection / depthwise-kernel gradients live in ONE region behind the last block (ParamStore):
        # final after the deferred launches - or after block 0's backward when nothing was deferred - and released as one bucket.
        # Nothing on the chain waits for the deferred launches (~0.3 ms: one cast + column-sum, two grouped products, one LayerNorm fold, the
        # depthwise weight-gradient pair moving 0.8 GB), so they run on the SIDE stream beside the subsampling's backward (1.6 ms of chain
        # left) and are joined in front of the optimizer (_backward_from_joint); TFASR_DEFER_SIDE=0: on the chain as in round 4.
        if hb is not None and self.aux_stream is not None and os.environ.get("TFASR_DEFER_SIDE", "1") != "0":
            main = torch.cuda.current_stream()
            self.aux_stream.wait_stream(main)  # (the table gradients of the blocks are already in this stream's order)
            # what the launches read was allocated on the main stream: alive until the join, not handed back to its allocator before
            self._defer_keep = (list(hb["dw"]), hb["ln_part"], self._zero_pool.get("bwd"), list(hb["ctx"]))
            with torch.cuda.stream(self.aux_stream):
                self._deferred_block_grads(hb, e["T"])
                if self.ps.defer_lo is not None:
                    self.dp.grads_ready(self.ps.defer_lo, self.ps.defer_hi)  # (ordered behind the launches: queued from their stream)
            self._aux_pending = True
        else:
            if hb is not None:
                if hb.get("aux_used"):
                    torch.cuda.current_stream().wait_stream(self.aux_stream)  # the table gradients accumulated on the auxiliary stream
                self._deferred_block_grads(hb, e["T"])
            if self.ps.defer_lo is not None:
                self.dp.grads_ready(self.ps.defer_lo, self.ps.defer_hi)
        t0 = self._tick("subsampling_bwd")
        self._subsampling_bwd(dx, ctx)
        self._tock("subsampling_bwd", t0)
        if getattr(self, "_defer_keep", None) is not None:
            # join of the deferred launches (every caller of encoder_bwd - transducer, CTC head - gets complete gradients back)
            torch.cuda.current_stream().wait_stream(self.aux_stream)
            self._defer_keep = None

    def _bwd_pool_off(self):
        """floats in front of a block's table-gradient accumulator in the zeroed backward pool: its BatchNorm backward sums, all copies"""
        return -(-self._bn_stats_copies * 2 * self.cfg.dmodel // 64) * 64

    def _deferred_block_grads(self, hb, T):
        """What the blocks left to the caller (tfasr_block_io.defer_pos_grad / ln_part_ext): gWpos_i += pe^T dpext_i and gbpos_i +=
        colsum(dpext_i) for every block - one cast + column-sum launch over all the f32 tables, the products in grouped launches - and one
        fold for the LayerNorm gamma / beta gradients of all blocks; the depthwise-conv weight gradients of all blocks as one launch pair."""
        ps, c = self.ps, self.cfg
        d, HD, R1 = c.dmodel, c.num_heads * ps.head_phys, 2 * T
        if hb["pos"]:
            pool = self._zero_pool["bwd"]
            off = self._bwd_pool_off()
            # ONE launch: bf16 copies of every block's f32 table gradient (the weight-gradient operands) + the bias gradients from the f32 values
            pool_t = torch.empty(pool.shape, dtype=self.dtype, device=self.device)
            nb = pool.shape[0]
            sums = [ps.g(f"enc/block{i}/mhsa/pos/b") if i in hb["pos"] else None for i in range(nb)]
            K.cast_colsum_many(pool[:, off:], pool_t[:, off:], pool.shape[1], nb, R1, HD, sums)
            pe = self._pe_ext(T)
            calls = []
            for i in hb["pos"]:
                pfx = f"enc/block{i}/mhsa/"
                calls.append(dict(A=pe, B=pool_t[i, off:off + R1 * HD], out=ps.g2d(pfx + "pos/w"), M=d, N=HD, K=R1, lda=d, ldb=HD, ldd=HD, trans_a=True,
                                  accumulate=True, split_k=1))
            for j in range(0, len(calls), 8):
                grp = calls[j:j + 8]
                if len(grp) == 1:
                    K.gemm(**grp[0])
                else:
                    K.gemm_group(grp)
        if hb["ctx"]:
            K.block_ln_fold_all(hb["ctx"], d)
        if hb["dw"]:
            K.block_dwconv_wgrad_all(hb["dw"][0][0], [t[1] for t in hb["dw"]], [t[2] for t in hb["dw"]], [t[3] for t in hb["dw"]], self.device)
            hb["dw"].clear()

    def _bucket_after_block(self, i):
        """block i's slice of the flat gradient (its Dense / BatchNorm variables: contiguous, ParamStore) is final: release its bucket"""
        lo = self.ps.offsets[f"enc/block{i}/ff1/d1/w"]
        hi = self.ps.offsets[f"enc/block{i + 1}/ff1/d1/w"] if i + 1 < self.cfg.num_blocks else self.ps.defer_lo
        self.dp.grads_ready(lo, hi)

    # =================================================================================== prediction network
    def prediction_fwd(self, tokens_dev, plen_dev, ctx, h0=None, c0=None):
        """TransducerPrediction.call (base_transducer.py:123-132): Embedding -> LSTM -> LayerNorm. tokens [B,U1] int32."""
        ps, c = self.ps, self.cfg
        B, U1 = tokens_dev.shape
        E, P = c.embed_dim, c.rnn_units
        emb = K.embedding_fwd(tokens_dev, ps.p("pred/emb"), self.dtype).view(B * U1, E)
        xg = K.matmul(emb, ps.w2d("pred/lstm/k"), bias=ps.p("pred/lstm/b")).view(B, U1, 4 * P)
        gates = torch.empty(B, U1, 4 * P, dtype=self.dtype, device=self.device)
        cseq = torch.empty(B, U1, P, dtype=torch.float32, device=self.device)
        hseq = torch.empty(B, U1, P, dtype=self.dtype, device=self.device)
        yseq = torch.empty(B, U1, P, dtype=self.dtype, device=self.device)
        hr = torch.empty(B, 4 * P, dtype=torch.float32, device=self.device)
        Wrk = ps.w2d("pred/lstm/rk")
        if self._lstm_persist_auto:
            K.lstm_set_persist(0 if (self.use_pred_stream and torch.cuda.current_stream(self.device) == self.pred_stream) else 1)
        K.lstm_seq_fwd(xg, Wrk, h0, c0, plen_dev, gates, cseq, hseq, yseq, hr)  # the U1 steps queued from C (one host call)
        y2 = yseq.view(B * U1, P)
        if c.prediction_layer_norm:
            pred, mean, rstd = K.layernorm_fwd(y2, ps.p("pred/ln/g"), ps.p("pred/ln/b"))
        else:  # prediction_layer_norm: False (contextnet/small.yml.j2)
            pred, mean, rstd = y2, None, None
        if ctx is not None:
            ctx["pred"] = dict(tokens=tokens_dev, plen=plen_dev, emb=emb, gates=gates, cseq=cseq, hseq=hseq, y2=y2, mean=mean, rstd=rstd, B=B, U1=U1)
        return pred  # [B*U1, P]

    # ---- the prediction network queued in SLICES between the encoder blocks --------------------------------------------------------------
    # Its recurrence is ~2 x U1 tiny launches per direction on a stream of its own.  Queued in one go, a host that is slower than the GPU
    # (under rocprofv3: tools/prof_streams.py, profiles/r05_streams_single_slices0.txt) blocks in hipLaunchKernel on that stream's full launch
    # queue and starves the ENCODER's stream meanwhile - the main queue idles for the whole duration of the prediction network, once per
    # direction, 21 % of the traced span.  A slice per encoder block keeps both queues fed: traced span 155 -> 128 ms for six steps.
    # Unprofiled the host runs ~3 steps ahead of the GPU (6.5 ms of pure host time per 21.7-ms step), so the gain there is small:
    # 21.75 -> 21.68 ms/step (two interleaved pairs, profiles/r05_dp_queue_sweep.txt).  TFASR_PRED_SLICES=0: in one go.
    def _pred_nslices(self, U1):
        n = int(os.environ.get("TFASR_PRED_SLICES", "8"))
        return max(1, min(n, U1))

    def _prediction_fwd_gen(self, tokens_dev, plen_dev, ctx):
        """prediction_fwd as a generator: every next() queues one slice (on the current stream); the last one sets self._side_result"""
        ps, c = self.ps, self.cfg
        B, U1 = tokens_dev.shape
        E, P = c.embed_dim, c.rnn_units
        emb = K.embedding_fwd(tokens_dev, ps.p("pred/emb"), self.dtype).view(B * U1, E)
        xg = K.matmul(emb, ps.w2d("pred/lstm/k"), bias=ps.p("pred/lstm/b")).view(B, U1, 4 * P)
        gates = torch.empty(B, U1, 4 * P, dtype=self.dtype, device=self.device)
        cseq = torch.empty(B, U1, P, dtype=torch.float32, device=self.device)
        hseq = torch.empty(B, U1, P, dtype=self.dtype, device=self.device)
        yseq = torch.empty(B, U1, P, dtype=self.dtype, device=self.device)
        hr = torch.empty(B, 4 * P, dtype=torch.float32, device=self.device)
        Wrk = ps.w2d("pred/lstm/rk")
        step = -(-U1 // self._pred_nslices(U1))
        for t0 in range(0, U1, step):
            K.lstm_seq_fwd_range(xg, Wrk, None, None, plen_dev, gates, cseq, hseq, yseq, hr, t0, min(U1, t0 + step))
            if t0 + step < U1:
                yield
        y2 = yseq.view(B * U1, P)
        if c.prediction_layer_norm:
            pred, mean, rstd = K.layernorm_fwd(y2, ps.p("pred/ln/g"), ps.p("pred/ln/b"))
        else:
            pred, mean, rstd = y2, None, None
        if ctx is not None:
            ctx["pred"] = dict(tokens=tokens_dev, plen=plen_dev, emb=emb, gates=gates, cseq=cseq, hseq=hseq, y2=y2, mean=mean, rstd=rstd, B=B, U1=U1)
        self._side_result = pred

    def _prediction_bwd_gen(self, dpred, ctx):
        """prediction_bwd as a generator (slices of the backward recurrence in descending time order, then the weight gradients)"""
        ps, c = self.ps, self.cfg
        s = ctx["pred"]
        B, U1 = s["B"], s["U1"]
        E, P = c.embed_dim, c.rnn_units
        if c.prediction_layer_norm:
            dy = K.layernorm_bwd(dpred, s["y2"], ps.p("pred/ln/g"), s["mean"], s["rstd"], ps.g("pred/ln/g"), ps.g("pred/ln/b")).view(B, U1, P)
        else:
            dy = dpred.view(B, U1, P)
        dy = dy.contiguous()
        dz = torch.empty(B, U1, 4 * P, dtype=self.dtype, device=self.device)
        dh_carry = torch.zeros(B, P, dtype=torch.float32, device=self.device)
        dc_carry = torch.zeros(B, P, dtype=torch.float32, device=self.device)
        dhr = torch.empty(B, P, dtype=torch.float32, device=self.device)
        Wrk = ps.w2d("pred/lstm/rk")
        step = -(-U1 // self._pred_nslices(U1))
        t1 = U1
        while t1 > 0:
            t0 = max(0, t1 - step)
            K.lstm_seq_bwd_range(dy, Wrk, s["gates"], s["cseq"], s["plen"], dz, dh_carry, dc_carry, dhr, t0, t1)
            t1 = t0
            yield
        dz2 = dz.view(B * U1, 4 * P)
        if U1 > 1:
            K.gemm(s["hseq"], dz[:, 1:], ps.g2d("pred/lstm/rk"), P, 4 * P, U1 - 1, P, 4 * P, 4 * P, trans_a=True, nb1=B,
                   sA=(U1 * P, 0), sB=(U1 * 4 * P, 0), sD=(0, 0), accumulate=True)
        demb = self._dense_bwd(dz2, s["emb"], "pred/lstm/k", "pred/lstm/b")
        K.embedding_bwd(s["tokens"], demb, ps.g("pred/emb"))

    def _side_tick(self):
        """queue the next slice of the prediction network (if one is pending) on its stream"""
        g = getattr(self, "_side_gen", None)
        if g is None:
            return
        with torch.cuda.stream(self.pred_stream):
            try:
                next(g)
            except StopIteration:
                self._side_gen = None

    def _side_drain(self):
        while getattr(self, "_side_gen", None) is not None:
            self._side_tick()

    def _pred_sliced(self):
        # the per-step kernels are what the overlapped step runs (see _lstm_persist_auto); a forced persistent launch has nothing to slice
        return self.use_pred_stream and self._lstm_persist_auto and os.environ.get("TFASR_PRED_SLICES", "8") != "0"

    def prediction_bwd(self, dpred, ctx):
        ps, c = self.ps, self.cfg
        s = ctx["pred"]
        B, U1 = s["B"], s["U1"]
        E, P = c.embed_dim, c.rnn_units
        if c.prediction_layer_norm:
            dy = K.layernorm_bwd(dpred, s["y2"], ps.p("pred/ln/g"), s["mean"], s["rstd"], ps.g("pred/ln/g"), ps.g("pred/ln/b")).view(B, U1, P)
        else:
            dy = dpred.view(B, U1, P)
        dz = torch.empty(B, U1, 4 * P, dtype=self.dtype, device=self.device)
        dh_carry = torch.zeros(B, P, dtype=torch.float32, device=self.device)
        dc_carry = torch.zeros(B, P, dtype=torch.float32, device=self.device)
        dhr = torch.empty(B, P, dtype=torch.float32, device=self.device)
        Wrk = ps.w2d("pred/lstm/rk")
        if self._lstm_persist_auto:
            K.lstm_set_persist(0 if (self.use_pred_stream and torch.cuda.current_stream(self.device) == self.pred_stream) else 1)
        K.lstm_seq_bwd(dy.contiguous(), Wrk, s["gates"], s["cseq"], s["plen"], dz, dh_carry, dc_carry, dhr)
        dz2 = dz.view(B * U1, 4 * P)
        # recurrent kernel: gR += sum_b h[b, :-1]^T @ dz[b, 1:]
        if U1 > 1:
            K.gemm(s["hseq"], dz[:, 1:], ps.g2d("pred/lstm/rk"), P, 4 * P, U1 - 1, P, 4 * P, 4 * P, trans_a=True, nb1=B,
                   sA=(U1 * P, 0), sB=(U1 * 4 * P, 0), sD=(0, 0), accumulate=True)
        demb = self._dense_bwd(dz2, s["emb"], "pred/lstm/k", "pred/lstm/b")
        K.embedding_bwd(s["tokens"], demb, ps.g("pred/emb"))

    # =================================================================================== joint + loss
    def joint_fwd(self, enc, pred, B, T, U1, ctx):
        """TransducerJoint.call (base_transducer.py:280-293) -> logits [B,T,U1,V]."""
        ps, c = self.ps, self.cfg
        J, V = c.joint_dim, c.vocab_size
        e = K.matmul(enc, ps.w2d("joint/enc/w"), bias=ps.p("joint/enc/b"))
        p = K.matmul(pred, ps.w2d("joint/pred/w"), bias=ps.p("joint/pred/b"))
        h = K.joint_fwd(e.view(B, T, J), p.view(B, U1, J))
        t0 = self._tick("joint_vocab_gemm")
        logits = K.matmul(h.view(B * T * U1, J), ps.w2d("joint/vocab/w"), bias=ps.p("joint/vocab/b")).view(B, T, U1, V)
        self._tock("joint_vocab_gemm", t0, 2.0 * B * T * U1 * J * V)
        if ctx is not None:
            ctx["joint"] = dict(enc=enc, pred=pred, h=h, B=B, T=T, U1=U1)
        return logits

    def joint_bwd(self, dlogits, ctx):
        ps, c = self.ps, self.cfg
        s = ctx["joint"]
        B, T, U1 = s["B"], s["T"], s["U1"]
        J, V = c.joint_dim, c.vocab_size
        h2 = s["h"].view(B * T * U1, J)
        dh = self._dense_bwd(dlogits.view(B * T * U1, V), h2, "joint/vocab/w", "joint/vocab/b")
        de, dp = K.joint_bwd(s["h"], dh.view(B, T, U1, J))
        denc = self._dense_bwd(de.view(B * T, J), s["enc"], "joint/enc/w", "joint/enc/b")
        dpred = self._dense_bwd(dp.view(B * U1, J), s["pred"], "joint/pred/w", "joint/pred/b")
        return denc, dpred

    # =================================================================================== public API
    def __call__(self, inputs: TrainInput, training=False):
        """Transducer.call (base_transducer.py:427-435)."""
        logits, elen, _ = self._forward(inputs, training, None)
        return TrainOutput(logits=logits, logits_length=torch.tensor(elen, dtype=torch.int32))

    def _forward(self, inputs: TrainInput, training, ctx, masks=None, joint=True):
        dev = self.device
        self._drop_epoch += 1
        sig = inputs.inputs.to(dev, non_blocking=True)
        slen = [int(v) for v in inputs.inputs_length.tolist()]
        tokens = inputs.predictions.to(dev, non_blocking=True).to(torch.int32).contiguous()
        plen = inputs.predictions_length.to(dev, non_blocking=True).to(torch.int32)
        B, U1 = tokens.shape
        main = torch.cuda.current_stream()
        # Host order: the launch queue holds only ~1 ms of work, so (1) the ~1 ms of Python that draws the SpecAugment masks runs FIRST,
        # while the previous step's tail is still executing (between log-mel and the mask kernel it left the main stream idle for 0.85 ms
        # per step under the profiler's slower host; neutral in an unprofiled same-box A/B: 23.81 vs 23.85 ms), and (2) the main stream
        # gets its front end BEFORE the ~170 launches of the prediction network go to the second stream.
        if training and masks is None:
            masks = self.draw_specaugment([-(-int(n) // self.cfg.frame_step) for n in slen])
        # The front end (log-mel + SpecAugment) reads nothing a previous step wrote - no parameters, only this batch's samples - so it does
        # not have to queue behind the previous step's tail on the main stream (subsampling backward + optimizer, ~1.8 ms): it goes to the
        # prediction network's stream, idle at this point, and the main stream picks the features up (round 5; TFASR_FRONT_EARLY=0: in line).
        # Only when the CALLER says its batches are complete in HBM before train_step is called (`prefetched_inputs`, what a double-buffered
        # input pipeline - the reference's tf.data prefetch - provides and what bench.py does): a batch still being produced by earlier work
        # on the current stream would otherwise be read too early.
        if (self.prefetched_inputs and self.use_pred_stream and os.environ.get("TFASR_FRONT_EARLY", "1") != "0"
                and inputs.inputs.device.type == "cuda" and inputs.inputs.device == sig.device):
            sig.record_stream(self.pred_stream)
            with torch.cuda.stream(self.pred_stream):
                feats, flen = self.frontend(sig, slen, training, masks)
            main.wait_stream(self.pred_stream)
            feats.record_stream(main)
        else:
            feats, flen = self.frontend(sig, slen, training, masks)
        self._side_gen = None
        if self.use_pred_stream:
            self.pred_stream.wait_stream(main)
            tokens.record_stream(self.pred_stream)
            plen.record_stream(self.pred_stream)
            if self._pred_sliced():
                K.lstm_set_persist(0)
                self._side_gen = self._prediction_fwd_gen(tokens, plen, ctx)
                self._side_tick()  # embedding, input product and the first slice now; the rest between the encoder blocks
            else:
                with torch.cuda.stream(self.pred_stream):
                    pred = self.prediction_fwd(tokens, plen, ctx)
                pred.record_stream(main)
        enc, T, elen, elen_dev = self.encoder_fwd(feats, flen, training, ctx)
        if self.use_pred_stream:
            if self._side_gen is not None or getattr(self, "_side_result", None) is not None:
                self._side_drain()
                pred, self._side_result = self._side_result, None
                pred.record_stream(main)
            main.wait_stream(self.pred_stream)
        else:
            pred = self.prediction_fwd(tokens, plen, ctx)
        if not joint:
            return enc, pred, (B, T, U1), elen, elen_dev
        logits = self.joint_fwd(enc, pred, B, T, U1, ctx)
        return logits, elen, elen_dev

    def loss_and_backward(self, data: TrainData, training=True, masks=None, want_backward=True, packed=True, reduce=True):
        """BaseModel._train_step (base_model.py:149-183): forward, RnntLoss (mean over the batch, rnnt_loss.py:34) and
        the full backward into the flat gradient buffer (gradients ACCUMULATE; zero_grad() first).

        packed=True evaluates the joint network and the loss only on the valid lattice nodes (t < logit_len_b,
        u <= label_len_b): padded nodes carry exactly zero gradient in the reference (impl/rnnt.py:218-224), so skipping them
        changes no result while removing the padding's share of the largest GEMMs of the step.

        reduce=False keeps this micro-step's gradients local (train_step_ga, base_model.py:200-209: the replicas exchange
        gradients only when the accumulated gradient is applied)."""
        self.dp.set_reduce(bool(reduce))
        ctx = {} if want_backward else None
        dev = self.device
        ps, c = self.ps, self.cfg
        enc, pred, (B, T, U1), elen, elen_dev = self._forward(data.inputs, training, ctx, masks, joint=False)
        J, V = c.joint_dim, c.vocab_size
        labels = data.labels.labels.to(dev, non_blocking=True).to(torch.int32).contiguous()
        llen_host = [int(v) for v in data.labels.labels_length.tolist()]
        # BaseLoss.call: logit_length = max(logit_length, label_length)  (losses/base_loss.py:36)
        tl = [min(max(int(a), b), T) for a, b in zip(elen, llen_host)]
        ul = [min(b, U1 - 1) for b in llen_host]
        tl_dev = self._h2d(tl)
        ul_dev = self._h2d(ul)
        gkey = ("gscale", B, self.dp.world)
        if gkey not in self._consts:
            self._consts[gkey] = torch.full((B,), 1.0 / (B * self.dp.world), dtype=torch.float32, device=dev)
        gscale = self._consts[gkey]
        if not packed:
            logits = self.joint_fwd(enc, pred, B, T, U1, ctx)
            t0 = self._tick("rnnt_loss")
            costs, dlogits = K.rnnt_loss_fwd_bwd(logits, labels, ul_dev, tl_dev, grad_scale=gscale, grads=logits, want_grads=want_backward)
            self._tock("rnnt_loss", t0)
            if not want_backward:
                return costs
            denc, dpred = self.joint_bwd(dlogits, ctx)
        else:
            off = np.zeros(B + 1, np.int64)
            off[1:] = np.cumsum([t * (u + 1) for t, u in zip(tl, ul)])
            total = int(off[-1])
            off_dev = self._h2d(off, torch.int64)
            e = K.matmul(enc, ps.w2d("joint/enc/w"), bias=ps.p("joint/enc/b"))
            p = K.matmul(pred, ps.w2d("joint/pred/w"), bias=ps.p("joint/pred/b"))
            h = K.joint_fwd_packed(e.view(B, T, J), p.view(B, U1, J), off_dev, ul_dev, total)
            # The vocabulary projection's epilogue also emits the log-softmax statistics of every lattice row (max / sum-exp per
            # 64-column slice, blank and label logits) from its f32 accumulators, so the loss skips its first pass over the logits
            # (impl/rnnt.py:211 tf.nn.log_softmax + the gathers of :94-105).  bf16 fast path only; otherwise the plain route.
            stats = None
            logits = None
            if self.dtype == torch.bfloat16 and self.fuse_joint_stats:
                parts = -(-V // 128) * 2
                lse_part = torch.empty(total, parts, 2, dtype=torch.float32, device=dev)
                pick = torch.empty(total, 2, dtype=torch.float32, device=dev)
                row_label = K.rnnt_row_labels(labels, ul_dev, tl_dev, off_dev, total, T, V)
                t0 = self._tick("joint_vocab_gemm")
                if self.joint_recompute and want_backward:
                    try:
                        K.gemm(h, ps.w2d("joint/vocab/w"), None, total, V, J, J, V, V, bias=ps.p("joint/vocab/b"), lse=(lse_part, row_label, pick))
                        self._tock("joint_vocab_gemm", t0, 2.0 * total * J * V)
                        t0 = self._tick("rnnt_loss")
                        costs, coef = K.rnnt_loss_packed_coef(labels, ul_dev, tl_dev, off_dev, total, T, V, (lse_part, pick), grad_scale=gscale)
                        dlogits = torch.empty(total, V, dtype=self.dtype, device=dev)
                        K.gemm(h, ps.w2d("joint/vocab/w"), dlogits, total, V, J, J, V, V, bias=ps.p("joint/vocab/b"), rgrad=(coef, row_label))
                        # algorithmic bytes of this variant: the gradient tensor is written once (the logits never exist)
                        self._tock("rnnt_loss", t0, 1.0 * total * V * dlogits.element_size())
                        return self._joint_backward_tail(costs, dlogits, h, enc, pred, off_dev, ul_dev, tl_dev, B, T, U1, ctx)
                    except K._lib.TfasrUnsupported:
                        t0 = self._tick("joint_vocab_gemm")  # shapes outside the 256-row kernel: the materialised route below
                try:
                    logits = torch.empty(total, V, dtype=self.dtype, device=dev)
                    K.gemm(h, ps.w2d("joint/vocab/w"), logits, total, V, J, J, V, V, bias=ps.p("joint/vocab/b"), lse=(lse_part, row_label, pick))
                    stats = (lse_part, pick)
                    self._tock("joint_vocab_gemm", t0, 2.0 * total * J * V)
                except K._lib.TfasrUnsupported:
                    logits = None
                    if self.timers is not None:  # (keep the section timers paired: ADVICE r02)
                        self._tock("joint_vocab_gemm", t0, 0.0)
            if logits is None:
                t0 = self._tick("joint_vocab_gemm")
                logits = K.matmul(h, ps.w2d("joint/vocab/w"), bias=ps.p("joint/vocab/b"))  # [total, V]
                self._tock("joint_vocab_gemm", t0, 2.0 * total * J * V)
            t0 = self._tick("rnnt_loss")
            costs, dlogits = K.rnnt_loss_packed(logits, labels, ul_dev, tl_dev, off_dev, total, T, grad_scale=gscale, grads=logits,
                                                want_grads=want_backward, stats=stats)
            self._tock("rnnt_loss", t0, 2.0 * total * V * logits.element_size())
            if not want_backward:
                return costs
            return self._joint_backward_tail(costs, dlogits, h, enc, pred, off_dev, ul_dev, tl_dev, B, T, U1, ctx)
        return self._backward_from_joint(costs, denc, dpred, ctx)

    def _joint_backward_tail(self, costs, dlogits, h, enc, pred, off_dev, ul_dev, tl_dev, B, T, U1, ctx):
        """packed lattice: gradient of the lattice logits -> gradients of the joint network's inputs (+ its weight gradients)"""
        J = self.cfg.joint_dim
        ps = self.ps
        # tanh' folded into the data gradient's epilogue (dact = TANH_OUT: times 1 - h^2): the segment sums read one tensor, not two
        if self.joint_wgrad_aux and self.aux_stream is not None and isinstance(self.dp, SingleProcess) and self.dtype == torch.bfloat16:
            # one GPU: the vocabulary projection's weight gradient (nothing waits for it before the optimizer) on the auxiliary stream,
            # beside its data gradient and the start of the backward chain
            main = torch.cuda.current_stream()
            self.aux_stream.wait_stream(main)
            W = ps.w2d("joint/vocab/w")
            din, dout = W.shape
            rows = dlogits.shape[0]
            for t in (dlogits, h):
                t.record_stream(self.aux_stream)
            with torch.cuda.stream(self.aux_stream):
                K.gemm(h, dlogits, ps.g2d("joint/vocab/w"), din, dout, rows, h.stride(0), dlogits.stride(0), dout, trans_a=True, accumulate=True,
                       split_k=_split_k(din, dout, rows), colsum=ps.g("joint/vocab/b"))
            self._aux_pending = True
            dh = K.matmul(dlogits, W, trans_b=True, dact_z=h, dact=ACT_TANH_OUT)
        else:
            dh = self._dense_bwd(dlogits, h, "joint/vocab/w", "joint/vocab/b", dact_z=h, dact=ACT_TANH_OUT)
        de, dp = K.joint_bwd_packed(None, dh, off_dev, ul_dev, tl_dev, B, T, U1)
        denc = self._dense_bwd(de.view(B * T, J), enc, "joint/enc/w", "joint/enc/b")
        dpred = self._dense_bwd(dp.view(B * U1, J), pred, "joint/pred/w", "joint/pred/b")
        return self._backward_from_joint(costs, denc, dpred, ctx)

    def _backward_from_joint(self, costs, denc, dpred, ctx):
        main = torch.cuda.current_stream()
        self.dp.grads_ready(self.ps.offsets["joint/enc/w"], self.ps.n_reg)
        self._side_gen = None
        if self.use_pred_stream:
            self.pred_stream.wait_stream(main)
            dpred.record_stream(self.pred_stream)
            if self._pred_sliced():
                K.lstm_set_persist(0)
                self._side_gen = self._prediction_bwd_gen(dpred, ctx)
                self._side_tick()
            else:
                with torch.cuda.stream(self.pred_stream):
                    self.prediction_bwd(dpred, ctx)
        else:
            self.prediction_bwd(dpred, ctx)
        self.encoder_bwd(denc, ctx)
        if self.use_pred_stream:
            self._side_drain()
            main.wait_stream(self.pred_stream)
        if getattr(self, "_aux_pending", False):
            main.wait_stream(self.aux_stream)
            self._aux_pending = False
        self.dp.finish_grads()
        return costs

    def zero_grad(self):
        self.ps.grad.zero_()

    def learning_rate(self, step):
        s = self.optimizer["schedule"]
        if isinstance(s, (int, float)):
            return float(s)
        from .configs import transformer_schedule

        return transformer_schedule(step, **s)

    def apply_gradients(self, grad_scale=1.0):
        """BaseModel._apply_gradients -> keras Adam (base_model.py:185-192; small.yml.j2:73-87) + L2 regulariser gradient."""
        o = self.optimizer
        self._gradient_noise()  # gradn_config (base_model.py:185-191): no-op unless compiled with one
        self.step += 1
        # keras evaluates the schedule at `iterations` BEFORE the increment (0 at the first update: the reference's first step has
        # lr = 0 with TransformerSchedule) and bias-corrects with iterations + 1 [ext: keras BaseOptimizer._get_current_learning_rate,
        # Adam.update_step]
        lr = self.learning_rate(self.step - 1)
        ps = self.ps
        # bf16 models: the optimizer writes the bf16 shadow of the parameters itself
        fused = ps.shadow is not ps.flat and ps.shadow.dtype == torch.bfloat16
        K.adam(ps.flat, ps.grad, ps.adam_m, ps.adam_v, ps.n_reg, lr, self.step, o["beta1"], o["beta2"], o["eps"], o["weight_decay"],
               self.cfg.l2, grad_scale, shadow=ps.shadow if fused else None)
        if not fused:
            ps.refresh_shadow()
        return lr

    def regularization_loss(self):
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        K.sumsq(self.ps.flat, self.ps.n_reg, out)
        return out * self.cfg.l2

    def train_step(self, data: TrainData, masks=None):
        """BaseModel.train_step / train_step_ga (base_model.py:194-209): returns {'loss': per-utterance costs [B]}.
        With ga_steps > 1 the optimizer is applied every ga_steps-th call with (sum of micro-grads)/ga_steps
        (optimizers/accumulation.py:64-70)."""
        if self._ga_count == 0:
            self.zero_grad()
        # the flat buffer accumulates LOCAL micro-gradients; it is all-reduced once, on the apply micro-step, where the
        # bucketed slices still overlap that micro-step's backward (base_model.py:200-209)
        original_weights = self.apply_gwn()  # gwn_config (base_model.py:156-159): no-op unless compiled with one
        costs = self.loss_and_backward(data, True, masks, reduce=self._ga_count + 1 >= self.ga_steps)
        self.remove_gwn(original_weights)
        self._ga_count += 1
        if self._ga_count >= self.ga_steps:
            self.apply_gradients(1.0 / self.ga_steps)
            self._ga_count = 0
        return {"loss": costs}

    # =================================================================================== greedy inference
    def get_initial_decoder_states(self, batch_size=1):
        """TransducerPrediction.get_initial_state (base_transducer.py:109-121): zeros [B, num_rnns=1, 2 (h,c), P]."""
        return torch.zeros(batch_size, 1, 2, self.cfg.rnn_units, dtype=torch.float32, device=self.device)

    def get_initial_tokens(self, batch_size=1):
        return torch.full((batch_size, 1), self.blank, dtype=torch.int32, device=self.device)

    def inference_twin(self):
        """This model in f32 parity mode over the SAME parameter buffers (f32 master weights, BatchNorm moving statistics): no copy,
        always current.  The twin of an f32 model is the model itself."""
        if self.dtype == torch.float32:
            return self
        if self._twin is None:
            t = object.__new__(type(self))
            t.__dict__.update(self.__dict__)
            t.dtype, t.ps = torch.float32, self.ps.alias(torch.float32)
            t._consts, t._blk_params, t._blk_sizes, t._zero_pool, t._twin = {}, {}, {}, {}, None
            t.timers, t.timer_work = None, {}
            self._twin = t
        return self._twin

    @torch.no_grad()
    def encode(self, signals, signals_length, precision=None):
        """frontend + ConformerEncoder.call_next (conformer.py:703-718), inference mode (moving BN statistics).
        precision "f32" (default, self.decode_precision) runs it on the f32 master weights; "bf16" on the training kernels."""
        if (precision or self.decode_precision) == "f32" and self.dtype != torch.float32:
            return self.inference_twin().encode(signals, signals_length)
        sig = signals.to(self.device, non_blocking=True)
        slen = [int(v) for v in signals_length.tolist()]
        feats, flen = self.frontend(sig, slen, training=False)
        enc, T, elen, _ = self.encoder_fwd(feats, flen, False, None)
        return enc.view(sig.shape[0], T, self.cfg.dmodel), elen

    @torch.no_grad()
    def recognize(self, inputs: PredictInput, max_tokens_per_frame=3, check_every=16, precision=None):
        """Transducer.recognize (base_transducer.py:474-494): batch size 1 -> recognize_single (<=3 symbols per frame),
        otherwise recognize_batch — the two variants are NOT equivalent in the reference and both are reproduced."""
        enc, elen = self.encode(inputs.inputs, inputs.inputs_length, precision)
        return self.recognize_encoded(enc, elen, inputs.previous_tokens, inputs.previous_decoder_states, max_tokens_per_frame,
                                      check_every)

    def recognize_beam(self, inputs: PredictInput, beam_width=10, **kw):
        """The reference's recognize_beam falls back to greedy (base_transducer.py:841-842)."""
        return self.recognize(inputs, **kw)

    @torch.no_grad()
    def recognize_encoded(self, enc, elen, previous_tokens=None, previous_decoder_states=None, max_tokens_per_frame=3,
                          check_every=16):
        ps, c, dev = self.ps, self.cfg, self.device
        B, T, d = enc.shape
        E, P, J, V = c.embed_dim, c.rnn_units, c.joint_dim, c.vocab_size
        mode = 1 if B == 1 else 0
        # The search arithmetic (embedding, LSTM cell, LayerNorm, joint, vocabulary projection, log-softmax, arg-max) runs in
        # f32 on the f32 master weights whatever the model's storage type: the GEMMs of one step are [B, <=640] x [640, <=2560]
        # (launch-bound, not MFMA-bound), and every token decision then carries the reference's f32 arithmetic
        # (base_transducer.py:437-464) - a bf16 model's tokens are bit-exact against the reference search applied to ITS
        # encoder output (tests/test_parity_baseline_gpu.py).
        f32 = torch.float32
        nframes = torch.tensor([int(v) for v in elen], dtype=torch.int32).to(dev)
        enc32 = enc.reshape(B * T, d)
        if enc32.dtype != f32:
            enc32 = K.cast(enc32.contiguous(), torch.empty(B * T, d, dtype=f32, device=dev))
        encj = K.matmul(enc32, ps.p2d("joint/enc/w"), bias=ps.p("joint/enc/b")).view(B, T, J)
        max_tokens = int(elen[0]) * max_tokens_per_frame if mode == 1 else 2 * T + 1
        tokens = torch.full((B, max(max_tokens, 1)), self.blank, dtype=torch.int32, device=dev)
        frame_idx = torch.zeros(B, dtype=torch.int32, device=dev)
        tok_idx = torch.full((B,), -1 if mode == 1 else 1, dtype=torch.int32, device=dev)
        prev_tok = (torch.full((B,), self.blank, dtype=torch.int32, device=dev) if previous_tokens is None
                    else previous_tokens.to(dev).to(torch.int32).reshape(B).clone())  # (updated in place by the search: never the caller's tensor)
        if previous_decoder_states is None:
            h = torch.zeros(B, P, dtype=f32, device=dev)
            cst = torch.zeros(B, P, dtype=f32, device=dev)
        else:
            st = previous_decoder_states.to(dev)
            h = st[:, 0, 0].float().clone()
            cst = st[:, 0, 1].float().clone()
        per_frame = torch.zeros(max(int(elen[0]), 1), dtype=torch.int32, device=dev) if mode == 1 else None
        active = torch.ones(1, dtype=torch.int32, device=dev)
        ecur = torch.empty(B, J, dtype=f32, device=dev)
        xg = torch.empty(B, 4 * P, dtype=f32, device=dev)
        hr = torch.empty(B, 4 * P, dtype=f32, device=dev)
        h_new = torch.empty(B, P, dtype=f32, device=dev)
        c_new = torch.empty(B, P, dtype=f32, device=dev)
        pj = torch.empty(B, J, dtype=f32, device=dev)
        logits = torch.empty(B, V, dtype=f32, device=dev)
        Wk, Wrk, Wjp, Wv = ps.p2d("pred/lstm/k"), ps.p2d("pred/lstm/rk"), ps.p2d("joint/pred/w"), ps.p2d("joint/vocab/w")
        # every useful iteration advances a frame or appends a token; the reference's while_loop is unbounded and can
        # spin forever once a sample saturates its token buffer — cap the trip count instead
        max_iters = T + max_tokens + 2
        it = 0
        zbuf = torch.empty(B, J, dtype=f32, device=dev)
        lng, lnb = (ps.p("pred/ln/g"), ps.p("pred/ln/b")) if c.prediction_layer_norm else (None, None)
        fused = B <= 64 and self.decode_fused
        packed = K.decode_pack(ps.p("pred/emb"), Wk, Wrk, Wjp, Wv) if fused else None  # tile order of the MFMA step kernels
        # Iterations queued per host check (fused route).  A row leaves the loop only after it has consumed its frames, one per blank, so
        # max_b(frames left) iterations are certain to be needed: that many are queued without looking (the first batch is ~T' long), then
        # the counters are read back ONCE per batch (one device -> host sync) - 4-5 syncs per search instead of one per 64 iterations; the
        # no-op tail of the last batch is at most `check_every` iterations.
        need = int(elen[0]) - 1 if mode == 1 else max(int(v) for v in elen) - 1
        while it < max_iters:
            n = min(max(need, check_every) if fused else check_every, max_iters - it)
            if fused:
                # `n` iterations = 3 skinny-product launches + the bookkeeping kernel each (csrc/decode_step.hip), queued by one host call
                fused = K.decode_steps(ps.p("pred/emb"), Wk, Wrk, ps.p("pred/lstm/b"), lng, lnb, Wjp, ps.p("joint/pred/b"), Wv,
                                       ps.p("joint/vocab/b"), encj, nframes, frame_idx, tok_idx, prev_tok, h, cst, active, h_new, c_new, zbuf,
                                       logits, tokens, per_frame, max_tokens, self.blank, mode, max_tokens_per_frame, n, packed=packed)
                if fused:
                    it += n
                    left = (nframes - 1 - frame_idx).clamp_(min=0).max()
                    act_h, need = (int(v) for v in torch.stack([active[0], left]).tolist())  # (one sync)
                    if act_h == 0:
                        break
                    continue
            for _ in range(n):
                K.decode_prepare(encj, nframes, frame_idx, tok_idx, active, ecur, max_tokens, mode)
                emb = K.embedding_fwd(prev_tok, ps.p("pred/emb"), f32)
                K.matmul(emb, Wk, bias=ps.p("pred/lstm/b"), out=xg)
                K.gemm(h, Wrk, hr, B, 4 * P, P, P, 4 * P, 4 * P)
                K.lstm_step_fwd(xg, hr, h, cst, None, 0, None, c_new, h_new, None, B, P)
                if c.prediction_layer_norm:
                    y, _, _ = K.layernorm_fwd(h_new, ps.p("pred/ln/g"), ps.p("pred/ln/b"), save_stats=False)
                else:  # prediction_layer_norm: False (contextnet/small.yml.j2), as prediction_fwd
                    y = h_new
                K.matmul(y, Wjp, bias=ps.p("joint/pred/b"), out=pj)
                z = K.joint_fwd(ecur.view(B, 1, J), pj.view(B, 1, J))
                K.matmul(z.view(B, J), Wv, bias=ps.p("joint/vocab/b"), out=logits)
                K.decode_update(logits, active, nframes, frame_idx, prev_tok, tok_idx, tokens, per_frame, h_new, c_new, h, cst,
                                max_tokens, self.blank, mode, max_tokens_per_frame)
                it += 1
            if int(active.item()) == 0:
                break
        self.last_search_iterations = it  # (queued iterations incl. the no-op tail of the last batch; bench.py reports it)
        states = torch.stack([h, cst], dim=1).unsqueeze(1)  # [B, 1, 2, P]
        return PredictOutput(tokens=tokens[:, :max_tokens], next_tokens=prev_tok.view(B, 1), next_encoder_states=None,
                             next_decoder_states=states)

    # ----------------------------------------------------------------- timing hooks (bench.py)
    def _tick(self, name):
        if self.timers is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _tock(self, name, start, work=0.0):
        if self.timers is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.timers.setdefault(name, []).append((start, e))
        self.timer_work.setdefault(name, []).append(work)


def _mel_weight_matrix(num_mel_bins, num_spectrogram_bins, sample_rate, lower, upper):
    """tf.signal.linear_to_mel_weight_matrix semantics (feature_extraction.py:222-229): HTK mel scale, DC row zero,
    un-normalised triangles, float32 arithmetic."""
    f32 = np.float32

    def mel(f):
        return (f32(1127.0) * np.log(f32(1.0) + np.asarray(f, f32) / f32(700.0))).astype(f32)

    nyquist = f32(sample_rate) / f32(2.0)
    linear = np.linspace(f32(0.0), nyquist, num_spectrogram_bins, dtype=f32)[1:]
    spec_mel = mel(linear)[:, None]
    edges = np.linspace(mel(lower), mel(upper), num_mel_bins + 2, dtype=f32)
    lo, ce, hi = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    w = np.maximum(f32(0.0), np.minimum((spec_mel - lo) / (ce - lo), (hi - spec_mel) / (hi - ce))).astype(f32)
    return np.pad(w, [[1, 0], [0, 0]]).astype(f32)
