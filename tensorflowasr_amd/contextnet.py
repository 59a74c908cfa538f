"""ContextNet transducer (BASELINE.json configs[3]) on the same HIP kernels and C ABI as the
Conformer path: the encoder of tensorflow_asr/models/encoders/contextnet.py:40-341 -

    ConvModule   = SeparableConv1D(k, stride, causal) -> BatchNormalization(synchronized) -> activation        (:40-109)
    SEModule     = ConvModule -> masked global average pool -> Dense(C/8) -> swish -> Dense(C) -> sigmoid -> scale  (:111-176)
    ConvBlock    = (nlayers-1) ConvModules -> ConvModule(stride) -> SEModule [-> + ConvModule(stride, linear)(input)] -> swish (:179-298)
    ContextNetEncoder = reshape [B,T,F,1] -> [B,T,F], then the blocks; time reduction = product of the strides   (:301-341)

with an explicit hand-written backward.  The frontend, prediction network, joint network, RNN-T loss, optimizer, decoding
and data-parallel hooks are inherited from ConformerTransducer (models/transducer/contextnet.py wires the same pieces).

Kernel mapping: depthwise part = tfasr_dwconv_* (causal, stride 1; a strided causal conv is the stride-1 one sampled every
`stride` frames: tfasr_rows_subsample_*), pointwise part = tfasr_gemm (bias fused, weight + bias gradients in one launch),
BatchNorm = tfasr_bn_* with the sync-BN all-reduce hook, squeeze-excite = tfasr_se_* + two tiny GEMMs, block tail =
tfasr_add_act_*.  Everything is HBM-bound except the pointwise GEMMs.
"""
import os

import torch

from . import kernels as K
from .conformer import ConformerTransducer, _split_k
from .kernels import ACT_NONE, ACT_SIGMOID, ACT_SWISH
from .params import contextnet_modules

_ACT = {"swish": ACT_SWISH, "none": ACT_NONE}


class ContextNetTransducer(ConformerTransducer):
    def __init__(self, cfg, device=None, dtype=torch.bfloat16, seed=0, dp=None):
        if cfg.encoder != "contextnet":
            raise ValueError("ContextNetTransducer needs a config with encoder='contextnet' (configs.contextnet())")
        super().__init__(cfg, device, dtype, seed, dp)
        self.blocks = contextnet_modules(cfg)
        self.native_blocks = False  # the native executor (csrc/block.hip) is the Conformer block

    def _encoder_length(self, t):
        for blk in self.blocks:
            t = -(-int(t) // int(blk["stride"]))
        return t

    # ------------------------------------------------------------------------------- grouped weight gradients
    # The pointwise weight gradients (x^T dy, 137 per ContextNet-L step) are independent of everything downstream in the backward
    # pass: they are queued and issued several per launch (tfasr_gemm_group shares one k-split / XCD placement over the group)
    # instead of one split-K launch each.  The queue holds the operands alive until it is flushed.
    _WG_GROUP = 8

    def _dense_bwd(self, dy, x, wname, bname, alpha=1.0, dact_z=None, dact=ACT_NONE, need_dx=True, drop=(0.0, 0)):
        q = getattr(self, "_wg_queue", None)
        W = self.ps.w2d(wname)
        din, dout = W.shape
        rows = dy.shape[0]
        if q is None or self.dtype != torch.bfloat16 or rows < 2048 or dout <= 64 or alpha != 1.0:
            return super()._dense_bwd(dy, x, wname, bname, alpha, dact_z, dact, need_dx, drop)
        q.append(dict(A=x, B=dy, out=self.ps.g2d(wname), M=din, N=dout, K=rows, lda=x.stride(0), ldb=dy.stride(0), ldd=dout, trans_a=True,
                      accumulate=True, split_k=_split_k(din, dout, rows), colsum=self.ps.g(bname) if bname is not None else None))
        if len(q) >= self._WG_GROUP:
            self._wg_flush()
        if not need_dx:
            return None
        return K.matmul(dy, W, trans_b=True, dact_z=dact_z, dact=dact, drop_p=drop[0], drop_seed=drop[1])

    def _wg_flush(self):
        q = self._wg_queue
        if q:
            if len(q) == 1:
                K.gemm(**q[0])
            else:
                K.gemm_group(q)
            q.clear()

    def _dw_flush(self):
        q = self._dw_queue
        if not q:
            return
        groups = {}
        for it in q:
            groups.setdefault((tuple(it[0].shape), it[2].shape[0]), []).append(it)
        for items in groups.values():
            for j in range(0, len(items), 32):
                K.dwconv_bwd_weight_many(items[j:j + 32])
        q.clear()

    # ------------------------------------------------------------------------------- one ConvModule
    def _cm_fwd(self, x, mod, B, T, training, ctx):
        """x [B*T, Cin] -> y [B*T2, Cout]."""
        name, ci, co, Kk, stride, act = mod
        ps = self.ps
        t0 = self._tick("cn_dwconv_fwd")
        dw = K.dwconv_fwd(x.view(B, T, ci), ps.p(name + "/dw"), None)
        self._tock("cn_dwconv_fwd", t0, 2.0 * B * T * ci * x.element_size())  # algorithmic bytes: read x once, write y once
        T2 = -(-T // stride)
        if stride > 1:
            dw = K.rows_subsample_fwd(dw, stride)
        dw2 = dw.view(B * T2, ci)
        pw = K.matmul(dw2, ps.w2d(name + "/pw/w"), bias=ps.p(name + "/pw/b"))
        y, bn = self._bn_fwd(pw, name + "/bn", training, _ACT[act])
        if ctx is not None:
            ctx[name] = dict(x=x, dw=dw2, pw=pw, bn=bn, T=T, T2=T2)
        return y, T2

    def _cm_bwd(self, dy, mod, B, ctx):
        name, ci, co, Kk, stride, act = mod
        ps = self.ps
        s = ctx.pop(name)
        dpw = self._bn_bwd(s["pw"], dy, name + "/bn", s["bn"], _ACT[act])
        ddw = self._dense_bwd(dpw, s["dw"], name + "/pw/w", name + "/pw/b")
        ddw = ddw.view(B, s["T2"], ci)
        if stride > 1:
            ddw = K.rows_subsample_bwd(ddw, s["T"], stride)
        x3 = s["x"].view(B, s["T"], ci)
        dq = getattr(self, "_dw_queue", None)
        if dq is not None and self.dtype == torch.bfloat16:
            # nothing on the backward chain waits for a depthwise WEIGHT gradient (two launches at ~0.3 of HBM per layer, 151 layers): the
            # operands are kept and the layers of one shape go out as one launch pair at the end of the encoder's backward
            dq.append((x3, ddw, ps.g(name + "/dw").view(Kk, ci), None))
        else:
            K.dwconv_bwd_weight(x3, ddw, ps.g(name + "/dw"), None)
        return K.dwconv_bwd_data(ddw, ps.p(name + "/dw")).view(B * s["T"], ci)

    # ------------------------------------------------------------------------------- squeeze-and-excite
    def _se_fwd(self, o, pfx, B, T, C, len_dev, ctx):
        ps = self.ps
        pool = K.se_pool(o.view(B, T, C), len_dev)                                  # [B, C] f32, masked mean
        pool_c = pool if self.dtype == torch.float32 else K.cast(pool, torch.empty(B, C, dtype=self.dtype, device=self.device))
        z1 = torch.empty(B, C // 8, dtype=self.dtype, device=self.device)
        h1 = K.matmul(pool_c, ps.w2d(pfx + "fc1/w"), bias=ps.p(pfx + "fc1/b"), act=ACT_SWISH, prez=z1)
        z2 = torch.empty(B, C, dtype=self.dtype, device=self.device)
        sg = K.matmul(h1, ps.w2d(pfx + "fc2/w"), bias=ps.p(pfx + "fc2/b"), act=ACT_SIGMOID, prez=z2)
        scale = sg if self.dtype == torch.float32 else K.cast(sg, torch.empty(B, C, dtype=torch.float32, device=self.device))
        y = K.se_scale_fwd(o.view(B, T, C), scale).view(B * T, C)
        if ctx is not None:
            ctx[pfx] = dict(o=o, pool_c=pool_c, z1=z1, h1=h1, z2=z2, scale=scale, len_dev=len_dev)
        return y

    def _se_bwd(self, dy, pfx, B, T, C, ctx):
        s = ctx.pop(pfx)
        ds = K.se_scale_bwd_reduce(s["o"].view(B, T, C), dy.view(B, T, C))          # [B, C] f32
        ds_c = ds if self.dtype == torch.float32 else K.cast(ds, torch.empty(B, C, dtype=self.dtype, device=self.device))
        dz2 = K.add_act_bwd(s["z2"], None, ds_c, ACT_SIGMOID)
        dz1 = self._dense_bwd(dz2, s["h1"], pfx + "fc2/w", pfx + "fc2/b", dact_z=s["z1"], dact=ACT_SWISH)
        dpool_c = self._dense_bwd(dz1, s["pool_c"], pfx + "fc1/w", pfx + "fc1/b")
        dpool = dpool_c if self.dtype == torch.float32 else K.cast(dpool_c, torch.empty(B, C, dtype=torch.float32, device=self.device))
        return K.se_bwd_apply(dy.view(B, T, C), s["scale"], dpool, s["len_dev"]).view(B * T, C)

    # ------------------------------------------------------------------------------- ConvBlock
    def _block_fwd_cn(self, x, blk, B, T, lens, training, ctx):
        x0, T0 = x, T
        for mod in blk["convs"]:
            x, T = self._cm_fwd(x, mod, B, T, training, ctx)
        lens2 = [-(-n // blk["stride"]) for n in lens]
        len_dev = self._h2d(lens2)
        C = blk["C"]
        se = self._se_fwd(x, blk["prefix"] + "se/", B, T, C, len_dev, ctx)
        res = None
        if blk["res"] is not None:
            res, _ = self._cm_fwd(x0, blk["res"], B, T0, training, ctx)
        y = K.add_act_fwd(se, res, ACT_SWISH)
        if ctx is not None:
            ctx[blk["prefix"]] = dict(se=se, res=res, T=T)
        return y, T, lens2

    def _block_bwd_cn(self, dy, blk, B, ctx):
        s = ctx.pop(blk["prefix"])
        T, C = s["T"], blk["C"]
        dz = K.add_act_bwd(s["se"], s["res"], dy, ACT_SWISH)
        dx = self._se_bwd(dz, blk["prefix"] + "se/", B, T, C, ctx)
        for mod in reversed(blk["convs"]):
            dx = self._cm_bwd(dx, mod, B, ctx)
        if blk["res"] is not None:
            dres = self._cm_bwd(dz, blk["res"], B, ctx)
            dx = K.add_act_fwd(dx, dres, ACT_NONE)
        return dx

    # ------------------------------------------------------------------------------- encoder
    def encoder_fwd(self, feats, flen, training, ctx):
        """ContextNetEncoder.call (contextnet.py:313-318): features [B, T0, F] -> [B*T', dmodel], T', lengths."""
        B, T, F = feats.shape
        x = feats.reshape(B * T, F)
        lens = [int(n) for n in flen]
        for blk in self.blocks:
            x, T, lens = self._block_fwd_cn(x, blk, B, T, lens, training, ctx)
            self._side_tick()  # (a slice of the prediction network on its own stream, if one is pending)
        elen_dev = self._h2d(lens)
        if ctx is not None:
            ctx["enc"] = dict(B=B, T=T, elen_dev=elen_dev)
        return x, T, lens, elen_dev

    def encoder_bwd(self, dx, ctx):
        B = ctx["enc"]["B"]
        self._wg_queue = []
        # one GPU: the depthwise weight gradients of all layers batched by shape after the loop (a data-parallel group releases a block's
        # gradient range right behind the block); dw_batch=False: per layer
        from .conformer import SingleProcess
        self._dw_queue = [] if (isinstance(self.dp, SingleProcess) and getattr(self, "dw_batch", True)) else None
        try:
            for i in reversed(range(len(self.blocks))):
                dx = self._block_bwd_cn(dx, self.blocks[i], B, ctx)
                if self._wg_queue is not None:
                    self._wg_flush()  # this block's gradients are complete before its range is handed to the all-reduce
                lo = self.ps.offsets[self.blocks[i]["convs"][0][0] + "/dw"]
                hi = self.ps.offsets[self.blocks[i + 1]["convs"][0][0] + "/dw"] if i + 1 < len(self.blocks) else self.ps.offsets["pred/emb"]
                self.dp.grads_ready(lo, hi)
                self._side_tick()
            self._dw_flush()
        finally:
            self._wg_queue = None  # the prediction / joint networks' gradients are issued directly
            self._dw_queue = None
        # (the input features need no gradient)
