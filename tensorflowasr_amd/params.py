"""Flat parameter / gradient / optimizer-state store.

All trainable variables live in ONE contiguous f32 buffer (regularised variables first, each group in forward
order), mirrored by one f32 gradient buffer and — in bf16 mode — one bf16 shadow copy that the MFMA GEMMs read.
Consequences: the optimizer is a single kernel launch, the shadow refresh is a single cast, the data-parallel
gradient exchange is a handful of all-reduces over contiguous slices (no per-tensor collectives), and the L2
kernel-regulariser (small.yml.j2:67-69) is the prefix [0, n_reg) of the buffer.

Variable names / layouts are the Keras ones of the reference except that the attention q/k/v kernels
[d,H,dh] are stored fused as one [d, 3*H*dh] matrix (columns q|k|v) so the projection is one GEMM; `export_keras()`
/ `import_keras()` split / fuse them, so a checkpoint in the reference's layout is a rename away.
"""
import math

import numpy as np
import torch

from . import kernels as K


def head_padded(name):
    """variables that carry the attention head dimension (physically padded to 64 in bf16 models, see ParamStore)"""
    return name.endswith(("mhsa/qkv/w", "mhsa/qkv/b", "mhsa/pos/w", "mhsa/pos/b", "mhsa/o/w", "mhsa/u", "mhsa/v")) or name in ("enc/u", "enc/v")


_CHAN = ("enc/sub/conv0/w", "enc/sub/conv0/b", "enc/sub/bn0/g", "enc/sub/bn0/b", "enc/sub/conv1/w", "enc/sub/conv1/b", "enc/sub/bn1/g",
         "enc/sub/bn1/b", "enc/linear/w", "enc/sub/bn0/mm", "enc/sub/bn0/mv", "enc/sub/bn1/mm", "enc/sub/bn1/mv")


def deferred_grad(name):
    """regularised Conformer-block variables whose gradients are finished by launches that run once for ALL blocks after the last block's
    backward (conformer._deferred_block_grads): every LayerNorm's gamma / beta, the positional projection's kernel, the depthwise kernel"""
    return name.startswith("enc/block") and name.endswith(("/ln/g", "/ln/b", "mhsa/pos/w", "conv/dw/w"))


def chan_padded(name):
    """variables (and BatchNorm state) that carry the subsampling's channel dimension (physically padded to a multiple of 64 in bf16
    models, see ParamStore)"""
    return name in _CHAN


def param_specs(cfg, head_phys=None, filt_phys=None):
    """[(name, shape, regularized, init)] in forward order. init in {glorot, zeros, ones, embed, orth, lstm_bias}.
    head_phys: physical (stored) head dimension of the attention variables, >= cfg.head_size (ParamStore: zero padding);
    filt_phys: physical channel count of the convolutional subsampling, >= cfg.filters (glorot fans stay the reference's)."""
    d, H, dh, Cl = cfg.dmodel, cfg.num_heads, cfg.head_size, cfg.filters
    C = filt_phys or Cl
    Kk, V, E, P, J = cfg.kernel_size, cfg.vocab_size, cfg.embed_dim, cfg.rnn_units, cfg.joint_dim
    F2 = -(-(-(-cfg.num_feature_bins // 2)) // 2)
    HD = H * (head_phys or dh)
    s = []

    def add(name, shape, reg, init, fans=None):
        s.append((name, tuple(shape), reg, init, fans))

    if getattr(cfg, "encoder", "conformer") == "contextnet":
        contextnet_specs(cfg, add)
        return _tail_specs(cfg, add, s)
    add("enc/sub/conv0/w", (3, 3, 1, C), True, "glorot", (9, 9 * Cl))
    add("enc/sub/conv0/b", (C,), False, "zeros")
    add("enc/sub/bn0/b", (C,), True, "zeros")
    add("enc/sub/bn0/g", (C,), True, "ones")
    add("enc/sub/conv1/w", (3, 3, C, C), True, "glorot", (9 * Cl, 9 * Cl))
    add("enc/sub/conv1/b", (C,), False, "zeros")
    add("enc/sub/bn1/b", (C,), True, "zeros")
    add("enc/sub/bn1/g", (C,), True, "ones")
    add("enc/linear/w", (F2 * C, d), True, "glorot", (F2 * Cl, d))
    add("enc/linear/b", (d,), False, "zeros")
    per_layer_bias = bool(getattr(cfg, "mhsam_use_attention_bias", False))
    if not per_layer_bias:
        add("enc/u", (HD,), False, "zeros")
        add("enc/v", (HD,), False, "zeros")
    for i in range(cfg.num_blocks):
        p = f"enc/block{i}/"
        for ff in ("ff1/", "ff2/"):
            add(p + ff + "ln/g", (d,), True, "ones"); add(p + ff + "ln/b", (d,), True, "zeros")
            add(p + ff + "d1/w", (d, cfg.ffm_scale * d), True, "glorot"); add(p + ff + "d1/b", (cfg.ffm_scale * d,), False, "zeros")
            add(p + ff + "d2/w", (cfg.ffm_scale * d, d), True, "glorot"); add(p + ff + "d2/b", (d,), False, "zeros")
            if ff == "ff1/":
                m = p + "mhsa/"
                add(m + "ln/g", (d,), True, "ones"); add(m + "ln/b", (d,), True, "zeros")
                add(m + "qkv/w", (d, 3 * HD), True, "glorot", (H * d, dh * d)); add(m + "qkv/b", (3 * HD,), False, "zeros")
                add(m + "pos/w", (d, HD), True, "glorot", (H * d, dh * d)); add(m + "pos/b", (HD,), False, "zeros")
                add(m + "o/w", (HD, d), True, "glorot", (dh * H, d * H)); add(m + "o/b", (d,), False, "zeros")
                if per_layer_bias:  # multihead_attention.py:522-538 (bias_regularizer: not regularised)
                    add(m + "u", (HD,), False, "zeros"); add(m + "v", (HD,), False, "zeros")
                c = p + "conv/"
                add(c + "ln/g", (d,), True, "ones"); add(c + "ln/b", (d,), True, "zeros")
                add(c + "pw1/w", (d, 2 * d), True, "glorot"); add(c + "pw1/b", (2 * d,), False, "zeros")
                add(c + "dw/w", (Kk, d), True, "glorot", (Kk * d, Kk)); add(c + "dw/b", (d,), False, "zeros")
                add(c + "bn/b", (d,), True, "zeros"); add(c + "bn/g", (d,), True, "ones")
                add(c + "pw2/w", (d, d), True, "glorot"); add(c + "pw2/b", (d,), False, "zeros")
        add(p + "ln/g", (d,), True, "ones"); add(p + "ln/b", (d,), True, "zeros")
    return _tail_specs(cfg, add, s)


def _tail_specs(cfg, add, s):
    """prediction + joint networks (shared by every encoder family)"""
    d, V, E, P, J = cfg.dmodel, cfg.vocab_size, cfg.embed_dim, cfg.rnn_units, cfg.joint_dim
    if getattr(cfg, "head", "transducer") == "ctc":  # ConformerDecoder (models/ctc/conformer.py:21-47): Dense(vocab_size) "logits"
        add("dec/logits/w", (d, V), True, "glorot"); add("dec/logits/b", (V,), False, "zeros")
        return s
    add("pred/emb", (V, E), True, "embed")
    add("pred/lstm/k", (E, 4 * P), True, "glorot")
    add("pred/lstm/rk", (P, 4 * P), False, "orth")
    add("pred/lstm/b", (4 * P,), False, "lstm_bias")
    if getattr(cfg, "prediction_layer_norm", True):
        add("pred/ln/g", (P,), True, "ones"); add("pred/ln/b", (P,), True, "zeros")
    add("joint/enc/w", (d, J), True, "glorot"); add("joint/enc/b", (J,), False, "zeros")
    add("joint/pred/w", (P, J), True, "glorot"); add("joint/pred/b", (J,), False, "zeros")
    add("joint/vocab/w", (J, V), True, "glorot"); add("joint/vocab/b", (V,), False, "zeros")
    return s


def contextnet_modules(cfg):
    """[(prefix, Cin, Cout, kernel, stride, activation)] of every ConvModule in forward order, per block, plus SE sizes:
    returns a list of blocks: dict(convs=[...], se=(prefix, C), res=module-or-None, C=Cout, stride=s)."""
    blocks, cin = [], cfg.num_feature_bins
    for i, b in enumerate(cfg.contextnet_blocks):
        C = int(b["filters"] * cfg.contextnet_alpha)
        K, s, n = int(b["kernel_size"]), int(b.get("strides", 1)), int(b["nlayers"])
        p = f"enc/block{i}/"
        convs, c = [], cin
        for j in range(n - 1):
            convs.append((p + f"conv{j}", c, C, K, 1, "swish"))
            c = C
        convs.append((p + f"conv{n - 1}", c, C, K, s, "swish"))           # last_conv carries the block's stride
        convs.append((p + "se/conv", C, C, K, 1, "swish"))               # SEModule's own conv module
        res = (p + "res", cin, C, K, s, "none") if b.get("residual", True) else None
        blocks.append(dict(prefix=p, convs=convs, res=res, C=C, stride=s))
        cin = C
    return blocks


def contextnet_specs(cfg, add):
    for blk in contextnet_modules(cfg):
        for (m, ci, co, K, _s, _a) in blk["convs"] + ([blk["res"]] if blk["res"] else []):
            add(m + "/dw", (K, ci), True, "glorot", (K * ci, K))
            add(m + "/pw/w", (ci, co), True, "glorot")
            add(m + "/pw/b", (co,), False, "zeros")
            add(m + "/bn/b", (co,), True, "zeros")
            add(m + "/bn/g", (co,), True, "ones")
        C = blk["C"]
        p = blk["prefix"] + "se/"
        add(p + "fc1/w", (C, C // 8), True, "glorot"); add(p + "fc1/b", (C // 8,), False, "zeros")
        add(p + "fc2/w", (C // 8, C), True, "glorot"); add(p + "fc2/b", (C,), False, "zeros")


def bn_names(cfg):
    if getattr(cfg, "encoder", "conformer") == "contextnet":
        out = []
        for blk in contextnet_modules(cfg):
            out += [m[0] + "/bn" for m in blk["convs"]] + ([blk["res"][0] + "/bn"] if blk["res"] else [])
        return out
    # encoder_convm_dw_norm_type: layer (encoders/conformer.py:334-340) puts a LayerNormalization in the depthwise-norm slot: no moving
    # statistics exist for it (a keras LayerNormalization holds gamma and beta only)
    blocks = [] if getattr(cfg, "convm_dw_norm", "batch") == "layer" else [f"enc/block{i}/conv/bn" for i in range(cfg.num_blocks)]
    # encoder_subsampling.norms: layer (subsampling.py:205-213): LayerNormalization in the bn0 / bn1 slots, no moving statistics either
    sub = [] if getattr(cfg, "sub_norm", "batch") == "layer" else ["enc/sub/bn0", "enc/sub/bn1"]
    return sub + blocks


class ParamStore:
    ALIGN = 64  # elements; keeps every variable 256-B aligned in f32 and 128-B aligned in bf16

    def __init__(self, cfg, device, dtype, seed=0, head_phys=None, filt_phys=None):
        """head_phys: PHYSICAL head dimension of the attention variables (q/k/v/position projections, output projection rows, u / v
        biases).  A bf16 model whose heads are narrower than the fused attention kernels' 64 (the reference ships head 36 and 44:
        small.yml.j2:39, ctc/conformer/small.yml.j2:39) stores those variables zero-padded to 64 per head: the padded q / k / v /
        position columns are exactly zero, so every score, probability and context value is unchanged, the padded context columns are
        zero, and every gradient into a padded element is exactly zero (dq_pad = dS k_pad = 0, dk_pad = dS^T q_pad = 0,
        dv_pad = P^T (dy Wo_pad^T) = 0), so Adam / L2 / weight decay leave the padding at zero forever.  import_keras / export_keras
        speak the reference's (unpadded) layouts; the softmax scale stays 1 / sqrt(cfg.head_size).
        filt_phys: PHYSICAL channel count of Conv2dSubsampling (subsampling.py:163-254).  The K-segmented conv2 GEMMs over the haloed
        space-to-depth layout need whole 64-channel slabs; the reference's small model has 144 filters (small.yml.j2:27), stored here as
        192 with zero weights / biases / BatchNorm gamma and beta in the padded channels: those channels carry exact zeros forward
        (conv of zero weights, x_hat = 0, swish(0) = 0) and exact zero gradients backward (every path into them multiplies a zero
        activation or a zero weight), so they stay zero; the linear layer's padded input rows are zero as well."""
        self.cfg, self.device, self.dtype = cfg, device, dtype
        self.head_phys = int(head_phys or cfg.head_size)
        if self.head_phys < cfg.head_size:
            raise ValueError("head_phys must be >= head_size")
        conformer = getattr(cfg, "encoder", "conformer") == "conformer"
        self.filt_phys = int(filt_phys or cfg.filters) if conformer else int(cfg.filters)
        if self.filt_phys < cfg.filters:
            raise ValueError("filt_phys must be >= filters")
        specs = param_specs(cfg, self.head_phys, self.filt_phys)
        ordered = [x for x in specs if x[2]] + [x for x in specs if not x[2]]
        # PHYSICAL order = the canonical order above (regularised variables first, forward order) except that the Conformer blocks' small
        # regularised variables whose gradients the train step finishes AFTER the last block's backward (LayerNorm gamma / beta, the
        # positional projection, the depthwise kernel: conformer._deferred_block_grads) sit together in ONE region behind the last block:
        # a block's remaining variables are then a contiguous slice that is final when the block's backward is (the data-parallel bucket
        # released right behind it), and the deferred region is one more slice released once, after the deferred launches - so a
        # data-parallel rank keeps the hoisted step.  `names`, the initialisation order and the logical layout of
        # state files stay canonical: only offsets move.
        phys, late = [], []
        for x in ordered:
            if x[2] and conformer and deferred_grad(x[0]):
                late.append(x)
                continue
            if late is not None and x[2] and not x[0].startswith("enc/"):
                phys, late = phys + late, None  # first regularised variable behind the encoder: the deferred region goes in front of it
            if late is not None and not x[2]:
                phys, late = phys + late, None  # (an encoder-only store)
            phys.append(x)
        if late:
            phys += late
        self.offsets, self.shapes = {}, {}
        self.defer_lo = self.defer_hi = None
        off = 0
        for name, shape, reg, init, fans in phys:
            if reg and conformer and deferred_grad(name) and self.defer_lo is None:
                self.defer_lo = off
            self.offsets[name] = off
            self.shapes[name] = shape
            off += -(-int(np.prod(shape)) // self.ALIGN) * self.ALIGN
            if reg:
                self.n_reg = off
                if conformer and deferred_grad(name):
                    self.defer_hi = off
        self.n = off
        self.names = [x[0] for x in ordered]
        self._views = {}
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.adam_m = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.adam_v = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.shadow = self.flat if dtype == torch.float32 else torch.zeros(self.n, dtype=dtype, device=device)
        # non-trainable BatchNorm moving statistics (keras: moving_mean zeros, moving_variance ones)
        self.state, self.shapes_state = {}, {}
        for nm in bn_names(cfg):
            C = self.shapes[nm + "/g"][0]
            self.state[nm + "/mm"] = torch.zeros(C, dtype=torch.float32, device=device)
            self.state[nm + "/mv"] = torch.ones(C, dtype=torch.float32, device=device)
            self.shapes_state[nm + "/mm"] = self.shapes_state[nm + "/mv"] = (C,)
        self._init(ordered, seed)
        self.refresh_shadow()

    def alias(self, dtype=torch.float32):
        """A second store over the SAME master / gradient / optimizer / BatchNorm-state buffers whose compute-type copy is the f32
        master itself: what a bf16-trained model's f32 inference twin reads (conformer.ConformerTransducer.inference_twin)."""
        if dtype != torch.float32:
            raise ValueError("only the f32 master can be aliased")
        o = object.__new__(ParamStore)
        o.__dict__.update(self.__dict__)
        o.dtype, o.shadow, o._views = dtype, self.flat, {}
        return o

    # ------------------------------------------------------------------ views
    def _view(self, buf, name, two_d=False):
        # views into the flat buffers never move: build each once (this is on the per-launch host path)
        key = (id(buf), name, two_d)
        v = self._views.get(key)
        if v is None:
            o, shp = self.offsets[name], self.shapes[name]
            v = buf[o:o + int(np.prod(shp))].view(*shp)
            if two_d:
                v = v.view(-1, shp[-1])
            self._views[key] = v
        return v

    def p(self, name):  # f32 master
        return self._view(self.flat, name)

    def w(self, name):  # compute-dtype copy read by the GEMMs
        return self._view(self.shadow, name)

    def g(self, name):  # f32 gradient
        return self._view(self.grad, name)

    def w2d(self, name):
        return self._view(self.shadow, name, True)

    def p2d(self, name):  # f32 master as a matrix (the greedy search's prediction / joint arithmetic stays f32)
        return self._view(self.flat, name, True)

    def g2d(self, name):
        return self._view(self.grad, name, True)

    # ------------------------------------------------------------------ head padding
    def _head_view(self, t, name, dh):
        """view of a head-carrying variable with the head axis split as [..., H, dh, ...] (dh = physical or logical)"""
        H, d = self.cfg.num_heads, self.cfg.dmodel
        if name.endswith("qkv/w"):
            return t.reshape(d, 3, H, dh)
        if name.endswith("qkv/b"):
            return t.reshape(3, H, dh)
        if name.endswith("pos/w"):
            return t.reshape(d, H, dh)
        if name.endswith("o/w"):
            return t.reshape(H, dh, d)
        return t.reshape(H, dh)  # pos/b, u, v

    def _chan_view(self, t, name, C):
        """view of a subsampling variable with its channel axes explicit (C = physical or logical channel count)"""
        if name == "enc/sub/conv0/w":
            return t.reshape(3, 3, 1, C)
        if name == "enc/sub/conv1/w":
            return t.reshape(3, 3, C, C)
        if name == "enc/linear/w":
            return t.reshape(-1, C, self.cfg.dmodel)  # rows = (ff, c): math_util.merge_two_last_dims
        return t.reshape(C)

    def _chan_axes(self, name):
        return {"enc/sub/conv0/w": (3,), "enc/sub/conv1/w": (2, 3), "enc/linear/w": (1,)}.get(name, (0,))

    def _pad(self, name, t):
        """logical layout (head dim cfg.head_size, cfg.filters channels) -> stored layout (head_phys / filt_phys, zero padded)"""
        Cl, Cp = self.cfg.filters, self.filt_phys
        if Cp != Cl and chan_padded(name):
            v = self._chan_view(t, name, Cl)
            shp = list(v.shape)
            for ax in self._chan_axes(name):
                shp[ax] = Cp
            out = torch.zeros(shp, dtype=v.dtype, device=v.device)
            sl = out
            for ax in self._chan_axes(name):
                sl = sl.narrow(ax, 0, Cl)
            sl.copy_(v)
            return out.reshape(self.shapes[name] if name in self.shapes else self.shapes_state[name])
        dh, dp = self.cfg.head_size, self.head_phys
        if dp == dh or not head_padded(name):
            return t
        v = self._head_view(t, name, dh)
        ax = 1 if name.endswith("o/w") else v.dim() - 1
        shp = list(v.shape)
        shp[ax] = dp
        out = torch.zeros(shp, dtype=v.dtype, device=v.device)
        out.narrow(ax, 0, dh).copy_(v)
        return out.reshape(self.shapes[name])

    def _unpad(self, name, t):
        Cl, Cp = self.cfg.filters, self.filt_phys
        if Cp != Cl and chan_padded(name):
            v = self._chan_view(t, name, Cp)
            for ax in self._chan_axes(name):
                v = v.narrow(ax, 0, Cl)
            v = v.contiguous()
            return v.reshape(-1, self.cfg.dmodel) if name == "enc/linear/w" else v
        dh, dp = self.cfg.head_size, self.head_phys
        if dp == dh or not head_padded(name):
            return t
        v = self._head_view(t, name, dp)
        ax = 1 if name.endswith("o/w") else v.dim() - 1
        v = v.narrow(ax, 0, dh).contiguous()
        if name.endswith("qkv/w"):
            return v.reshape(self.cfg.dmodel, -1)
        if name.endswith("pos/w"):
            return v.reshape(self.cfg.dmodel, -1)
        if name.endswith("o/w"):
            return v.reshape(-1, self.cfg.dmodel)
        return v.reshape(-1)

    def to_logical(self, buf):
        """A flat parameter-shaped buffer (parameters, Adam moments, accumulated gradients) as ONE f32 CPU vector in the LOGICAL layout
        (head size cfg.head_size, cfg.filters channels): independent of the zero padding the device layout carries, so a state file
        resumes into a model built with another storage type or other TFASR_HEAD_PAD / TFASR_FILTER_PAD settings."""
        return torch.cat([self._unpad(name, self._view(buf, name).detach().float().cpu()).reshape(-1) for name in self.names])

    def from_logical(self, vec, buf):
        """inverse of to_logical: scatter (and zero-pad) `vec` into the device buffer `buf`"""
        vec = torch.as_tensor(vec).float().cpu().reshape(-1)
        off = 0
        for name in self.names:
            n = self._unpad(name, torch.empty(self.shapes[name])).numel()
            if off + n > vec.numel():
                raise ValueError("logical state vector is shorter than this model's parameters")
            piece = vec[off:off + n]
            off += n
            phys = self._pad(name, piece.reshape(self._unpad(name, torch.empty(self.shapes[name])).shape))
            self._view(buf, name).copy_(phys.reshape(self.shapes[name]).to(buf.device))
        if off != vec.numel():
            raise ValueError("logical state vector is longer than this model's parameters")

    def rezero_pads(self, buf=None):
        """restore the zero padding of `buf` (default: the master parameters) after something wrote whole-buffer noise into it"""
        dh, dp = self.cfg.head_size, self.head_phys
        Cl, Cp = self.cfg.filters, self.filt_phys
        if dp == dh and Cp == Cl:
            return
        buf = self.flat if buf is None else buf
        for name in self.names:
            if dp != dh and head_padded(name):
                v = self._head_view(self._view(buf, name), name, dp)
                v.narrow(1 if name.endswith("o/w") else v.dim() - 1, dh, dp - dh).zero_()
            if Cp != Cl and chan_padded(name):
                v = self._chan_view(self._view(buf, name), name, Cp)
                for ax in self._chan_axes(name):
                    v.narrow(ax, Cl, Cp - Cl).zero_()

    def refresh_shadow(self):
        if self.shadow is not self.flat:
            K.cast(self.flat, self.shadow)

    def num_trainable(self):
        """number of trainable variables' elements in the REFERENCE's layouts (the zero padding of the heads is not a parameter)"""
        if self.head_phys != self.cfg.head_size or self.filt_phys != self.cfg.filters:
            return sum(int(np.prod(x[1])) for x in param_specs(self.cfg))
        return sum(int(np.prod(s)) for s in self.shapes.values())

    # ------------------------------------------------------------------ init (Keras defaults)
    def _init(self, ordered, seed):
        rng = np.random.default_rng(seed)
        host = np.zeros(self.n, np.float32)
        logical = {x[0]: x[1] for x in param_specs(self.cfg)} if (self.head_phys != self.cfg.head_size or self.filt_phys != self.cfg.filters) else {}
        for name, shape, reg, init, fans in ordered:
            n = int(np.prod(shape))
            stored = shape
            shape = logical.get(name, shape)  # the draws are those of the unpadded model (same seed -> same weights), padded afterwards
            if init == "zeros":
                w = np.zeros(shape, np.float32)
            elif init == "ones":
                w = np.ones(shape, np.float32)
            elif init == "embed":
                w = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
            elif init == "orth":
                P = shape[0]
                q, r = np.linalg.qr(rng.standard_normal((shape[1], P)))
                w = (q * np.sign(np.diag(r))).T.astype(np.float32)
            elif init == "lstm_bias":
                P = shape[0] // 4
                w = np.zeros(shape, np.float32)
                w[P:2 * P] = 1.0  # unit_forget_bias
            else:
                fi, fo = fans if fans else (shape[0], shape[-1])
                lim = math.sqrt(6.0 / (fi + fo))
                w = rng.uniform(-lim, lim, shape).astype(np.float32)
            if shape != stored:
                w = self._pad(name, torch.from_numpy(w)).numpy()
            host[self.offsets[name]:self.offsets[name] + n] = w.reshape(-1)
        self.flat.copy_(torch.from_numpy(host))

    # ------------------------------------------------------------------ Keras-layout import / export
    def import_keras(self, W):
        """W: name -> tensor in the reference/Keras layouts (the oracle's naming, oracle/conformer_ref.py)."""
        H, dh = self.cfg.num_heads, self.cfg.head_size
        host = self.flat.cpu().clone()

        def put(name, t):
            t = self._pad(name, torch.as_tensor(t).detach().float().cpu()).reshape(-1)
            o = self.offsets[name]
            assert t.numel() == int(np.prod(self.shapes[name])), name
            host[o:o + t.numel()] = t

        for name in self.names:
            if name.endswith("qkv/w"):
                base = name[:-len("qkv/w")]
                put(name, torch.cat([W[base + k + "/w"].reshape(-1, H * dh) for k in ("q", "k", "v")], dim=1))
            elif name.endswith("qkv/b"):
                base = name[:-len("qkv/b")]
                put(name, torch.cat([W[base + k + "/b"].reshape(-1) for k in ("q", "k", "v")]))
            else:
                put(name, W[name])
        self.flat.copy_(host)
        for k in self.state:
            if k in W:
                self.state[k].copy_(self._pad(k, torch.as_tensor(W[k]).detach().float().cpu()))
        self.refresh_shadow()

    def export_keras(self, buf=None):
        """name -> CPU tensor in Keras layouts (q/k/v split into [d,H,dh], biases [H,dh], o [H,dh,d])."""
        H, dh, d = self.cfg.num_heads, self.cfg.head_size, self.cfg.dmodel
        buf = self.flat if buf is None else buf
        out = {}
        for name in self.names:
            t = self._unpad(name, self._view(buf, name).detach().float().cpu())
            if name.endswith("qkv/w"):
                base = name[:-len("qkv/w")]
                for i, k in enumerate(("q", "k", "v")):
                    out[base + k + "/w"] = t[:, i * H * dh:(i + 1) * H * dh].reshape(d, H, dh).clone()
            elif name.endswith("qkv/b"):
                base = name[:-len("qkv/b")]
                for i, k in enumerate(("q", "k", "v")):
                    out[base + k + "/b"] = t[i * H * dh:(i + 1) * H * dh].reshape(H, dh).clone()
            elif name.endswith("pos/w"):
                out[name] = t.reshape(d, H, dh).clone()
            elif name.endswith("pos/b") or name in ("enc/u", "enc/v") or name.endswith("mhsa/u") or name.endswith("mhsa/v"):
                out[name] = t.reshape(H, dh).clone()
            elif name.endswith("mhsa/o/w"):
                out[name] = t.reshape(H, dh, d).clone()
            else:
                out[name] = t.clone()
        if buf is self.flat:
            for k, v in self.state.items():
                out[k] = self._unpad(k, v.detach().cpu().clone())
        return out
