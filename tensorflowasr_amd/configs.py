"""Model configuration with the reference's keyword surface.

`ConformerConfig.from_reference(config)` accepts the `model_config.config` mapping of
examples/models/transducer/conformer/small.yml.j2:3-69 (= the kwargs of
tensorflow_asr/models/transducer/conformer.py:23-79) unchanged.  `conformer_s()` is that file's values;
`conformer_m()` is the Conformer-M of arXiv:2005.08100 Table 1 expressed through the same keys (the reference ships
no M config: SURVEY.md §0.1).
"""
import math
from dataclasses import dataclass, field


def speech_kwargs(sc):
    """`speech_config` of the reference's YAML (FeatureExtraction.__init__, models/layers/feature_extraction.py:32-130) -> the front-end
    fields of ConformerConfig.  Options the MI355X front end does not implement fail loudly instead of being ignored; the epsilon range
    check is the reference's (feature_extraction.py:113)."""
    sc = dict(sc or {})
    only = {"feature_type": ("log_mel_spectrogram",), "pad_end": (True,), "use_librosa_like_stft": (False,), "log_base": ("e",),
            "normalize_signal": (False,), "normalize_zscore": (False,), "normalize_min_max": (False,), "padding": (0,)}
    for k, ok in only.items():
        if k in sc and sc[k] not in ok:
            raise NotImplementedError(f"speech_config.{k}={sc[k]!r}: only {ok} is on the MI355X hot path")
    nfft = sc.get("nfft", 512)
    if nfft is None:  # the reference falls back to the frame length (feature_extraction.py:121)
        nfft = int(round(sc.get("sample_rate", 16000) * sc.get("frame_ms", 25) / 1000.0))
    if nfft != 512 or int(round(sc.get("sample_rate", 16000) * sc.get("frame_ms", 25) / 1000.0)) > 512:
        raise NotImplementedError(f"speech_config: nfft={nfft} / frame_ms={sc.get('frame_ms', 25)}: the front-end kernel is a 512-point transform (frames of up to 512 samples)")
    eps = float(sc.get("epsilon", 1e-6))
    assert eps > 1e-9 and eps <= 0.001, "epsilon must be in (1e-9, 0.001]"
    return dict(sample_rate=sc.get("sample_rate", 16000), frame_ms=sc.get("frame_ms", 25), stride_ms=sc.get("stride_ms", 10), nfft=nfft,
                num_feature_bins=sc.get("num_feature_bins", 80), preemphasis=sc.get("preemphasis", 0.97), epsilon=eps,
                lower_edge_hertz=float(sc.get("lower_edge_hertz", 0.0)), upper_edge_hertz=float(sc.get("upper_edge_hertz", 8000.0)))


@dataclass
class ConformerConfig:
    # speech_config (models/layers/feature_extraction.py:34-55)
    sample_rate: int = 16000
    frame_ms: int = 25
    stride_ms: int = 10
    nfft: int = 512
    num_feature_bins: int = 80
    preemphasis: float = 0.97
    epsilon: float = 1e-6
    lower_edge_hertz: float = 0.0
    upper_edge_hertz: float = 8000.0
    # augmentation (small.yml.j2:11-24)
    time_masking: dict = field(default_factory=lambda: dict(prob=1.0, num_masks=10, mask_factor=-1, p_upperbound=0.05, mask_value=0))
    freq_masking: dict = field(default_factory=lambda: dict(prob=1.0, num_masks=1, mask_factor=27, mask_value=0))
    # encoder
    filters: int = 144
    dmodel: int = 144
    num_blocks: int = 16
    head_size: int = 36
    num_heads: int = 4
    kernel_size: int = 31
    ffm_scale: int = 4
    ffm_residual: float = 0.5
    mhsam_residual: float = 1.0
    convm_residual: float = 1.0
    dropout: float = 0.1
    use_attention_auto_mask: bool = True
    # streaming Conformer (examples/models/transducer/conformer/small-streaming.yml.j2:33,38-39): chunked attention mask
    # (multihead_attention.py:104-143,331-345) and LayerNormalization after the depthwise conv (encoders/conformer.py:334-340)
    chunk_size: int = None
    history_size: int = None
    convm_dw_norm: str = "batch"
    sub_norm: str = "batch"  # Conv2dSubsampling `norms` (subsampling.py:197-213): "batch" (small.yml.j2) or "layer" (small-streaming.yml.j2)
    # encoder_mhsam_use_attention_bias: True (examples/models/ctc/conformer/small.yml.j2:45): per-layer content / positional
    # attention biases (multihead_attention.py:522-538) instead of the encoder-level shared pair (encoders/conformer.py:647-663)
    mhsam_use_attention_bias: bool = False
    # model head: "transducer" (prediction + joint networks) or "ctc" (models/ctc/conformer.py:21-47: one Dense(vocab) decoder)
    head: str = "transducer"
    # prediction / joint
    embed_dim: int = 320
    rnn_units: int = 320
    joint_dim: int = 320
    vocab_size: int = 1000
    blank: int = 0
    l2: float = 1e-6
    prediction_layer_norm: bool = True
    # encoder family: "conformer" (default) or "contextnet" (models/encoders/contextnet.py)
    encoder: str = "conformer"
    contextnet_blocks: list = None   # [dict(nlayers, kernel_size, filters, strides, residual)], the yml's encoder_blocks
    contextnet_alpha: float = 1.0

    @property
    def frame_length(self):
        return int(round(self.sample_rate * self.frame_ms / 1000.0))

    @property
    def frame_step(self):
        return int(round(self.sample_rate * self.stride_ms / 1000.0))

    @property
    def time_reduction_factor(self):
        if self.encoder == "contextnet":
            f = 1
            for b in self.contextnet_blocks:
                f *= int(b.get("strides", 1))
            return f
        return 4

    @classmethod
    def from_reference(cls, config: dict, class_name: str = None):
        """Map the reference's Conformer kwargs (models/transducer/conformer.py:23-79; models/ctc/conformer.py:57-100 when
        `class_name` is the YAML's "tensorflow_asr.models.ctc.conformer>Conformer") onto this dataclass."""
        c = dict(config)
        sc = dict(c.get("speech_config", {}))
        aug = (sc.get("augmentation_config") or {}).get("feature_augment", {}) or {}
        sub = (c.get("encoder_subsampling") or {}).get("config", {})
        unsupported = {
            "encoder_mha_type": ("relmha",), "encoder_padding": ("causal",), "prediction_rnn_type": ("lstm",),
            "prediction_num_rnns": (1,), "joint_activation": ("tanh",), "joint_mode": ("add",),
            "encoder_convm_dw_norm_type": ("batch", "layer"), "prediction_label_encode_mode": ("embedding",),
            "encoder_memory_length": (None,), "encoder_use_attention_causal_mask": (False,),
            # constructor options of models/transducer/conformer.py:23-77 (models/ctc/conformer.py for the CTC head) that change the
            # network and are not built: a non-default value must not be ignored silently
            "encoder_interleave_relpe": (True,), "encoder_mhsam_causal": (False,), "encoder_convm_scale_factor": (2,),
            "encoder_convm_use_group_conv": (False,), "encoder_module_norm_position": ("pre",), "encoder_block_norm_position": ("post",),
            "encoder_trainable": (True,), "prediction_trainable": (True,), "joint_trainable": (True,),
            "prediction_projection_units": (0,), "prejoint_encoder_linear": (True,), "prejoint_prediction_linear": (True,),
            "postjoint_linear": (False,), "bias_regularizer": (None,), "activity_regularizer": (None,), "recurrent_regularizer": (None,),
        }
        if "prediction_layer_norm" in c and not c["prediction_layer_norm"] and not (class_name and ".ctc." in class_name):
            raise NotImplementedError("prediction_layer_norm=False: the Conformer transducer's prediction network ends in a LayerNorm here "
                                      "(base_transducer.py:123-132 with the YAML default)")
        for k, ok in unsupported.items():
            if k in c and c[k] not in ok:
                raise NotImplementedError(f"{k}={c[k]!r}: only {ok} is on the MI355X hot path")
        # A key the mapping omits takes Conv2dSubsampling's OWN default (subsampling.py:163-176: strides [[2, 1], [2, 1]], kernels
        # [[3, 3], [3, 3]], paddings causal, norms none, activations relu) - which is not the network of small.yml.j2 and is not built, so
        # an incomplete mapping raises instead of silently becoming BatchNorm + swish.  No `encoder_subsampling` at all (the
        # reference's constructor requires one, models/transducer/conformer.py:28): the shipped small.yml.j2 block.
        ref_def = dict(kernels=[[3, 3], [3, 3]], strides=[[2, 1], [2, 1]], paddings=["causal", "causal"], norms=["none", "none"],
                       activations=["relu", "relu"])
        yml_def = dict(kernels=[3, 3], strides=[2, 2], paddings=["causal", "causal"], norms=["batch", "batch"], activations=["swish", "swish"])
        getd = (lambda k: sub.get(k, ref_def[k])) if sub else (lambda k: yml_def[k])

        def _pairs(v):  # [3, 3] (both layers 3 -> keras broadcasts an int to both axes) and [[3, 3], [3, 3]] describe the same convolutions
            return [list(x) if isinstance(x, (list, tuple)) else [x, x] for x in v]

        if _pairs(getd("kernels")) != [[3, 3], [3, 3]] or _pairs(getd("strides")) != [[2, 2], [2, 2]]:
            raise NotImplementedError(f"encoder_subsampling kernels={getd('kernels')} strides={getd('strides')}: only the 3x3 stride-2 causal "
                                      f"Conv2dSubsampling of small.yml.j2:26-33 is supported")
        norms = [str(n) for n in getd("norms")]
        if norms not in (["batch", "batch"], ["layer", "layer"]):
            raise NotImplementedError(f"encoder_subsampling.norms={norms}: both blocks 'batch' (small.yml.j2:31) or both 'layer' "
                                      f"(small-streaming.yml.j2) are built (subsampling.py:197-213)")
        if [str(a) for a in getd("activations")] not in (["swish", "swish"], ["silu", "silu"]):
            raise NotImplementedError(f"encoder_subsampling.activations={getd('activations')}: only swish (small.yml.j2:32)")
        if [str(a) for a in getd("paddings")] != ["causal", "causal"]:
            raise NotImplementedError(f"encoder_subsampling.paddings={getd('paddings')}: only causal (small.yml.j2:30)")
        reg = c.get("kernel_regularizer") or {}
        l2 = float((reg.get("config") or {}).get("l2", 1e-6)) if isinstance(reg, dict) else 1e-6
        kw = dict(
            **speech_kwargs(sc),
            filters=(sub.get("filters") or [c.get("encoder_dmodel", 144)])[0], dmodel=c.get("encoder_dmodel", 144),
            num_blocks=c.get("encoder_num_blocks", 16), head_size=c.get("encoder_head_size", 36),
            num_heads=c.get("encoder_num_heads", 4), kernel_size=c.get("encoder_kernel_size", 31),
            ffm_scale=c.get("encoder_ffm_scale_factor", 4), ffm_residual=c.get("encoder_ffm_residual_factor", 0.5),
            mhsam_residual=c.get("encoder_mhsam_residual_factor", 1.0), convm_residual=c.get("encoder_convm_residual_factor", 1.0),
            dropout=c.get("encoder_dropout", 0.1), use_attention_auto_mask=c.get("encoder_use_attention_auto_mask", True),
            embed_dim=c.get("prediction_embed_dim", 512), rnn_units=c.get("prediction_rnn_units", 320),
            joint_dim=c.get("joint_dim", 1024), vocab_size=int(c.get("vocab_size", 1000)), blank=c.get("blank", 0), l2=l2,
            chunk_size=c.get("encoder_chunk_size"), history_size=c.get("encoder_history_size"),
            convm_dw_norm=c.get("encoder_convm_dw_norm_type", "batch"), sub_norm=norms[0],
            mhsam_use_attention_bias=bool(c.get("encoder_mhsam_use_attention_bias", False)))
        if class_name and ".ctc." in class_name:
            kw["head"] = "ctc"
        if (kw["chunk_size"] is None) != (kw["history_size"] is None):  # multihead_attention.py:339 needs both
            kw["chunk_size"] = kw["history_size"] = None
        if "time_masking" in aug:
            kw["time_masking"] = dict(aug["time_masking"])
        if "freq_masking" in aug:
            kw["freq_masking"] = dict(aug["freq_masking"])
        return cls(**kw)


def conformer_ctc_s(vocab_size=1000, **over):
    """examples/models/ctc/conformer/small.yml.j2:26-46: d=176, 4 heads of 44, per-layer attention biases, Dense(vocab) decoder."""
    kw = dict(filters=176, dmodel=176, head_size=44, num_heads=4, mhsam_use_attention_bias=True, head="ctc", vocab_size=vocab_size)
    kw.update(over)
    return ConformerConfig(**kw)


def conformer_s(vocab_size=1000, **over):
    return ConformerConfig(vocab_size=vocab_size, **over)


def conformer_m(vocab_size=1000, **over):
    kw = dict(filters=256, dmodel=256, head_size=64, num_heads=4, embed_dim=640, rnn_units=640, joint_dim=640, vocab_size=vocab_size)
    kw.update(over)
    return ConformerConfig(**kw)


CONTEXTNET_BLOCKS = (  # examples/models/transducer/contextnet/small.yml.j2:25-190 (nlayers, kernel, filters, stride, residual)
    [(1, 5, 256, 1, False)] + [(5, 5, 256, 1, True)] * 2 + [(5, 5, 256, 2, True)] + [(5, 5, 256, 1, True)] * 3 + [(5, 5, 256, 2, True)]
    + [(5, 5, 256, 1, True)] * 3 + [(5, 5, 512, 1, True)] * 3 + [(5, 5, 512, 2, True)] + [(5, 5, 512, 1, True)] * 7 + [(1, 5, 640, 1, False)])


def _cn_blocks(spec):
    return [dict(nlayers=n, kernel_size=k, filters=f, strides=s, residual=r) for n, k, f, s, r in spec]


def contextnet(vocab_size=1000, alpha=0.5, **over):
    """The reference's ContextNet transducer (contextnet/small.yml.j2: alpha 0.5, 23 blocks, time reduction 8, prediction
    LSTM 512 without LayerNorm, joint 512); alpha 1 / 2 give the paper's M / L widths."""
    blocks = _cn_blocks(CONTEXTNET_BLOCKS)
    kw = dict(encoder="contextnet", contextnet_blocks=blocks, contextnet_alpha=alpha, dmodel=int(blocks[-1]["filters"] * alpha), embed_dim=640,
              rnn_units=512, joint_dim=512, prediction_layer_norm=False, vocab_size=vocab_size, dropout=0.0)
    kw.update(over)
    return ConformerConfig(**kw)


def contextnet_from_reference(config: dict):
    """The reference's ContextNet kwargs (models/transducer/contextnet.py:23-52; contextnet/small.yml.j2:1-220) -> config."""
    c = dict(config)
    for k, ok in {"prediction_rnn_type": ("lstm",), "prediction_num_rnns": (1,), "joint_activation": ("tanh",), "joint_mode": ("add",),
                  "prediction_label_encode_mode": ("embedding",), "prediction_projection_units": (0,)}.items():
        if k in c and c[k] not in ok:
            raise NotImplementedError(f"{k}={c[k]!r}: only {ok} is on the MI355X hot path")
    blocks = []
    for b in c["encoder_blocks"]:
        if str(b.get("activation", "silu")) not in ("silu", "swish") or str(b.get("padding", "causal")) != "causal":
            raise NotImplementedError(f"ContextNet block {b}: only silu activations / causal padding are built")
        blocks.append(dict(nlayers=int(b["nlayers"]), kernel_size=int(b["kernel_size"]), filters=int(b["filters"]), strides=int(b.get("strides", 1)),
                           residual=bool(b.get("residual", True))))
    alpha = float(c.get("encoder_alpha", 0.5))
    sc = dict(c.get("speech_config", {}))
    aug = (sc.get("augmentation_config") or {}).get("feature_augment", {}) or {}
    reg = c.get("kernel_regularizer") or {}
    kw = dict(encoder="contextnet", contextnet_blocks=blocks, contextnet_alpha=alpha, dmodel=int(blocks[-1]["filters"] * alpha),
              embed_dim=c.get("prediction_embed_dim", 512), rnn_units=c.get("prediction_rnn_units", 320), joint_dim=c.get("joint_dim", 1024),
              prediction_layer_norm=bool(c.get("prediction_layer_norm", True)), vocab_size=int(c.get("vocab_size", 1000)), blank=c.get("blank", 0),
              dropout=0.0, l2=float((reg.get("config") or {}).get("l2", 1e-6)) if isinstance(reg, dict) else 1e-6,
              **speech_kwargs(sc))
    if "time_masking" in aug:
        kw["time_masking"] = dict(aug["time_masking"])
    if "freq_masking" in aug:
        kw["freq_masking"] = dict(aug["freq_masking"])
    return ConformerConfig(**kw)


def contextnet_tiny(vocab_size=29, **over):
    blocks = _cn_blocks([(1, 5, 32, 1, False), (3, 5, 32, 1, True), (3, 5, 32, 2, True), (2, 3, 48, 2, True), (1, 5, 64, 1, False)])
    kw = dict(encoder="contextnet", contextnet_blocks=blocks, contextnet_alpha=0.5, dmodel=32, embed_dim=24, rnn_units=24, joint_dim=40,
              prediction_layer_norm=False, vocab_size=vocab_size, dropout=0.0)
    kw.update(over)
    return ConformerConfig(**kw)


def conformer_tiny(vocab_size=29, **over):
    kw = dict(dropout=0.0, filters=32, dmodel=32, head_size=8, num_heads=4, embed_dim=24, rnn_units=24, joint_dim=40, num_blocks=2,
              kernel_size=7, vocab_size=vocab_size)
    kw.update(over)
    return ConformerConfig(**kw)


def transformer_schedule(step, dmodel, warmup_steps=10000, scale=2.0, max_lr=None, min_lr=None):
    """TransformerSchedule.__call__ (optimizers/schedules.py:28-37)."""
    step = float(step)
    # step 0 (keras' `iterations` at the first update): tf gives min(inf, 0) = 0
    lr = scale * dmodel ** -0.5 * min(step ** -0.5, step * warmup_steps ** -1.5) if step > 0 else 0.0
    if max_lr is not None:
        lr = min(max_lr, lr)
    if min_lr is not None:
        lr = max(min_lr, lr)
    return lr
