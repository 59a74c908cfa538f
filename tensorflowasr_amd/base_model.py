"""The reference's model surface (tensorflow_asr/models/base_model.py) on the HIP path: what a `tensorflow_asr` user calls on a model
object besides `__call__` / `train_step` / `recognize`, with the reference's names, argument meaning and error behaviour:

    tokenizer (property)                       base_model.py:41-47
    make(input_shape, prediction_shape, batch_size) -> TrainOutput of output SHAPES      :68-100
    compile(optimizer, output_shapes, ga_steps, gwn_config, gradn_config, ...)           :102-125, transducer/base_transducer.py:378-380
    test_step / predict_step                   :212-243
    save_weights / load_weights                :55-61   (`.weights.h5` = Keras 3 container, anything else = `.npz` under Keras paths)
    get_initial_tokens / _encoder_states / _decoder_states                                :316-323, base_transducer.py:466-470
    apply_gwn / remove_gwn (variational weight noise)                                     base_transducer.py:382-425, ctc/base_ctc.py:41-77
    gradient noise in _apply_gradients         :185-192
and `model_from_config({"class_name": "tensorflow_asr.models.transducer.conformer>Conformer", "config": {...}})`
(utils/keras_util.py model_from_config; YAML `model_config`, small.yml.j2:1-69).

Host-side logic only: arithmetic stays in libtfasr_hip.so (the noise is `tfasr_gauss_noise`).
"""
import math

import torch

from . import kernels as K
from .schemas import PredictInput, TrainData, TrainOutput

_COMPONENTS = {"encoder": ("enc/",), "predict_net": ("pred/",), "joint_net": ("joint/",), "decoder": ("dec/",)}


def _eval_number(v):
    """YAML numbers such as `max_lr: 0.05/(144**0.5)` arrive as strings (small.yml.j2:80).  Only arithmetic is accepted: the expression is
    walked node by node (numbers, + - * / ** and unary signs, sqrt(...)); anything else raises instead of being evaluated."""
    if not isinstance(v, str):
        return v
    import ast
    import operator as op

    ops = {ast.Add: op.add, ast.Sub: op.sub, ast.Mult: op.mul, ast.Div: op.truediv, ast.Pow: op.pow, ast.FloorDiv: op.floordiv, ast.Mod: op.mod}

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)) and not isinstance(n.value, bool):
            return n.value
        if isinstance(n, ast.BinOp) and type(n.op) in ops:
            return ops[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.UAdd, ast.USub)):
            x = ev(n.operand)
            return -x if isinstance(n.op, ast.USub) else x
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "sqrt" and len(n.args) == 1 and not n.keywords:
            return math.sqrt(ev(n.args[0]))
        raise ValueError(f"not a number expression: {v!r}")

    return float(ev(ast.parse(v.strip(), mode="eval")))


def optimizer_from_config(cfg, dmodel):
    """`learning_config.optimizer_config` ({class_name: Adam, config: {learning_rate: <float | {class_name: ...TransformerSchedule,
    config}>, beta_1, beta_2, epsilon, weight_decay}}, small.yml.j2:73-87) -> the optimizer dict of the train step."""
    if cfg is None:
        return None
    if not isinstance(cfg, dict):
        raise TypeError("optimizer must be None or the reference's optimizer_config mapping")
    name = str(cfg.get("class_name", "Adam"))
    if name.rsplit(">", 1)[-1].lower() != "adam":
        raise NotImplementedError(f"optimizer {name!r}: the shipped configs use Adam; only Adam is on the MI355X hot path")
    c = dict(cfg.get("config") or {})
    lr = c.get("learning_rate", 1e-3)
    if isinstance(lr, dict):
        sname = str(lr.get("class_name", "")).rsplit(">", 1)[-1]
        if sname != "TransformerSchedule":
            raise NotImplementedError(f"learning-rate schedule {sname!r}: only TransformerSchedule (optimizers/schedules.py:28-37) is built")
        sc = dict(lr.get("config") or {})
        sched = dict(dmodel=int(sc.get("dmodel", dmodel)), warmup_steps=int(sc.get("warmup_steps", 4000)), scale=float(sc.get("scale", 1.0)),
                     max_lr=None if sc.get("max_lr") is None else float(_eval_number(sc["max_lr"])))
        if sc.get("min_lr") is not None:
            sched["min_lr"] = float(_eval_number(sc["min_lr"]))
    else:
        sched = float(_eval_number(lr))
    return dict(beta1=float(c.get("beta_1", 0.9)), beta2=float(c.get("beta_2", 0.999)), eps=float(c.get("epsilon", 1e-7)),
                weight_decay=float(c.get("weight_decay") or 0.0), schedule=sched)


class BaseModel:
    """Mixin of the concrete models (ConformerTransducer, ConformerCTC, ContextNetTransducer)."""

    _tokenizer = None
    gwn_config = None
    gradn_config = None
    tfasr_loss = None
    _batch_size = None
    _per_replica_batch_size = None

    # ----------------------------------------------------------------------------------------------- tokenizer
    @property
    def tokenizer(self):
        return self._tokenizer

    @tokenizer.setter
    def tokenizer(self, tokenizer):
        self._tokenizer = tokenizer

    # ----------------------------------------------------------------------------------------------- make / compile
    def make(self, input_shape=[None], prediction_shape=[None], batch_size=None, **kwargs) -> TrainOutput:
        """BaseModel.make (base_model.py:68-100): records the (global / per-replica) batch size and returns the OUTPUT SHAPES of
        `__call__` per replica (None for unknown dimensions).  Nothing has to be built here: parameters exist from construction."""
        assert batch_size is not None and batch_size > 0
        world = int(self.dp.world)
        self._per_replica_batch_size = int(batch_size / world)
        self._batch_size = batch_size
        c = self.cfg
        n = input_shape[0] if len(input_shape) else None
        t_out = None
        if n is not None:
            t = -(-int(n) // c.frame_step)
            t_out = self._encoder_length(t)
        u1 = prediction_shape[0] if len(prediction_shape) else None
        b = self._per_replica_batch_size
        if getattr(c, "head", "transducer") == "ctc":
            logits = [b, t_out, c.vocab_size]
        else:
            logits = [b, t_out, u1, c.vocab_size]
        return TrainOutput(logits=logits, logits_length=[b])

    def _encoder_length(self, t):
        """frames -> encoder frames (math_util.conv_output_length "causal" twice for the Conformer; models override)."""
        return -(-(-(-int(t) // 2)) // 2)

    def compile(self, optimizer=None, output_shapes=None, loss=None, run_eagerly=False, ga_steps=None, gwn_config=None,
                gradn_config=None, **kwargs):
        """Transducer.compile / BaseModel.compile (base_transducer.py:378-380, base_model.py:102-125).  `optimizer` is the YAML's
        optimizer_config mapping (or None: keep the small.yml.j2 defaults); the loss object is built like the reference builds it."""
        from . import losses

        opt = optimizer_from_config(optimizer, self.cfg.dmodel) if isinstance(optimizer, dict) or optimizer is None else optimizer
        if opt is not None:
            if not isinstance(opt, dict) or "schedule" not in opt:
                raise TypeError("optimizer: pass the reference's optimizer_config mapping")
            self.optimizer = opt
        if isinstance(ga_steps, int) and ga_steps > 1:
            self.ga_steps = ga_steps
        else:
            self.ga_steps = 1
        self._ga_count = 0
        self.gwn_config = dict(gwn_config) if gwn_config else None
        self.gradn_config = dict(gradn_config) if gradn_config else None
        if loss is None:
            if getattr(self.cfg, "head", "transducer") == "ctc":
                loss = losses.CtcLoss(blank=self.blank, name="ctc_loss")
            else:
                loss = losses.RnntLoss(blank=self.blank, output_shapes=output_shapes, name="rnnt_loss")
        self.tfasr_loss = loss
        return self

    # ----------------------------------------------------------------------------------------------- noise hooks
    def _component_ranges(self, prefixes):
        """merged [lo, hi) element ranges of the flat parameter buffer holding the variables whose name starts with `prefixes`"""
        ps = self.ps
        spans = []
        for nm in ps.names:
            if nm.startswith(prefixes):
                lo = ps.offsets[nm]
                n = 1
                for s in ps.shapes[nm]:
                    n *= int(s)
                spans.append((lo, lo + -(-n // ps.ALIGN) * ps.ALIGN))
        spans.sort()
        out = []
        for lo, hi in spans:
            if out and out[-1][1] == lo:
                out[-1][1] = hi
            else:
                out.append([lo, hi])
        return [(a, b) for a, b in out]

    def apply_gwn(self):
        """Variational weight noise (base_transducer.py:382-405, ctc/base_ctc.py:41-58, utils/layer_util.py:42-52): from optimizer
        step `<part>_step` on, N(0, `<part>_stddev`) is added to every trainable weight of that part for the forward AND backward of
        this step.  Returns what remove_gwn needs (the original values)."""
        g = self.gwn_config
        if not g:
            return {}
        saved = {}
        self._gwn_epoch = getattr(self, "_gwn_epoch", 0) + 1
        for part, prefixes in _COMPONENTS.items():
            st, sd = g.get(part + "_step"), g.get(part + "_stddev")
            if st is None or sd is None or self.step < int(st):
                continue
            ranges = self._component_ranges(prefixes)
            if not ranges:
                continue
            keep = []
            for k, (lo, hi) in enumerate(ranges):
                view = self.ps.flat[lo:hi]
                keep.append((lo, hi, view.clone()))
                seed = (self._gwn_epoch * 64 + len(saved) * 8 + k) * 1_000_003 + 17 + ((int(self.dp.rank) & 0x7F) << 48)
                K.gauss_noise(view, float(sd), seed)
            saved[part] = keep
        if saved:
            self.ps.rezero_pads()  # the zero padding of narrow attention heads is not a weight
            self.ps.refresh_shadow()
        return saved

    def remove_gwn(self, original_weights):
        if not original_weights:
            return
        for keep in original_weights.values():
            for lo, hi, orig in keep:
                self.ps.flat[lo:hi].copy_(orig)
        self.ps.refresh_shadow()

    def _gradient_noise(self):
        """math_util.add_gauss_noise on every gradient once optimizer.iterations >= gradn_config["step"] (base_model.py:185-191; order
        pinned by tests/golden/wiring_train_step.npz: after the GA average, BEFORE `optimizer.apply`, whose cross-replica SUM then adds
        the replicas' independent draws).  This runs after the gradient all-reduce (the bucketed exchange overlaps the backward), so
        every rank adds the SAME draw - the seed carries no rank - of the sum of `world` independent N(0, stddev) draws, i.e.
        N(0, stddev * sqrt(world)): the replicas' parameters stay bit-identical and the update has the reference's
        distribution."""
        g = self.gradn_config
        if not g or self.step < int(g["step"]):
            return
        self._gradn_epoch = getattr(self, "_gradn_epoch", 0) + 1
        world = max(1, int(getattr(self.dp, "world", 1)))
        K.gauss_noise(self.ps.grad, float(g["stddev"]) * math.sqrt(world), self._gradn_epoch * 7_368_787 + 5)
        self.ps.rezero_pads(self.ps.grad)

    # ----------------------------------------------------------------------------------------------- steps
    @torch.no_grad()
    def test_step(self, data: TrainData):
        """BaseModel.test_step (base_model.py:212-233): forward with training=False + loss; {'loss': per-utterance costs [B]} like
        train_step here (the reference's metric is their mean)."""
        return {"loss": self.loss_and_backward(data, training=False, want_backward=False)}

    @torch.no_grad()
    def predict_step(self, data: TrainData):
        """BaseModel.predict_step (base_model.py:235-250): greedy and beam transcripts + the labels."""
        x, y_true = data
        batch_size = int(x.inputs.shape[0])
        inputs = PredictInput(inputs=x.inputs, inputs_length=x.inputs_length, previous_tokens=self.get_initial_tokens(batch_size=batch_size),
                              previous_encoder_states=self.get_initial_encoder_states(batch_size=batch_size),
                              previous_decoder_states=self.get_initial_decoder_states(batch_size=batch_size))
        _tokens = self.recognize(inputs=inputs).tokens
        _beam_tokens = self.recognize_beam(inputs=inputs).tokens
        return {"tokens": _tokens, "beam_tokens": _beam_tokens, "labels": y_true.labels}

    # ----------------------------------------------------------------------------------------------- states
    def get_initial_encoder_states(self, batch_size=1):
        return []

    # ----------------------------------------------------------------------------------------------- weights
    def save_weights(self, filepath, overwrite=True):
        """BaseModel.save_weights (base_model.py:55-57): `*.h5` -> Keras 3 `.weights.h5` container, otherwise `.npz`; both under the
        reference's variable paths / layouts."""
        import os

        from . import checkpoint

        filepath = str(filepath)
        if not overwrite and os.path.exists(filepath):
            raise FileExistsError(filepath)
        if filepath.endswith((".h5", ".hdf5")):
            return checkpoint.save_weights_h5(self, filepath)
        return checkpoint.save_weights(self, filepath)

    def load_weights(self, filepath, skip_mismatch=False, **kwargs):
        """BaseModel.load_weights (base_model.py:59-61).  skip_mismatch=True loads what matches and keeps the rest (`.npz`)."""
        from . import checkpoint

        filepath = str(filepath)
        if filepath.endswith((".h5", ".hdf5")):
            return checkpoint.load_weights_h5(self, filepath, strict=not skip_mismatch)
        return checkpoint.load_weights(self, filepath, strict=not skip_mismatch)


# ----------------------------------------------------------------------------------------------------- registry
def model_from_config(model_config, device=None, dtype=torch.bfloat16, seed=0, dp=None):
    """utils/keras_util.py model_from_config on the YAML's `model_config` ({class_name, config}): the reference resolves
    `tensorflow_asr.models.transducer.conformer>Conformer` through keras' registry; the same strings select the HIP models here."""
    from . import configs

    name = str(model_config["class_name"])
    conf = model_config.get("config") or {}
    mod, _, cls = name.partition(">")
    key = (mod.rsplit(".", 2)[-2:] if mod.count(".") >= 2 else [mod]) + [cls]
    key = ".".join(key)
    if key == "transducer.conformer.Conformer":
        from .conformer import ConformerTransducer

        return ConformerTransducer(configs.ConformerConfig.from_reference(conf, class_name=name), device, dtype=dtype, seed=seed, dp=dp)
    if key == "ctc.conformer.Conformer":
        from .ctc_model import ConformerCTC

        return ConformerCTC(configs.ConformerConfig.from_reference(conf, class_name=name), device, dtype=dtype, seed=seed, dp=dp)
    if key == "transducer.contextnet.Contextnet" or key == "transducer.contextnet.ContextNet":
        from .contextnet import ContextNetTransducer

        if not hasattr(configs, "contextnet_from_reference"):
            raise NotImplementedError("ContextNet from a reference mapping: use configs.contextnet(alpha=...)")
        return ContextNetTransducer(configs.contextnet_from_reference(conf), device, dtype=dtype, seed=seed, dp=dp)
    raise NotImplementedError(f"{name}: not on the MI355X hot path (SURVEY.md section 8: Conformer transducer / CTC, ContextNet)")
