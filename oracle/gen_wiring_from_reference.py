"""Generate tests/golden/wiring_*.npz by CONSTRUCTING and RUNNING the reference's own model classes from /root/reference:
`tensorflow_asr.models.transducer.conformer.Conformer` (-> Transducer.call base_transducer.py:427-435 -> FeatureExtraction.call
feature_extraction.py:255-303, ConformerEncoder.call encoders/conformer.py:672-701 -> Conv2dSubsampling.call subsampling.py:218-230,
RelativeSinusoidalPositionalEncoding.call, ConformerBlock.call :504-535, FFModule.call :101-109, MHSAModule.call :209-239 ->
MultiHeadRelativeAttention.call multihead_attention.py:584-667, ConvModule.call :366-377, Residual.call residual.py:58-62;
TransducerPrediction.call base_transducer.py:123-132; TransducerJoint.call :280-293) and
`tensorflow_asr.models.transducer.contextnet.ContextNet` (-> ContextNetEncoder / ConvBlock / SEModule / ConvModule .call,
encoders/contextnet.py:74-90,152-165,251-263,306-311), over oracle/tf_shim.py (NumPy tf) and oracle/keras_shim.py (miniature Keras
with Layer.__call__'s mask plumbing).  Constructors, layer order, dropout sites, residual factors, norm positions, which layers
support masking and what `compute_mask` returns are all the REFERENCE's code; the Keras library layers are restatements (see
keras_shim's header).  Runs in the build container only (the GPU box has no /root/reference); the .npz files are committed.

    python oracle/gen_wiring_from_reference.py
"""
import importlib
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import keras_shim as K  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------- name maps (keras path -> oracle key)
def conformer_key(path):
    """keras variable path of the reference Conformer transducer -> oracle/conformer_ref.py weight name (+ layout fix)."""
    p = path.split("/", 1)[1]  # drop the model name
    leaf = {"kernel": "w", "bias": "b", "gamma": "g", "beta": "b", "moving_mean": "mm", "moving_variance": "mv"}
    m = re.fullmatch(r"encoder/subsampling/block_(\d)/conv_\d/(kernel|bias)", p)
    if m:
        return f"enc/sub/conv{m[1]}/{leaf[m[2]]}", None
    m = re.fullmatch(r"encoder/subsampling/block_(\d)/(?:bn|ln)_\d/(\w+)", p)  # norms: layer -> LayerNormalization in the same slot
    if m:
        return f"enc/sub/bn{m[1]}/{leaf[m[2]]}", None
    m = re.fullmatch(r"encoder/linear/(kernel|bias)", p)
    if m:
        return f"enc/linear/{leaf[m[1]]}", None
    if p == "encoder/content_attention_bias":
        return "enc/u", None
    if p == "encoder/positional_attention_bias":
        return "enc/v", None
    m = re.fullmatch(r"encoder/block_(\d+)/(.+)", p)
    if m:
        i, rest = m[1], m[2]
        pre = f"enc/block{i}/"
        mm = re.fullmatch(r"ff_module_(\d)/(ln|dense_1|dense_2)/(\w+)", rest)
        if mm:
            sub = {"ln": "ln", "dense_1": "d1", "dense_2": "d2"}[mm[2]]
            return f"{pre}ff{mm[1]}/{sub}/{leaf[mm[3]]}", None
        mm = re.fullmatch(r"mhsa_module/ln/(\w+)", rest)
        if mm:
            return f"{pre}mhsa/ln/{leaf[mm[1]]}", None
        mm = re.fullmatch(r"mhsa_module/mhsa/(query|key|value|encoding|attention_output)/(kernel|bias)", rest)
        if mm:
            sub = {"query": "q", "key": "k", "value": "v", "encoding": "pos", "attention_output": "o"}[mm[1]]
            return f"{pre}mhsa/{sub}/{leaf[mm[2]]}", None
        mm = re.fullmatch(r"mhsa_module/mhsa/(content|positional)_attention_bias", rest)
        if mm:
            return f"{pre}mhsa/{'u' if mm[1] == 'content' else 'v'}", None
        mm = re.fullmatch(r"conv_module/(ln|pw_conv_1|dw_conv|dw_bn|dw_ln|pw_conv_2)/(\w+)", rest)
        if mm:
            sub = {"ln": "ln", "pw_conv_1": "pw1", "dw_conv": "dw", "dw_bn": "bn", "dw_ln": "bn", "pw_conv_2": "pw2"}[mm[1]]
            fix = "squeeze0" if mm[1] in ("pw_conv_1", "pw_conv_2") and mm[2] == "kernel" else ("squeeze2" if mm[1] == "dw_conv" and mm[2] == "kernel" else None)
            return f"{pre}conv/{sub}/{leaf[mm[2]]}", fix
        mm = re.fullmatch(r"ln/(\w+)", rest)
        if mm:
            return f"{pre}ln/{leaf[mm[1]]}", None
    m = re.fullmatch(r"prediction/embedding/embeddings", p)
    if m:
        return "pred/emb", None
    m = re.fullmatch(r"prediction/lstm_0/(kernel|recurrent_kernel|bias)", p)
    if m:
        return "pred/lstm/" + {"kernel": "k", "recurrent_kernel": "rk", "bias": "b"}[m[1]], None
    m = re.fullmatch(r"prediction/ln_0/(\w+)", p)
    if m:
        return f"pred/ln/{leaf[m[1]]}", None
    m = re.fullmatch(r"joint/(enc|pred|vocab)/(kernel|bias)", p)
    if m:
        return f"joint/{m[1]}/{leaf[m[2]]}", None
    raise KeyError(path)


def contextnet_key(path):
    """keras variable path of the reference ContextNet transducer -> product / oracle name (params.contextnet_specs)."""
    p = path.split("/", 1)[1]
    leaf = {"gamma": "g", "beta": "b", "moving_mean": "mm", "moving_variance": "mv"}
    m = re.fullmatch(r"encoder/block_(\d+)/(.+)", p)
    if m:
        pre, rest = f"enc/block{m[1]}/", m[2]
        mod = None
        mm = re.fullmatch(r"conv_module_(\d+)/(.+)", rest)
        if mm:
            mod, rest2 = f"{pre}conv{mm[1]}", mm[2]
        mm = mm or re.fullmatch(r"se/conv_module/(.+)", rest)
        if mod is None and mm:
            mod, rest2 = f"{pre}se/conv", mm[1]
        mm = mm or re.fullmatch(r"residual/(.+)", rest)
        if mod is None and mm:
            mod, rest2 = f"{pre}res", mm[1]
        if mod is not None:
            if rest2 == "conv/depthwise_kernel":
                return f"{mod}/dw", "squeeze2"
            if rest2 == "conv/pointwise_kernel":
                return f"{mod}/pw/w", "squeeze0"
            if rest2 == "conv/bias":
                return f"{mod}/pw/b", None
            m2 = re.fullmatch(r"bn/(\w+)", rest2)
            if m2:
                return f"{mod}/bn/{leaf[m2[1]]}", None
        mm = re.fullmatch(r"se/(fc1|fc2)/(kernel|bias)", rest)
        if mm:
            return f"{pre}se/{mm[1]}/{'w' if mm[2] == 'kernel' else 'b'}", None
    # prediction / joint networks share the conformer names
    return conformer_key(path)


def _to_keras_layout(value, var, fix):
    v = np.asarray(value, np.float32)
    if fix == "squeeze0":
        v = v[None]
    elif fix == "squeeze2":
        v = v[..., None]
    assert v.shape == var.shape, (v.shape, var.shape)
    return v


def _randomise(model, keyfn, seed):
    """Seeded values for EVERY variable (so nothing is a silent zero / one), returned under the oracle's names and layouts."""
    rng = np.random.default_rng(seed)
    W = {}
    for path, var in model.named_weights().items():
        key, fix = keyfn(path)
        shape = var.shape
        if path.endswith("moving_variance"):
            v = rng.uniform(0.5, 1.5, shape)
        elif path.endswith("gamma"):
            v = 1.0 + rng.uniform(-0.2, 0.2, shape)
        elif path.endswith(("bias", "beta", "moving_mean", "attention_bias")):
            v = rng.uniform(-0.1, 0.1, shape)
        elif path.endswith("embeddings"):
            v = rng.uniform(-0.5, 0.5, shape)
        else:
            fan_in = int(np.prod(shape[:-1])) if "attention_output" not in path else int(np.prod(shape[:-1]))
            v = rng.standard_normal(shape) * (1.0 / max(1.0, fan_in)) ** 0.5
        var.assign(v.astype(np.float32))
        o = np.asarray(var, np.float32).copy()
        if fix == "squeeze0":
            o = o[0]
        elif fix == "squeeze2":
            o = o[..., 0]
        assert key not in W, key
        W[key] = o
    return W


def _tap(layer):
    """the LAST recorded output of `layer` in this run."""
    for ev, lyr, out in reversed(K.RECORD):
        if ev == "out" and lyr is layer:
            return out
    raise KeyError(layer.name)


def _mask_seen(layer):
    return [info for ev, lyr, info in K.RECORD if ev == "call" and lyr is layer]


def _m2a(m, shape):
    """a recorded mask (or None) as an int8 array: -1 = no mask reached the layer."""
    return np.full((1,), -1, np.int8) if m is None else np.asarray(m, np.int8)


# ---------------------------------------------------------------------------------------------- Conformer transducer
TINY = dict(encoder_dmodel=32, encoder_num_blocks=2, encoder_head_size=8, encoder_num_heads=4, encoder_kernel_size=7,
            prediction_embed_dim=24, prediction_rnn_units=24, joint_dim=40, vocab_size=29)


def _conformer_kwargs(which, **over):
    cfgs = json.load(open(os.path.join(OUT, "reference_configs.json")))
    kw = dict(cfgs[which]["model_config"]["config"])
    kw.pop("kernel_regularizer", None)  # the yml's {class_name: l2, config: 1e-6} = the constructor's default L2
    kw.update(TINY)
    sub = json.loads(json.dumps(kw["encoder_subsampling"]))
    sub["config"]["filters"] = [32, 32]
    kw["encoder_subsampling"] = sub
    sc = dict(kw["speech_config"])
    sc["augmentation_config"] = {}  # SpecAugment is pinned by specaugment_reference.npz; it draws from tf.random
    kw["speech_config"] = sc
    kw.update(over)
    return kw


def gen_conformer(name, which, lens, ulens, seed, dropout, **over):
    with K.reference_runtime() as (tf, keras):
        mod = importlib.import_module("tensorflow_asr.models.transducer.conformer")
        importlib.import_module("tensorflow_asr.models.layers.subsampling")  # the package __init__ imports (= registers) every module
        schemas = importlib.import_module("tensorflow_asr.schemas")
        kw = _conformer_kwargs(which, encoder_dropout=dropout, **over)
        model = mod.Conformer(**kw)
        K.assign_paths(model)
        rng = np.random.default_rng(seed)
        B, N, U = len(lens), max(lens), max(ulens)
        sig = np.clip(rng.standard_normal((B, N)) * 0.1, -1, 1).astype(np.float32)
        for b, n in enumerate(lens):
            sig[b, n:] = 0.0
        labels = rng.integers(1, kw["vocab_size"], (B, U)).astype(np.int32)
        for b, u in enumerate(ulens):
            labels[b, u:] = 0
        preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)
        plen = np.asarray([u + 1 for u in ulens], np.int32)
        c = tf.convert_to_tensor
        inputs = schemas.TrainInput(inputs=c(sig), inputs_length=c(np.asarray(lens, np.int32)), predictions=c(preds), predictions_length=c(plen))
        model(inputs, training=False)  # builds every variable
        W = _randomise(model, conformer_key, seed + 1)
        out = {"signals": sig, "signals_length": np.asarray(lens, np.int32), "predictions": preds, "predictions_length": plen,
               "labels": labels, "labels_length": np.asarray(ulens, np.int32)}
        out.update({f"W/{k}": v for k, v in W.items()})
        enc = model.encoder
        for mode in ("train", "eval"):  # eval runs AFTER the training call: it sees the moving statistics that call updated
            del K.RECORD[:]
            K.DROPOUT.reset(seed + 2 if dropout > 0 else None)
            res = model(inputs, training=(mode == "train"))
            t = f"{mode}/"
            out[t + "logits"], out[t + "logits_length"] = np.asarray(res.logits, np.float32), np.asarray(res.logits_length, np.int32)
            feats, flen = _tap(model.feature_extraction)
            out[t + "features"], out[t + "features_length"] = np.asarray(feats, np.float32), np.asarray(flen, np.int32)
            sub, sublen = _tap(enc.conv_subsampling)
            out[t + "subsampling"] = np.asarray(sub, np.float32)
            out[t + "linear"] = np.asarray(_tap(enc.linear), np.float32)
            x_pe, pe = _tap(enc.relpe)
            out[t + "relpe"] = np.asarray(pe, np.float32)
            for i, blk in enumerate(enc.conformer_blocks):
                out[t + f"block{i}"] = np.asarray(_tap(blk)[0], np.float32)
                out[t + f"block{i}/ffm1"] = np.asarray(_tap(blk.ffm1), np.float32)
                out[t + f"block{i}/mhsam"] = np.asarray(_tap(blk.mhsam)[0], np.float32)
                out[t + f"block{i}/convm"] = np.asarray(_tap(blk.convm), np.float32)
                # the mask that reached the attention's softmax ([B, 1, T, 1] = padded QUERY rows only) and the depthwise BatchNorm
                sm = _mask_seen(blk.mhsam.mha._softmax)
                out[t + f"block{i}/softmax_mask"] = _m2a(sm[-1].get("mask"), None)
                if hasattr(blk.convm.dw_norm, "masks_seen"):
                    out[t + f"block{i}/dw_bn_mask"] = _m2a(blk.convm.dw_norm.masks_seen[-1], None)
            for j, sb in enumerate(enc.conv_subsampling.convs):
                if hasattr(sb.layers[1], "masks_seen"):
                    out[t + f"sub_bn{j}_mask"] = _m2a(sb.layers[1].masks_seen[-1], None)
            encoded = _tap(enc)[0]
            out[t + "encoder"], out[t + "encoder_mask"] = np.asarray(encoded, np.float32), _m2a(K.get_keras_mask(encoded), None)
            pred = _tap(model.predict_net)[0]
            out[t + "prediction"] = np.asarray(pred, np.float32)
            out[t + "lstm_mask"] = _m2a(model.predict_net.rnns[0].masks_seen[-1], None)
            out[t + "logits_mask"] = _m2a(K.get_keras_mask(res.logits), None)
            if mode == "train":
                for path, keep in K.DROPOUT.masks.items():
                    out[f"drop/{_drop_site(path)}"] = keep
                for path, var in model.named_weights().items():  # moving statistics after ONE training call (momentum 0.99)
                    if path.endswith(("moving_mean", "moving_variance")):
                        out["after_train/" + conformer_key(path)[0]] = np.asarray(var, np.float32).copy()
        out["dropout_rate"] = np.asarray(dropout, np.float32)
        out["kwargs"] = np.asarray(json.dumps({k: v for k, v in kw.items() if k != "speech_config"}, sort_keys=True))
        np.savez_compressed(os.path.join(OUT, f"wiring_{name}.npz"), **out)
        print(f"wiring_{name}: logits {out['train/logits'].shape}, T' lengths {out['train/logits_length']}, "
              f"{sum(1 for k in out if k.startswith('W/'))} variables, {sum(1 for k in out if k.startswith('drop/'))} dropout sites")


def _drop_site(path):
    """keras Dropout layer path -> the oracle's site number (conformer_ref.encoder: 0 after the linear layer, block i: 16 + 8 i + k)."""
    p = path.split("/", 1)[1]
    if p == "encoder/dropout":
        return 0
    m = re.fullmatch(r"encoder/block_(\d+)/(.+)", p)
    k = {"ff_module_1/dropout_1": 0, "ff_module_1/dropout_2": 1, "mhsa_module/dropout": 2, "conv_module/dropout": 3,
         "ff_module_2/dropout_1": 4, "ff_module_2/dropout_2": 5}[m[2]]
    return 16 + 8 * int(m[1]) + k


# ---------------------------------------------------------------------------------------------- ContextNet transducer
CN_BLOCKS = [(1, 5, 32, 1, False), (3, 5, 32, 1, True), (3, 5, 32, 2, True), (2, 3, 48, 2, True), (1, 5, 64, 1, False)]  # configs.contextnet_tiny


def gen_contextnet(name, lens, ulens, seed):
    with K.reference_runtime() as (tf, keras):
        mod = importlib.import_module("tensorflow_asr.models.transducer.contextnet")
        schemas = importlib.import_module("tensorflow_asr.schemas")
        cfgs = json.load(open(os.path.join(OUT, "reference_configs.json")))
        kw = dict(cfgs["transducer/contextnet/small"]["model_config"]["config"])
        kw.pop("kernel_regularizer", None)
        kw["encoder_blocks"] = [dict(nlayers=n, kernel_size=k, filters=f, strides=s, residual=r, activation="silu") for n, k, f, s, r in CN_BLOCKS]
        kw.update(encoder_alpha=0.5, prediction_embed_dim=24, prediction_rnn_units=24, joint_dim=40, vocab_size=29, prediction_layer_norm=False)
        sc = dict(kw["speech_config"])
        sc["augmentation_config"] = {}
        kw["speech_config"] = sc
        model = mod.ContextNet(**kw)
        rng = np.random.default_rng(seed)
        B, N, U = len(lens), max(lens), max(ulens)
        sig = np.clip(rng.standard_normal((B, N)) * 0.1, -1, 1).astype(np.float32)
        for b, n in enumerate(lens):
            sig[b, n:] = 0.0
        labels = rng.integers(1, kw["vocab_size"], (B, U)).astype(np.int32)
        for b, u in enumerate(ulens):
            labels[b, u:] = 0
        preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)
        plen = np.asarray([u + 1 for u in ulens], np.int32)
        c = tf.convert_to_tensor
        inputs = schemas.TrainInput(inputs=c(sig), inputs_length=c(np.asarray(lens, np.int32)), predictions=c(preds), predictions_length=c(plen))
        model(inputs, training=False)
        W = _randomise(model, contextnet_key, seed + 1)
        out = {"signals": sig, "signals_length": np.asarray(lens, np.int32), "predictions": preds, "predictions_length": plen,
               "labels": labels, "labels_length": np.asarray(ulens, np.int32)}
        out.update({f"W/{k}": v for k, v in W.items()})
        enc = model.encoder
        for mode in ("train", "eval"):
            del K.RECORD[:]
            res = model(inputs, training=(mode == "train"))
            t = f"{mode}/"
            out[t + "logits"], out[t + "logits_length"] = np.asarray(res.logits, np.float32), np.asarray(res.logits_length, np.int32)
            feats, flen = _tap(model.feature_extraction)
            out[t + "features"], out[t + "features_length"] = np.asarray(feats, np.float32), np.asarray(flen, np.int32)
            for i, blk in enumerate(enc.blocks):
                y, yl = _tap(blk)
                out[t + f"block{i}"], out[t + f"block{i}_length"] = np.asarray(y, np.float32), np.asarray(yl, np.int32)
                out[t + f"block{i}/se"] = np.asarray(_tap(blk.se)[0], np.float32)
                out[t + f"block{i}/pool_mask"] = _m2a(blk.se.global_avg_pool.masks_seen[-1], None)
                out[t + f"block{i}/last_conv_bn_mask"] = _m2a(blk.last_conv.bn.masks_seen[-1], None)
                out[t + f"block{i}/se_conv_bn_mask"] = _m2a(blk.se.conv.bn.masks_seen[-1], None)
            out[t + "encoder"] = np.asarray(_tap(enc)[0], np.float32)
            if mode == "train":
                for path, var in model.named_weights().items():
                    if path.endswith(("moving_mean", "moving_variance")):
                        out["after_train/" + contextnet_key(path)[0]] = np.asarray(var, np.float32).copy()
        np.savez_compressed(os.path.join(OUT, f"wiring_{name}.npz"), **out)
        print(f"wiring_{name}: logits {out['train/logits'].shape}, encoder lengths {out['train/logits_length']}, "
              f"{sum(1 for k in out if k.startswith('W/'))} variables")


def gen_train_step():
    """models/base_model.py: the bodies of BaseModel._train_step (:149-183), _apply_gradients (:185-192), train_step (:194-198) and
    train_step_ga (:200-209) executed with a recording stand-in `self` (the Keras trainer pieces they call - `_compute_loss`, the loss
    tracker, `optimizer.scale_loss / apply`, `tf.GradientTape` - are [ext]): pins the ORDER of the step - weight noise on, forward in
    training mode, weight noise off, loss, tracker update with the un-scaled loss, loss scaling, gradients w.r.t. the trainable
    weights, gradient noise gate, optimizer apply; and the accumulate / gradients + apply + reset split of the GA step."""
    import types

    from oracle import tf_shim

    with K.reference_runtime() as (tf, keras):
        events = []
        ev = lambda *a: events.append(" ".join(str(x) for x in a))

        class Tape:
            def __enter__(self):
                ev("tape.enter")
                return self

            def __exit__(self, *a):
                ev("tape.exit")

            def watch(self, x):
                ev("tape.watch", x)

            def gradient(self, loss, wrt):
                ev("tape.gradient", loss, "wrt", wrt)
                return "grads"

        tf.GradientTape = Tape
        tree = types.SimpleNamespace(flatten=lambda x: [np.zeros((5, 3))])
        loss_module = types.SimpleNamespace(unscale_loss_for_distribution=lambda l: f"unscaled({l})")
        math_util = types.SimpleNamespace(add_gauss_noise=lambda g, stddev: f"noisy({g},{stddev})")
        ns = {"tf": tf, "tree": tree, "loss_module": loss_module, "math_util": math_util, "schemas": types.SimpleNamespace(TrainData=object, TrainOutput=object)}
        fns = tf_shim.extract_functions("tensorflow_asr/models/base_model.py",
                                        ["BaseModel._train_step", "BaseModel._apply_gradients", "BaseModel.train_step", "BaseModel.train_step_ga"], ns)
        x = types.SimpleNamespace(inputs="x.inputs")
        y_pred = types.SimpleNamespace(logits="y_pred.logits")

        def make_self(gradn, iterations):
            me = types.SimpleNamespace(trainable_weights="trainable_weights", gradn_config=gradn)
            me.apply_gwn = lambda: (ev("apply_gwn"), "orig")[1]
            me.remove_gwn = lambda o: ev("remove_gwn", o)
            me.__call__ = None
            me.tfasr_compute_loss = lambda **kw: (ev("tfasr_compute_loss training=%s" % kw["training"]), "loss")[1]
            me._loss_tracker = types.SimpleNamespace(update_state=lambda l, sample_weight=None: ev("loss_tracker.update_state", l, "count", int(np.asarray(sample_weight))))
            me.optimizer = types.SimpleNamespace(scale_loss=lambda l: (ev("optimizer.scale_loss", l), f"scaled({l})")[1], iterations=iterations,
                                                 apply=lambda g, w: ev("optimizer.apply", g, w))
            me.get_metrics_result = lambda: "metrics"
            me.ga = types.SimpleNamespace(accumulate=lambda g, w: ev("ga.accumulate", g), gradients=lambda g, w: (ev("ga.gradients", g), "ga_grads")[1],
                                          reset=lambda: ev("ga.reset"))
            me._train_step = lambda data: fns["BaseModel._train_step"](me, data)
            me._apply_gradients = lambda g: fns["BaseModel._apply_gradients"](me, g)
            return me

        class Callable(types.SimpleNamespace):
            def __call__(self, inp, training=False):
                ev(f"forward training={training}")
                return y_pred

        out = {}
        for name, gradn, it in (("plain", None, 0), ("gradn_before", {"step": 10, "stddev": 0.5}, 3), ("gradn_after", {"step": 10, "stddev": 0.5}, 10)):
            del events[:]
            me = make_self(gradn, it)
            me2 = Callable(**vars(me))
            me2._train_step = lambda data, me2=me2: fns["BaseModel._train_step"](me2, data)
            me2._apply_gradients = lambda g, me2=me2: fns["BaseModel._apply_gradients"](me2, g)
            res = fns["BaseModel.train_step"](me2, (x, "y"))
            assert res == "metrics"
            out[f"train_step_{name}"] = np.asarray(list(events))
        for name, do_apply in (("accumulate", None), ("apply", True)):
            del events[:]
            me = make_self(None, 0)
            me2 = Callable(**vars(me))
            me2._train_step = lambda data, me2=me2: fns["BaseModel._train_step"](me2, data)
            me2._apply_gradients = lambda g, me2=me2: fns["BaseModel._apply_gradients"](me2, g)
            fns["BaseModel.train_step_ga"](me2, (x, "y"), do_apply)
            out[f"train_step_ga_{name}"] = np.asarray(list(events))
        np.savez_compressed(os.path.join(OUT, "wiring_train_step.npz"), **out)
        for k, v in out.items():
            print(k, list(v))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_train_step()
    # ragged batches: T0 = ceil(n / 160) frames, T' = ceil(ceil(T0 / 2) / 2)
    gen_conformer("conformer", "transducer/conformer/small", [9000, 5500, 2100], [6, 3, 5], seed=71, dropout=0.0)
    gen_conformer("conformer_dropout", "transducer/conformer/small", [4000, 2500, 1300], [6, 3, 5], seed=72, dropout=0.1)
    gen_conformer("conformer_streaming", "transducer/conformer/small-streaming", [9000, 5500, 7100], [6, 3, 5], seed=73, dropout=0.0,
                  encoder_history_size=4, encoder_chunk_size=2)
    gen_contextnet("contextnet", [4000, 2900, 1700], [6, 3, 5], seed=81)
