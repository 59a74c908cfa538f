"""Weight checkpoints keyed by the reference's Keras variable paths (checkpoint interop).

`save_weights` / `load_weights` mirror BaseModel.save_weights / load_weights (tensorflow_asr/models/base_model.py:55-61) with an
`.npz` container: one array per Keras variable, named by the layer path the reference builds (encoders/conformer.py:57-657,
layers/subsampling.py:174-214, transducer/base_transducer.py:56-273) and stored in the Keras layout:

    conformer_encoder/subsampling/block_{i}/conv_{i}/{kernel [3,3,cin,cout], bias}      .../bn_{i}/{gamma, beta, moving_mean, moving_variance}
    conformer_encoder/linear/{kernel, bias}      conformer_encoder/{content_attention_bias, positional_attention_bias} [H, dh]
    conformer_encoder/block_{i}/ff_module_{1,2}/{ln/{gamma,beta}, dense_1/{kernel,bias}, dense_2/{kernel,bias}}
    conformer_encoder/block_{i}/mhsa_module/{ln, mhsa/{query,key,value}/{kernel [d,H,dh], bias [H,dh]}, mhsa/encoding/{kernel,bias},
                                              mhsa/attention_output/{kernel [H,dh,d], bias}}
    conformer_encoder/block_{i}/conv_module/{ln, pw_conv_1/{kernel [1,d,2d], bias}, dw_conv/{kernel [K,d,1], bias}, dw_bn/..., pw_conv_2/...}
    conformer_encoder/block_{i}/ln/{gamma, beta}
    prediction/embedding/embeddings      prediction/lstm_0/lstm_cell/{kernel [E,4P], recurrent_kernel [P,4P], bias [4P]}      prediction/ln_0
    joint/{enc, pred, vocab}/{kernel, bias}

The Keras 3 `.weights.h5` CONTAINER is read and written by the pure-Python HDF5 subset in h5lite.py (h5py is not installable in the
product image; reader and writer are validated against the real HDF5 library, tests/test_h5lite.py + tests/test_checkpoint.py):
`load_weights_h5` / `save_weights_h5` below.  The variable PATHS inside the container are restated from the layer attributes in the
reference source and keras' saving_lib, not from a file Keras wrote (no Keras here: parity unpinned), which is why the loader
resolves variables by anchor tokens instead of hard-coded full paths.  Names, layouts, q/k/v split and BatchNorm moving statistics
are exercised by tests/test_checkpoint.py.
"""
import re

import numpy as np

_BLOCK = re.compile(r"^enc/block(\d+)/(.*)$")
_SUB = re.compile(r"^enc/sub/(conv|bn)(\d+)/(.*)$")

_FF = {"ln/g": "ln/gamma", "ln/b": "ln/beta", "d1/w": "dense_1/kernel", "d1/b": "dense_1/bias", "d2/w": "dense_2/kernel", "d2/b": "dense_2/bias"}
_MHSA = {
    "ln/g": "ln/gamma", "ln/b": "ln/beta",
    "q/w": "mhsa/query/kernel", "q/b": "mhsa/query/bias", "k/w": "mhsa/key/kernel", "k/b": "mhsa/key/bias",
    "v/w": "mhsa/value/kernel", "v/b": "mhsa/value/bias", "pos/w": "mhsa/encoding/kernel", "pos/b": "mhsa/encoding/bias",
    "o/w": "mhsa/attention_output/kernel", "o/b": "mhsa/attention_output/bias",
    # encoder_mhsam_use_attention_bias: True -> every attention layer owns its pair (multihead_attention.py:522-538)
    "u": "mhsa/content_attention_bias", "v": "mhsa/positional_attention_bias",
}
_CONV = {
    "ln/g": "ln/gamma", "ln/b": "ln/beta", "pw1/w": "pw_conv_1/kernel", "pw1/b": "pw_conv_1/bias", "dw/w": "dw_conv/kernel",
    "dw/b": "dw_conv/bias", "bn/g": "dw_bn/gamma", "bn/b": "dw_bn/beta", "bn/mm": "dw_bn/moving_mean", "bn/mv": "dw_bn/moving_variance",
    "pw2/w": "pw_conv_2/kernel", "pw2/b": "pw_conv_2/bias",
}
_TAIL = {
    "enc/linear/w": "conformer_encoder/linear/kernel", "enc/linear/b": "conformer_encoder/linear/bias",
    "enc/u": "conformer_encoder/content_attention_bias", "enc/v": "conformer_encoder/positional_attention_bias",
    "pred/emb": "prediction/embedding/embeddings", "pred/lstm/k": "prediction/lstm_0/lstm_cell/kernel",
    "pred/lstm/rk": "prediction/lstm_0/lstm_cell/recurrent_kernel", "pred/lstm/b": "prediction/lstm_0/lstm_cell/bias",
    "pred/ln/g": "prediction/ln_0/gamma", "pred/ln/b": "prediction/ln_0/beta",
    "joint/enc/w": "joint/enc/kernel", "joint/enc/b": "joint/enc/bias", "joint/pred/w": "joint/pred/kernel", "joint/pred/b": "joint/pred/bias",
    "joint/vocab/w": "joint/vocab/kernel", "joint/vocab/b": "joint/vocab/bias",
    # Conformer-CTC head (models/ctc/conformer.py:21-47): ConformerDecoder "conformer_decoder" -> Dense "logits"
    "dec/logits/w": "conformer_decoder/logits/kernel", "dec/logits/b": "conformer_decoder/logits/bias",
}
_DW_LN = {"bn/g": "dw_ln/gamma", "bn/b": "dw_ln/beta"}  # encoder_convm_dw_norm_type: layer -> the layer is named dw_ln (encoders/conformer.py:334-340)


def keras_path(name, dw_norm="batch", sub_norm="batch"):
    """Keras variable path of one tensor of ParamStore.export_keras() (q/k/v already split; BN state as .../mm, .../mv).
    dw_norm = cfg.convm_dw_norm: the depthwise-norm slot of the conv module is `dw_bn` (BatchNormalization) or `dw_ln`."""
    if name in _TAIL:
        return _TAIL[name]
    m = _SUB.match(name)
    if m:
        kind, i, leaf = m.group(1), m.group(2), m.group(3)
        if kind == "bn" and sub_norm == "layer":  # `norms: layer` -> the layer is named ln_{i} (subsampling.py:205-213)
            kind = "ln"
        leaf = {"w": "kernel", "b": "bias" if kind == "conv" else "beta", "g": "gamma", "mm": "moving_mean", "mv": "moving_variance"}[leaf]
        return f"conformer_encoder/subsampling/block_{i}/{kind}_{i}/{leaf}"
    m = _BLOCK.match(name)
    if m:
        i, rest = m.group(1), m.group(2)
        base = f"conformer_encoder/block_{i}/"
        if dw_norm == "layer" and rest.startswith("conv/") and rest[5:] in _DW_LN:
            return base + "conv_module/" + _DW_LN[rest[5:]]
        for pfx, module, table in (("ff1/", "ff_module_1/", _FF), ("ff2/", "ff_module_2/", _FF), ("mhsa/", "mhsa_module/", _MHSA), ("conv/", "conv_module/", _CONV)):
            if rest.startswith(pfx) and rest[len(pfx):] in table:
                return base + module + table[rest[len(pfx):]]
        if rest in ("ln/g", "ln/b"):
            return base + ("ln/gamma" if rest.endswith("g") else "ln/beta")
    raise KeyError(f"no Keras path is known for {name!r}")


def _to_keras_layout(name, a):
    if name.endswith("conv/pw1/w") or name.endswith("conv/pw2/w"):
        return a.reshape(1, *a.shape)            # Conv1D kernel [1, cin, cout]
    if name.endswith("conv/dw/w"):
        return a.reshape(*a.shape, 1)            # DepthwiseConv1D kernel [K, C, 1]
    return a


def _from_keras_layout(name, a, shape, path=None):
    a = np.asarray(a)
    if int(np.prod(a.shape)) != int(np.prod(shape)):
        raise ValueError(f"{path or name}: checkpoint shape {tuple(a.shape)} does not match {tuple(shape)}")
    return a.reshape(shape)


def to_keras(exported, dw_norm="batch", sub_norm="batch"):
    """ParamStore.export_keras() dict (name -> tensor) -> {Keras path: float32 ndarray in the Keras layout}."""
    out = {}
    for name, t in exported.items():
        a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
        out[keras_path(name, dw_norm, sub_norm)] = _to_keras_layout(name, a)
    return out


def from_keras(arrays, template, strict=True, dw_norm="batch", sub_norm="batch"):
    """{Keras path: array} -> dict in the layout of `template` (= ParamStore.export_keras(), which fixes names and shapes).
    strict: every variable of the model must be present and no unknown array may remain."""
    arrays = dict(arrays)
    out, missing = {}, []
    for name, t in template.items():
        path = keras_path(name, dw_norm, sub_norm)
        if path not in arrays:
            missing.append(path)
            continue
        out[name] = _from_keras_layout(name, arrays.pop(path), tuple(t.shape), path)
    if strict and (missing or arrays):
        raise KeyError(f"checkpoint does not match the model: missing {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                       f"unexpected {sorted(arrays)[:5]}{'...' if len(arrays) > 5 else ''}")
    return out


def save_weights(model, filepath):
    """BaseModel.save_weights (base_model.py:55-57): every trainable variable + BatchNorm moving statistics -> `.npz`."""
    arrays = to_keras(model.ps.export_keras(), getattr(model.cfg, "convm_dw_norm", "batch"), getattr(model.cfg, "sub_norm", "batch"))
    with open(filepath, "wb") as f:  # (np.savez would append ".npz" to a bare name)
        np.savez(f, **{k.replace("/", "|"): v for k, v in arrays.items()})
    return sorted(arrays)


def load_weights(model, filepath, strict=True):
    """BaseModel.load_weights (base_model.py:59-61): restores the variables and refreshes the compute-dtype weight shadow."""
    import torch

    with np.load(filepath) as z:
        arrays = {k.replace("|", "/"): z[k] for k in z.files}
    template = model.ps.export_keras()
    got = from_keras(arrays, template, strict=strict, dw_norm=getattr(model.cfg, "convm_dw_norm", "batch"), sub_norm=getattr(model.cfg, "sub_norm", "batch"))
    merged = {k: (torch.from_numpy(np.ascontiguousarray(got[k])) if k in got else v) for k, v in template.items()}
    model.ps.import_keras(merged)
    return sorted(got)


# ------------------------------------------------------------------------------------------------------------------------
# Keras 3 `.weights.h5` containers (base_model.py:55-61 -> keras.Model.save_weights -> saving_lib.H5IOStore), read with the
# pure-Python reader h5lite (h5py is not installable in the product image).
#
# keras stores every layer's variables as datasets `<object path>/vars/<i>` (i = position in layer.weights: Dense kernel, bias;
# BatchNormalization gamma, beta, moving_mean, moving_variance; LSTM cell kernel, recurrent_kernel, bias; ...) where the object
# path is built by walking the model's ATTRIBUTES (saving_lib._walk_saveable: `encoder`, `conformer_blocks`, `ffm1`, `ffn1`,
# `_query_dense` ...), list members being named after their class (`conformer_block`, `conformer_block_1`, ...).  No Keras
# installation exists here to confirm the exact spelling of those paths for tensorflow_asr's models ("parity unpinned"), so the
# loader does not hard-code full paths: each model variable is located by an ordered list of anchor tokens, every anchor with the
# attribute-name AND the layer-name spelling, plus the variable's position and its Keras shape; a variable must match exactly one
# dataset or the load fails loudly with the candidates listed.
# ------------------------------------------------------------------------------------------------------------------------
_H5_MODULE_TOKENS = ("ffm1", "ffm2", "mhsam", "convm", "ff_module_1", "ff_module_2", "mhsa_module", "conv_module")


def _blk(i):
    return (f"conformer_block_{i}" if i else "conformer_block", f"block_{i}")


def h5_anchors(name):
    """(anchor groups, variable index, forbidden tokens) of one ParamStore.export_keras() name."""
    idx = {"w": 0, "b": 1, "g": 0, "mm": 2, "mv": 3}
    m = _SUB.match(name)
    if m:
        kind, i, leaf = m.group(1), int(m.group(2)), m.group(3)
        seq = ("sequential" if i == 0 else f"sequential_{i}", f"block_{i}")
        if kind == "conv":
            return [("conv_subsampling", "subsampling"), seq, ("conv2d", f"conv_{i}", "conv2d_" + str(i))], idx[leaf], ()
        return [("conv_subsampling", "subsampling"), seq, ("batch_normalization", f"bn_{i}", "layer_normalization", f"ln_{i}")], {"g": 0, "b": 1, "mm": 2, "mv": 3}[leaf], ()
    if name in ("enc/linear/w", "enc/linear/b"):
        return [("linear",)], idx[name[-1]], ("conv_subsampling", "subsampling") + _H5_MODULE_TOKENS
    if name in ("enc/u", "enc/v"):
        return [("encoder", "conformer_encoder")], 0 if name.endswith("u") else 1, ("<own>",)
    m = _BLOCK.match(name)
    if m:
        i, rest = int(m.group(1)), m.group(2)
        a = [_blk(i)]
        if rest in ("ln/g", "ln/b"):
            return a + [("post_norm", "ln")], 0 if rest.endswith("g") else 1, _H5_MODULE_TOKENS
        mod, sub = rest.split("/", 1)
        a.append({"ff1": ("ffm1", "ff_module_1"), "ff2": ("ffm2", "ff_module_2"), "mhsa": ("mhsam", "mhsa_module"), "conv": ("convm", "conv_module")}[mod])
        layer, leaf = sub.rsplit("/", 1) if "/" in sub else (sub, None)
        if leaf is None:  # per-layer attention biases: the attention layer's own variables come after nothing else of its own
            return a + [("mha", "mhsa")], 0 if layer == "u" else 1, ("<own>",)
        table = {
            "ln": ("pre_norm", "ln"), "d1": ("ffn1", "dense_1"), "d2": ("ffn2", "dense_2"),
            "q": ("_query_dense", "query"), "k": ("_key_dense", "key"), "v": ("_value_dense", "value"), "pos": ("_relpe_dense", "encoding"),
            "o": ("_output_dense", "attention_output"), "pw1": ("pw_conv_1",), "dw": ("dw_conv",), "bn": ("dw_norm", "dw_bn", "dw_ln"), "pw2": ("pw_conv_2",),
        }
        vi = {"g": 0, "b": 1, "mm": 2, "mv": 3}[leaf] if layer in ("ln", "bn") else idx[leaf]
        forb = ("post_norm",) if layer == "ln" else ()
        return a + [table[layer]], vi, forb
    tail = {
        "pred/emb": ([("predict_net", "transducer_prediction", "prediction"), ("label_encoder", "embedding")], 0),
        "pred/lstm/k": ([("predict_net", "transducer_prediction", "prediction"), ("rnns", "lstm", "lstm_0")], 0),
        "pred/lstm/rk": ([("predict_net", "transducer_prediction", "prediction"), ("rnns", "lstm", "lstm_0")], 1),
        "pred/lstm/b": ([("predict_net", "transducer_prediction", "prediction"), ("rnns", "lstm", "lstm_0")], 2),
        "pred/ln/g": ([("predict_net", "transducer_prediction", "prediction"), ("lns", "layer_normalization", "ln_0")], 0),
        "pred/ln/b": ([("predict_net", "transducer_prediction", "prediction"), ("lns", "layer_normalization", "ln_0")], 1),
        "joint/enc/w": ([("joint_net", "transducer_joint", "joint"), ("ffn_enc", "enc")], 0), "joint/enc/b": ([("joint_net", "transducer_joint", "joint"), ("ffn_enc", "enc")], 1),
        "joint/pred/w": ([("joint_net", "transducer_joint", "joint"), ("ffn_pred", "pred")], 0), "joint/pred/b": ([("joint_net", "transducer_joint", "joint"), ("ffn_pred", "pred")], 1),
        "joint/vocab/w": ([("joint_net", "transducer_joint", "joint"), ("ffn_out", "vocab")], 0), "joint/vocab/b": ([("joint_net", "transducer_joint", "joint"), ("ffn_out", "vocab")], 1),
        "dec/logits/w": ([("decoder", "conformer_decoder"), ("vocab", "logits")], 0), "dec/logits/b": ([("decoder", "conformer_decoder"), ("vocab", "logits")], 1),
    }
    if name in tail:
        return tail[name][0], tail[name][1], ()
    raise KeyError(f"no .weights.h5 anchors are known for {name!r}")


def _h5_match(tokens, anchors, vindex, forbidden):
    if len(tokens) < 2 or tokens[-2] != "vars" or tokens[-1] != str(vindex):
        return False
    body = tokens[:-2]
    if any(t in forbidden for t in body):
        return False
    if "<own>" in forbidden and (not body or body[-1] not in anchors[-1]):  # the anchor layer's OWN variables: .../<layer>/vars/i
        return False
    pos = 0
    for alts in anchors:
        hit = next((k for k in range(pos, len(body)) if body[k] in alts), None)
        if hit is None:
            return False
        pos = hit + 1
    return True


def from_weights_h5(datasets, template):
    """{h5 dataset path: ndarray} (h5lite.H5File.datasets()) -> dict in the layout of `template` (ParamStore.export_keras())."""
    toks = {p: p.split("/") for p in datasets}
    out, used = {}, set()
    for name, t in template.items():
        anchors, vi, forb = h5_anchors(name)
        shp = tuple(_to_keras_layout(name, np.zeros(tuple(t.shape), np.float32)).shape)
        cands = [p for p, tk in toks.items() if _h5_match(tk, anchors, vi, forb) and tuple(datasets[p].shape) in (shp, tuple(t.shape))]
        if len(cands) != 1:
            raise KeyError(f"{name} (Keras shape {shp}): expected exactly one dataset matching anchors {anchors} / vars/{vi}, found {sorted(cands)[:6]}")
        out[name] = _from_keras_layout(name, datasets[cands[0]], tuple(t.shape))
        used.add(cands[0])
    return out, sorted(set(datasets) - used)


def _h5_conformer_only(model):
    """The `.weights.h5` path table (h5_anchors / keras3_h5_path) is written for the Conformer transducer / CTC variables only.  EXPERIMENTAL
    either way: the attribute-walk spelling has not been checked against a file written by real Keras (no Keras / h5py in this image);
    `.npz` (export_keras names) is the tested interchange format."""
    if getattr(model.cfg, "encoder", "conformer") != "conformer":
        raise NotImplementedError(f"Keras .weights.h5 import / export is built for the Conformer models only (encoder = {model.cfg.encoder!r}); "
                                  "use the .npz weights format")


def load_weights_h5(model, filepath, strict=True):
    """BaseModel.load_weights (base_model.py:59-61) for a Keras 3 `.weights.h5` file.  strict: every float dataset of the file
    outside `optimizer/` must have been consumed (so a silently ignored layer cannot go unnoticed)."""
    _h5_conformer_only(model)
    import torch

    from .h5lite import H5File

    with H5File(filepath) as f:
        datasets = f.datasets()
    template = model.ps.export_keras()
    got, unused = from_weights_h5(datasets, template)
    left = [p for p in unused if not p.startswith("optimizer") and datasets[p].dtype.kind == "f" and datasets[p].size > 1]
    if strict and left:
        raise KeyError(f"{filepath}: {len(left)} weight datasets were not mapped onto the model, e.g. {left[:5]}")
    model.ps.import_keras({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in got.items()})
    return sorted(got)


def save_weights_h5(model, filepath):
    """BaseModel.save_weights (base_model.py:55-57 -> keras.Model.save_weights -> saving_lib.save_weights_only -> H5IOStore) as a
    Keras 3 `.weights.h5` container, written by the pure-Python HDF5 writer (h5lite.write_h5; validated against the real HDF5
    library): every layer's variables as `<attribute-walk path>/vars/<i>` in the Keras layouts, plus the model's own empty `vars`
    group.  Returns the dataset paths.  (The attribute-walk spelling is restated from the reference's layer attributes and keras'
    saving_lib, not confirmed against a Keras-written file: see the caveat above.)"""
    from .h5lite import write_h5

    _h5_conformer_only(model)
    datasets = {}
    for name, t in model.ps.export_keras().items():
        a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
        path = keras3_h5_path(name)
        if path in datasets:
            raise KeyError(f"two variables map to {path}")
        datasets[path] = _to_keras_layout(name, a)
    write_h5(filepath, datasets, groups=["vars"])
    return sorted(datasets)


def keras3_h5_path(name, cfg=None):
    """The attribute-walk path keras' saving_lib is expected to give the variable `name` (used to WRITE test fixtures; see the
    caveat above: unconfirmed against a Keras-written file)."""
    anchors, vi, _ = h5_anchors(name)
    first = [a[0] for a in anchors]
    extra = {"rnns": ["rnns", "lstm", "cell"], "lns": ["lns", "layer_normalization"], "conv2d": ["layers", "conv2d"],
             "batch_normalization": ["layers", "batch_normalization"], "conv_subsampling": ["conv_subsampling", "convs"]}
    first = [t for tok in first for t in extra.get(tok, [tok])]
    if any(tok.startswith("_") and tok.endswith("_dense") for tok in first):
        first.insert(len(first) - 1, "mha")
    m = _BLOCK.match(name)
    if m or name.startswith("enc/"):
        if first[0] != "encoder":
            first = ["encoder"] + (["conformer_blocks"] if m else []) + first
    return "/".join(first + ["vars", str(vi)])


def save_state(model, filepath):
    """Everything a resumed run needs beyond the weights: the flat f32 parameters, Adam first / second moments, the
    optimizer step (= position in the learning-rate schedule), the dropout epoch and the BatchNorm moving statistics.  Parameter-shaped
    vectors are stored in the LOGICAL layout (ParamStore.to_logical: no head / filter zero padding), so the file does not depend on the
    storage type or on TFASR_HEAD_PAD / TFASR_FILTER_PAD of the run that wrote it."""
    ps = model.ps
    arrays = {"layout": np.asarray("logical"), "flat": ps.to_logical(ps.flat).numpy(), "adam_m": ps.to_logical(ps.adam_m).numpy(),
              "adam_v": ps.to_logical(ps.adam_v).numpy(), "step": np.asarray(model.step, np.int64),
              "drop_epoch": np.asarray(model._drop_epoch, np.int64), "ga_count": np.asarray(model._ga_count, np.int64),
              "names": np.asarray(ps.names),
              # counters behind the gradient / weight noise seeds (base_model._gradient_noise / apply_gwn): a resumed run continues the
              # sequence of draws instead of replaying the ones from the start of training
              "gradn_epoch": np.asarray(getattr(model, "_gradn_epoch", 0), np.int64), "gwn_epoch": np.asarray(getattr(model, "_gwn_epoch", 0), np.int64)}
    if model._ga_count > 0:  # saved in the middle of a gradient-accumulation cycle: the partial sum of micro-gradients belongs to the state
        arrays["grad"] = ps.to_logical(ps.grad).numpy()
    rng = getattr(model, "_rng", None)
    if rng is not None:  # SpecAugment draws
        import json

        arrays["rng"] = np.asarray(json.dumps(rng.bit_generator.state))
    for k, v in ps.state.items():
        arrays["state|" + k.replace("/", "|")] = ps._unpad(k, v.detach().cpu().clone()).numpy()
    with open(filepath, "wb") as f:
        np.savez(f, **arrays)


def load_state(model, filepath):
    import torch

    ps = model.ps
    with np.load(filepath) as z:
        if list(z["names"]) != list(ps.names):
            raise ValueError(f"{filepath} was written by a different model configuration (parameter names differ)")
        logical = "layout" in z.files and str(z["layout"]) == "logical"
        if not logical:
            # A device-layout file (round 3 or earlier) stores the flat buffer in the PHYSICAL order of the run that wrote it.  That order is
            # not a contract: round 5 moved every block's LayerNorm / positional / depthwise variables into a region behind the last block
            # (ParamStore.defer_lo), with `names` and the element count unchanged - a raw copy would land parameters and Adam moments at
            # permuted offsets without any error.  Such files are refused; weights alone travel through save_weights / load_weights
            # (per-variable, layout-free).
            raise ValueError(f"{filepath} stores the training state in the physical device layout of the build that wrote it (no 'layout: logical' "
                             f"marker).  That layout has changed since; the file cannot be mapped safely.  Re-save it with the build that wrote it "
                             f"using a version that writes the logical layout, or restore the weights only (load_weights).")

        def put(key, buf):
            try:
                ps.from_logical(z[key], buf)
            except ValueError as e:
                raise ValueError(f"{filepath}: {key}: {e} (different model configuration)") from e

        put("flat", ps.flat)
        put("adam_m", ps.adam_m)
        put("adam_v", ps.adam_v)
        model.step, model._drop_epoch, model._ga_count = int(z["step"]), int(z["drop_epoch"]), int(z["ga_count"])
        if "gradn_epoch" in z.files:  # (older state files: the counters restart, as before)
            model._gradn_epoch, model._gwn_epoch = int(z["gradn_epoch"]), int(z["gwn_epoch"])
        if model._ga_count > 0:
            if "grad" in z.files:
                put("grad", ps.grad)
            else:  # a state file without the partial sum: restart the accumulation cycle instead of applying a stale buffer
                model._ga_count = 0
                ps.grad.zero_()
        if "rng" in z.files and getattr(model, "_rng", None) is not None:
            import json

            model._rng.bit_generator.state = json.loads(str(z["rng"]))
        for k in ps.state:
            v = torch.from_numpy(z["state|" + k.replace("/", "|")])
            ps.state[k].copy_(ps._pad(k, v))
    ps.refresh_shadow()
