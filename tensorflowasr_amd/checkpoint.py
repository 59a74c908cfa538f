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

The `.weights.h5` CONTAINER itself cannot be read or written here (h5py is not installed and there is no network), and the paths
above are restated from the layer names in the reference source, not from a file Keras wrote (parity unpinned; INTEGRATION.md
shows the h5py loop that copies between an `.h5` store and this `.npz` by matching path suffix and shape).  Everything below
the container - names, layouts, q/k/v split, BatchNorm moving statistics - is exercised by tests/test_checkpoint.py.
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
}


def keras_path(name):
    """Keras variable path of one tensor of ParamStore.export_keras() (q/k/v already split; BN state as .../mm, .../mv)."""
    if name in _TAIL:
        return _TAIL[name]
    m = _SUB.match(name)
    if m:
        kind, i, leaf = m.group(1), m.group(2), m.group(3)
        leaf = {"w": "kernel", "b": "bias" if kind == "conv" else "beta", "g": "gamma", "mm": "moving_mean", "mv": "moving_variance"}[leaf]
        return f"conformer_encoder/subsampling/block_{i}/{kind}_{i}/{leaf}"
    m = _BLOCK.match(name)
    if m:
        i, rest = m.group(1), m.group(2)
        base = f"conformer_encoder/block_{i}/"
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


def _from_keras_layout(name, a, shape):
    a = np.asarray(a)
    if int(np.prod(a.shape)) != int(np.prod(shape)):
        raise ValueError(f"{keras_path(name)}: checkpoint shape {tuple(a.shape)} does not match {tuple(shape)}")
    return a.reshape(shape)


def to_keras(exported):
    """ParamStore.export_keras() dict (name -> tensor) -> {Keras path: float32 ndarray in the Keras layout}."""
    out = {}
    for name, t in exported.items():
        a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
        out[keras_path(name)] = _to_keras_layout(name, a)
    return out


def from_keras(arrays, template, strict=True):
    """{Keras path: array} -> dict in the layout of `template` (= ParamStore.export_keras(), which fixes names and shapes).
    strict: every variable of the model must be present and no unknown array may remain."""
    arrays = dict(arrays)
    out, missing = {}, []
    for name, t in template.items():
        path = keras_path(name)
        if path not in arrays:
            missing.append(path)
            continue
        out[name] = _from_keras_layout(name, arrays.pop(path), tuple(t.shape))
    if strict and (missing or arrays):
        raise KeyError(f"checkpoint does not match the model: missing {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                       f"unexpected {sorted(arrays)[:5]}{'...' if len(arrays) > 5 else ''}")
    return out


def save_weights(model, filepath):
    """BaseModel.save_weights (base_model.py:55-57): every trainable variable + BatchNorm moving statistics -> `.npz`."""
    arrays = to_keras(model.ps.export_keras())
    with open(filepath, "wb") as f:  # (np.savez would append ".npz" to a bare name)
        np.savez(f, **{k.replace("/", "|"): v for k, v in arrays.items()})
    return sorted(arrays)


def load_weights(model, filepath, strict=True):
    """BaseModel.load_weights (base_model.py:59-61): restores the variables and refreshes the compute-dtype weight shadow."""
    import torch

    with np.load(filepath) as z:
        arrays = {k.replace("|", "/"): z[k] for k in z.files}
    template = model.ps.export_keras()
    got = from_keras(arrays, template, strict=strict)
    merged = {k: (torch.from_numpy(np.ascontiguousarray(got[k])) if k in got else v) for k, v in template.items()}
    model.ps.import_keras(merged)
    return sorted(got)
