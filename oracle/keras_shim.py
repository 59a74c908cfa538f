"""ORACLE tooling (test infrastructure only; nothing here is shipped or imported by the product).

A miniature stand-in for the parts of Keras 3 (keras-nightly 3.9, `requirements.txt:11` of the reference) that the
reference's model files touch, so that the reference's OWN classes - `tensorflow_asr/models/encoders/conformer.py`,
`models/layers/{subsampling,convolution,multihead_attention,positional_encoding,residual,general,embedding}.py`,
`models/activations/glu.py`, `models/encoders/contextnet.py`, `models/transducer/{base_transducer,conformer,contextnet}.py`,
`models/layers/feature_extraction.py`, `models/base_layer.py` - are IMPORTED and EXECUTED from /root/reference (constructors
and `call` bodies) over `oracle/tf_shim.py`'s NumPy `tf`.

What is reference source here and what is a restatement:
  * every class above, its constructor arguments, layer order, names, `supports_masking`, `compute_mask` overrides: REFERENCE
    source, executed;
  * the Keras library layers those classes instantiate (`Dense`, `EinsumDense`, `LayerNormalization`, `BatchNormalization`,
    `Conv1D/2D`, `DepthwiseConv1D`, `SeparableConv1D`, `LSTM`, `Embedding`, `Dropout`, `Softmax`, `MultiHeadAttention` base,
    `GlobalAveragePooling1D`, `Sequential`) and `Layer.__call__`'s build / mask plumbing: [ext] restatements of Keras 3's published
    behaviour, written here from the documented definitions and independently of `oracle/conformer_ref.py` (two restatements
    that must agree: `tests/test_reference_wiring.py`).

Mask plumbing ([ext] keras/src/layers/layer.py `Layer.__call__`, restated):
  * a mask is the `_keras_mask` attribute of a tensor OBJECT (`backend.get_keras_mask / set_keras_mask`);
  * before `call`: with exactly one tensor argument and a `mask` parameter that was not passed, `mask` = the argument's masks;
    with several tensor arguments, each `<arg>_mask` parameter is filled likewise;
  * after `call`: if `layer.supports_masking` (default: `compute_mask` is overridden), outputs that carry no mask yet receive
    `compute_mask(first_arg, masks_of_first_arg)`; a layer without mask support attaches nothing (a fresh output tensor therefore
    carries no mask: "destroys the mask"), but never strips a mask an inner layer has already attached to the returned object;
  * library layers with `supports_masking = True`: Dense, LayerNormalization, BatchNormalization (masked moments), Dropout,
    Activation, Identity, Softmax, Embedding, RNN/LSTM, GlobalAveragePooling1D, MultiHeadAttention; WITHOUT: every convolution
    (BaseConv / BaseDepthwiseConv / BaseSeparableConv), EinsumDense, Model / Sequential (unless `compute_mask` is overridden).
"""
import contextlib
import inspect
import sys
import types

import numpy as np

from oracle import tf_shim
from oracle.tf_shim import T, _t

REFERENCE_ROOT = tf_shim.REFERENCE_ROOT


# ------------------------------------------------------------------------------------------------------------------ helpers
class Variable(T):
    """A weight: ndarray subclass (so `x + v`, `tf.cast(v, ...)`, `tf.gather_nd(v, ...)` work) with assign() and a name."""

    def assign(self, value):
        self[...] = np.asarray(value, self.dtype)
        return self

    def assign_add(self, value):
        self[...] = np.asarray(self) + np.asarray(value, self.dtype)
        return self

    @property
    def value(self):
        return _t(np.asarray(self))


def make_variable(value, name):
    v = np.array(value, dtype=np.float32).view(Variable)
    v.var_name = name
    return v


def is_tensor(x):
    return isinstance(x, np.ndarray) and not isinstance(x, Variable)


def flatten(struct):
    if isinstance(struct, (list, tuple)):
        out = []
        for s in struct:
            out.extend(flatten(s))
        return out
    if isinstance(struct, dict):
        out = []
        for k in sorted(struct):
            out.extend(flatten(struct[k]))
        return out
    return [struct]


def map_structure(fn, struct):
    if isinstance(struct, tuple) and hasattr(struct, "_fields"):
        return type(struct)(*[map_structure(fn, s) for s in struct])
    if isinstance(struct, (list, tuple)):
        return type(struct)(map_structure(fn, s) for s in struct)
    if isinstance(struct, dict):
        return {k: map_structure(fn, v) for k, v in struct.items()}
    return fn(struct)


def get_keras_mask(x):
    return getattr(x, "_keras_mask", None)


def set_keras_mask(x, mask):
    try:
        x._keras_mask = mask
    except AttributeError:
        pass


def _default(fn):
    fn._is_default = True
    return fn


def _is_default(method):
    return getattr(method, "_is_default", False)


def _shape_of(x):
    return tuple(int(s) for s in np.shape(x)) if isinstance(x, np.ndarray) else None


_ACT = {
    None: lambda x: x,
    "linear": lambda x: x,
    "tanh": lambda x: _t(np.tanh(np.asarray(x))),
    "relu": lambda x: _t(np.maximum(np.asarray(x), 0)),
    "sigmoid": lambda x: _t((1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))).astype(np.asarray(x).dtype)),
}
_ACT["swish"] = _ACT["silu"] = lambda x: _t((np.asarray(x, np.float64) / (1.0 + np.exp(-np.asarray(x, np.float64)))).astype(np.asarray(x).dtype))


def get_activation(a):
    if callable(a):
        return a
    return _ACT[a.lower() if isinstance(a, str) else a]


class _Init:
    """Initialisers only need to produce SOME value of the right shape: the generator overwrites every weight with seeded draws."""

    def __init__(self, kind):
        self.kind = kind

    def __call__(self, shape):
        if self.kind == "ones":
            return np.ones(shape, np.float32)
        return np.zeros(shape, np.float32)


# ------------------------------------------------------------------------------------------------------------------ Layer
_AUTO_NAMES = {}


class Layer:
    def __init__(self, *, trainable=True, name=None, dtype=None, activity_regularizer=None, autocast=True, **kwargs):
        if kwargs:
            raise TypeError(f"{type(self).__name__}: unexpected keyword arguments {sorted(kwargs)}")
        if name is None:
            base = "".join("_" + c.lower() if c.isupper() else c for c in type(self).__name__).lstrip("_")
            n = _AUTO_NAMES.get(base, 0)
            _AUTO_NAMES[base] = n + 1
            name = base if n == 0 else f"{base}_{n}"
        self.name = name
        self.trainable = trainable
        self.built = False
        self.activity_regularizer = activity_regularizer
        self._own_weights = []
        self._losses = []
        # [ext] keras Layer.__init__: `self._supports_masking = not utils.is_default(self.compute_mask)`
        self.supports_masking = not _is_default(self.compute_mask)
        self._call_params = list(inspect.signature(self.call).parameters)

    # ---- dtype surface (float32 policy throughout)
    dtype = property(lambda self: "float32")
    compute_dtype = property(lambda self: "float32")
    variable_dtype = property(lambda self: "float32")
    dtype_policy = property(lambda self: "float32")

    @property
    def _call_has_mask_arg(self):
        return "mask" in self._call_params

    @property
    def _call_has_training_arg(self):
        return "training" in self._call_params

    # ---- weights
    def add_weight(self, shape=None, initializer="zeros", dtype=None, trainable=True, autocast=True, regularizer=None, constraint=None,
                   aggregation="none", name=None, **_k):
        kind = initializer if isinstance(initializer, str) else getattr(initializer, "kind", "zeros")
        v = make_variable(_Init(kind)(tuple(int(s) for s in (shape or ()))), name)
        v.trainable = bool(trainable) and self.trainable
        v.regularizer = regularizer
        self._own_weights.append(v)
        return v

    def _sublayers(self):
        out, seen = [], set()

        def visit(o):
            if isinstance(o, Layer):
                if id(o) not in seen:
                    seen.add(id(o))
                    out.append(o)
            elif isinstance(o, (list, tuple)):
                for e in o:
                    visit(e)
            elif isinstance(o, dict):
                for e in o.values():
                    visit(e)

        for k, v in vars(self).items():
            if k.startswith("__"):
                continue
            visit(v)
        return out

    def named_weights(self, prefix=""):
        """{keras-style path: Variable}, depth first in attribute (= creation) order."""
        here = f"{prefix}{self.name}"
        out = {f"{here}/{v.var_name}": v for v in self._own_weights}
        for sub in self._sublayers():
            out.update(sub.named_weights(here + "/"))
        return out

    @property
    def weights(self):
        return list(self.named_weights().values())

    @property
    def trainable_weights(self):
        return [v for v in self.weights if v.trainable]

    def add_loss(self, loss):
        self._losses.append(loss)

    # ---- protocol
    def build(self, input_shape=None):
        self.built = True

    def call(self, *a, **k):
        raise NotImplementedError

    @_default
    def compute_mask(self, inputs, previous_mask):
        return previous_mask

    def compute_output_shape(self, *a, **k):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        params = inspect.signature(self.call).parameters
        names = list(params)
        # `training` is accepted by every layer call; it only reaches `call` if `call` declares it
        if "training" in kwargs and "training" not in names and not any(p.kind == p.VAR_KEYWORD for p in params.values()):
            kwargs.pop("training")
        bound = inspect.signature(self.call).bind(*args, **kwargs)
        arguments = dict(bound.arguments)
        first_arg = args[0] if args else next(iter(arguments.values()))
        # 1. build on first use with the shape(s) of the first argument
        if not self.built:
            shapes = map_structure(_shape_of, first_arg)
            self._build_called_with = shapes
            self.build(shapes)
            self.built = True
        # 2. populate mask arguments
        tensor_args = {k: v for k, v in arguments.items()
                       if k not in ("args", "kwargs") and flatten(v) and all(is_tensor(e) for e in flatten(v))}
        if len(tensor_args) == 1:
            if "mask" in names and arguments.get("mask") is None:
                kwargs["mask"] = map_structure(get_keras_mask, next(iter(tensor_args.values())))
        elif len(tensor_args) > 1:
            for k, v in tensor_args.items():
                if f"{k}_mask" in names and arguments.get(f"{k}_mask") is None:
                    kwargs[f"{k}_mask"] = map_structure(get_keras_mask, v)
        RECORD.append(("call", self, {k: v for k, v in kwargs.items() if k.endswith("mask")}))
        outputs = self.call(*args, **kwargs)
        RECORD.append(("out", self, outputs))
        # 3. mask metadata on the outputs
        previous_mask = map_structure(get_keras_mask, first_arg)
        if self.supports_masking:
            self._set_mask_metadata(first_arg, outputs, previous_mask)
        return outputs

    def _set_mask_metadata(self, inputs, outputs, previous_mask):
        flat_outputs = flatten(outputs)
        if all(get_keras_mask(x) is not None for x in flat_outputs):
            return
        output_masks = self.compute_mask(inputs, previous_mask)
        if output_masks is None:
            return
        for tensor, mask in zip(flat_outputs, flatten(output_masks)):
            if get_keras_mask(tensor) is None and mask is not None:
                set_keras_mask(tensor, mask)


RECORD = []  # (event, layer, info) log of a run: which masks reached which layer


class Model(Layer):
    pass


class Sequential(Model):
    """[ext] keras Sequential before a functional graph exists: layers applied in order, `mask` handed on explicitly."""

    def __init__(self, layers=None, trainable=True, name=None):
        super().__init__(trainable=trainable, name=name)
        self._layers = []
        for layer in layers or []:
            self.add(layer)

    def add(self, layer, rebuild=True):
        self._layers.append(layer)

    @property
    def layers(self):
        return list(self._layers)

    def call(self, inputs, training=None, mask=None):
        outputs = inputs
        for layer in self._layers:
            kw = {}
            if layer._call_has_mask_arg:
                kw["mask"] = mask
            if layer._call_has_training_arg and training is not None:
                kw["training"] = training
            outputs = layer(inputs, **kw)
            inputs = outputs
            mask = map_structure(get_keras_mask, outputs)
        return outputs


# ------------------------------------------------------------------------------------------------------------------ library layers
class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 lora_rank=None, **kwargs):
        super().__init__(activity_regularizer=activity_regularizer, **kwargs)
        self.units, self.activation, self.use_bias = int(units), get_activation(activation), use_bias
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.supports_masking = True

    def build(self, input_shape):
        self.kernel = self.add_weight(name="kernel", shape=(input_shape[-1], self.units), regularizer=self.kernel_regularizer)
        self.bias = self.add_weight(name="bias", shape=(self.units,), regularizer=self.bias_regularizer) if self.use_bias else None
        self.built = True

    def call(self, inputs, training=None):
        x = np.matmul(np.asarray(inputs), np.asarray(self.kernel))
        if self.bias is not None:
            x = x + np.asarray(self.bias)
        return self.activation(_t(x))

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.units,)


class EinsumDense(Layer):
    def __init__(self, equation, output_shape, activation=None, bias_axes=None, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 kernel_regularizer=None, bias_regularizer=None, kernel_constraint=None, bias_constraint=None, lora_rank=None, **kwargs):
        super().__init__(**kwargs)
        self.equation, self.bias_axes, self.activation = equation, bias_axes, get_activation(activation)
        self.partial_output_shape = tuple(output_shape) if isinstance(output_shape, (list, tuple)) else (output_shape,)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer

    def build(self, input_shape):
        ins, out = self.equation.split("->")
        inp, ker = ins.split(",")
        full_out = (input_shape[0],) + self.partial_output_shape
        dims = {}
        for c, s in zip(inp, input_shape):
            dims[c] = s
        for c, s in zip(out, full_out):
            if s is not None:
                dims.setdefault(c, s)
        self.kernel = self.add_weight(name="kernel", shape=tuple(dims[c] for c in ker), regularizer=self.kernel_regularizer)
        if self.bias_axes:
            first = min(out.index(c) for c in self.bias_axes)
            self.bias = self.add_weight(name="bias", shape=tuple(dims[c] if c in self.bias_axes else 1 for c in out[first:]),
                                        regularizer=self.bias_regularizer)
        else:
            self.bias = None
        self.full_output_shape = tuple(dims.get(c) for c in out)
        self.built = True

    def call(self, inputs, training=None):
        x = np.einsum(self.equation, np.asarray(inputs), np.asarray(self.kernel))
        if self.bias is not None:
            x = x + np.asarray(self.bias)
        return self.activation(_t(x.astype(np.float32)))

    def compute_output_shape(self, input_shape):
        return (input_shape[0],) + tuple(self.full_output_shape[1:])


class LayerNormalization(Layer):
    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, rms_scaling=False, beta_initializer="zeros", gamma_initializer="ones",
                 beta_regularizer=None, gamma_regularizer=None, beta_constraint=None, gamma_constraint=None, **kwargs):
        super().__init__(**kwargs)
        assert axis == -1 and center and scale and not rms_scaling
        self.epsilon, self.gamma_regularizer, self.beta_regularizer = epsilon, gamma_regularizer, beta_regularizer
        self.supports_masking = True

    def build(self, input_shape):
        self.gamma = self.add_weight(name="gamma", shape=(input_shape[-1],), initializer="ones", regularizer=self.gamma_regularizer)
        self.beta = self.add_weight(name="beta", shape=(input_shape[-1],), regularizer=self.beta_regularizer)
        self.built = True

    def call(self, inputs):
        x = np.asarray(inputs, np.float64)
        mean = x.mean(-1, keepdims=True)
        var = ((x - mean) ** 2).mean(-1, keepdims=True)
        y = (x - mean) / np.sqrt(var + self.epsilon) * np.asarray(self.gamma, np.float64) + np.asarray(self.beta, np.float64)
        return _t(y.astype(np.float32))


class BatchNormalization(Layer):
    """[ext] keras BatchNormalization: training-mode batch moments (biased variance) over every axis but the last, weighted by the
    mask WHEN ONE ARRIVES (keras `_moments(inputs, mask)`); moving statistics updated with momentum; inference uses them."""

    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros", gamma_initializer="ones",
                 moving_mean_initializer="zeros", moving_variance_initializer="ones", beta_regularizer=None, gamma_regularizer=None,
                 beta_constraint=None, gamma_constraint=None, synchronized=False, **kwargs):
        super().__init__(**kwargs)
        assert axis == -1 and center and scale
        self.momentum, self.epsilon, self.synchronized = momentum, epsilon, synchronized
        self.gamma_regularizer, self.beta_regularizer = gamma_regularizer, beta_regularizer
        self.supports_masking = True
        self.masks_seen = []

    def build(self, input_shape):
        c = input_shape[-1]
        self.gamma = self.add_weight(name="gamma", shape=(c,), initializer="ones", regularizer=self.gamma_regularizer)
        self.beta = self.add_weight(name="beta", shape=(c,), regularizer=self.beta_regularizer)
        self.moving_mean = self.add_weight(name="moving_mean", shape=(c,), trainable=False)
        self.moving_variance = self.add_weight(name="moving_variance", shape=(c,), initializer="ones", trainable=False)
        self.built = True

    def call(self, inputs, training=None, mask=None):
        x = np.asarray(inputs, np.float64)
        axes = tuple(range(x.ndim - 1))
        self.masks_seen.append(None if mask is None else np.asarray(mask).copy())
        if training and self.trainable:
            if mask is None:
                mean = x.mean(axes)
                var = ((x - mean) ** 2).mean(axes)
            else:
                w = np.asarray(mask, np.float64)[..., None]
                for _ in range(x.ndim - w.ndim):
                    w = w[..., None]
                sw = w.sum(axes) + 1e-7
                mean = (w * x).sum(axes) / sw
                var = (w * (x - mean) ** 2).sum(axes) / sw
            self.moving_mean.assign(np.asarray(self.moving_mean) * self.momentum + mean * (1 - self.momentum))
            self.moving_variance.assign(np.asarray(self.moving_variance) * self.momentum + var * (1 - self.momentum))
        else:
            mean, var = np.asarray(self.moving_mean, np.float64), np.asarray(self.moving_variance, np.float64)
        y = (x - mean) / np.sqrt(var + self.epsilon) * np.asarray(self.gamma, np.float64) + np.asarray(self.beta, np.float64)
        return _t(y.astype(np.float32))


class Dropout(Layer):
    """Dropout with INJECTED masks: `DROPOUT_MASKS[layer path]` (keep / (1 - rate) scaling per keras) when training and rate > 0.
    The path is assigned by `assign_paths()` after the model is constructed."""

    def __init__(self, rate, noise_shape=None, seed=None, **kwargs):
        super().__init__(**kwargs)
        self.rate, self.seed = rate, seed
        self.supports_masking = True
        self.built = True
        self.path = None

    def call(self, inputs, training=False):
        if training and self.rate > 0:
            keep = DROPOUT.draw(self.path or self.name, np.shape(inputs), self.rate)
            return _t((np.asarray(inputs) * keep / (1.0 - self.rate)).astype(np.float32))
        return inputs


class _DropoutSource:
    def __init__(self):
        self.rng, self.masks = None, {}

    def reset(self, seed=None):
        self.rng, self.masks = (None if seed is None else np.random.default_rng(seed)), {}

    def draw(self, key, shape, rate):
        if self.rng is None:
            raise RuntimeError("dropout with rate > 0 in training mode needs DROPOUT.reset(seed)")
        keep = (self.rng.random(shape) >= rate).astype(np.float32)
        n = sum(1 for k in self.masks if k == key or k.startswith(key + "#"))
        self.masks[key if n == 0 else f"{key}#{n}"] = keep
        return keep


DROPOUT = _DropoutSource()


class Identity(Layer):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.supports_masking = True
        self.built = True

    def call(self, inputs):
        return inputs


class Activation(Layer):
    def __init__(self, activation, **kwargs):
        super().__init__(**kwargs)
        self.activation = get_activation(activation)
        self.supports_masking = True
        self.built = True

    def call(self, inputs):
        return self.activation(inputs)


class Softmax(Layer):
    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis
        self.supports_masking = True
        self.built = True

    def call(self, inputs, mask=None):  # the reference overrides this
        raise NotImplementedError


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, embeddings_initializer="uniform", embeddings_regularizer=None, embeddings_constraint=None,
                 mask_zero=False, weights=None, lora_rank=None, **kwargs):
        super().__init__(**kwargs)
        self.input_dim, self.output_dim, self.mask_zero = input_dim, output_dim, mask_zero
        self.embeddings_regularizer = embeddings_regularizer

    def build(self, input_shape=None):
        if not getattr(self, "embeddings", None) is not None:
            self.embeddings = self.add_weight(name="embeddings", shape=(self.input_dim, self.output_dim), regularizer=self.embeddings_regularizer)
        self.built = True

    def call(self, inputs):
        return _t(np.asarray(self.embeddings)[np.asarray(inputs).astype(np.int64)])

    def compute_mask(self, inputs, mask=None):
        return None if not self.mask_zero else _t(np.asarray(inputs) != 0)

    def compute_output_shape(self, input_shape):
        return tuple(input_shape) + (self.output_dim,)


def _conv_nd(x, w, strides, padding, dilation=None):
    """channels-last N-d cross-correlation (N = 1, 2): x [B, *S, Cin], w [*K, Cin, Cout], float64 accumulation."""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    nd = w.ndim - 2
    K = w.shape[:nd]
    dilation = dilation or (1,) * nd
    assert all(d == 1 for d in dilation)
    if padding == "same":
        pads = []
        for i in range(nd):
            L = x.shape[1 + i]
            out = -(-L // strides[i])
            tot = max((out - 1) * strides[i] + K[i] - L, 0)
            pads.append((tot // 2, tot - tot // 2))
        x = np.pad(x, [(0, 0)] + pads + [(0, 0)])
    elif padding == "causal":
        assert nd == 1
        x = np.pad(x, [(0, 0), (K[0] - 1, 0), (0, 0)])
    else:
        assert padding == "valid", padding
    outs = [(x.shape[1 + i] - K[i]) // strides[i] + 1 for i in range(nd)]
    y = np.zeros((x.shape[0],) + tuple(outs) + (w.shape[-1],), np.float64)
    for idx in np.ndindex(*K):
        sl = (slice(None),) + tuple(slice(idx[i], idx[i] + (outs[i] - 1) * strides[i] + 1, strides[i]) for i in range(nd))
        y += np.tensordot(x[sl], w[idx], axes=([x.ndim - 1], [0]))
    return y


def _tup(v, n):
    return tuple(int(e) for e in v) if isinstance(v, (list, tuple)) else (int(v),) * n


class _BaseConv(Layer):
    rank = None

    def __init__(self, filters, kernel_size, strides=1, padding="valid", data_format=None, dilation_rate=1, groups=1, activation=None,
                 use_bias=True, kernel_initializer="glorot_uniform", bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None, **kwargs):
        super().__init__(activity_regularizer=activity_regularizer, **kwargs)
        self.filters, self.kernel_size, self.strides = int(filters), _tup(kernel_size, self.rank), _tup(strides, self.rank)
        self.padding, self.data_format, self.dilation_rate = padding, data_format or "channels_last", _tup(dilation_rate, self.rank)
        self.groups, self.activation, self.use_bias = groups, get_activation(activation), use_bias
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        assert groups == 1 and self.data_format == "channels_last"

    def build(self, input_shape):
        self.kernel = self.add_weight(name="kernel", shape=self.kernel_size + (input_shape[-1], self.filters), regularizer=self.kernel_regularizer)
        self.bias = self.add_weight(name="bias", shape=(self.filters,), regularizer=self.bias_regularizer) if self.use_bias else None
        self.built = True

    def call(self, inputs):
        y = _conv_nd(inputs, self.kernel, self.strides, self.padding, self.dilation_rate)
        if self.bias is not None:
            y = y + np.asarray(self.bias, np.float64)
        return self.activation(_t(y.astype(np.float32)))


class Conv1D(_BaseConv):
    rank = 1


class Conv2D(_BaseConv):
    rank = 2


class DepthwiseConv1D(Layer):
    rank = 1

    def __init__(self, kernel_size, strides=1, padding="valid", depth_multiplier=1, data_format=None, dilation_rate=1, activation=None,
                 use_bias=True, depthwise_initializer="glorot_uniform", bias_initializer="zeros", depthwise_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, depthwise_constraint=None, bias_constraint=None, **kwargs):
        super().__init__(activity_regularizer=activity_regularizer, **kwargs)
        self.kernel_size, self.strides, self.padding = _tup(kernel_size, 1), _tup(strides, 1), padding
        self.depth_multiplier, self.data_format, self.dilation_rate = depth_multiplier, data_format or "channels_last", _tup(dilation_rate, 1)
        self.activation, self.use_bias = get_activation(activation), use_bias
        self.depthwise_regularizer, self.bias_regularizer = depthwise_regularizer, bias_regularizer
        assert depth_multiplier == 1

    def build(self, input_shape):
        c = input_shape[-1]
        self.kernel = self.add_weight(name="kernel", shape=self.kernel_size + (c, self.depth_multiplier), regularizer=self.depthwise_regularizer)
        self.bias = self.add_weight(name="bias", shape=(c,), regularizer=self.bias_regularizer) if self.use_bias else None
        self.built = True

    def call(self, inputs):
        x, w = np.asarray(inputs, np.float64), np.asarray(self.kernel, np.float64)[:, :, 0]
        assert self.padding == "valid"
        K, s = self.kernel_size[0], self.strides[0]
        out = (x.shape[1] - K) // s + 1
        y = np.zeros((x.shape[0], out, x.shape[2]), np.float64)
        for k in range(K):
            y += x[:, k:k + (out - 1) * s + 1:s] * w[k]
        if self.bias is not None:
            y = y + np.asarray(self.bias, np.float64)
        return self.activation(_t(y.astype(np.float32)))

    def _get_input_channel(self, input_shape):
        return input_shape[-1]


class SeparableConv1D(Layer):
    rank = 1

    def __init__(self, filters, kernel_size, strides=1, padding="valid", data_format=None, dilation_rate=1, depth_multiplier=1, activation=None,
                 use_bias=True, depthwise_initializer="glorot_uniform", pointwise_initializer="glorot_uniform", bias_initializer="zeros",
                 depthwise_regularizer=None, pointwise_regularizer=None, bias_regularizer=None, activity_regularizer=None,
                 depthwise_constraint=None, pointwise_constraint=None, bias_constraint=None, **kwargs):
        super().__init__(activity_regularizer=activity_regularizer, **kwargs)
        self.filters, self.kernel_size, self.strides, self.padding = int(filters), _tup(kernel_size, 1), _tup(strides, 1), padding
        self.data_format, self.dilation_rate, self.depth_multiplier = data_format or "channels_last", _tup(dilation_rate, 1), depth_multiplier
        self.activation, self.use_bias = get_activation(activation), use_bias
        self.depthwise_regularizer, self.pointwise_regularizer, self.bias_regularizer = depthwise_regularizer, pointwise_regularizer, bias_regularizer
        assert depth_multiplier == 1

    def build(self, input_shape):
        c = input_shape[-1]
        self.depthwise_kernel = self.add_weight(name="depthwise_kernel", shape=self.kernel_size + (c, 1), regularizer=self.depthwise_regularizer)
        self.pointwise_kernel = self.add_weight(name="pointwise_kernel", shape=(1, c, self.filters), regularizer=self.pointwise_regularizer)
        self.bias = self.add_weight(name="bias", shape=(self.filters,), regularizer=self.bias_regularizer) if self.use_bias else None
        self.built = True

    def call(self, inputs):
        x, w = np.asarray(inputs, np.float64), np.asarray(self.depthwise_kernel, np.float64)[:, :, 0]
        assert self.padding == "valid"
        K, s = self.kernel_size[0], self.strides[0]
        out = (x.shape[1] - K) // s + 1
        y = np.zeros((x.shape[0], out, x.shape[2]), np.float64)
        for k in range(K):
            y += x[:, k:k + (out - 1) * s + 1:s] * w[k]
        y = y @ np.asarray(self.pointwise_kernel, np.float64)[0]
        if self.bias is not None:
            y = y + np.asarray(self.bias, np.float64)
        return self.activation(_t(y.astype(np.float32)))


class GlobalAveragePooling1D(Layer):
    def __init__(self, data_format=None, keepdims=False, **kwargs):
        super().__init__(**kwargs)
        self.keepdims = keepdims
        self.supports_masking = True
        self.masks_seen = []

    def call(self, inputs, mask=None):
        x = np.asarray(inputs, np.float64)
        self.masks_seen.append(None if mask is None else np.asarray(mask).copy())
        if mask is not None:
            m = np.asarray(mask, np.float64)[:, :, None]
            y = (x * m).sum(1, keepdims=self.keepdims) / m.sum(1, keepdims=self.keepdims)
        else:
            y = x.mean(1, keepdims=self.keepdims)
        return _t(y.astype(np.float32))

    def compute_mask(self, inputs, mask=None):
        return None


class LSTM(Layer):
    """[ext] keras LSTM: gates i, f, c, o; kernel [in, 4u], recurrent_kernel [u, 4u], ONE bias [4u]; sigmoid / tanh; masked steps carry the
    state forward and emit zeros when `zero_output_for_mask` (else the previous output)."""

    def __init__(self, units, activation="tanh", recurrent_activation="sigmoid", use_bias=True, kernel_initializer="glorot_uniform",
                 recurrent_initializer="orthogonal", bias_initializer="zeros", unit_forget_bias=True, kernel_regularizer=None,
                 recurrent_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, recurrent_constraint=None,
                 bias_constraint=None, dropout=0.0, recurrent_dropout=0.0, seed=None, return_sequences=False, return_state=False,
                 go_backwards=False, stateful=False, unroll=False, use_cudnn="auto", implementation=2, zero_output_for_mask=False, **kwargs):
        super().__init__(activity_regularizer=activity_regularizer, **kwargs)
        assert activation == "tanh" and recurrent_activation == "sigmoid" and use_bias and not go_backwards and not stateful
        self.units, self.return_sequences, self.return_state, self.zero_output_for_mask = int(units), return_sequences, return_state, zero_output_for_mask
        self.kernel_regularizer, self.recurrent_regularizer, self.bias_regularizer = kernel_regularizer, recurrent_regularizer, bias_regularizer
        self.supports_masking = True
        self.masks_seen = []

    def build(self, input_shape):
        u = self.units
        self.kernel = self.add_weight(name="kernel", shape=(input_shape[-1], 4 * u), regularizer=self.kernel_regularizer)
        self.recurrent_kernel = self.add_weight(name="recurrent_kernel", shape=(u, 4 * u), regularizer=self.recurrent_regularizer)
        self.bias = self.add_weight(name="bias", shape=(4 * u,), regularizer=self.bias_regularizer)
        self.built = True

    def get_initial_state(self, batch_size):
        return [_t(np.zeros((batch_size, self.units), np.float32)) for _ in range(2)]

    def call(self, sequences, initial_state=None, mask=None, training=False):
        x = np.asarray(sequences, np.float64)
        B, Tn, _ = x.shape
        u = self.units
        self.masks_seen.append(None if mask is None else np.asarray(mask).copy())
        h, c = (np.asarray(s, np.float64) for s in initial_state) if initial_state is not None else (np.zeros((B, u)), np.zeros((B, u)))
        W, R, b = (np.asarray(v, np.float64) for v in (self.kernel, self.recurrent_kernel, self.bias))
        sig = lambda z: 1.0 / (1.0 + np.exp(-z))
        outs, last = [], np.zeros((B, u))
        for t in range(Tn):
            z = x[:, t] @ W + h @ R + b
            i, f, g, o = sig(z[:, :u]), sig(z[:, u:2 * u]), np.tanh(z[:, 2 * u:3 * u]), sig(z[:, 3 * u:])
            cn = f * c + i * g
            hn = o * np.tanh(cn)
            if mask is not None:
                m = np.asarray(mask)[:, t, None]
                out = np.where(m, hn, 0.0 if self.zero_output_for_mask else last)
                h, c = np.where(m, hn, h), np.where(m, cn, c)
            else:
                out, h, c = hn, hn, cn
            last = out
            outs.append(out)
        seq = _t(np.stack(outs, 1).astype(np.float32))
        y = seq if self.return_sequences else _t(outs[-1].astype(np.float32))
        if self.return_state:
            return y, _t(h.astype(np.float32)), _t(c.astype(np.float32))
        return y

    def compute_mask(self, _, mask):
        mask = flatten(mask)[0] if mask is not None else None
        out = mask if self.return_sequences else None
        return [out, None, None] if self.return_state else out


# ---- keras.src.layers.attention.multi_head_attention (module-level helpers + the base class)
def _index_to_einsum_variable(i):
    return chr(97 + i)


def _build_attention_equation(rank, attn_axes):
    target = "".join(_index_to_einsum_variable(i) for i in range(rank))
    batch_dims = tuple(np.delete(range(rank), tuple(attn_axes) + (rank - 1,)))
    letter_offset, source = rank, ""
    for i in range(rank):
        if i in batch_dims or i == rank - 1:
            source += target[i]
        else:
            source += _index_to_einsum_variable(letter_offset)
            letter_offset += 1
    product = "".join([target[i] for i in batch_dims] + [target[i] for i in attn_axes] + [source[i] for i in attn_axes])
    return f"{source},{target}->{product}", f"{product},{source}->{target}", len(product)


def _build_proj_equation(free_dims, bound_dims, output_dims):
    inp = ker = out = bias = ""
    off = 0
    for i in range(free_dims):
        c = _index_to_einsum_variable(i + off)
        inp += c
        out += c
    off += free_dims
    for i in range(bound_dims):
        c = _index_to_einsum_variable(i + off)
        inp += c
        ker += c
    off += bound_dims
    for i in range(output_dims):
        c = _index_to_einsum_variable(i + off)
        ker += c
        out += c
        bias += c
    return f"{inp},{ker}->{out}", bias, len(out)


def _get_output_shape(output_rank, known_last_dims):
    return [None] * (output_rank - len(known_last_dims)) + list(known_last_dims)


class MultiHeadAttention(Layer):
    def __init__(self, num_heads, key_dim, value_dim=None, dropout=0.0, use_bias=True, output_shape=None, attention_axes=None,
                 flash_attention=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", kernel_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None, seed=None, **kwargs):
        super().__init__(activity_regularizer=activity_regularizer, **kwargs)
        self.supports_masking = True
        self._num_heads, self._key_dim, self._value_dim = num_heads, key_dim, value_dim if value_dim else key_dim
        self._inverse_sqrt_key_dim = None
        self._dropout, self._use_bias, self._output_shape = dropout, use_bias, output_shape
        self._kernel_regularizer, self._bias_regularizer = kernel_regularizer, bias_regularizer
        self._attention_axes = attention_axes
        self.seed = seed
        self._inverse_sqrt_key_dim = 1.0 / np.sqrt(float(self._key_dim))

    dropout = property(lambda self: self._dropout)

    def _get_common_kwargs_for_sublayer(self):
        return dict(kernel_regularizer=self._kernel_regularizer, bias_regularizer=self._bias_regularizer, dtype=self.dtype_policy)

    def build(self, query_shape, value_shape, key_shape=None):
        key_shape = value_shape if key_shape is None else key_shape
        eq, bias_axes, out_rank = _build_proj_equation(len(query_shape) - 1, bound_dims=1, output_dims=2)
        self._query_dense = EinsumDense(eq, output_shape=_get_output_shape(out_rank - 1, [self._num_heads, self._key_dim]),
                                        bias_axes=bias_axes if self._use_bias else None, name="query", **self._get_common_kwargs_for_sublayer())
        self._query_dense.build(query_shape)
        eq, bias_axes, out_rank = _build_proj_equation(len(key_shape) - 1, bound_dims=1, output_dims=2)
        self._key_dense = EinsumDense(eq, output_shape=_get_output_shape(out_rank - 1, [self._num_heads, self._key_dim]),
                                      bias_axes=bias_axes if self._use_bias else None, name="key", **self._get_common_kwargs_for_sublayer())
        self._key_dense.build(key_shape)
        eq, bias_axes, out_rank = _build_proj_equation(len(value_shape) - 1, bound_dims=1, output_dims=2)
        self._value_dense = EinsumDense(eq, output_shape=_get_output_shape(out_rank - 1, [self._num_heads, self._value_dim]),
                                        bias_axes=bias_axes if self._use_bias else None, name="value", **self._get_common_kwargs_for_sublayer())
        self._value_dense.build(value_shape)
        self._build_attention(out_rank)
        out_shape = list(self._output_shape) if self._output_shape else [query_shape[-1]]
        eq, bias_axes, out_rank = _build_proj_equation(len(query_shape) - 1, bound_dims=2, output_dims=len(out_shape))
        self._output_dense = EinsumDense(eq, output_shape=_get_output_shape(out_rank - 1, out_shape),
                                         bias_axes=bias_axes if self._use_bias else None, name="attention_output", **self._get_common_kwargs_for_sublayer())
        od_in = list(self._query_dense.compute_output_shape(query_shape))
        od_in[-1] = self._value_dim
        self._output_dense.build(tuple(od_in))
        self.built = True

    def _masked_softmax(self, attention_scores, attention_mask=None):
        if attention_mask is not None:
            axis = -len(self._attention_axes) * 2 - 1
            for _ in range(np.ndim(attention_scores) - np.ndim(attention_mask)):
                attention_mask = _t(np.expand_dims(np.asarray(attention_mask), axis))
        return self._softmax(attention_scores, mask=attention_mask)

    def _compute_attention_mask(self, query, value, query_mask=None, value_mask=None, key_mask=None, attention_mask=None, use_causal_mask=False):
        auto = None
        if query_mask is not None:
            auto = np.asarray(query_mask, bool)[..., None]           # [B, T, 1]
        for m in (value_mask, key_mask):
            if m is not None:
                mm = np.asarray(m, bool)[..., None, :]                # [B, 1, S]
                auto = mm if auto is None else auto & mm
        if use_causal_mask:
            Tq, Sv = np.shape(query)[1], np.shape(value)[1]
            mm = np.tril(np.ones((1, Tq, Sv), bool))
            auto = mm if auto is None else auto & mm
        if auto is not None:
            attention_mask = _t(auto) if attention_mask is None else _t(np.asarray(attention_mask, bool) & auto)
        return attention_mask

    def compute_output_shape(self, query_shape, value_shape=None, key_shape=None):
        if self._output_shape:
            return tuple(query_shape[:-1]) + tuple(self._output_shape)
        return tuple(query_shape)


# ------------------------------------------------------------------------------------------------------------------ module assembly
class _Registry(dict):
    pass


REGISTRY = _Registry()


def make_keras(tf):
    keras = types.ModuleType("keras")
    keras.Model, keras.Sequential = Model, Sequential
    keras.Variable = Variable
    keras.layers = types.SimpleNamespace(
        Layer=Layer, Dense=Dense, EinsumDense=EinsumDense, LayerNormalization=LayerNormalization, BatchNormalization=BatchNormalization,
        Dropout=Dropout, Identity=Identity, Activation=Activation, Softmax=Softmax, Embedding=Embedding, Conv1D=Conv1D, Conv2D=Conv2D,
        DepthwiseConv1D=DepthwiseConv1D, DepthwiseConv2D=type("DepthwiseConv2D", (Layer,), {}), SeparableConv1D=SeparableConv1D,
        SeparableConv2D=type("SeparableConv2D", (Layer,), {}), GlobalAveragePooling1D=GlobalAveragePooling1D, LSTM=LSTM,
        GRU=type("GRU", (Layer,), {}), SimpleRNN=type("SimpleRNN", (Layer,), {}), MultiHeadAttention=MultiHeadAttention,
        MaxPool2D=type("MaxPool2D", (Layer,), {}), MaxPooling2D=type("MaxPooling2D", (Layer,), {}), Bidirectional=type("Bidirectional", (Layer,), {}),
        Reshape=type("Reshape", (Layer,), {}))
    keras.regularizers = types.SimpleNamespace(l2=lambda v: ("l2", float(v)), Regularizer=object)
    keras.initializers = types.SimpleNamespace(Initializer=object)
    keras.activations = types.SimpleNamespace(linear=lambda x: x, get=get_activation)
    keras.optimizers = types.SimpleNamespace(Optimizer=object, LossScaleOptimizer=object)
    keras.metrics = types.SimpleNamespace(Metric=object)
    keras.losses = types.SimpleNamespace(Loss=object)
    keras.KerasTensor = object

    def register_keras_serializable(package="Custom", name=None):
        def deco(cls):
            REGISTRY[f"{package}>{name or cls.__name__}"] = cls
            return cls
        return deco

    keras.utils = types.SimpleNamespace(register_keras_serializable=register_keras_serializable,
                                        get_registered_object=lambda name, **_k: REGISTRY[name])
    # keras.src.*
    src = types.ModuleType("keras.src")
    backend = types.ModuleType("keras.src.backend")
    backend.get_keras_mask, backend.set_keras_mask = get_keras_mask, set_keras_mask
    backend.standardize_dtype = lambda d: d if isinstance(d, str) else np.dtype(getattr(d, "np", d)).name
    backend.numpy = types.SimpleNamespace(exp=lambda x: _t(np.exp(np.asarray(x))))
    backend.math = types.SimpleNamespace(logsumexp=lambda x, axis=None, keepdims=False: tf_shim._logsumexp(np.asarray(x), axis, keepdims))
    backend.epsilon = lambda: 1e-7
    activations = types.ModuleType("keras.src.activations")
    activations.softmax = lambda x, axis=-1: tf.nn.softmax(x, axis=axis)
    mha = types.ModuleType("keras.src.layers.attention.multi_head_attention")
    mha._build_attention_equation, mha._build_proj_equation, mha._get_output_shape = _build_attention_equation, _build_proj_equation, _get_output_shape
    opu = types.ModuleType("keras.src.ops.operation_utils")
    opu.compute_conv_output_shape = lambda *a, **k: None
    tree = types.ModuleType("keras.src.tree")
    tree.flatten, tree.map_structure = flatten, map_structure
    src.backend, src.activations, src.tree = backend, activations, tree
    keras.src = src
    mods = {"keras": keras, "keras.src": src, "keras.src.backend": backend, "keras.src.activations": activations,
            "keras.src.layers": types.ModuleType("keras.src.layers"), "keras.src.layers.attention": types.ModuleType("keras.src.layers.attention"),
            "keras.src.layers.attention.multi_head_attention": mha, "keras.src.ops": types.ModuleType("keras.src.ops"),
            "keras.src.ops.operation_utils": opu, "keras.src.tree": tree}
    mods["keras.src.layers.attention"].multi_head_attention = mha
    return keras, mods


def extend_tf_for_models(tf):
    """tf primitives the model files need beyond tf_shim.make_tf(): identity-preserving cast / convert (TensorFlow returns the SAME
    tensor object when nothing changes, so an attached `_keras_mask` survives), activations, tf.signal ([ext] restated)."""
    A = np.asarray

    def cast(x, dtype, name=None):
        d = np.dtype(tf_shim._npd(dtype) if not isinstance(dtype, (str, np.dtype)) else dtype)
        if isinstance(x, T) and x.dtype == d:
            return x
        return _t(A(x).astype(d))

    def convert_to_tensor(x, dtype=None, name=None):
        if isinstance(x, T) and not isinstance(x, Variable) and (dtype is None or x.dtype == np.dtype(tf_shim._npd(dtype))):
            return x
        return _t(A(x), tf_shim._npd(dtype) if dtype is not None else None) if not isinstance(x, np.ndarray) else _t(A(x).astype(tf_shim._npd(dtype)) if dtype is not None else A(x))

    tf.cast, tf.convert_to_tensor = cast, convert_to_tensor
    tf.nn.swish = _ACT["swish"]
    tf.nn.relu = _ACT["relu"]
    tf.nn.tanh = _ACT["tanh"]
    tf.tanh = _ACT["tanh"]
    tf.identity = lambda x, name=None: x
    tf.Variable = Variable
    import logging
    tf.get_logger = lambda: logging.getLogger("tensorflow_shim")
    tf.nest = types.SimpleNamespace(map_structure=map_structure, flatten=flatten)
    class TensorSpec:
        def __init__(self, shape=None, dtype=None, name=None):
            self.shape, self.dtype = shape, dtype

        @classmethod
        def from_tensor(cls, tensor, name=None):
            return cls(np.shape(tensor), np.asarray(tensor).dtype)

    tf.TensorSpec = TensorSpec
    tf.gather = lambda params, indices, axis=None, batch_dims=0, name=None: tf_shim._gather(A(params), A(indices), axis, batch_dims)
    tf.squeeze = lambda input, axis=None, name=None: _t(np.squeeze(A(input), axis=None if axis is None else tuple(np.atleast_1d(axis))))  # noqa: A002

    def hann_window(n, periodic=True, dtype=None):
        return _t((0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / (n if periodic else n - 1))).astype(np.float32))

    def frame(signal, frame_length, frame_step, pad_end=False, pad_value=0, axis=-1):
        x = A(signal)
        n = x.shape[-1]
        nf = -(-n // frame_step) if pad_end else max(0, 1 + (n - frame_length) // frame_step)
        need = (nf - 1) * frame_step + frame_length
        if need > n:
            x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, need - n)], constant_values=pad_value)
        idx = np.arange(nf)[:, None] * frame_step + np.arange(frame_length)[None, :]
        return _t(x[..., idx])

    def stft(signals, frame_length, frame_step, fft_length=None, window_fn=None, pad_end=False, name=None):
        fr = A(frame(signals, frame_length, frame_step, pad_end=pad_end), np.float32) * A(hann_window(frame_length, periodic=True))
        return _t(np.fft.rfft(fr.astype(np.float64), n=fft_length or frame_length, axis=-1).astype(np.complex64))

    def linear_to_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000, lower_edge_hertz=125.0,
                                    upper_edge_hertz=3800.0, dtype=None, name=None):
        nsb = int(A(num_spectrogram_bins))
        hz2mel = lambda f: 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
        lin = np.linspace(0.0, sample_rate / 2.0, nsb)[1:]
        spec_mel = hz2mel(lin)[:, None]
        edges = np.linspace(hz2mel(lower_edge_hertz), hz2mel(upper_edge_hertz), num_mel_bins + 2)
        lo, ce, hi = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
        w = np.maximum(0.0, np.minimum((spec_mel - lo) / (ce - lo), (hi - spec_mel) / (hi - ce)))
        return _t(np.pad(w, [(1, 0), (0, 0)]).astype(np.float32))

    tf.signal = types.SimpleNamespace(hann_window=hann_window, frame=frame, stft=stft, linear_to_mel_weight_matrix=linear_to_mel_weight_matrix)
    tf.abs = lambda x, name=None: _t(np.abs(A(x)))
    tf.random.normal = None  # bind a generator before running weight-noise bodies
    return tf


@contextlib.contextmanager
def reference_runtime(stubs=None):
    """Install `tensorflow`, `keras` (+ keras.src.*) and a `tensorflow_asr` package whose SUBMODULES are found under /root/reference
    (its own `__init__`, which imports the real TensorFlow and walks every subpackage, is replaced), with stubs for the modules
    that need libraries absent here (env_util: TF internals; data_util: librosa; tokenizers: tensorflow_text; file_util; gammatone).
    Yields (tf, keras).  sys.modules is restored on exit."""
    tf = extend_tf_for_models(tf_shim.make_tf())
    keras, kmods = make_keras(tf)
    pkg = types.ModuleType("tensorflow_asr")
    pkg.__path__ = [f"{REFERENCE_ROOT}/tensorflow_asr"]
    pkg.tf, pkg.keras = tf, keras
    env = types.ModuleType("tensorflow_asr.utils.env_util")
    env.TF_CUDNN = False
    env.has_devices = lambda *_a, **_k: False
    pkg.env_util = env
    data_util = types.ModuleType("tensorflow_asr.utils.data_util")

    def get(obj, path, default=None):  # utils/data_util.py:38-60 (the module itself imports librosa)
        cur = obj
        for key in str(path).split("."):
            if isinstance(cur, dict):
                cur = cur.get(key, default)
            elif isinstance(cur, list):
                try:
                    cur = cur[int(key)]
                except (IndexError, ValueError):
                    return default
            else:
                return default
        return cur

    data_util.get = get
    tok = types.ModuleType("tensorflow_asr.tokenizers")
    tok.Tokenizer = object
    file_util = types.ModuleType("tensorflow_asr.utils.file_util")
    gamma = types.ModuleType("tensorflow_asr.features.gammatone")
    trainer = types.ModuleType("keras.src.backend.tensorflow.trainer")
    trainer.TensorFlowTrainer = type("TensorFlowTrainer", (), {})
    trainer.reduce_per_replica = None
    loss_mod = types.ModuleType("keras.src.losses.loss")
    losses_pkg = types.ModuleType("keras.src.losses")
    losses_pkg.loss = loss_mod
    opt_pkg, base_opt = types.ModuleType("keras.src.optimizers"), types.ModuleType("keras.src.optimizers.base_optimizer")
    base_opt.BaseOptimizer = object
    opt_pkg.base_optimizer = base_opt
    new = dict(kmods)
    new.update({"tensorflow": tf, "tensorflow_asr": pkg, "tensorflow_asr.utils.env_util": env, "tensorflow_asr.utils.data_util": data_util,
                "tensorflow_asr.tokenizers": tok, "tensorflow_asr.utils.file_util": file_util, "tensorflow_asr.features.gammatone": gamma,
                "keras.src.backend.tensorflow": types.ModuleType("keras.src.backend.tensorflow"), "keras.src.backend.tensorflow.trainer": trainer,
                "keras.src.losses": losses_pkg, "keras.src.losses.loss": loss_mod,
                "keras.src.optimizers": opt_pkg, "keras.src.optimizers.base_optimizer": base_opt})
    # every reference sub-package as a bare namespace (their __init__.py files glob-import all siblings)
    import os
    for dirpath, dirnames, _files in os.walk(f"{REFERENCE_ROOT}/tensorflow_asr"):
        dirnames[:] = [d for d in dirnames if not d.startswith("__")]
        rel = os.path.relpath(dirpath, REFERENCE_ROOT).replace(os.sep, ".")
        if rel != "tensorflow_asr" and os.path.isfile(os.path.join(dirpath, "__init__.py")):
            m = types.ModuleType(rel)
            m.__path__ = [dirpath]
            new[rel] = m
    new.update(stubs or {})
    saved = {k: v for k, v in sys.modules.items() if k == "tensorflow" or k.startswith(("tensorflow.", "tensorflow_asr", "keras"))}
    for k in saved:
        del sys.modules[k]
    sys.modules.update(new)
    REGISTRY.clear()
    _AUTO_NAMES.clear()
    del RECORD[:]
    try:
        yield tf, keras
    finally:
        for k in [k for k in sys.modules if k == "tensorflow" or k.startswith(("tensorflow.", "tensorflow_asr", "keras"))]:
            del sys.modules[k]
        sys.modules.update(saved)


def assign_paths(model, prefix=""):
    """give every Dropout its path (for the injected-mask dictionary) - call once after construction."""
    here = f"{prefix}{model.name}"
    for sub in model._sublayers():
        if isinstance(sub, Dropout):
            sub.path = f"{here}/{sub.name}"
        assign_paths(sub, here + "/")
