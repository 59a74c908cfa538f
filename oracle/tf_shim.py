"""ORACLE tooling (test infrastructure only).

A NumPy stand-in for the handful of `tf.*` primitives that the reference's pure-Python hot-path files call,
so that the reference's OWN source files can be executed in this container (TensorFlow is not installed):
    tensorflow_asr/losses/impl/rnnt.py                    (RNN-T loss + closed-form gradient)
    tensorflow_asr/models/layers/multihead_attention.py   (rel_left_shift, compute_streaming_mask — function bodies)
    tensorflow_asr/models/layers/positional_encoding.py   (compute_sinusoid_position_encoding)
Each primitive follows the documented TF semantics (eager mode, float32 default).  Used only by
oracle/gen_golden_from_reference.py to write tests/golden/*.npz; nothing here is shipped.
"""
import contextlib
import importlib.util
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


class _DType:
    def __init__(self, np_dtype, name):
        self.np = np_dtype
        self.name = name

    def __eq__(self, o):
        return np.dtype(getattr(o, "np", o)) == np.dtype(self.np) if not isinstance(o, str) else self.name == o

    def __hash__(self):
        return hash(self.name)

    @property
    def min(self):
        return np.finfo(self.np).min

    @property
    def max(self):
        return np.finfo(self.np).max


def _npd(dt):
    if dt is None:
        return None
    return getattr(dt, "np", dt)


class _Shape(tuple):
    def as_list(self):
        return list(self)


class T(np.ndarray):
    """ndarray whose .shape has .as_list() and whose dtype compares with shim dtypes."""

    @property
    def shape(self):  # noqa: D401
        return _Shape(np.ndarray.shape.__get__(self))

    def numpy(self):
        return np.asarray(self)


def _t(x, dtype=None):
    a = np.asarray(x, dtype=_npd(dtype))
    if a.dtype == np.float64 and dtype is None and not isinstance(x, np.ndarray):
        a = a.astype(np.float32)  # python floats become float32 like tf.constant
    return a.view(T)


def _diag_part_v2(input, k, padding_value):  # noqa: A002
    """tf.raw_ops.MatrixDiagPartV2: out[..., d, :] = diagonal (k_hi - d), LEFT-aligned, padded on the right."""
    x = np.asarray(input)
    k_lo, k_hi = (int(k[0]), int(k[1])) if isinstance(k, (tuple, list)) else (int(k), int(k))
    M, N = x.shape[-2], x.shape[-1]
    lens = [min(M + min(d, 0), N - max(d, 0)) for d in range(k_lo, k_hi + 1)]
    maxlen = max(lens)
    out = np.full(x.shape[:-2] + (k_hi - k_lo + 1, maxlen), padding_value, dtype=x.dtype)
    for row, d in enumerate(range(k_hi, k_lo - 1, -1)):
        ln = min(M + min(d, 0), N - max(d, 0))
        if ln <= 0:
            continue
        i = np.arange(ln)
        out[..., row, :ln] = x[..., i - min(d, 0), i + max(d, 0)]
    if k_lo == k_hi:
        out = out[..., 0, :]
    return out.view(T)


def _scan(fn, elems, initializer, reverse=False):
    n = len(elems[0]) if isinstance(elems, (tuple, list)) else len(elems)
    order = range(n - 1, -1, -1) if reverse else range(n)
    acc = initializer
    outs = [None] * n
    for i in order:
        e = tuple(x[i] for x in elems) if isinstance(elems, (tuple, list)) else elems[i]
        acc = fn(acc, e)
        outs[i] = acc
    return _t(np.stack([np.asarray(o) for o in outs], axis=0))


def _gather_nd(params, indices, batch_dims=0):
    p, idx = np.asarray(params), np.asarray(indices)
    if batch_dims == 0:
        return _t(p[tuple(np.moveaxis(idx, -1, 0))])
    assert batch_dims == 1
    return _t(np.stack([p[b][tuple(np.moveaxis(idx[b], -1, 0))] for b in range(p.shape[0])], axis=0))


def _scatter_nd(indices, updates, shape, name=None):
    out = np.zeros([int(s) for s in shape], dtype=np.asarray(updates).dtype)
    idx = np.asarray(indices)
    np.add.at(out, tuple(np.moveaxis(idx, -1, 0)), np.asarray(updates))
    return _t(out)


def _sequence_mask(lengths, maxlen=None, dtype=None):
    lengths = np.asarray(lengths)
    maxlen = int(lengths.max()) if maxlen is None else int(maxlen)
    m = np.arange(maxlen)[(None,) * lengths.ndim] < lengths[..., None]
    return _t(m.astype(_npd(dtype) or np.bool_))


def _one_hot(indices, depth, dtype=None):
    idx = np.asarray(indices)
    out = (idx[..., None] == np.arange(int(depth))).astype(_npd(dtype) or np.float32)
    return _t(out)


def _pad(x, paddings, mode="CONSTANT", constant_values=0):
    return _t(np.pad(np.asarray(x), [(int(a), int(b)) for a, b in paddings], constant_values=constant_values))


def _slice(x, begin, size):
    x = np.asarray(x)
    sl = tuple(slice(int(b), None if int(s) == -1 else int(b) + int(s)) for b, s in zip(begin, size))
    return _t(x[sl])


def _logsumexp(x, axis=None, keepdims=False):
    x = np.asarray(x)
    with np.errstate(invalid="ignore", divide="ignore"):
        m = np.max(x, axis=axis, keepdims=True)
        m0 = np.where(np.isfinite(m), m, 0)
        r = np.log(np.sum(np.exp(x - m0), axis=axis, keepdims=True)) + m0
    return _t(r if keepdims else np.squeeze(r, axis=axis))


def _log_softmax(x, axis=-1):
    x = np.asarray(x)
    m = x.max(axis=axis, keepdims=True)
    return _t(x - m - np.log(np.exp(x - m).sum(axis=axis, keepdims=True)))


def make_tf():
    tf = types.ModuleType("tensorflow")
    for nm, d in [("float32", np.float32), ("float64", np.float64), ("float16", np.float16), ("int32", np.int32),
                  ("int64", np.int64), ("bool", np.bool_)]:
        setattr(tf, nm, _DType(d, nm))
    tf.bfloat16 = _DType(np.float16, "bfloat16_unused")  # only ever compared against, never produced
    tf.dtypes = types.SimpleNamespace(float32=tf.float32)
    tf.Tensor = T
    np_err = dict(invalid="ignore", divide="ignore", over="ignore")

    def w(f):
        def g(*a, **k):
            k.pop("name", None)
            with np.errstate(**np_err):
                return _t(f(*a, **k))
        return g

    tf.convert_to_tensor = lambda x, dtype=None, name=None: _t(x, dtype)
    tf.constant = lambda x, dtype=None, shape=None: _t(x, dtype)
    tf.cast = lambda x, dtype: _t(np.asarray(x).astype(_npd(dtype)))
    tf.shape = lambda x, out_type=None: _t(np.array(np.asarray(x).shape, dtype=_npd(out_type) or np.int32))
    tf.reshape = lambda x, shape: _t(np.reshape(np.asarray(x), [int(s) for s in np.asarray(shape).reshape(-1)]
                                                if not isinstance(shape, (list, tuple)) else [int(s) for s in shape]))
    tf.transpose = lambda x, perm=None: _t(np.transpose(np.asarray(x), perm))
    tf.reverse = lambda x, axis: _t(np.flip(np.asarray(x), axis=tuple(axis)))
    tf.concat = lambda xs, axis: _t(np.concatenate([np.asarray(v) for v in xs], axis=axis))
    tf.stack = lambda xs, axis=0: _t(np.stack([np.asarray(v) for v in xs], axis=axis))
    tf.unstack = lambda x, axis=0: [_t(v) for v in np.moveaxis(np.asarray(x), axis, 0)]
    tf.expand_dims = lambda x, axis: _t(np.expand_dims(np.asarray(x), axis))
    tf.tile = lambda x, multiples: _t(np.tile(np.asarray(x), [int(m) for m in multiples]))
    tf.repeat = lambda x, repeats, axis=None: _t(np.repeat(np.asarray(x), repeats, axis=axis))
    tf.range = lambda *a, dtype=None, **k: _t(np.arange(*[np.asarray(v).item() if np.ndim(v) == 0 else v for v in a],
                                                        dtype=_npd(dtype) or _npd(k.get("dtype"))))
    _shp = lambda shape: [int(s) for s in np.atleast_1d(np.asarray(shape))]
    tf.ones = lambda shape, dtype=None, name=None: _t(np.ones(_shp(shape), _npd(dtype) or np.float32))
    tf.zeros = lambda shape, dtype=None, name=None: _t(np.zeros(_shp(shape), _npd(dtype) or np.float32))
    tf.fill = lambda shape, v: _t(np.full([int(s) for s in shape], v))
    tf.zeros_like = lambda x, dtype=None: _t(np.zeros_like(np.asarray(x), dtype=_npd(dtype)))
    tf.ones_like = lambda x, dtype=None: _t(np.ones_like(np.asarray(x), dtype=_npd(dtype)))
    tf.where = w(lambda c, x=None, y=None: np.where(np.asarray(c), np.asarray(x), np.asarray(y)))
    tf.equal = w(lambda a, b: np.asarray(a) == np.asarray(b))
    tf.less = w(lambda a, b: np.asarray(a) < np.asarray(b))
    tf.multiply = w(lambda a, b: np.asarray(a) * np.asarray(b))
    tf.add = w(lambda a, b: np.asarray(a) + np.asarray(b))
    tf.exp = w(lambda x: np.exp(np.asarray(x)))
    tf.sin = w(lambda x: np.sin(np.asarray(x)))
    tf.cos = w(lambda x: np.cos(np.asarray(x)))
    tf.pow = w(lambda a, b: np.power(np.asarray(a), np.asarray(b)))
    tf.maximum = w(lambda a, b: np.maximum(np.asarray(a), np.asarray(b)))
    tf.minimum = w(lambda a, b: np.minimum(np.asarray(a), np.asarray(b)))
    tf.reduce_max = w(lambda x, axis=None, keepdims=False: np.max(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.reduce_sum = w(lambda x, axis=None, keepdims=False: np.sum(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.reduce_all = w(lambda x, axis=None: np.all(np.asarray(x), axis=axis))
    tf.einsum = w(lambda eq, *ops, **k: np.einsum(eq, *[np.asarray(o) for o in ops]))
    tf.pad = _pad
    tf.slice = _slice
    tf.scan = _scan
    tf.one_hot = _one_hot
    tf.sequence_mask = _sequence_mask
    tf.gather_nd = _gather_nd
    tf.scatter_nd = _scatter_nd
    tf.ensure_shape = lambda x, shape: x
    def _map_fn(fn, elems, dtype=None, fn_output_signature=None):
        if isinstance(elems, (tuple, list)):
            outs = [fn(tuple(_t(e[i]) for e in elems)) for i in range(len(elems[0]))]
        else:
            outs = [fn(_t(e)) for e in elems]
        if isinstance(outs[0], (tuple, list)):
            return tuple(_t(np.stack([np.asarray(o[k]) for o in outs])) for k in range(len(outs[0])))
        return _t(np.stack([np.asarray(o) for o in outs]))

    tf.map_fn = _map_fn
    tf.roll = lambda input, shift, axis: _t(np.roll(np.asarray(input), int(shift), axis=axis))  # noqa: A002
    tf.device = lambda *_a, **_k: contextlib.nullcontext()
    tf.name_scope = lambda *_a, **_k: contextlib.nullcontext()
    tf.newaxis = None
    tf.math = types.SimpleNamespace(
        is_nan=w(lambda x: np.isnan(np.asarray(x))), log=w(lambda x: np.log(np.asarray(x))),
        reduce_logsumexp=lambda x, axis=None, keepdims=False: _logsumexp(x, axis, keepdims),
        floordiv=w(lambda a, b: np.asarray(a) // np.asarray(b)), logical_not=w(lambda a: ~np.asarray(a)),
        logical_and=w(lambda a, b: np.asarray(a) & np.asarray(b)), minimum=tf.minimum, maximum=tf.maximum)
    tf.nn = types.SimpleNamespace(log_softmax=lambda x, axis=-1: _log_softmax(x, axis))
    tf.raw_ops = types.SimpleNamespace(MatrixDiagPartV2=_diag_part_v2)
    tf.compat = types.SimpleNamespace(dimension_value=lambda d: None if d is None else int(d))
    tf.linalg = types.SimpleNamespace(
        band_part=w(lambda x, lo, hi: np.asarray(x) & (np.tril(np.ones(np.asarray(x).shape[-2:], bool), hi if hi >= 0 else 10**9)
                                                        & np.triu(np.ones(np.asarray(x).shape[-2:], bool), -lo if lo >= 0 else -10**9))))
    tf.custom_gradient = lambda f: f
    _extend(tf, w)
    return tf


class _TensorArray:
    """tf.TensorArray (fixed size, clear_after_read=False): write returns the array, read/stack as documented."""

    def __init__(self, dtype=None, size=0, dynamic_size=False, clear_after_read=True, element_shape=None, **_k):
        self._d = _npd(dtype)
        self._v = [np.zeros((), self._d) for _ in range(int(np.asarray(size)))]

    def write(self, index, value):
        self._v[int(np.asarray(index))] = np.asarray(value, self._d)
        return self

    def read(self, index):
        return _t(self._v[int(np.asarray(index))])

    def stack(self):
        return _t(np.stack(self._v) if self._v else np.zeros((0,), self._d))


class InjectedUniform:
    """tf.random.uniform(shape=[]) stand-in backed by a numpy Generator: float dtypes -> rng.uniform(minval, maxval),
    integer dtypes -> rng.integers(minval, maxval) (maxval exclusive, like TF).  Every draw is recorded."""

    def __init__(self, rng):
        self.rng, self.log = rng, []

    def __call__(self, shape=(), minval=0, maxval=None, dtype=None, seed=None, name=None):
        assert list(shape) == []
        d = np.dtype(_npd(dtype) or np.float32)
        lo, hi = np.asarray(minval).item(), (None if maxval is None else np.asarray(maxval).item())
        if np.issubdtype(d, np.integer):
            if hi <= lo:
                raise ValueError(f"tf.random.uniform: maxval {hi} must be > minval {lo} (InvalidArgumentError in TF)")
            v = int(self.rng.integers(lo, hi))
        else:
            v = float(self.rng.uniform(lo, 1.0 if hi is None else hi))
        self.log.append(v)
        return _t(np.asarray(v, d))


def _extend(tf, w):
    """Primitives used by the greedy-search loops (base_transducer.py:496-712), SpecAugment (specaugment.py:58-137),
    the schedule / accumulator / loss-length / frontend helpers and MultiHeadRelativeAttention._compute_attention."""
    tf.greater_equal = w(lambda a, b: np.asarray(a) >= np.asarray(b))
    tf.greater = w(lambda a, b: np.asarray(a) > np.asarray(b))
    tf.less_equal = w(lambda a, b: np.asarray(a) <= np.asarray(b))
    tf.not_equal = w(lambda a, b: np.asarray(a) != np.asarray(b))
    tf.logical_not = w(lambda a: ~np.asarray(a, bool))
    tf.logical_or = w(lambda a, b: np.asarray(a, bool) | np.asarray(b, bool))
    tf.logical_and = w(lambda a, b: np.asarray(a, bool) & np.asarray(b, bool))
    tf.abs = w(lambda x: np.abs(np.asarray(x)))
    tf.square = w(lambda x: np.square(np.asarray(x)))
    tf.sqrt = w(lambda x: np.sqrt(np.asarray(x)))
    tf.floor = w(lambda x: np.floor(np.asarray(x)))
    tf.divide = w(lambda a, b: np.asarray(a) / np.asarray(b))
    tf.subtract = w(lambda a, b: np.asarray(a) - np.asarray(b))
    tf.reduce_mean = w(lambda x, axis=None, keepdims=False: np.mean(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.reduce_min = w(lambda x, axis=None, keepdims=False: np.min(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.broadcast_to = w(lambda x, shape: np.broadcast_to(np.asarray(x), [int(v) for v in shape]))
    tf.split = lambda x, n, axis=-1: [_t(v) for v in np.split(np.asarray(x), n, axis=axis)]
    tf.argmax = lambda x, axis=None, output_type=None: _t(np.argmax(np.asarray(x), axis=axis).astype(_npd(output_type) or np.int64))
    tf.matmul = w(lambda a, b: np.matmul(np.asarray(a), np.asarray(b)))
    tf.einsum = w(lambda eq, *ops, **k: np.einsum(eq, *[np.asarray(o) for o in ops]))  # `optimize=` is a scheduling hint
    tf.TensorArray = _TensorArray
    tf.TensorShape = lambda dims=None: list(dims or [])
    tf.cond = lambda pred, true_fn, false_fn: (true_fn() if bool(np.asarray(pred)) else false_fn())

    def while_loop(cond, body, loop_vars, back_prop=True, maximum_iterations=None, **_k):
        v, n = tuple(loop_vars), 0
        while bool(np.asarray(cond(*v)).all()):
            v = tuple(body(*v))
            n += 1
            if maximum_iterations is not None and n >= maximum_iterations:
                break
            if n > getattr(tf, "_while_cap", 1_000_000):
                raise RuntimeError("tf.while_loop stand-in: runaway loop")
        tf._last_while_iterations = n
        return v

    tf.while_loop = while_loop

    def tensor_scatter_nd_update(tensor, indices, updates):
        out = np.array(np.asarray(tensor), copy=True)
        idx = np.asarray(indices)
        out[tuple(np.moveaxis(idx, -1, 0))] = np.asarray(updates)
        return _t(out)

    tf.tensor_scatter_nd_update = tensor_scatter_nd_update
    tf.math.greater_equal, tf.math.less, tf.math.reduce_all = tf.greater_equal, tf.less, tf.reduce_all
    tf.math.ceil = w(lambda x: np.ceil(np.asarray(x)))
    tf.math.reduce_variance = w(lambda x, axis=None, keepdims=False: np.var(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.nn.sigmoid = w(lambda x: (1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))).astype(np.asarray(x).dtype))
    tf.nn.softmax = w(lambda x, axis=-1: (lambda e: e / e.sum(axis=axis, keepdims=True))(np.exp(np.asarray(x) - np.asarray(x).max(axis=axis, keepdims=True))))
    tf.random = types.SimpleNamespace(uniform=None)  # bind an InjectedUniform before running SpecAugment bodies


def extract_functions(rel_path, names, namespace):
    """exec only the named top-level functions / class methods (`Class.method`) of a reference source file inside
    `namespace` (used when the module itself imports Keras internals that do not exist here)."""
    import ast

    src = open(f"{REFERENCE_ROOT}/{rel_path}").read()
    tree = ast.parse(src)
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), rel_path, "exec")
            exec(code, namespace)
            out[node.name] = namespace[node.name]
        if isinstance(node, ast.ClassDef):
            for sub in node.body:
                key = f"{node.name}.{sub.name}" if isinstance(sub, ast.FunctionDef) else None
                if key in names:
                    sub.decorator_list = []
                    code = compile(ast.Module(body=[sub], type_ignores=[]), rel_path, "exec")
                    ns = dict(namespace)
                    exec(code, ns)
                    out[key] = ns[sub.name]
    missing = set(names) - set(out)
    if missing:
        raise KeyError(f"{missing} not found in {rel_path}")
    return out


def load_reference_module(rel_path, mod_name, extra_modules=None):
    """Execute one reference source file (by path) with `tensorflow_asr.tf` bound to the shim."""
    tf = make_tf()
    pkg = types.ModuleType("tensorflow_asr")
    pkg.__path__ = []
    pkg.tf = tf
    pkg.schemas = types.ModuleType("tensorflow_asr.schemas")
    pkg.schemas.TrainOutput = tuple  # only used as a type annotation by impl/rnnt.py:187
    pkg.keras = types.ModuleType("keras")
    utils = types.ModuleType("tensorflow_asr.utils")
    utils.__path__ = []
    saved = {k: sys.modules.get(k) for k in ("tensorflow", "tensorflow_asr", "tensorflow_asr.schemas",
                                             "tensorflow_asr.utils", "tensorflow_asr.utils.shape_util")}
    sys.modules.update({"tensorflow": tf, "tensorflow_asr": pkg, "tensorflow_asr.schemas": pkg.schemas,
                        "tensorflow_asr.utils": utils})
    try:
        spec = importlib.util.spec_from_file_location("tensorflow_asr.utils.shape_util",
                                                      f"{REFERENCE_ROOT}/tensorflow_asr/utils/shape_util.py")
        su = importlib.util.module_from_spec(spec)
        sys.modules["tensorflow_asr.utils.shape_util"] = su
        spec.loader.exec_module(su)
        utils.shape_util = su
        for k, v in (extra_modules or {}).items():
            sys.modules[k] = v
        spec = importlib.util.spec_from_file_location(mod_name, f"{REFERENCE_ROOT}/{rel_path}")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod, tf
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def extend_for_ctc_tpu(tf):
    """Primitives (with TensorFlow's KEYWORD names) that tensorflow_asr/losses/impl/ctc_tpu.py calls, so that the reference's pure-TF
    CTC (`ctc_loss_tpu`, :1295; ClassicCtcLossData :821-1290) runs over NumPy.  tf.custom_gradient stays the identity (the decorated
    methods return (value, backprop)): the generator reads `loss` / `gradient` from the loss-data object directly."""
    A = np.asarray
    err = dict(invalid="ignore", divide="ignore", over="ignore")

    def ints(v):
        return [int(A(s)) for s in (v if isinstance(v, (list, tuple)) else A(v).reshape(-1))]

    def w(f):
        def g(*a, **k):
            k.pop("name", None)
            with np.errstate(**err):
                return _t(f(*a, **k))
        return g

    tf.inf = np.inf
    tf.Variable = T
    tf.stop_gradient = lambda x, name=None: _t(x)
    tf.constant = lambda value, dtype=None, shape=None, name=None: _t(value, dtype)
    tf.reduce_logsumexp = lambda input_tensor, axis=None, keepdims=False, name=None: _logsumexp(A(input_tensor), tuple(axis) if isinstance(axis, list) else axis, keepdims)
    tf.reduce_max = w(lambda input_tensor, axis=None, keepdims=False: np.max(A(input_tensor), axis=tuple(axis) if isinstance(axis, list) else axis, keepdims=keepdims))
    tf.reduce_sum = w(lambda input_tensor, axis=None, keepdims=False: np.sum(A(input_tensor), axis=tuple(axis) if isinstance(axis, list) else axis, keepdims=keepdims))
    tf.where = w(lambda condition, x=None, y=None: np.where(A(condition), A(x), A(y)))
    tf.reshape = lambda tensor, shape, name=None: _t(np.reshape(A(tensor), ints(shape)))
    tf.transpose = lambda a, perm=None, name=None: _t(np.transpose(A(a), None if perm is None else ints(perm)))
    tf.expand_dims = lambda input, axis, name=None: _t(np.expand_dims(A(input), int(axis)))  # noqa: A002
    tf.squeeze = lambda input, axis=None, name=None: _t(np.squeeze(A(input), axis=None if axis is None else tuple(np.atleast_1d(axis))))  # noqa: A002
    tf.tile = lambda input, multiples, name=None: _t(np.tile(A(input), ints(multiples)))  # noqa: A002
    tf.stack = lambda values, axis=0, name=None: _t(np.stack([A(v) for v in values], axis=axis))
    tf.concat = lambda values, axis, name=None: _t(np.concatenate([A(v) for v in values], axis=axis))
    tf.roll = lambda input, shift, axis, name=None: _t(np.roll(A(input), int(A(shift)), axis=int(axis)))  # noqa: A002
    tf.gather = lambda params, indices, axis=None, batch_dims=0, name=None: _gather(A(params), A(indices), axis, batch_dims)
    tf.one_hot = lambda indices, depth, dtype=None, name=None: _t((A(indices)[..., None] == np.arange(int(A(depth)))).astype(_npd(dtype) or np.float32))
    tf.sequence_mask = lambda lengths, maxlen=None, dtype=None, name=None: _sequence_mask(lengths, None if maxlen is None else int(A(maxlen)), dtype)
    tf.cumsum = w(lambda x, axis=0, exclusive=False, reverse=False: np.cumsum(A(x), axis=axis))
    tf.meshgrid = lambda *a, indexing="xy": [_t(v) for v in np.meshgrid(*[A(v) for v in a], indexing=indexing)]
    tf.eye = lambda num_rows, num_columns=None, dtype=None, name=None: _t(np.eye(int(A(num_rows)), None if num_columns is None else int(A(num_columns)), dtype=_npd(dtype) or np.float32))
    tf.zeros = lambda shape, dtype=None, name=None: _t(np.zeros(ints(shape), _npd(dtype) or np.float32))
    tf.ones = lambda shape, dtype=None, name=None: _t(np.ones(ints(shape), _npd(dtype) or np.float32))
    tf.cast = lambda x, dtype, name=None: _t(A(x).astype(_npd(dtype)))
    tf.maximum = w(lambda x, y: np.maximum(A(x), A(y)))
    tf.scatter_nd = lambda indices, updates, shape, name=None: _scatter_nd(A(indices), A(updates), ints(shape))
    tf.cond = lambda pred, true_fn, false_fn, name=None: (true_fn() if bool(A(pred)) else false_fn())

    def pad(tensor, paddings, mode="CONSTANT", constant_values=0, name=None):
        p = [[int(A(a)), int(A(b))] for a, b in paddings]
        return _t(np.pad(A(tensor), p, constant_values=A(constant_values).item()))

    tf.pad = pad

    def band_part(input, num_lower, num_upper, name=None):  # noqa: A002
        x = A(input)
        n, m = x.shape[-2:]
        i, j = np.arange(n)[:, None], np.arange(m)[None, :]
        keep = ((num_lower < 0) | (i - j <= num_lower)) & ((num_upper < 0) | (j - i <= num_upper))
        return _t(np.where(keep, x, np.zeros((), x.dtype)))

    def set_diag(input, diagonal, name=None):  # noqa: A002
        x = np.array(A(input), copy=True)
        n = min(x.shape[-2:])
        idx = np.arange(n)
        x[..., idx, idx] = A(diagonal)
        return _t(x)

    tf.linalg = types.SimpleNamespace(band_part=band_part, set_diag=set_diag)

    def seg(op, init):
        def f(data, segment_ids, num_segments, name=None):
            d, s, n = A(data), A(segment_ids), int(A(num_segments))
            out = np.full((n,) + d.shape[s.ndim:], init(d.dtype), d.dtype)
            op.at(out, s.reshape(-1), d.reshape((-1,) + d.shape[s.ndim:]))
            return _t(out)
        return f

    tf.math.softplus = w(lambda features: np.logaddexp(0.0, A(features)))
    tf.math.expm1 = w(lambda x: np.expm1(A(x)))
    tf.math.log = w(lambda x: np.log(A(x)))
    # TF's kernel starts every segment at numeric_limits<T>::lowest() (FINITE: -3.4e38 for f32) and reduces with max, so a segment
    # whose data are all -inf reports lowest(), not -inf: ctc_tpu.py:109-117 relies on it (-inf - lowest() = -inf, never NaN)
    tf.math.unsorted_segment_max = seg(np.maximum, lambda dt: np.finfo(dt).min if np.issubdtype(dt, np.floating) else np.iinfo(dt).min)
    tf.math.unsorted_segment_sum = seg(np.add, lambda dt: 0)
    return tf


def _gather(params, indices, axis, batch_dims):
    if batch_dims == 0:
        return _t(np.take(params, indices, axis=0 if axis is None else axis))
    # batch_dims = b: leading b dims of params / indices are paired; gather along `axis` (default = batch_dims)
    axis = batch_dims if axis is None else axis
    lead = params.shape[:batch_dims]
    out = []
    for idx in np.ndindex(*lead):
        out.append(np.take(params[idx], indices[idx], axis=axis - batch_dims))
    first = out[0]
    return _t(np.stack(out).reshape(lead + first.shape))
