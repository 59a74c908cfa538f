"""Generate tests/golden/*.npz by executing the reference's OWN source files (from /root/reference) over
oracle/tf_shim (NumPy stand-in for tf.*).  Run in the build container only (the GPU box has no
/root/reference); the .npz fixtures are committed.

    python oracle/gen_golden_from_reference.py
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tf_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_rnnt():
    """tensorflow_asr/losses/impl/rnnt.py: compute_rnnt_loss_and_grad_helper (use_cpu=True)."""
    mod, tf = tf_shim.load_reference_module("tensorflow_asr/losses/impl/rnnt.py", "tensorflow_asr.losses.impl.rnnt")
    cases = {
        # name: (seed, B, T, U, V, logit_len, label_len)
        "small": (11, 2, 5, 3, 4, [5, 4], [3, 2]),
        "mid": (12, 3, 12, 7, 16, [12, 9, 10], [7, 3, 5]),
        "equal": (13, 2, 9, 4, 32, [9, 9], [4, 4]),
        "u1": (14, 2, 6, 1, 8, [6, 3], [1, 1]),
        "t1": (15, 1, 1, 2, 8, [1], [2]),  # logit_len < label_len is clamped by BaseLoss before the helper: T stays 1 -> use clamp
    }
    for name, (seed, B, T, U, V, tl, ul) in cases.items():
        rng = np.random.default_rng(seed)
        if name == "t1":
            T = 2
            tl = [2]
        logits = (rng.standard_normal((B, T, U + 1, V)) * 2.0).astype(np.float32)
        labels = rng.integers(1, V, (B, U)).astype(np.int32)
        tl = np.asarray(tl, np.int32)
        ul = np.asarray(ul, np.int32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss, grads = mod.compute_rnnt_loss_and_grad_helper(
                tf.convert_to_tensor(logits), tf.convert_to_tensor(labels), tf.convert_to_tensor(ul),
                tf.convert_to_tensor(tl), use_cpu=True)
        np.savez_compressed(os.path.join(OUT, f"rnnt_reference_{name}.npz"), logits=logits, labels=labels,
                            label_len=ul, logit_len=tl, loss=np.asarray(loss, np.float32),
                            grads=np.asarray(grads, np.float32))
        print(f"rnnt_reference_{name}: loss={np.asarray(loss)}")


def _shape_util(tf):
    import types

    su = types.ModuleType("shape_util")

    def shape_list(x, out_type=None):
        return [int(v) for v in np.asarray(x).shape]

    su.shape_list = shape_list
    return su


def gen_attention():
    """multihead_attention.py: rel_left_shift (:27-77) and compute_streaming_mask (:104-143) function bodies."""
    tf = tf_shim.make_tf()
    ns = {"tf": tf, "shape_util": _shape_util(tf)}
    fns = tf_shim.extract_functions("tensorflow_asr/models/layers/multihead_attention.py",
                                    ["rel_left_shift", "compute_streaming_mask"], ns)
    rng = np.random.default_rng(21)
    out = {}
    for T in (3, 5, 8):
        x = rng.standard_normal((2, 2, T, 2 * T - 1)).astype(np.float32)
        out[f"shift_in_{T}"] = x
        out[f"shift_out_{T}"] = np.asarray(fns["rel_left_shift"](tf.convert_to_tensor(x), causal=False))
    out["mask_2_2_8"] = np.asarray(fns["compute_streaming_mask"](2, 2, tf.zeros([5, 8, 8])))
    out["mask_3_3_14"] = np.asarray(fns["compute_streaming_mask"](3, 3, tf.zeros([5, 14, 14])))
    out["mask_4_m1_10"] = np.asarray(fns["compute_streaming_mask"](4, -1, tf.zeros([1, 10, 10])))
    np.savez_compressed(os.path.join(OUT, "attention_reference.npz"), **out)
    print("attention_reference:", {k: v.shape for k, v in out.items()})


def gen_posenc():
    """positional_encoding.py: compute_position/compute_sinusoid_position_encoding (:22-52) and the body of
    RelativeSinusoidalPositionalEncoding.call (:114-174) with a stand-in `self` (scale None, memory 0, causal False)."""
    import types

    tf = tf_shim.make_tf()
    ns = {"tf": tf, "shape_util": _shape_util(tf)}
    fns = tf_shim.extract_functions("tensorflow_asr/models/layers/positional_encoding.py",
                                    ["compute_position", "compute_sinusoid_position_encoding",
                                     "RelativeSinusoidalPositionalEncoding.call"], ns)
    out = {}
    for (B, T, d, lens) in [(3, 6, 8, [6, 4, 1]), (2, 9, 16, [9, 5])]:
        me = types.SimpleNamespace(_scale=None, _memory_length=0, _interleave=True, _causal=False,
                                   do=lambda pe, training=False: pe)
        x = tf.zeros([B, T, d])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, pe = fns["RelativeSinusoidalPositionalEncoding.call"](me, (x, tf.convert_to_tensor(np.asarray(lens, np.int32))))
        out[f"pe_{B}_{T}_{d}"] = np.asarray(pe, np.float32)
        out[f"len_{B}_{T}_{d}"] = np.asarray(lens, np.int32)
    np.savez_compressed(os.path.join(OUT, "relpe_reference.npz"), **out)
    print("relpe_reference:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_rnnt()
    if "--all" in sys.argv or len(sys.argv) == 1:
        for fn in ("gen_attention", "gen_posenc"):
            if fn in globals():
                globals()[fn]()
