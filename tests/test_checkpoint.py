"""Checkpoint interop: Keras variable paths / layouts of tensorflowasr_amd/checkpoint.py."""
import numpy as np
import pytest
import torch

from tensorflowasr_amd import checkpoint as ck
from tensorflowasr_amd import configs, params


def _exported_template(cfg, seed=0):
    """What ParamStore.export_keras() returns (names, Keras layouts), built on the host from the parameter specs."""
    g = torch.Generator().manual_seed(seed)
    H, dh, d = cfg.num_heads, cfg.head_size, cfg.dmodel
    out = {}
    for spec in params.param_specs(cfg):
        name, shape = spec[0], tuple(spec[1])
        t = torch.randn(*shape, generator=g)
        if name.endswith("qkv/w"):
            for i, k in enumerate("qkv"):
                out[name[:-5] + k + "/w"] = t[:, i * H * dh:(i + 1) * H * dh].reshape(d, H, dh).clone()
        elif name.endswith("qkv/b"):
            for i, k in enumerate("qkv"):
                out[name[:-5] + k + "/b"] = t[i * H * dh:(i + 1) * H * dh].reshape(H, dh).clone()
        elif name.endswith("pos/w"):
            out[name] = t.reshape(d, H, dh)
        elif name.endswith("pos/b") or name in ("enc/u", "enc/v") or name.endswith(("mhsa/u", "mhsa/v")):
            out[name] = t.reshape(H, dh)
        elif name.endswith("mhsa/o/w"):
            out[name] = t.reshape(H, dh, d)
        else:
            out[name] = t
    for bn in params.bn_names(cfg):
        C = out[bn + "/g"].shape[0]
        out[bn + "/mm"], out[bn + "/mv"] = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    return out


def test_keras_paths_and_layouts():
    cfg = configs.conformer_tiny()
    tpl = _exported_template(cfg)
    arrays = ck.to_keras(tpl)
    assert len(arrays) == len(tpl)                                   # one Keras variable per tensor, no collisions
    d, H, dh, K = cfg.dmodel, cfg.num_heads, cfg.head_size, cfg.kernel_size
    b0 = "conformer_encoder/block_0/"
    want = {
        "conformer_encoder/subsampling/block_0/conv_0/kernel": (3, 3, 1, cfg.filters),
        "conformer_encoder/subsampling/block_1/bn_1/moving_variance": (cfg.filters,),
        "conformer_encoder/linear/kernel": tuple(tpl["enc/linear/w"].shape),
        "conformer_encoder/content_attention_bias": (H, dh),
        b0 + "ff_module_1/dense_1/kernel": (d, 4 * d),
        b0 + "ff_module_2/ln/gamma": (d,),
        b0 + "mhsa_module/mhsa/query/kernel": (d, H, dh),
        b0 + "mhsa_module/mhsa/value/bias": (H, dh),
        b0 + "mhsa_module/mhsa/encoding/kernel": (d, H, dh),
        b0 + "mhsa_module/mhsa/attention_output/kernel": (H, dh, d),
        b0 + "conv_module/pw_conv_1/kernel": (1, d, 2 * d),           # Conv1D kernel
        b0 + "conv_module/dw_conv/kernel": (K, d, 1),                  # DepthwiseConv1D kernel
        b0 + "conv_module/dw_bn/moving_mean": (d,),
        b0 + "ln/beta": (d,),
        "prediction/embedding/embeddings": (cfg.vocab_size, cfg.embed_dim),
        "prediction/lstm_0/lstm_cell/recurrent_kernel": (cfg.rnn_units, 4 * cfg.rnn_units),
        "prediction/ln_0/gamma": (cfg.rnn_units,),
        "joint/vocab/kernel": (cfg.joint_dim, cfg.vocab_size),
    }
    for path, shape in want.items():
        assert path in arrays and arrays[path].shape == shape and arrays[path].dtype == np.float32, path
    back = ck.from_keras(arrays, tpl)
    assert set(back) == set(tpl)
    for k in tpl:
        assert back[k].shape == tuple(tpl[k].shape) and np.array_equal(back[k], tpl[k].numpy()), k


def test_strict_loading_errors():
    cfg = configs.conformer_tiny()
    tpl = _exported_template(cfg)
    arrays = ck.to_keras(tpl)
    missing = dict(arrays)
    missing.pop("joint/vocab/bias")
    with pytest.raises(KeyError, match="joint/vocab/bias"):
        ck.from_keras(missing, tpl)
    assert "joint/vocab/b" not in ck.from_keras(missing, tpl, strict=False)
    extra = dict(arrays, **{"optimizer/iterations": np.zeros(1)})
    with pytest.raises(KeyError, match="optimizer/iterations"):
        ck.from_keras(extra, tpl)
    bad = dict(arrays)
    bad["joint/enc/kernel"] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError, match="joint/enc/kernel"):
        ck.from_keras(bad, tpl)
    with pytest.raises(KeyError):
        ck.keras_path("enc/cn/block0/whatever")


def test_ctc_head_per_layer_biases_and_layernorm_dw_variant_have_keras_paths():
    """Conformer-CTC (Dense "logits" head, per-layer attention biases) and the streaming Conformer (LayerNormalization in
    the depthwise-norm slot, named dw_ln, gamma / beta only) round-trip through the Keras-path checkpoint like the transducer."""
    cfg = configs.conformer_ctc_s(num_blocks=2)
    tpl = _exported_template(cfg)
    arrays = ck.to_keras(tpl)
    assert len(arrays) == len(tpl)
    H, dh = cfg.num_heads, cfg.head_size
    assert arrays["conformer_decoder/logits/kernel"].shape == (cfg.dmodel, cfg.vocab_size)
    assert arrays["conformer_encoder/block_1/mhsa_module/mhsa/content_attention_bias"].shape == (H, dh)
    assert arrays["conformer_encoder/block_0/mhsa_module/mhsa/positional_attention_bias"].shape == (H, dh)
    assert "conformer_encoder/content_attention_bias" not in arrays and "joint/vocab/kernel" not in arrays
    back = ck.from_keras(arrays, tpl)
    assert all(np.array_equal(back[k], tpl[k].numpy()) for k in tpl)
    # streaming: dw_ln, no moving statistics anywhere in the conv modules
    scfg = configs.conformer_tiny(convm_dw_norm="layer", chunk_size=2, history_size=4)
    assert params.bn_names(scfg) == ["enc/sub/bn0", "enc/sub/bn1"]
    stpl = _exported_template(scfg)
    assert not [k for k in stpl if "/conv/bn/m" in k]
    sarr = ck.to_keras(stpl, dw_norm="layer")
    assert len(sarr) == len(stpl)
    assert "conformer_encoder/block_0/conv_module/dw_ln/gamma" in sarr and "conformer_encoder/block_0/conv_module/dw_ln/beta" in sarr
    assert not [p for p in sarr if "dw_bn" in p or ("conv_module" in p and "moving_" in p)]
    sback = ck.from_keras(sarr, stpl, dw_norm="layer")
    assert all(np.array_equal(sback[k], stpl[k].numpy()) for k in stpl)
    with pytest.raises(KeyError):  # a batch-norm checkpoint does not load into the layer-norm model silently
        ck.from_keras(ck.to_keras(stpl, dw_norm="batch"), stpl, dw_norm="layer")


class _HostModel:
    """export_keras() of a model without a device (the container / path logic is host code)."""

    def __init__(self, cfg, seed=0):
        self.cfg = cfg
        tpl = _exported_template(cfg, seed)

        class PS:
            def export_keras(self_inner):
                return tpl

        self.ps, self.tpl = PS(), tpl


@pytest.mark.parametrize("make", [lambda: configs.conformer_tiny(), lambda: configs.conformer_ctc_s(num_blocks=2),
                                  lambda: configs.conformer_tiny(convm_dw_norm="layer", chunk_size=2, history_size=4)])
def test_weights_h5_writer_round_trip(tmp_path, make):
    """BaseModel.save_weights -> `.weights.h5` (callbacks.py:190-239, base_model.py:55-61): the file written by save_weights_h5 is read
    back by the reader, resolves onto every model variable exactly once, and - where the real HDF5 library is available (h5py under
    /opt/conda in the build container) - reads back identically through libhdf5."""
    import os
    import subprocess

    from tensorflowasr_amd.h5lite import H5File

    m = _HostModel(make())
    path = str(tmp_path / "model.weights.h5")
    names = ck.save_weights_h5(m, path)
    assert len(names) == len(m.tpl) and all("/vars/" in n for n in names)
    with H5File(path) as f:
        datasets = f.datasets()
    assert sorted(datasets) == names
    got, unused = ck.from_weights_h5(datasets, m.tpl)
    assert unused == [] and sorted(got) == sorted(m.tpl)
    for k, t in m.tpl.items():
        np.testing.assert_array_equal(got[k], t.numpy(), err_msg=k)
    conda = "/opt/conda/bin/python3.9"
    if os.path.exists(conda) and subprocess.run([conda, "-c", "import h5py"], capture_output=True).returncode == 0:
        npz = str(tmp_path / "model.npz")
        np.savez(npz, **{k.replace("/", "|"): v for k, v in datasets.items()})
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([conda, os.path.join(root, "oracle", "check_h5_roundtrip.py"), path, npz], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_h5_writer_wide_and_deep_groups(tmp_path):
    """more children than one symbol node / one B-tree node holds (two-level group B-tree), scalars, empty and 64-bit datasets"""
    from tensorflowasr_amd.h5lite import H5File, write_h5

    rng = np.random.default_rng(0)
    ds = {f"wide/n{i}": np.arange(i % 5 + 1, dtype=np.int32) for i in range(300)}
    ds.update({"a/b/c/d/e/f32": rng.standard_normal((3, 4)).astype(np.float32), "a/f64": rng.standard_normal((2, 2, 2)),
               "a/scalar": np.float32(3.25), "a/i64": np.int64(-7), "a/empty": np.zeros((0, 4), np.float32), "a/u8": np.arange(17, dtype=np.uint8)})
    p = str(tmp_path / "t.h5")
    write_h5(p, ds, groups=["vars", "optimizer/vars"])
    with H5File(p) as f:
        got = f.datasets()
    assert sorted(got) == sorted(ds)
    for k, a in ds.items():
        assert got[k].dtype == np.asarray(a).dtype and got[k].shape == np.asarray(a).shape, k
        np.testing.assert_array_equal(got[k], a)


@pytest.mark.gpu
def test_save_and_load_weights_round_trip(tmp_path):
    from tensorflowasr_amd.conformer import ConformerTransducer

    dev = torch.device("cuda", 0)
    cfg = configs.conformer_tiny()
    a = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=1)
    for k, v in a.ps.state.items():                                   # non-trivial BatchNorm moving statistics
        v.copy_(torch.rand_like(v) + (0.5 if k.endswith("mv") else 0.0))
    path = tmp_path / "model.weights.npz"
    names = ck.save_weights(a, str(path))
    assert "conformer_encoder/block_1/conv_module/dw_bn/moving_variance" in names
    b = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=2)
    assert not torch.equal(a.ps.flat, b.ps.flat)
    ck.load_weights(b, str(path))
    assert torch.equal(a.ps.flat, b.ps.flat) and torch.equal(a.ps.shadow, b.ps.shadow)
    ea, eb = a.ps.export_keras(), b.ps.export_keras()  # (reference layouts: the zero padding of heads / channels is not checkpointed)
    for k in a.ps.state:
        assert torch.equal(ea[k], eb[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ctc", "streaming"])
def test_save_and_load_weights_round_trip_ctc_and_streaming(tmp_path, kind):
    """npz AND `.weights.h5` save -> load for the Conformer-CTC model (per-layer attention biases, Dense head) and for the
    streaming configuration (LayerNormalization depthwise norm: no moving statistics)."""
    from tensorflowasr_amd.conformer import ConformerTransducer
    from tensorflowasr_amd.ctc_model import ConformerCTC

    dev = torch.device("cuda", 0)
    if kind == "ctc":
        cfg = configs.conformer_ctc_s(num_blocks=2)
        mk = lambda seed: ConformerCTC(cfg, dev, dtype=torch.bfloat16, seed=seed)
    else:
        cfg = configs.conformer_tiny(convm_dw_norm="layer", chunk_size=2, history_size=4)
        mk = lambda seed: ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=seed)
    a = mk(1)
    for k, v in a.ps.state.items():
        v.copy_(torch.rand_like(v) + (0.5 if k.endswith("mv") else 0.0))
    for ext, save, load in ((".npz", ck.save_weights, ck.load_weights), (".weights.h5", ck.save_weights_h5, ck.load_weights_h5)):
        path = str(tmp_path / ("model" + ext))
        names = save(a, path)
        assert not [n for n in names if "dw_bn" in n] or kind == "ctc"
        b = mk(2)
        assert not torch.equal(a.ps.flat, b.ps.flat)
        load(b, path)
        assert torch.equal(a.ps.flat, b.ps.flat) and torch.equal(a.ps.shadow, b.ps.shadow)
        ea, eb = a.ps.export_keras(), b.ps.export_keras()
        for k in a.ps.state:
            assert torch.equal(ea[k], eb[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["weights_h5_tiny_attrpaths.weights.h5", "weights_h5_tiny_layernames.weights.h5"])
def test_load_weights_h5_into_the_model(golden_dir, fixture):
    """A `.weights.h5` container written by the real HDF5 library is read by the pure-Python reader and lands in the flat parameter
    buffer exactly like the same arrays imported directly (callbacks.py:190-239 / base_model.py:59-61 interop)."""
    import os

    from tensorflowasr_amd.conformer import ConformerTransducer

    dev = torch.device("cuda", 0)
    cfg = configs.conformer_tiny()
    z = np.load(os.path.join(golden_dir, "weights_h5_tiny_expected.npz"))
    W = {k.replace("|", "/"): torch.from_numpy(z[k]) for k in z.files}
    a = ConformerTransducer(cfg, dev, dtype=torch.float32, seed=1)
    a.ps.import_keras(W)
    b = ConformerTransducer(cfg, dev, dtype=torch.float32, seed=2)
    got = ck.load_weights_h5(b, os.path.join(golden_dir, fixture))
    assert len(got) == len(W)
    assert torch.equal(a.ps.flat, b.ps.flat)
    for k in a.ps.state:
        assert torch.equal(a.ps.state[k], b.ps.state[k]), k


@pytest.mark.gpu
def test_training_state_round_trip(tmp_path):
    """save_state / load_state: weights + Adam moments + optimizer step + dropout epoch, so a resumed run continues the schedule."""
    from tensorflowasr_amd import _smoke_model
    from tensorflowasr_amd.conformer import ConformerTransducer

    dev = torch.device("cuda", 0)
    cfg, a, data, *_ = _smoke_model.make(dev)
    for _ in range(3):
        a.train_step(data)
    p = tmp_path / "state.npz"
    a._gradn_epoch, a._gwn_epoch = 7, 5  # (counters behind the gradient / weight noise seeds: part of the state)
    ck.save_state(a, str(p))
    b = ConformerTransducer(cfg, dev, dtype=a.dtype, seed=99)
    ck.load_state(b, str(p))
    assert b.step == a.step == 3 and b._drop_epoch == a._drop_epoch
    assert (b._gradn_epoch, b._gwn_epoch) == (7, 5)
    for name in ("flat", "adam_m", "adam_v", "shadow"):
        assert torch.equal(getattr(a.ps, name), getattr(b.ps, name)), name
    la = a.train_step(data, masks=(None, None))["loss"]
    lb = b.train_step(data, masks=(None, None))["loss"]
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-5)


@pytest.mark.gpu
def test_training_state_is_layout_independent(tmp_path):
    """The state file holds the LOGICAL layout: written by a bf16 model (heads 8 -> 64 and filters 32 -> 64 zero-padded on the device), it
    resumes into an f32 model (no padding) with the same logical parameters, Adam moments and moving statistics - and back."""
    from tensorflowasr_amd import _smoke_model
    from tensorflowasr_amd.conformer import ConformerTransducer

    dev = torch.device("cuda", 0)
    cfg, a, data, *_ = _smoke_model.make(dev, dtype=torch.bfloat16)
    assert a.ps.head_phys != cfg.head_size or a.ps.filt_phys != cfg.filters
    for _ in range(2):
        a.train_step(data)
    p = tmp_path / "state.npz"
    ck.save_state(a, str(p))
    b = ConformerTransducer(cfg, dev, dtype=torch.float32, seed=5)
    assert b.ps.n != a.ps.n  # different device layouts
    ck.load_state(b, str(p))
    assert b.step == a.step
    for buf in ("flat", "adam_m", "adam_v"):
        assert torch.equal(a.ps.to_logical(getattr(a.ps, buf)), b.ps.to_logical(getattr(b.ps, buf))), buf
    ea, eb = a.ps.export_keras(), b.ps.export_keras()
    assert ea.keys() == eb.keys() and all(torch.equal(ea[k], eb[k]) for k in ea)
    p2 = tmp_path / "state2.npz"
    ck.save_state(b, str(p2))
    c = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=6)
    ck.load_state(c, str(p2))
    for buf in ("flat", "adam_m", "adam_v"):
        assert torch.equal(getattr(a.ps, buf), getattr(c.ps, buf)), buf


def test_h5_weights_refuse_contextnet(tmp_path):
    """the .weights.h5 path table covers the Conformer variables; ContextNet raises NotImplementedError (not a KeyError from the matcher)"""
    import types

    fake = types.SimpleNamespace(cfg=types.SimpleNamespace(encoder="contextnet"))
    with pytest.raises(NotImplementedError):
        ck.save_weights_h5(fake, str(tmp_path / "w.weights.h5"))
    with pytest.raises(NotImplementedError):
        ck.load_weights_h5(fake, str(tmp_path / "w.weights.h5"))


def test_device_layout_state_files_are_refused(tmp_path):
    """a state file without the logical-layout marker holds the flat buffer in the physical order of the build that wrote it;
    that order changed (deferred region) while names and count did not, so such a file must be refused, not copied raw."""
    import pytest

    from tensorflowasr_amd import checkpoint

    class PS:
        names = ["a", "b"]
        n = 4

    class M:
        ps = PS()

    f = tmp_path / "old_state.npz"
    np.savez(f, names=np.asarray(["a", "b"]), n=np.asarray(4), flat=np.zeros(4, np.float32), adam_m=np.zeros(4, np.float32), adam_v=np.zeros(4, np.float32),
             step=np.asarray(3), drop_epoch=np.asarray(0), ga_count=np.asarray(0))
    with pytest.raises(ValueError, match="physical device layout"):
        checkpoint.load_state(M(), str(f))
