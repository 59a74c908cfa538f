"""The HIP path against goldens produced by executing the reference's own source files (oracle/gen_golden_from_reference.py):
greedy search loops (base_transducer.py:496-712), SpecAugment application (specaugment.py:58-137),
_compute_attention (multihead_attention.py:543-582), GLU (glu.py:25-28).  CPU counterparts: tests/test_reference_pins.py."""
import os

import numpy as np
import pytest
import torch

from tensorflowasr_amd import configs
from tensorflowasr_amd import kernels as K
from tensorflowasr_amd.conformer import ConformerTransducer
from tests.test_reference_pins import _masks_from_draws, greedy_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(golden_dir):
    return lambda name: np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_greedy_search_matches_reference_loop_bodies(dev, G, dtype):
    """The search arithmetic is f32 for every storage type, so both models must reproduce the reference loops' tokens."""
    g = G("greedy_reference.npz")
    cfg = configs.conformer_tiny()
    for name in [str(n) for n in g["names"]]:
        model = ConformerTransducer(cfg, dev, dtype=dtype, seed=0)
        model.ps.import_keras(greedy_weights(g, name))
        enc = torch.from_numpy(g[f"{name}_enc"]).to(dev).to(dtype)
        if dtype != torch.float32:  # the golden encoder output is f32: hand the bf16 model exactly those values
            enc = torch.from_numpy(g[f"{name}_enc"]).to(dev)
        out = model.recognize_encoded(enc, g[f"{name}_len"].tolist())
        np.testing.assert_array_equal(out.tokens.cpu().numpy(), g[f"{name}_tokens"], err_msg=name)
        np.testing.assert_array_equal(out.next_tokens.cpu().numpy().reshape(-1), g[f"{name}_next_tokens"].reshape(-1), err_msg=name)
        np.testing.assert_allclose(out.next_decoder_states.cpu().numpy(), g[f"{name}_next_states"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", ["full", "padded", "prob"])
def test_hip_specaugment_matches_reference_augment_bodies(dev, G, case):
    g = G("specaugment_reference.npz")
    feat, length, want, prob = g[f"{case}_in"], int(g[f"{case}_len"]), g[f"{case}_out"], float(g[f"{case}_prob"])
    fm, tm = _masks_from_draws(g[f"{case}_draws"], length, prob=prob)
    x = torch.from_numpy(feat[None, :, :, 0].copy()).to(dev)
    K.specaugment(x, torch.from_numpy(fm).to(dev), torch.from_numpy(tm).to(dev), 0.0)
    np.testing.assert_array_equal(x.cpu().numpy()[0], want[:, :, 0])


@pytest.mark.parametrize("case,dtype", [("eq", torch.float32), ("ragged", torch.float32), ("head64", torch.float32), ("head64", torch.bfloat16)])
def test_hip_attention_core_matches_reference_compute_attention(dev, G, case, dtype):
    """f32: unfused kernels (any head size); bf16 + head 64: the fused flash-style kernel bench.py runs."""
    g = G("attention_core_reference.npz")
    q, k, v, table = (g[f"{case}_{n}"] for n in ("q", "k", "v", "table"))
    B, T, H, dh = q.shape
    HD = H * dh
    cfg = configs.conformer_tiny(num_heads=H, head_size=dh, dmodel=max(HD, 8))
    model = ConformerTransducer(cfg, dev, dtype=dtype, seed=0)
    assert model._fused_attention() == (dtype == torch.bfloat16)
    model.ps.p("enc/u").copy_(torch.from_numpy(g[f"{case}_cb"].reshape(-1)))
    model.ps.p("enc/v").copy_(torch.from_numpy(g[f"{case}_pb"].reshape(-1)))
    qkv = torch.from_numpy(np.concatenate([q.reshape(B * T, HD), k.reshape(B * T, HD), v.reshape(B * T, HD)], 1)).to(dev).to(dtype).contiguous()
    pext = torch.from_numpy(np.concatenate([table.reshape(2 * T - 1, HD), np.zeros((1, HD), np.float32)], 0)).to(dev).to(dtype).contiguous()
    lens = torch.from_numpy(g[f"{case}_lens"]).to(dev)
    att, _ = model.attention_core(qkv, pext, B, T, lens)
    torch.cuda.synchronize()
    want = g[f"{case}_out"].reshape(B * T, HD)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(att.float().cpu().numpy(), want, rtol=tol, atol=tol)


def test_hip_glu_matches_reference_body(dev, G):
    g = G("misc_reference.npz")
    y = K.glu_fwd(torch.from_numpy(g["glu_in"]).to(dev))
    np.testing.assert_allclose(y.cpu().numpy(), g["glu_out"], rtol=1e-6, atol=1e-7)
