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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_ctc_matches_reference_pure_tf_ctc(dev, G, dtype):
    """tfasr_ctc_loss (through the C ABI) against the reference's own pure-TF CTC (losses/impl/ctc_tpu.py over the shim): loss within
    the north star's 1e-3 relative, gradient w.r.t. the logits, +inf loss / zero gradient for an infeasible sample."""
    g = G("ctc_tpu_reference.npz")
    for name in [str(n) for n in g["names"]]:
        logits, labels = g[f"{name}_logits"], g[f"{name}_labels"]
        ll, tl = g[f"{name}_label_len"], g[f"{name}_logit_len"]
        U = labels.shape[1] - 1
        xin = torch.from_numpy(logits).to(dtype)
        if dtype != torch.float32 and not np.isfinite(g[f"{name}_loss"]).all():
            continue  # (bf16 rounding of the INPUT changes nothing about feasibility, but the finite sample is covered by the other cases)
        costs, grads = K.ctc_loss_fwd_bwd(xin.to(dev), torch.from_numpy(np.ascontiguousarray(labels[:, :U])).to(dev), torch.from_numpy(ll).to(dev),
                                          torch.from_numpy(tl).to(dev))
        torch.cuda.synchronize()
        want, wg = g[f"{name}_loss"], g[f"{name}_grad_logits"]
        ok = np.isfinite(want)
        got = costs.cpu().numpy()
        assert np.array_equal(np.isfinite(got), ok), name
        if dtype == torch.float32:
            np.testing.assert_allclose(got[ok], want[ok], rtol=2e-5, err_msg=name)
            np.testing.assert_allclose(grads.cpu().numpy(), wg, rtol=2e-4, atol=2e-5, err_msg=name)
        else:  # logits rounded to bf16 on the way in: the loss still has to sit within 1e-3 relative of the f32 reference ... usually
            np.testing.assert_allclose(got[ok], want[ok], rtol=2e-2, err_msg=name)
            np.testing.assert_allclose(grads.float().cpu().numpy(), wg, rtol=5e-2, atol=2e-2, err_msg=name)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_joint_matches_reference_joint_call(dev, G, dtype):
    """TransducerJoint.call + TransducerJointMerge.call bodies (base_transducer.py:199-207,280-293) vs the HIP joint network."""
    g = G("joint_callnext_reference.npz")
    from oracle import conformer_ref as R

    cfg = configs.conformer_tiny()
    model = ConformerTransducer(cfg, dev, dtype=dtype, seed=0)
    model.ps.import_keras(R.init_weights(R.conformer_config("tiny"), seed=int(g["wseed"]), scale_bias=0.1))
    enc, pred = g["joint_enc"], g["joint_pred"]
    B, T, d = enc.shape
    U1 = pred.shape[1]
    logits = model.joint_fwd(torch.from_numpy(enc).to(dev).to(dtype).view(B * T, d).contiguous(),
                             torch.from_numpy(pred).to(dev).to(dtype).view(B * U1, -1).contiguous(), B, T, U1, None)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(logits.float().cpu().numpy(), g["joint_logits"], rtol=tol, atol=tol)


def test_hip_call_next_matches_reference_body(dev, G):
    """Transducer.call_next (base_transducer.py:437-464) chained three times vs the HIP search step: one greedy iteration from the
    golden's (frame, previous token, state) must land on the arg-max of the golden's log-softmax and on its new state."""
    g = G("joint_callnext_reference.npz")
    from oracle import conformer_ref as R

    cfg = configs.conformer_tiny()
    model = ConformerTransducer(cfg, dev, dtype=torch.float32, seed=0)
    model.ps.import_keras(R.init_weights(R.conformer_config("tiny"), seed=int(g["wseed"]), scale_bias=0.1))
    for i in range(3):
        frames, toks, st = g["next_frames"][i], g["next_tokens"][i], g[f"next_state{i}"]
        for b in range(frames.shape[0]):  # recognize_single on a one-frame utterance performs exactly one call_next first
            out = model.recognize_encoded(torch.from_numpy(frames[b:b + 1]).to(dev), [1], previous_tokens=torch.from_numpy(toks[b:b + 1]).to(dev),
                                          previous_decoder_states=torch.from_numpy(st[b:b + 1]).to(dev), max_tokens_per_frame=1)
            want = int(np.argmax(g[f"next_ytu{i}"][b].reshape(-1)))
            got = out.tokens.cpu().numpy().reshape(-1)
            if want == 0:
                assert (got == 0).all()
            else:  # a non-blank arg-max is emitted and the state advances to the golden's new state
                assert got[0] == want, (i, b, got, want)
                np.testing.assert_allclose(out.next_decoder_states.cpu().numpy()[0], g[f"next_state{i + 1}"][b], rtol=1e-4, atol=1e-5)
