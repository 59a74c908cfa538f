"""The HIP path against the reference's OWN model classes (tests/golden/wiring_*.npz = `tensorflow_asr.models.transducer.conformer.
Conformer` / `...contextnet.ContextNet` constructed and run from /root/reference over the oracle's tf / keras shims, see
tests/test_reference_wiring.py): identical weights and signals, logits of `Transducer.call` in training mode (batch statistics over
every frame, attention masking padded query rows only - as the reference's classes did it), the moving statistics that call leaves
behind, and the inference-mode logits that use them."""
import os

import numpy as np
import pytest
import torch

from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer
from tensorflowasr_amd.contextnet import ContextNetTransducer
from tensorflowasr_amd.schemas import TrainInput

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(GOLD, f"wiring_{name}.npz"))
    W = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("W/")}
    inp = TrainInput(torch.from_numpy(z["signals"]), torch.from_numpy(z["signals_length"]), torch.from_numpy(z["predictions"]),
                     torch.from_numpy(z["predictions_length"]))
    return z, W, inp


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name,over", [("conformer", {}), ("conformer_streaming", dict(chunk_size=2, history_size=4, convm_dw_norm="layer", sub_norm="layer"))])
def test_conformer_logits_match_reference_classes(dev, dtype, name, over):
    z, W, inp = _load(name)
    cfg = configs.conformer_tiny(**over)
    cfg.time_masking, cfg.freq_masking = {}, {}  # the golden run has no augmentation (SpecAugment: specaugment_reference.npz)
    model = ConformerTransducer(cfg, dev, dtype=dtype, seed=0)
    model.ps.import_keras(W)
    # bf16: 3 x the measured error of this tiny model (d = 32: 1.06e-2 train / 8.3e-3 eval, streaming 8.2e-3); f32 measures 1.5e-6
    tol = 2e-3 if dtype == torch.float32 else 3.2e-2
    for native in (True, False):
        model.native_blocks = native
        model.ps.import_keras(W)  # (resets the moving statistics)
        logits, elen, _ = model._forward(inp, True, None)
        torch.cuda.synchronize()
        assert elen == z["train/logits_length"].tolist()
        got = logits.float().cpu().numpy()
        # rows past an utterance's encoder length hold whatever the (unmasked) network computes there in the reference too: compared as well
        print(f"\n[wiring] {name} {str(dtype)[6:]} native={native}: train logits rel L2 {_rel(got, z['train/logits']):.3e}")
        assert _rel(got, z["train/logits"]) < tol, (native, _rel(got, z["train/logits"]))
        if dtype == torch.float32:
            np.testing.assert_allclose(got, z["train/logits"], rtol=2e-3, atol=2e-3)
            for k in z.files:  # keras momentum 0.99 update with the batch moments over EVERY frame (no mask reached a BatchNorm)
                if k.startswith("after_train/"):
                    np.testing.assert_allclose(model.ps.state[k[len("after_train/"):]].cpu().numpy(), z[k], rtol=1e-3, atol=1e-5)
        ev, elen2, _ = model._forward(inp, False, None)
        torch.cuda.synchronize()
        assert elen2 == elen
        print(f"[wiring] {name} {str(dtype)[6:]} native={native}: eval logits rel L2 {_rel(ev.float().cpu().numpy(), z['eval/logits']):.3e}")
        assert _rel(ev.float().cpu().numpy(), z["eval/logits"]) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_contextnet_logits_match_reference_classes(dev, dtype):
    z, W, inp = _load("contextnet")
    cfg = configs.contextnet_tiny()
    cfg.time_masking, cfg.freq_masking = {}, {}
    model = ContextNetTransducer(cfg, dev, dtype=dtype, seed=0)
    for name, w in W.items():
        if name.endswith(("/mm", "/mv")):
            model.ps.state[name].copy_(w.to(model.device))
        else:
            model.ps.p(name).copy_(w.to(model.device).reshape(model.ps.p(name).shape))
    model.ps.refresh_shadow()
    logits, elen, _ = model._forward(inp, True, None)
    torch.cuda.synchronize()
    assert elen == z["train/logits_length"].tolist()
    got = logits.float().cpu().numpy()
    tol = 2e-3 if dtype == torch.float32 else 4e-2
    print(f"\n[wiring] contextnet {str(dtype)[6:]}: train logits rel L2 {_rel(got, z['train/logits']):.3e}")
    assert _rel(got, z["train/logits"]) < tol, _rel(got, z["train/logits"])
    if dtype == torch.float32:
        np.testing.assert_allclose(got, z["train/logits"], rtol=2e-3, atol=2e-3)
        for k in z.files:
            if k.startswith("after_train/"):
                np.testing.assert_allclose(model.ps.state[k[len("after_train/"):]].cpu().numpy(), z[k], rtol=1e-3, atol=1e-5)
    ev, _, _ = model._forward(inp, False, None)
    torch.cuda.synchronize()
    print(f"[wiring] contextnet {str(dtype)[6:]}: eval logits rel L2 {_rel(ev.float().cpu().numpy(), z['eval/logits']):.3e}")
    assert _rel(ev.float().cpu().numpy(), z["eval/logits"]) < tol
