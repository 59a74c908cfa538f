"""GPU: the EXACT batches bench.py times (make_batch seeds 10 and 23: 32 LibriSpeech-shaped utterances, Conformer-M, 16 blocks, sync-BN
moments over 14 784 / 23 776 rows, packed lattices of 343 k / 486 k cells) through the benchmarked bf16 path and through the GPU f32
parity mode, SpecAugment masks injected (the same draw into both), dropout 0.

The chain of evidence this closes: reference classes executed from /root/reference == oracle (tests/test_reference_wiring.py, CPU)
== GPU f32 parity mode at B <= 2, 16 blocks, T' = 462 / 743 (tests/test_parity_baseline_gpu.py) ; GPU f32 parity mode == GPU bf16 default
path AT THE BENCH SHAPE (this file).  The CPU oracle cannot run B = 32 at these lengths (a 44 GB dense lattice); the f32 mode can.
Tolerances: loss 1e-3 relative (north_star's bound for the RNN-T loss), gradient 2e-2 relative L2 over the whole flat gradient and
6e-2 per variable family (bf16 storage of activations over 16 blocks; the per-variable table of DESIGN section 4 has the B <= 2 figures).
Reference step: models/base_model.py:149-183."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [10, 23])
def test_bench_batch_bf16_step_matches_f32_parity_mode(dev, seed):
    import bench
    from tensorflowasr_amd import configs
    from tensorflowasr_amd.conformer import ConformerTransducer

    cfg = configs.conformer_m()
    cfg.dropout = 0.0
    batch = bench.make_batch(cfg, 32, seed=seed, padding="batch", size="LibriSpeech-shaped")
    data = bench.to_train_data(batch, dev)
    out = {}
    masks = None
    for dtype in (torch.float32, torch.bfloat16):
        model = ConformerTransducer(cfg, dev, dtype=dtype, seed=0)
        if masks is None:
            masks = model.draw_specaugment([-(-int(n) // cfg.frame_step) for n in batch["nsamp"]])
            assert masks[1] is not None and int(masks[1][..., 1].sum()) > 0  # time masks are really applied
        model.zero_grad()
        costs = model.loss_and_backward(data, True, masks)
        torch.cuda.synchronize()
        g = model.ps.export_keras(model.ps.grad)
        out[dtype] = (costs.float().cpu().numpy().astype(np.float64), {k: v.double().numpy() for k, v in g.items()})
        del model
        torch.cuda.empty_cache()
    c32, g32 = out[torch.float32]
    c16, g16 = out[torch.bfloat16]
    assert np.isfinite(c32).all() and np.isfinite(c16).all() and c32.shape == (32,)
    # loss: the batch mean (what the optimizer follows) and every utterance
    assert abs(c16.mean() - c32.mean()) <= 1e-3 * abs(c32.mean()), (c16.mean(), c32.mean())
    np.testing.assert_allclose(c16, c32, rtol=2e-3)
    # gradient: relative L2 over the whole gradient, and per family of variables
    num = sum(float(((g16[k] - g32[k]) ** 2).sum()) for k in g32)
    den = sum(float((g32[k] ** 2).sum()) for k in g32)
    rel = (num / den) ** 0.5
    assert rel <= 2e-2, rel
    fam = {}
    for k in g32:
        f = k.split("/")[0] if not k.startswith("enc/block") else "enc/blocks"
        a = fam.setdefault(f, [0.0, 0.0])
        a[0] += float(((g16[k] - g32[k]) ** 2).sum())
        a[1] += float((g32[k] ** 2).sum())
    for f, (n_, d_) in fam.items():
        assert (n_ / max(d_, 1e-300)) ** 0.5 <= 6e-2, (f, (n_ / d_) ** 0.5)
