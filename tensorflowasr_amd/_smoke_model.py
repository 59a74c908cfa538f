"""smoke(): one tiny Conformer-Transducer train step on the HIP path (called from __graft_entry__.smoke, which checks
the loss against the oracle; the oracle import lives THERE, never in the product package)."""
import numpy as np
import torch

from . import configs
from .conformer import ConformerTransducer
from .schemas import TrainData, TrainInput, TrainLabel


def make(dev, dtype=torch.float32, seed=0):
    cfg = configs.conformer_tiny()
    model = ConformerTransducer(cfg, dev, dtype=dtype, seed=seed)
    rng = np.random.default_rng(seed)
    B, N, U = 2, 4000, 5
    sig = np.clip(rng.standard_normal((B, N)) * 0.1, -1, 1).astype(np.float32)
    labels = rng.integers(1, cfg.vocab_size, (B, U)).astype(np.int32)
    preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)
    data = TrainData(TrainInput(torch.from_numpy(sig), torch.tensor([N, N], dtype=torch.int32), torch.from_numpy(preds),
                                torch.tensor([U + 1, U + 1], dtype=torch.int32)),
                     TrainLabel(torch.from_numpy(labels), torch.tensor([U, U], dtype=torch.int32)))
    return cfg, model, data, sig, labels, preds


def run(dev):
    cfg, model, data, *_ = make(dev)
    model.optimizer["schedule"] = 1e-3
    l0 = float(model.train_step(data, masks=(None, None))["loss"].mean())
    for _ in range(3):
        l1 = float(model.train_step(data, masks=(None, None))["loss"].mean())
    torch.cuda.synchronize()
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0, (l0, l1)
    return l0, l1
