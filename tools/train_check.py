import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); sys.argv=["x"]; spec.loader.exec_module(b)
from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer
cfg = configs.conformer_m()
dev = torch.device("cuda", 0)
model = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
model.optimizer["schedule"] = 3e-4
batch = b.make_batch(cfg, 16, seed=3, padding="batch", size="LibriSpeech-shaped")
data = b.to_train_data(batch, dev)
hist = []
for i in range(40):
    out = model.train_step(data)
    hist.append(float(out["loss"].float().mean()))
print("loss:", [round(h, 1) for h in hist[::4]], "finite", bool(np.isfinite(hist).all()))
print("max |param|", float(model.ps.flat.abs().max()), "grad finite", bool(torch.isfinite(model.ps.grad).all()))
