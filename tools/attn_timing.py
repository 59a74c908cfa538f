"""Per-phase shader clocks of the KEY-side attention backward (relattn_fused_bwd_k_kernel: it runs last and owns the probe buffer) inside a Conformer-M train step.
Probe build: tools/build_probe_lib.sh attn_fused.hip -DTFASR_ATTN_TIMING=3 (=1: the query-side backward owns the buffer: python tools/attn_timing.py q), then
  TFASR_LIB=$PWD/tools/hwprobe/libtfasr_probe.so python tools/attn_timing.py
Each workgroup's wave 0 sums the clocks of five phases of its key-block loop: 0 DMA issue + wait + barrier, 1 S^T / dP^T / G^T products +
strip store, 2 scores read back, exp, dS, skewed image, dS store, 3 the two dq products, 4 closing barrier; plus loop total, epilogue."""
import ctypes, importlib.util, os, sys
WHICH = sys.argv[1] if len(sys.argv) > 1 else "k"
sys.path.insert(0, os.getcwd())
import numpy as np, torch
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); sys.argv = ["x"]; spec.loader.exec_module(b)
from tensorflowasr_amd import configs, _lib
from tensorflowasr_amd.conformer import ConformerTransducer
cfg = configs.conformer_m()
dev = torch.device("cuda", 0)
model = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
batch = b.make_batch(cfg, 32, seed=3, padding="batch", size="LibriSpeech-shaped")
data = b.to_train_data(batch, dev)
for i in range(3):
    model.train_step(data)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["TFASR_LIB"])
n = 8 * 8192
buf = (ctypes.c_longlong * n)()
assert lib.tfasr_attn_timing_read(buf, n) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(8192, 8)
live = a[(a[:, 7] > 0) & (a[:, 5] > 0)]
print("workgroups with a key loop:", len(live), "key blocks each:", int(np.median(live[:, 7])))
per = live[:, :5] / live[:, 7:8]
names = (["DMA wait + barrier", "S^T dP^T G^T products + strip store", "scores, exp, dS, skewed image, dS store", "dq products", "-"] if WHICH == "q" else
         ["top wait + barrier", "S^T dP^T products, own window scores", "barrier + prefetch issue", "exp, dS, images", "dK dV products"])  # relattn_fused_bwd_k_kernel runs last in a step and owns the buffer
for i, nm in enumerate(names):
    print(f"  phase {i} {nm:32s} median {np.median(per[:, i]):8.0f} clocks per key block   mean {per[:, i].mean():8.0f}")
print(f"  loop total per key block: median {np.median(live[:, 5] / live[:, 7]):.0f}; loop median {np.median(live[:, 5]):.0f}, epilogue median {np.median(live[:, 6]):.0f} max {live[:, 6].max()}")
