"""Time the prediction network's recurrence per direction: persistent kernels (csrc/lstm_persist.hip) vs the step-kernel path.
    python tools/lstm_bench.py [B U1 P]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tensorflowasr_amd import kernels as K

B, U1, P = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 111, 640)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
xg = (torch.randn(B, U1, 4 * P, generator=g) * 0.7).to(bf).to(dev)
rk = (torch.randn(P, 4 * P, generator=g) / P ** 0.5).to(bf).to(dev)
lens = torch.full((B,), U1, dtype=torch.int32, device=dev)
dy = (torch.randn(B, U1, P, generator=g) * 0.5).to(bf).to(dev)
gates = torch.empty(B, U1, 4 * P, dtype=bf, device=dev)
cseq = torch.empty(B, U1, P, dtype=torch.float32, device=dev)
hseq = torch.empty(B, U1, P, dtype=bf, device=dev)
yseq = torch.empty(B, U1, P, dtype=bf, device=dev)
dz = torch.empty(B, U1, 4 * P, dtype=bf, device=dev)
hr = torch.empty(B, 4 * P, dtype=torch.float32, device=dev)
dhr = torch.empty(B, P, dtype=torch.float32, device=dev)
dhc = torch.zeros(B, P, dtype=torch.float32, device=dev)
dcc = torch.zeros(B, P, dtype=torch.float32, device=dev)
sync = K.lstm_persist_sync(dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


f = t(lambda: K.lstm_persist_fwd(xg, rk, None, None, lens, gates, cseq, hseq, yseq, sync))
b = t(lambda: K.lstm_persist_bwd(dy, rk, gates, cseq, lens, dz, dhc, dcc, sync))
print(f"persistent: B={B} U1={U1} P={P}: forward {f:.3f} ms ({f / U1 * 1e3:.2f} us/step), backward {b:.3f} ms ({b / U1 * 1e3:.2f} us/step), timeout flag {int(sync[1])}")
os.environ["TFASR_LSTM_PERSIST"] = os.environ.get("TFASR_LSTM_PERSIST", "1")
print("seq API (TFASR_LSTM_PERSIST=%s): forward %.3f ms, backward %.3f ms" % (
    os.environ["TFASR_LSTM_PERSIST"], t(lambda: K.lstm_seq_fwd(xg, rk, None, None, lens, gates, cseq, hseq, yseq, hr)),
    t(lambda: K.lstm_seq_bwd(dy, rk, gates, cseq, lens, dz, dhc, dcc, dhr))))
