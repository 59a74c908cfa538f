"""Log-mel kernel against the oracle (max abs error in the log domain, where it sits) and its time at the bench shape:
    python tools/logmel_check.py [B] [N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import conformer_ref as R  # noqa: E402
from tensorflowasr_amd import kernels as K  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = R.conformer_config("S")
    melw = R.mel_weight_matrix()
    args = (torch.from_numpy(R.hann_periodic(400)).to(dev), torch.from_numpy(melw).to(dev), torch.from_numpy(R.mel_bands(melw)).to(dev))
    for n in (4321, 160000):
        rng = np.random.default_rng(n)
        sig = np.clip(rng.standard_normal((3, n)) * 0.1, -1, 1).astype(np.float32)
        ref = R.log_mel(sig, cfg)
        out = K.logmel(torch.from_numpy(sig).to(dev), *args, 160, 512, 0.97, 1e-6, torch.float32).cpu().numpy()
        err = np.abs(out - ref)
        i = np.unravel_index(np.argmax(err), err.shape)
        print("n=%d max abs err %.3e at %s (out %.5f ref %.5f); per-bin max %s" % (n, err.max(), i, out[i], ref[i], np.round(err.max((0, 1))[:8], 5)), flush=True)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 476160
    sig = (torch.randn(B, N) * 0.1).to(dev)
    for dt in (torch.bfloat16,):
        for _ in range(3):
            K.logmel(sig, *args, 160, 512, 0.97, 1e-6, dt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.logmel(sig, *args, 160, 512, 0.97, 1e-6, dt)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        gb = (B * N * 4 + B * (N // 160) * 80 * 2) / 1e9
        print("logmel %d x %d: %.1f us  (%.2f TB/s of %.1f MB)" % (B, N, us, gb / us * 1e6 / 1e3, gb * 1e3))


if __name__ == "__main__":
    main()
