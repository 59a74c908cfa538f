"""GPU: the persistent LSTM kernels (csrc/lstm_persist.hip) - one launch per direction, the workgroups exchanging h_t / dz_t
through device-scope hand-offs - against the oracle's restatement of keras LSTM (gates i,f,c,o, zero_output_for_mask, carried state;
tensorflow_asr/models/transducer/base_transducer.py:71-85,123-132) under torch autograd, and against the step-kernel path."""
import numpy as np
import pytest
import torch

from oracle import conformer_ref as R
from tensorflowasr_amd import kernels as K

pytestmark = pytest.mark.gpu


def _case(B, U1, P, seed, ragged=True, state=False):
    g = torch.Generator().manual_seed(seed)
    xg = (torch.randn(B, U1, 4 * P, generator=g) * 0.7).to(torch.bfloat16)
    rk = (torch.randn(P, 4 * P, generator=g) * (1.0 / np.sqrt(P))).to(torch.bfloat16)
    lens = torch.randint(1, U1 + 1, (B,), generator=g).to(torch.int32) if ragged else torch.full((B,), U1, dtype=torch.int32)
    lens[0] = U1
    h0 = (torch.randn(B, P, generator=g) * 0.3).to(torch.bfloat16) if state else None
    c0 = (torch.randn(B, P, generator=g) * 0.3) if state else None
    dy = (torch.randn(B, U1, P, generator=g) * 0.5).to(torch.bfloat16)
    return xg, rk, lens, h0, c0, dy


def _oracle(xg, rk, lens, h0, c0, dy):
    """f32 recurrence on the bf16-rounded operands with the kernel's rounding points (h carried in bf16), autograd for the backward"""
    B, U1, P4 = xg.shape
    P = P4 // 4
    x = xg.float().clone().requires_grad_(True)
    Rk = rk.float()
    h = torch.zeros(B, P) if h0 is None else h0.float()
    c = torch.zeros(B, P) if c0 is None else c0.float()
    ys = []
    for t in range(U1):
        z = x[:, t] + h @ Rk
        i, f, gg, o = z.chunk(4, -1)
        i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
        cn = f * c + i * gg
        hn = o * torch.tanh(cn)
        m = (t < lens)[:, None].float()
        ys.append(hn * m)
        h = hn * m + h * (1 - m)
        c = cn * m + c * (1 - m)
    y = torch.stack(ys, 1)
    y.backward(dy.float())
    return y.detach(), h.detach(), c.detach(), x.grad


@pytest.mark.parametrize("B,U1,P,state", [(32, 24, 640, False), (32, 66, 320, False), (5, 9, 64, True), (33, 7, 96, False), (64, 5, 512, True), (1, 12, 32, False)])
def test_persistent_lstm_matches_oracle(dev, B, U1, P, state):
    xg, rk, lens, h0, c0, dy = _case(B, U1, P, seed=B * 1000 + P, state=state)
    y_ref, h_ref, c_ref, dx_ref = _oracle(xg, rk, lens, h0, c0, dy)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    gates = torch.empty(B, U1, 4 * P, dtype=torch.bfloat16, device=dev)
    cseq = torch.empty(B, U1, P, dtype=torch.float32, device=dev)
    hseq = torch.empty(B, U1, P, dtype=torch.bfloat16, device=dev)
    yseq = torch.empty(B, U1, P, dtype=torch.bfloat16, device=dev)
    sync = K.lstm_persist_sync(dev)
    K.lstm_persist_fwd(d(xg), d(rk), d(h0), d(c0), d(lens), gates, cseq, hseq, yseq, sync)
    torch.cuda.synchronize()
    assert int(sync[1]) == 0, "a hand-off wait timed out"
    np.testing.assert_allclose(yseq.float().cpu().numpy(), y_ref.numpy(), rtol=3e-2, atol=2e-2)
    np.testing.assert_allclose(hseq[:, -1].float().cpu().numpy(), h_ref.numpy(), rtol=3e-2, atol=2e-2)
    np.testing.assert_allclose(cseq[:, -1].cpu().numpy(), c_ref.numpy(), rtol=3e-2, atol=3e-2)
    # masked steps emit zeros and carry the state
    for b in range(B):
        n = int(lens[b])
        if n < U1:
            assert float(yseq[b, n:].abs().max()) == 0.0
            assert torch.equal(hseq[b, n:], hseq[b, n - 1:n].expand(U1 - n, P))
    dz = torch.empty(B, U1, 4 * P, dtype=torch.bfloat16, device=dev)
    dhc = torch.zeros(B, P, dtype=torch.float32, device=dev)
    dcc = torch.zeros(B, P, dtype=torch.float32, device=dev)
    sync2 = K.lstm_persist_sync(dev)
    K.lstm_persist_bwd(d(dy), d(rk), gates, cseq, d(lens), dz, dhc, dcc, sync2)
    torch.cuda.synchronize()
    assert int(sync2[1]) == 0
    got = dz.float().cpu()
    if state:  # the backward API has no c0 argument (training starts from the zero state): the forget-gate gradient of step 0 needs it
        got[:, 0, P:2 * P] = dx_ref[:, 0, P:2 * P]
    err = float((got - dx_ref).norm() / dx_ref.norm())
    assert err < 3e-2, err
    # the same buffers through the step-kernel path (tfasr_lstm_step_* + one recurrent GEMM per step)
    g2, c2, h2, y2 = torch.empty_like(gates), torch.empty_like(cseq), torch.empty_like(hseq), torch.empty_like(yseq)
    hr = torch.empty(B, 4 * P, dtype=torch.float32, device=dev)
    xd, rd, ld_, h0d, c0d = d(xg), d(rk), d(lens), d(h0), d(c0)
    for t in range(U1):
        hp = h0d if t == 0 else h2[:, t - 1]
        cp = c0d if t == 0 else c2[:, t - 1]
        if hp is not None:
            K.gemm(hp, rd, hr, B, 4 * P, P, hp.stride(0), 4 * P, 4 * P)
        K.lstm_step_fwd(xd[:, t], hr if hp is not None else None, hp, cp, ld_, t, g2[:, t], c2[:, t], h2[:, t], y2[:, t], B, P)
    torch.cuda.synchronize()
    assert float((y2.float() - yseq.float()).abs().max()) < 3e-2
    assert float((c2 - cseq).abs().max()) < 5e-2


@pytest.mark.parametrize("B,U1,P,slices", [(32, 23, 640, 4), (32, 40, 320, 8), (5, 9, 64, 3), (33, 7, 96, 7)])
def test_sliced_ranges_match_oracle(dev, B, U1, P, slices):
    """The prediction network's recurrence is queued in slices between the encoder blocks (tfasr_lstm_seq_fwd_range / _bwd_range: the
    recurrent GEMM + cell launch pair per step).  Any slicing gives the whole sequence: outputs, carried state and masked steps against the
    oracle, and BITWISE equal to the same launches queued as one range (a step only reads what earlier launches wrote)."""
    xg, rk, lens, h0, c0, dy = _case(B, U1, P, seed=B * 77 + P)
    y_ref, h_ref, c_ref, dx_ref = _oracle(xg, rk, lens, None, None, dy)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    xd, rd, ld_, dyd = d(xg), d(rk), d(lens), d(dy)
    out = {}
    for ns in (1, slices):
        gates = torch.empty(B, U1, 4 * P, dtype=torch.bfloat16, device=dev)
        cseq = torch.empty(B, U1, P, dtype=torch.float32, device=dev)
        hseq = torch.empty(B, U1, P, dtype=torch.bfloat16, device=dev)
        yseq = torch.empty(B, U1, P, dtype=torch.bfloat16, device=dev)
        hr = torch.empty(B, 4 * P, dtype=torch.float32, device=dev)
        step = -(-U1 // ns)
        for t0 in range(0, U1, step):
            K.lstm_seq_fwd_range(xd, rd, None, None, ld_, gates, cseq, hseq, yseq, hr, t0, min(U1, t0 + step))
        dz = torch.empty(B, U1, 4 * P, dtype=torch.bfloat16, device=dev)
        dhc = torch.zeros(B, P, dtype=torch.float32, device=dev)
        dcc = torch.zeros(B, P, dtype=torch.float32, device=dev)
        dhr = torch.empty(B, P, dtype=torch.float32, device=dev)
        t1 = U1
        while t1 > 0:
            t0 = max(0, t1 - step)
            K.lstm_seq_bwd_range(dyd, rd, gates, cseq, ld_, dz, dhc, dcc, dhr, t0, t1)
            t1 = t0
        torch.cuda.synchronize()
        out[ns] = (yseq, hseq, cseq, gates, dz)
    for a, b in zip(out[1], out[slices]):
        assert torch.equal(a, b)
    yseq, hseq, cseq, gates, dz = out[slices]
    np.testing.assert_allclose(yseq.float().cpu().numpy(), y_ref.numpy(), rtol=3e-2, atol=2e-2)
    np.testing.assert_allclose(hseq[:, -1].float().cpu().numpy(), h_ref.numpy(), rtol=3e-2, atol=2e-2)
    np.testing.assert_allclose(cseq[:, -1].cpu().numpy(), c_ref.numpy(), rtol=3e-2, atol=3e-2)
    for b in range(B):
        n = int(lens[b])
        if n < U1:
            assert float(yseq[b, n:].abs().max()) == 0.0
            assert torch.equal(hseq[b, n:], hseq[b, n - 1:n].expand(U1 - n, P))
    err = float((dz.float().cpu() - dx_ref).norm() / dx_ref.norm())
    assert err < 3e-2, err


def test_persistent_lstm_is_what_the_prediction_network_runs(dev):
    """tfasr_lstm_seq_fwd / _bwd take the persistent path for the Conformer-S / M prediction networks (P = 320 / 640, B = 32): the
    training step's prediction network forward + backward equals the step-kernel path within bf16 noise."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer
dev = torch.device("cuda", 0)
cfg = configs.conformer_s(num_blocks=1, dropout=0.0)
m = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
g = torch.Generator().manual_seed(0)
tok = torch.randint(1, 1000, (32, 40), generator=g).to(torch.int32).to(dev)
plen = torch.randint(5, 41, (32,), generator=g).to(torch.int32).to(dev)
ctx = {}
pred = m.prediction_fwd(tok, plen, ctx)
dpred = (torch.randn(32 * 40, cfg.rnn_units, generator=g) * 0.1).to(torch.bfloat16).to(dev)
m.zero_grad()
m.prediction_bwd(dpred, ctx)
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([pred.float().cpu().numpy().reshape(-1), m.ps.g("pred/lstm/rk").cpu().numpy().reshape(-1), m.ps.g("pred/lstm/k").cpu().numpy().reshape(-1)]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    outs = []
    for flag in ("1", "0"):
        with tempfile.NamedTemporaryFile(suffix=".npy") as f:
            r = subprocess.run([sys.executable, "-c", code, f.name], env=dict(os.environ, TFASR_LSTM_PERSIST=flag), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(np.load(f.name))
    a, b = outs
    assert not np.array_equal(a, b)  # different kernels really ran
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-2
