#!/usr/bin/env python
"""bench.py — audio-hours/sec of one full Conformer-Transducer train step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = log-mel + SpecAugment + Conformer encoder + LSTM prediction net + joint + RNN-T loss + full backward +
(RCCL gradient all-reduce) + Adam, on one batch of synthetic 16 kHz utterances already resident in HBM.
Workload (BASELINE.json metric "audio-hours/sec (train step) Conformer-M RNN-T"): Conformer-M, 32 utterances per GPU
(weak scaling: global batch = 32 x N), LibriSpeech-shaped durations (lognormal, mean ~12.3 s, clipped to [1.3, 29.7] s,
~3.7 BPE tokens/s), padded to the batch maximum.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

# before the HIP runtime initialises: enough hardware queues for main + prediction-network + RCCL streams (tensorflowasr_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3


def flush_c_stdio():
    """RCCL prints a five-line version banner through C stdio when its first communicator is created; in a pipe or file that buffer
    is written at exit, i.e. AFTER Python's own output.  Flushing it early keeps the JSON line the last line of rank 0's stdout."""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def box_snapshot(index=0):
    """Clocks / power / temperature of this rank's GPU as the driver's library reports them (amdsmi; `rocm-smi --json` if that
    fails).  Taken while a calibration loop is still executing ("under_load") and at the idle ends of the run, so that a box which
    clocks lower, is power-capped lower or runs hotter than the one behind the committed figures can be told from a code regression."""
    out = {}
    try:
        import amdsmi

        try:
            amdsmi.amdsmi_init()
        except Exception:
            pass
        h = amdsmi.amdsmi_get_processor_handles()[index]
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            gfx = [int(v) for v in (m.get("current_gfxclks") or []) if isinstance(v, (int, float)) and 0 < v < 60000]
            if gfx:
                out["gfxclk_mhz_mean"], out["gfxclk_mhz_min"], out["gfxclk_mhz_max"] = round(float(np.mean(gfx)), 0), min(gfx), max(gfx)
            elif isinstance(m.get("current_gfxclk"), (int, float)):
                out["gfxclk_mhz_mean"] = m["current_gfxclk"]
            for k_out, k_in in (("uclk_mhz", "current_uclk"), ("socket_power_w", "current_socket_power"), ("avg_socket_power_w", "average_socket_power"),
                                ("temp_hotspot_c", "temperature_hotspot"), ("temp_mem_c", "temperature_mem"), ("throttle_status", "throttle_status"),
                                ("gfx_activity", "average_gfx_activity")):
                v = m.get(k_in)
                if isinstance(v, (int, float)) and v < 65535:
                    out[k_out] = v
        except Exception as e:
            out["metrics_error"] = repr(e)[:120]
        try:
            c = amdsmi.amdsmi_get_power_cap_info(h)
            cap = c.get("power_cap")
            if isinstance(cap, (int, float)):
                out["power_cap_w"] = round(cap / 1e6, 0) if cap > 1e5 else cap
        except Exception:
            pass
    except Exception as e:
        out["amdsmi_error"] = repr(e)[:120]
    if "gfxclk_mhz_mean" not in out:
        try:
            import subprocess

            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
            card = next(iter(json.loads(r.stdout).values()))
            out["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "power", "junction", "memory"))}
        except Exception as e:
            out["rocm_smi_error"] = repr(e)[:120]
    return out


CALIBRATION_FILE = os.path.join(ROOT, "profiles", "r06_calibration.json")
CAL_CELLS, CAL_J, CAL_V = 343040, 640, 1000  # the lattice of the first bench batch (343 k cells, rounded to 256 rows), Conformer-M joint
CAL_COPY_BYTES = 1 << 30


def calibrate(dev, index=0):
    """Two fixed probes whose only variable is the box: (1) the bf16 vocabulary product at the fixed 343 k-cell
    shape through the same library entry point as the step, alone on the chip - clock / power bound; (2) a 1 GiB device-to-device copy -
    HBM bound.  Each is compared with the figure committed in profiles/r06_calibration.json (measured with this code on the box the
    README's numbers come from): `x_vs_committed` > 1 = this box is that much SLOWER.  Clocks are read while probe 1 is executing."""
    from tensorflowasr_amd import kernels as K

    g = torch.Generator().manual_seed(1)
    h = (torch.randn(CAL_CELLS, CAL_J, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    W = (torch.randn(CAL_J, CAL_V, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    out = torch.empty(CAL_CELLS, CAL_V, dtype=torch.bfloat16, device=dev)
    src = torch.empty(CAL_COPY_BYTES, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(3):
        K.gemm(h, W, out, CAL_CELLS, CAL_V, CAL_J, CAL_J, CAL_V, CAL_V)
        dst.copy_(src)
    torch.cuda.synchronize()
    n1, n2 = 30, 10
    ev[0].record()
    for _ in range(n1):
        K.gemm(h, W, out, CAL_CELLS, CAL_V, CAL_J, CAL_J, CAL_V, CAL_V)
    ev[1].record()
    time.sleep(0.004)  # (the queue holds ~20 ms of launches: the snapshot below is taken with the product running)
    under_load = box_snapshot(index)
    torch.cuda.synchronize()
    ev[2].record()
    for _ in range(n2):
        dst.copy_(src)
    ev[3].record()
    torch.cuda.synchronize()
    gemm_ms = ev[0].elapsed_time(ev[1]) / n1
    copy_ms = ev[2].elapsed_time(ev[3]) / n2
    res = {"joint_gemm_ms": round(gemm_ms, 4), "joint_gemm_tflops": round(2.0 * CAL_CELLS * CAL_J * CAL_V / (gemm_ms * 1e-3) / 1e12, 1),
           "copy_1gib_ms": round(copy_ms, 4), "copy_GBps": round(2.0 * CAL_COPY_BYTES / (copy_ms * 1e-3) / 1e9, 1),
           "shape": f"[{CAL_CELLS}, {CAL_J}] x [{CAL_J}, {CAL_V}] bf16 -> bf16; copy {CAL_COPY_BYTES >> 20} MiB read + write", "under_load": under_load}
    try:
        ref = json.load(open(CALIBRATION_FILE))
        res["committed"] = {k: ref[k] for k in ("joint_gemm_ms", "copy_1gib_ms", "train_ms_per_step", "where") if k in ref}
        res["gemm_x_vs_committed"] = round(gemm_ms / ref["joint_gemm_ms"], 4)
        res["copy_x_vs_committed"] = round(copy_ms / ref["copy_1gib_ms"], 4)
    except Exception:
        res["committed"] = None
    del h, W, out, src, dst
    return res


def sub_line(argv, timeout_s=150):
    """One more bench line (another workload of BASELINE.json's metric) from a child process: its JSON line parsed, the bulky keys dropped."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-extras", "--regions", "1"] + argv
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                d = json.loads(line)
                for k in ("roofline_by_time", "joint_recompute_variant", "box", "higher_is_better", "vs_baseline", "data", "scaling", "n_gpus"):
                    d.pop(k, None)
                return d
        return {"value": None, "error": (r.stderr or r.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"exceeded {timeout_s} s"}
    except Exception as e:
        return {"value": None, "error": repr(e)[:200]}


def make_batch(cfg, B, seed, padding, size):
    """LibriSpeech-shaped synthetic batch: returns host tensors + total audio seconds."""
    rng = np.random.default_rng(seed)
    if size == "S-10s":
        dur = np.full(B, 10.0)
        ulen = rng.integers(32, 65, B)
        Umax = 64
    else:
        sigma = 0.55
        dur = np.clip(rng.lognormal(math.log(12.3) - sigma * sigma / 2, sigma, B), 1.3, 29.7)
        ulen = np.clip(np.round(dur * 3.7), 1, 230).astype(np.int64)
        Umax = 230 if padding == "reference" else int(ulen.max())
    nsamp = (dur * 16000).astype(np.int64)
    N = 475760 if padding == "reference" and size != "S-10s" else int(nsamp.max())
    sig = np.clip(rng.standard_normal((B, N)).astype(np.float32) * 0.1, -1, 1)
    for b in range(B):
        sig[b, nsamp[b]:] = 0.0
    labels = rng.integers(1, cfg.vocab_size, (B, Umax)).astype(np.int32)
    for b in range(B):
        labels[b, ulen[b]:] = 0
    preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)
    return dict(sig=sig, nsamp=nsamp.astype(np.int32), labels=labels, preds=preds, ulen=ulen.astype(np.int32),
                seconds=float(nsamp.sum()) / 16000.0)


def to_train_data(batch, dev):
    from tensorflowasr_amd.schemas import TrainData, TrainInput, TrainLabel

    return TrainData(
        TrainInput(torch.from_numpy(batch["sig"]).to(dev), torch.from_numpy(batch["nsamp"]),
                   torch.from_numpy(batch["preds"]).to(dev), torch.from_numpy(batch["ulen"] + 1).to(dev)),
        TrainLabel(torch.from_numpy(batch["labels"]).to(dev), torch.from_numpy(batch["ulen"])))


def step_matmul_flops(cfg, batch):
    """Multiply-accumulate work (x2) of one train step on the MATRIX cores for this batch: every Dense / attention / conv2 /
    joint product of the forward, times 3 for the two gradient products, with the joint on the packed (valid) lattice only.
    Depthwise conv, norms and the loss are not MFMA work and are not counted."""
    nsamp = np.asarray(batch["nsamp"], np.int64)
    B = len(nsamp)
    T0 = -(-int(nsamp.max()) // cfg.frame_step)
    T1, T2 = -(-T0 // 2), -(-(-(-T0 // 2)) // 2)
    F2 = -(-(-(-cfg.num_feature_bins // 2)) // 2)
    rows = B * T2
    d, C, H, dh, J, V, P, E = cfg.dmodel, cfg.filters, cfg.num_heads, cfg.head_size, cfg.joint_dim, cfg.vocab_size, cfg.rnn_units, cfg.embed_dim
    enc_block = rows * (2 * (2 * d * 4 * d) + d * 3 * H * dh + H * dh * d + d * 2 * d + d * d) + 2 * T2 * H * dh * d  # per block MACs
    attn = B * H * T2 * T2 * dh * 3  # content, position (skewed), P@V
    enc = cfg.num_blocks * (enc_block + attn) + rows * F2 * 9 * C * C + rows * F2 * C * d
    ul = np.asarray(batch["ulen"], np.int64)
    U1 = int(ul.max()) + 1
    tl = np.minimum(np.maximum(-(-(-(-(-(-nsamp // cfg.frame_step)) // 2)) // 2), ul), T2)
    cells = int((tl * (ul + 1)).sum())
    pred = B * U1 * (E * 4 * P + P * 4 * P)
    joint = rows * d * J + B * U1 * P * J + cells * J * V
    return 2.0 * 3.0 * (enc + pred + joint)


def step_useful_flops(cfg, batch):
    """step_matmul_flops without the padding: every encoder product counted over the REAL frames of each utterance only
    (rows = sum_b T'_b, attention = sum_b T'_b^2) - what an implementation free to skip padded frames would have to do.  The
    reference (and this path) DOES compute the padded frames: they enter the BatchNorm moments and are attended to as keys."""
    nsamp = np.asarray(batch["nsamp"], np.int64)
    B = len(nsamp)
    t2 = -(-(-(-(-(-nsamp // cfg.frame_step)) // 2)) // 2)
    F2 = -(-(-(-cfg.num_feature_bins // 2)) // 2)
    rows = int(t2.sum())
    d, C, H, dh, J, V, P, E = cfg.dmodel, cfg.filters, cfg.num_heads, cfg.head_size, cfg.joint_dim, cfg.vocab_size, cfg.rnn_units, cfg.embed_dim
    enc_block = rows * (2 * (2 * d * 4 * d) + d * 3 * H * dh + H * dh * d + d * 2 * d + d * d) + 2 * int(t2.max()) * H * dh * d
    attn = int((t2 * t2).sum()) * H * dh * 3
    enc = cfg.num_blocks * (enc_block + attn) + rows * F2 * 9 * C * C + rows * F2 * C * d
    ul = np.asarray(batch["ulen"], np.int64)
    tl = np.minimum(np.maximum(t2, ul), int(t2.max()))
    cells = int((tl * (ul + 1)).sum())
    pred = int((ul + 1).sum()) * (E * 4 * P + P * 4 * P)
    joint = rows * d * J + int((ul + 1).sum()) * P * J + cells * J * V
    return 2.0 * 3.0 * (enc + pred + joint)


def wgrad_group_shapes(cfg):
    d, H, dh = cfg.dmodel, cfg.num_heads, max(cfg.head_size, 64)
    f = cfg.ffm_scale * d
    return [(d, f), (f, d), (d, f), (f, d), (d, 3 * H * dh), (H * dh, d), (d, 2 * d), (d, d)]


def wgrad_group_roofline(cfg, rows_list, dev, in_step=None, iters=24):
    """`roofline_by_time`: the kernel family with the largest share of the step's kernel time (profiles/r0*_step*_kernel_stats: the
    grouped weight gradients of a Conformer block, one launch per block).  Two measurements:
    * `in_step`: HIP events recorded by the block executor around every grouped launch ON THE STREAM IT RUNS ON (its second stream, beside
      the next block's backward chain) during real train steps (tfasr_block_wgrad_probe) - the number that describes the step;
    * isolated: the same launch alone on the chip, at the MEAN row count of the batches the step cycles through, rotating over enough
      operand sets that their total exceeds the 256 MiB Infinity Cache (every launch streams its operands from HBM)."""
    from tensorflowasr_amd import kernels as K

    shapes = wgrad_group_shapes(cfg)
    rows = int(round(float(np.mean(rows_list)) / 64.0)) * 64
    per_set = sum(rows * (m + n) * 2 for m, n in shapes)
    nsets = max(3, -(-(288 << 20) // per_set))
    g = torch.Generator().manual_seed(0)
    sets = []
    for _ in range(nsets):
        xs = [(torch.randn(rows, m, generator=g) * 0.1).to(dev).to(torch.bfloat16) for m, n in shapes]
        dys = [(torch.randn(rows, n, generator=g) * 0.1).to(dev).to(torch.bfloat16) for m, n in shapes]
        sets.append((xs, dys))
    outs = [torch.zeros(m, n, device=dev) for m, n in shapes]
    bs = [torch.zeros(n, device=dev) for m, n in shapes]
    calls = [[dict(A=xs[i], B=dys[i], out=outs[i], M=shapes[i][0], N=shapes[i][1], K=rows, lda=shapes[i][0], ldb=shapes[i][1], ldd=shapes[i][1],
                   trans_a=True, accumulate=True, split_k=8, colsum=bs[i]) for i in range(len(shapes))] for xs, dys in sets]
    for k in range(nsets):
        K.gemm_group(calls[k])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(iters):
        K.gemm_group(calls[it % nsets])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = sum(2.0 * rows * m * n for m, n in shapes)
    by = sum((rows * (m + n)) * 2.0 + m * n * 4.0 for m, n in shapes)
    ach = fl / (ms * 1e-3) / 1e12
    out = {"kernel": "wgrad_group_kernel (the 8 Dense-layer weight gradients of one Conformer block, one grouped launch)", "bound": "mfma",
           "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
           "ms_per_launch": round(ms, 4), "rows": rows, "algorithmic_bytes": by, "hbm_frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           "operand_sets": nsets, "operand_bytes_total": nsets * per_set,
           "how": f"isolated launches at the mean rows of the step's batches, rotating over {nsets} operand sets ({nsets * per_set >> 20} MiB > the 256 MiB "
                  f"Infinity Cache); `in_step` = HIP events of the block executor around the same launch inside real train steps"}
    if in_step is not None:
        ms_in, n_in, fl_in = in_step  # total ms, launches, flop of all of them
        if n_in > 0 and ms_in > 0:
            a2 = fl_in / (ms_in * 1e-3) / 1e12
            out["in_step"] = {"ms_per_launch": round(ms_in / n_in, 4), "launches": n_in, "achieved": round(a2, 1), "frac": round(a2 / MFMA_BF16_PEAK_TFLOPS, 4),
                              "note": "the launch runs on the executor's second stream beside the next block's backward chain (one GPU) / in line (data-parallel "
                                      "rank): its duration there, not the step time it costs"}
            out["in_step_frac"] = out["in_step"]["frac"]
    return out


def dp_route_line(argv_model, steps=10, warmup=3, timeout_s=150):
    """`dp_route_ms`: the data-parallel code path (bucketed gradient all-reduce hooks, sync-BN reductions, per-rank seeds) through a ONE-rank
    RCCL group on this GPU, in a child process (a process group cannot be added to this one after the fact): everything a rank of a
    multi-GPU job does except the wire time, so the cost of the route is visible in every default run."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--dp-hooks", "--no-cpu-baseline", "--no-extras", "--steps", str(steps), "--warmup", str(warmup)] + argv_model
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                return float(json.loads(line)["ms_per_step"])
    except Exception:
        return None
    return None


def pmc_traffic(flops_per_launch, J, V):
    """HBM bytes per launch of the joint vocabulary GEMM from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    profiles/r01_pmc_traffic.json, collected on this same command).  PMC counters cannot be read inside a timed run, so
    the figure is looked up: the forward vocabulary projection is the plain NN gemm_fast launch whose WRITE_SIZE equals its
    output (cells x V bf16) - no other launch of the step writes that much from that kernel.  None if it was not profiled."""
    here = os.path.dirname(os.path.abspath(__file__))
    path = next((q for q in (os.path.join(here, "profiles", f) for f in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")) if os.path.exists(q)), None)
    if path is None:
        return None
    rows = json.load(open(path))
    vals = []
    for fl in flops_per_launch:
        cells = fl / (2.0 * J * V)
        out_mb = cells * (V * 2 + (-(-V // 128) * 2) * 8 + 8) / 1e6  # bf16 logits + the epilogue's row statistics (lse_part, pick)
        m = [r["hbm_MB"] for r in rows if r["kernel"].startswith(("gemm_big_kernel<false, 256, 64,", "gemm_fast_kernel<false, false, 128, 64>", "gemm_fast_kernel<false, false, 128, 0>")) and abs(r["write_MB"] - out_mb) < 0.03 * out_mb]
        if m:
            vals.append(float(np.mean(m)) * 1e6)
    return round(float(np.mean(vals)), 0) if vals else None


def cpu_baseline_worker(size, vocab, batch=32, micro=4):
    """Reference-path stand-in: the oracle's torch-CPU restatement of the same train step (TensorFlow is not installable:
    BASELINE.md section 2), timed on this box's host cores over the FULL first batch of the GPU line (same generator and seed, every
    utterance padded to the batch maximum like the GPU step pads it).  Host memory bounds the dense f32 lattice the restatement
    materialises ([B, T', U+1, V] and its gradient: ~44 GB at B = 32), so the 32 utterances go through as micro-batches of `micro`
    whose gradients accumulate into one optimizer step - the reference's own train_step_ga route (base_model.py:200-209)."""
    from oracle import conformer_ref as R
    from oracle import rnnt_ref

    cores = min(os.cpu_count() or 1, 16)  # more threads than this only add contention for these op sizes
    torch.set_num_threads(cores)
    ocfg = R.conformer_config("M" if size.startswith("M") else "S", vocab)
    W = R.init_weights(ocfg, seed=3)
    Wg = {k: v.clone().requires_grad_(R.is_trainable(k)) for k, v in W.items()}
    from tensorflowasr_amd import configs as _cfgs

    pcfg = _cfgs.conformer_m(vocab) if size.startswith("M") else _cfgs.conformer_s(vocab)
    full = make_batch(pcfg, batch, seed=10, padding="batch", size="S-10s" if not size.startswith("M") else "LibriSpeech-shaped")
    N = int(full["nsamp"].max())
    secs_total = float(full["nsamp"].sum()) / 16000.0
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in Wg.items() if v.requires_grad}
    groups = [list(range(j, min(j + micro, batch))) for j in range(0, batch, micro)]

    def micro_step(idx):
        nsamp = full["nsamp"][idx]
        sig = full["sig"][idx][:, :N]
        ulen = full["ulen"][idx]
        U = int(ulen.max())
        labels = full["labels"][idx][:, :U].copy()
        preds = np.concatenate([np.zeros((len(idx), 1), np.int32), labels], 1)
        feat = R.log_mel(sig, ocfg)
        flen = R.get_nframes(nsamp)
        logits, elen = R.transducer_forward(torch.from_numpy(feat), flen, torch.from_numpy(preds), torch.from_numpy(ulen.astype(np.int64) + 1), Wg, ocfg, training=True)
        tl, ul = rnnt_ref.clamp_lengths(elen.numpy(), ulen)
        loss, g = rnnt_ref.rnnt_loss_and_grad(logits.detach().numpy(), labels, ul, np.minimum(tl, logits.shape[1]), np.float32)
        logits.backward(torch.from_numpy(g / batch))
        return float(loss.sum())

    def apply(i):
        with torch.no_grad():
            for k, v in Wg.items():
                if v.grad is None:
                    continue
                gk = v.grad + (2e-6 * v if R.is_regularized(k) else 0)
                p, m, vv = R.adam_step(v, gk, state[k][0], state[k][1], i + 1, 1e-4)
                v.copy_(p)
                state[k] = (m, vv)
                v.grad = None

    micro_step(groups[0])  # warm-up: one micro-batch, its gradient discarded
    for v in Wg.values():
        v.grad = None
    # bounded sample: micro-batches of the step in order until >= 2 are done and ~20 s are spent (all of them if the box is fast enough),
    # then ONE optimizer update; the update's time enters in proportion to the share of the step that was timed
    t0 = time.perf_counter()
    k = 0
    secs_timed = 0.0
    for idx in groups:
        micro_step(idx)
        k += 1
        secs_timed += float(full["nsamp"][idx].sum()) / 16000.0
        if k >= 2 and time.perf_counter() - t0 > 20.0:
            break
    t_micro = time.perf_counter() - t0
    t1 = time.perf_counter()
    apply(0)
    t_adam = time.perf_counter() - t1
    dt = t_micro + t_adam * k / len(groups)
    return dict(value=(secs_timed / 3600.0) / dt, unit="audio-hours/sec", cores=cores, kind="port", batch=batch, micro_batch=micro,
                micro_batches_timed=k, micro_batches_per_step=len(groups), s_per_micro_batch=round(t_micro / k, 3), s_adam=round(t_adam, 3),
                s_per_step_extrapolated=round(t_micro / k * len(groups) + t_adam, 2),
                sample=f"oracle (torch-CPU fp32 restatement of tensorflow_asr; TF unavailable) full train step, Conformer-{ocfg['dmodel']}d, on the {batch} "
                       f"utterances of the GPU line's first batch (same generator and seed: {secs_total:.1f} s of audio, every utterance padded to "
                       f"{N / 16000.0:.1f} s like the GPU step pads it), run as {len(groups)} accumulated micro-batches of {micro} + one Adam update (host memory bounds "
                       f"the dense f32 lattice; BatchNorm moments per micro-batch); bounded sample: the first {k} of the {len(groups)} micro-batches timed after a "
                       f"one-micro-batch warm-up ({secs_timed:.1f} s of audio in {t_micro:.1f} s) + the update's share; {cores} threads")


def cpu_baseline(size, vocab, batch=32, timeout_s=300):
    """Run the CPU baseline in a child process with a hard wall-clock bound so the default bench always finishes."""
    import subprocess

    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline_worker(%r, %d, %d)))" % (ROOT, size, vocab, batch))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "audio-hours/sec", "error": (r.stderr or r.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "audio-hours/sec", "error": f"CPU baseline exceeded {timeout_s} s"}


def bench_decode(args, model, cfg, dev):
    """Greedy transducer search (Transducer.recognize, base_transducer.py:474-575) RTF = wall time / audio duration on
    32 x 10 s synthetic utterances.  Headline = the token-exact mode (tests/test_parity_baseline_gpu.py: a bf16-trained model
    decodes on its f32 master weights with the exact-f32 kernels, tokens bit-equal to the f32 oracle); the bf16-encoder time is
    an extra key.  Random-init weights: the blank logit's bias is calibrated (bisection, untimed) so that the search emits a
    speech-like ~3.7 tokens per second instead of nothing or a saturated token buffer."""
    from tensorflowasr_amd.schemas import PredictInput

    rng = np.random.default_rng(0)
    B, secs = args.batch, 10.0
    sig = torch.from_numpy(np.clip(rng.standard_normal((B, int(secs * 16000))).astype(np.float32) * 0.1, -1, 1)).to(dev)
    lens = torch.full((B,), int(secs * 16000), dtype=torch.int32)
    inp = PredictInput(sig, lens)
    target = 3.7 * secs * B
    b0 = float(model.ps.p("joint/vocab/b")[0].item())
    # Calibration (untimed): the blank logit's bias is raised until EVERY row terminates by itself - no row saturates its token buffer, so
    # the loop ends after max_b(frames_b + tokens_b) iterations like a speech-like search does, not at this build's iteration cap (round 4's
    # line had one row of the random-weight model that never emitted blank: 753 iterations for 1 053 tokens) - and among
    # those biases the one whose token count is closest to ~3.7 tokens per second of audio.
    def probe(bias):
        model.ps.p("joint/vocab/b")[0] = b0 + bias
        model.ps.refresh_shadow()
        tok = model.recognize(inp).tokens
        per_row = (tok != 0).sum(1)
        return int(per_row.sum().item()), bool((per_row >= tok.shape[1] - 2).any().item())

    lo, hi = 0.0, 16.0  # lo: saturating (or too many tokens); hi: terminating
    best = None
    for _ in range(14):
        bias = 0.5 * (lo + hi)
        ntok, sat = probe(bias)
        if sat or ntok > 2.0 * target:
            lo = bias
        else:
            if best is None or abs(ntok - target) < abs(best[1] - target):
                best = (bias, ntok)
            if ntok >= 0.5 * target:
                break
            hi = bias
    bias, ntok = best if best is not None else (hi, -1)
    ntok, sat = probe(bias)
    res, toks = {}, {}
    for prec in ("f32", "bf16") if args.dtype == "bf16" else ("f32",):
        for _ in range(args.warmup):
            out = model.recognize(inp, precision=prec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model.recognize(inp, precision=prec)
        torch.cuda.synchronize()
        res[prec] = ((time.perf_counter() - t0) / args.steps, int((out.tokens != 0).sum().item()), int((out.tokens != 0).sum(1).max().item()))
        toks[prec] = out.tokens.clone()
    # breakdown of the token-exact mode: front end + encoder vs the greedy search, HIP events around each half
    twin = model.inference_twin() if model.dtype != torch.float32 else model
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    enc_ms = srch_ms = 0.0
    nrep = max(3, min(args.steps, 10))
    for _ in range(nrep):
        ev[0].record()
        enc, elen = twin.encode(inp.inputs, inp.inputs_length)
        ev[1].record()
        twin.recognize_encoded(enc, elen)
        ev[2].record()
        torch.cuda.synchronize()
        enc_ms += ev[0].elapsed_time(ev[1]) / nrep
        srch_ms += ev[1].elapsed_time(ev[2]) / nrep
    T_enc = int(enc.shape[1])
    # encoder flop of the f32 twin (same formula as the train step's forward share) against the exact-f32 MFMA peak
    d, C, H, dh, F2 = cfg.dmodel, cfg.filters, cfg.num_heads, cfg.head_size, -(-(-(-cfg.num_feature_bins // 2)) // 2)
    rows = B * T_enc
    enc_flop = 2.0 * (cfg.num_blocks * (rows * (2 * (2 * d * 4 * d) + d * 3 * H * dh + H * dh * d + d * 2 * d + d * d) + 2 * T_enc * H * dh * d
                                        + B * H * T_enc * T_enc * dh * 3) + rows * F2 * 9 * C * C + rows * F2 * C * d)
    dt, ntok, max_row_tokens = res["f32"]
    line = {"metric": "greedy-decode RTF Conformer-%s RNN-T" % args.model, "value": round(dt / (B * secs), 6), "unit": "RTF (wall s / audio s)",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": False,
            "dtype": "f32", "data": "synthetic", "vs_baseline": None,
            "config": {"workload": f"greedy search (recognize_batch) over {B} x {secs:.0f} s utterances incl. log-mel + encoder; token-exact mode "
                                   f"(f32 master weights, exact-f32 MFMA encoder; search arithmetic f32), blank bias +{bias:.3f} (calibrated)",
                       "tokens_emitted": ntok, "global_batch": B, "training_storage": args.dtype}}
    line["breakdown"] = {"frontend_encoder_ms": round(enc_ms, 3), "search_ms": round(srch_ms, 3),
                         "encoder_f32_tflops": round(enc_flop / (enc_ms * 1e-3) / 1e12, 1), "encoder_f32_mfma_frac": round(enc_flop / (enc_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                         "search_us_per_token": round(srch_ms * 1e3 / max(ntok, 1), 2),
                         "search_iterations": int(getattr(twin, "last_search_iterations", 0)),
                         "search_us_per_iteration": round(srch_ms * 1e3 / max(int(getattr(twin, "last_search_iterations", 0)), 1), 2),
                         "note": "HIP events around model.encode (log-mel + f32 encoder) and recognize_encoded (greedy search); fraction of the exact-f32 MFMA peak; "
                                 "search_iterations = iterations of the batch loop queued (one frame or one token per row each; incl. the no-op tail of the "
                                 "last queued batch); the blank bias is calibrated so that every row terminates by itself (`row_saturated` false)"}
    line["breakdown"]["row_saturated"] = bool(sat)
    # what the batch loop NEEDS: the slowest row advances one frame per blank and one token per non-blank
    useful = int(T_enc + max_row_tokens)
    line["breakdown"]["useful_iterations"] = useful
    line["breakdown"]["search_us_per_useful_iteration"] = round(srch_ms * 1e3 / max(useful, 1), 2)
    if "bf16" in res:
        # second reported mode: the bf16 training kernels for the encoder, the search arithmetic f32 as in the exact mode;
        # agreement with the token-exact mode = utterances whose whole token row is identical / token positions that are identical
        ta, tb = toks["f32"], toks["bf16"]
        Lc = min(ta.shape[1], tb.shape[1])
        same_pos = (ta[:, :Lc] == tb[:, :Lc])
        line["bf16_encoder"] = {"value": round(res["bf16"][0] / (B * secs), 6), "ms_per_step": round(res["bf16"][0] * 1e3, 3),
                                "tokens_emitted": res["bf16"][1],
                                "utterances_identical_to_exact_mode": int(same_pos.all(1).sum().item()), "utterances": int(B),
                                "token_positions_identical_frac": round(float(same_pos.float().mean().item()), 4),
                                "note": "training kernels (bf16 storage) for the encoder, f32 search; NOT token-exact vs the f32 reference: with these "
                                        "random-init weights near-ties of the argmax flip (a trained model's margins are wider)"}
    print(json.dumps(line))


def bench_ctc_decode(args, dev, dtype):
    """Conformer-CTC (examples/models/ctc/conformer/small.yml.j2 dimensions) inference RTF: CtcModel.recognize (greedy) and
    recognize_beam (tf.nn.ctc_beam_search_decoder semantics, host routine like the reference's) over 32 x 10 s utterances,
    log-mel + encoder included.  BASELINE configs[4] names Jasper + fp16; the reference's Jasper streaming path is broken
     and the MI355X path stores bf16, so the CTC family is benchmarked on its Conformer model."""
    from tensorflowasr_amd import configs
    from tensorflowasr_amd.ctc_model import ConformerCTC
    from tensorflowasr_amd.schemas import PredictInput

    cfg = configs.conformer_ctc_s()
    model = ConformerCTC(cfg, dev, dtype=dtype, seed=0)
    rng = np.random.default_rng(0)
    B, secs = args.batch, 10.0
    sig = torch.from_numpy(np.clip(rng.standard_normal((B, int(secs * 16000))).astype(np.float32) * 0.1, -1, 1)).to(dev)
    inp = PredictInput(sig, torch.full((B,), int(secs * 16000), dtype=torch.int32))
    res = {}
    for name, fn in (("greedy", lambda: model.recognize(inp)), ("beam10", lambda: model.recognize_beam(inp, beam_width=10))):
        for _ in range(args.warmup):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / args.steps
    dt = res["greedy"]
    print(json.dumps({"metric": "CTC greedy-decode RTF Conformer-CTC(S)", "value": round(dt / (B * secs), 6), "unit": "RTF (wall s / audio s)",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": False,
                      "dtype": args.dtype, "data": "synthetic", "vs_baseline": None,
                      "config": {"workload": f"CtcModel.recognize over {B} x {secs:.0f} s utterances incl. log-mel + encoder (random weights)",
                                 "global_batch": B, "beam_search_width10_rtf": round(res["beam10"] / (B * secs), 6),
                                 "beam_search_ms": round(res["beam10"] * 1e3, 2)}}))


def relaunch_cmd(gpus, argv, env):
    """`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself under torch.distributed.run, one rank
    per GPU over RCCL (the reference's MirroredStrategy spans every visible GPU from one process: utils/env_util.py:57-70).
    Returns the command line, or None when this process is already a rank (WORLD_SIZE set by a launcher) or N == 1."""
    if gpus <= 1 or env.get("WORLD_SIZE") is not None:
        return None
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


class _StubModel:
    """TFASR_BENCH_STUB=1 (tests/test_bench_launch.py): the launch / rendezvous / timing / reporting logic of this file on CPU
    with the gloo backend, the train step replaced by one small all-reduce.  Never a measurement."""

    class _PS:
        def num_trainable(self):
            return 0

    def __init__(self, dp):
        self.dp, self.ps, self.timers, self.timer_work, self.time_sections = dp, self._PS(), {}, {}, False
        self.buf = torch.ones(1024)

    def train_step(self, data):
        if self.dp:
            self.dp.allreduce_stats_(self.buf.clone())
        return {"loss": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="M", choices=["M", "S", "S-streaming", "contextnet"],
                    help="M / S = Conformer sizes; S-streaming = small-streaming.yml.j2 (chunk 16 / history 64, LayerNorm depthwise and subsampling norms); contextnet = BASELINE configs[3] family")
    ap.add_argument("--alpha", type=float, default=2.0, help="ContextNet width multiplier (0.5 small, 1 medium, 2 large)")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--padding", default="batch", choices=["batch", "reference"])
    ap.add_argument("--workload", default=None, help="'S-10s' = BASELINE cfg2 (10 s utterances); default LibriSpeech-shaped")
    ap.add_argument("--mode", default="train", choices=["train", "decode", "ctc-decode"],
                    help="decode = transducer greedy-search RTF (second half of BASELINE.json's metric); ctc-decode = Conformer-CTC greedy / beam RTF")
    ap.add_argument("--regions", type=int, default=3,
                    help="timed regions of --steps steps each, every one bracketed by barrier + synchronize; the line reports the MEDIAN region "
                         "(ms_per_step / value) plus every region, the minimum and the spread")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the reference-padding second measurement of the default run")
    ap.add_argument("--no-specaugment", action="store_true")
    ap.add_argument("--dp-hooks", action="store_true",
                    help="with --gpus 1: run the DATA-PARALLEL code path (split block phases around the sync-BN all-reduces, bucketed gradient "
                         "all-reduce) through a one-rank RCCL group - what that path costs by itself, wire time excluded")
    ap.add_argument("--dropout", type=float, default=None, help="override encoder dropout (default: reference value 0.1)")
    args = ap.parse_args()
    stub = os.environ.get("TFASR_BENCH_STUB") == "1"

    cmd = relaunch_cmd(args.gpus, sys.argv[1:], os.environ)
    if cmd is not None:
        import subprocess

        if not stub and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.stdout.flush()
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    from tensorflowasr_amd import configs, dp as dpmod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # never print a line whose n_gpus is not what was asked for
        raise SystemExit(f"--gpus {args.gpus} but this process group has WORLD_SIZE={world}: refusing to report a {world}-GPU run as {args.gpus}")
    dp = dpmod.init_from_env(backend="gloo" if stub else None) if world > 1 else None
    if args.dp_hooks and world == 1 and not stub:
        import socket

        import torch.distributed as dist

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        sock.close()
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dp = dpmod.DataParallel()
    if dp and dp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {dp.world} ranks")
    rank = dp.rank if dp else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if stub:
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a, **k: None
    else:
        from tensorflowasr_amd.conformer import ConformerTransducer

        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    if args.model == "contextnet":
        cfg = configs.contextnet(alpha=args.alpha)
    else:
        cfg = configs.conformer_m() if args.model == "M" else configs.conformer_s()
        if args.model == "S-streaming":  # examples/models/transducer/conformer/small-streaming.yml.j2:26,33,38-39
            cfg = configs.conformer_s(chunk_size=16, history_size=64, convm_dw_norm="layer", sub_norm="layer")
    if args.no_specaugment:
        cfg.time_masking, cfg.freq_masking = {}, {}
    if args.dropout is not None:
        cfg.dropout = args.dropout
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.mode == "ctc-decode":
        return bench_ctc_decode(args, dev, dtype)
    if stub:
        model = _StubModel(dp)
    elif args.model == "contextnet":
        from tensorflowasr_amd.contextnet import ContextNetTransducer

        model = ContextNetTransducer(cfg, dev, dtype=dtype, seed=0, dp=dp)
    else:
        model = ConformerTransducer(cfg, dev, dtype=dtype, seed=0, dp=dp)
    if dp and not stub:
        dp.attach(model.ps.grad)
    if args.mode == "decode":
        return bench_decode(args, model, cfg, dev)
    size = args.workload or ("S-10s" if args.model.startswith("S") else "LibriSpeech-shaped")
    # a few distinct batches per rank, resident in HBM before the timed region
    nb = 2
    # weak scaling with the per-GPU work EXACTLY fixed: every rank runs the same synthetic shard shapes (same seeds), so padded
    # lengths agree across ranks - which the synchronised BatchNorm's count (rows x world) assumes, as in the reference where
    # the global batch is padded as one (datasets.py:102-138,342-365) - and no rank waits for a longer batch on another.
    batches = [make_batch(cfg, 2 if stub else args.batch, seed=10 + 13 * i, padding=args.padding, size="S-10s" if stub else size) for i in range(nb)]
    data = [to_train_data(b, dev) for b in batches]
    model.timers = {}
    model.time_sections = bool(os.environ.get("TFASR_BENCH_SECTIONS"))
    if not stub:
        torch.cuda.synchronize()
        model.prefetched_inputs = True  # the batches above are complete in HBM before any step is queued (inputs resident: the bench contract)

    def one_step(i):
        return model.train_step(data[i % nb])

    box = None
    if not stub and rank == 0 and world == 1 and not args.no_extras:
        box = {"before": box_snapshot(local_rank)}
    for i in range(args.warmup):
        one_step(i)
    flush_c_stdio()  # RCCL's start-up banner (C stdio) leaves every rank's buffer now, long before rank 0's JSON line
    model.timers, model.timer_work = {}, {}
    n_launch0 = 0
    if not stub:
        from tensorflowasr_amd import kernels as _K

    # `--regions` timed regions of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and reduced with MAX over
    # the ranks; the reported step time is the MEDIAN region's (one region = the contract's single timed region: --regions 1)
    region_ms, region_host_ms, region_secs = [], [], []
    launches_per_step = None
    step_no = 0
    dp_acc = getattr(dp, "accounting", None) if dp else None
    if dp_acc is not None:
        dp_acc.enable()
    rank_ms = []
    for r in range(max(1, args.regions)):
        torch.cuda.synchronize()
        if dp:
            dp.barrier()
        if not stub:
            n_launch0 = _K.launch_count()
        t0 = time.perf_counter()
        secs_local = 0.0
        for i in range(args.steps):
            one_step(step_no)
            secs_local += batches[step_no % nb]["seconds"]
            step_no += 1
        t_host = time.perf_counter() - t0  # host-side enqueue time (the GPU may still be running)
        if not stub:
            launches_per_step = (_K.launch_count() - n_launch0) / float(args.steps)
        torch.cuda.synchronize()
        if dp:
            dp.barrier()
        dt_r = time.perf_counter() - t0
        if dp:
            rank_ms.append((dp.max_scalar(dt_r, dev) / args.steps * 1e3, dp.min_scalar(dt_r, dev) / args.steps * 1e3))
            dt_r = dp.max_scalar(dt_r, dev)
            secs_r = dp.mean_scalar(torch.tensor([secs_local], dtype=torch.float64, device=dev)).item() * world
        else:
            secs_r = secs_local
        region_ms.append(dt_r / args.steps * 1e3)
        region_host_ms.append(t_host / args.steps * 1e3)
        region_secs.append(secs_r)
    order = sorted(range(len(region_ms)), key=lambda k: region_ms[k])
    med = order[(len(order) - 1) // 2]  # the median region (the lower one of an even count)
    ms_per_step = region_ms[med]
    dt = ms_per_step * 1e-3 * args.steps
    secs_total = region_secs[med]
    value = (secs_total / 3600.0) / dt
    dp_report = None
    if dp_acc is not None:
        # where this rank's compute stream waited for the wire (dp.Accounting), rank 0's figures and the worst rank's
        acc = dp_acc.summary(args.steps * len(region_ms))
        dp_acc.enable(False)
        dp_report = {"rank0": acc, "max_over_ranks": {k: round(dp.max_scalar(v, dev), 4) for k, v in acc.items() if k.endswith("_ms")},
                     "rank_step_ms_max": round(rank_ms[med][0], 3), "rank_step_ms_min": round(rank_ms[med][1], 3),
                     "grad_wire": dp.grad_wire, "stats_communicator": "own" if dp.stats_group is not dp.group else "shared",
                     "note": "syncbn_wait = compute-stream time between queuing a sync-BN statistics all-reduce and holding its result, "
                             "grad_allreduce_exposed = compute-stream time in finish_grads (gradient all-reduce not hidden under backward); HIP events, per step"}
    if rank == 0 and os.environ.get("TFASR_BENCH_HOST"):
        sys.stderr.write(f"[host] enqueue {region_host_ms[med]:.2f} ms/step of {ms_per_step:.2f} ms/step\n")

    if rank == 0 and os.environ.get("TFASR_BENCH_SECTIONS"):
        torch.cuda.synchronize()
        for name, tm in sorted(model.timers.items()):
            sys.stderr.write(f"[section] {name:18s} {sum(a.elapsed_time(b) for a, b in tm) / args.steps:8.3f} ms/step over {len(tm) // args.steps} calls\n")
    if rank == 0:
        roof = None
        tm = model.timers.get("joint_vocab_gemm") or []
        cn = model.timers.get("cn_dwconv_fwd") or []
        if args.model == "contextnet" and cn:
            # ContextNet is depthwise-conv / BatchNorm heavy: report the HBM roofline of its depthwise convolution (all launches of
            # the timed region: achieved = algorithmic bytes (read x + write y) / measured time)
            torch.cuda.synchronize()
            ms_tot = float(np.sum([a.elapsed_time(b) for a, b in cn]))
            by = float(np.sum(model.timer_work["cn_dwconv_fwd"]))
            ach = by / (ms_tot * 1e-3) / 1e9
            roof = {"kernel": "dwconv_tile_kernel (causal depthwise Conv1D k=5, forward, all layers)", "bound": "hbm", "achieved": round(ach, 1),
                    "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                    "ms_per_launch": round(ms_tot / len(cn), 4)}
        elif tm:
            torch.cuda.synchronize()
            ms = float(np.mean([a.elapsed_time(b) for a, b in tm]))
            fl = float(np.mean(model.timer_work["joint_vocab_gemm"]))
            peak = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
            ach = fl / (ms * 1e-3) / 1e12
            roof = {"kernel": "gemm_big_kernel<false, 256, E_LSE> (joint vocabulary projection + log-softmax statistics, fwd)", "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": pmc_traffic(model.timer_work["joint_vocab_gemm"], cfg.joint_dim, cfg.vocab_size),
                    "traffic_source": "looked up from the committed rocprofv3 PMC passes of this command (profiles/r0*_pmc_traffic.json), not measured in this run",
                    "ms_per_launch": round(ms, 4)}
            # whole-step view (north_star asks for the step's fraction of the MFMA roofline as well): matrix-core flop of
            # the step / step time; mean over the batches the timed region cycles through
            if args.model != "contextnet":
                sf = float(np.mean([step_matmul_flops(cfg, batches[(med * args.steps + i) % nb]) for i in range(args.steps)]))
                roof["step_matmul_tflops"] = round(sf / (ms_per_step * 1e-3) / 1e12, 1)
                roof["step_frac"] = round(sf / (ms_per_step * 1e-3) / 1e12 / peak, 4)
                uf = float(np.mean([step_useful_flops(cfg, batches[(med * args.steps + i) % nb]) for i in range(args.steps)]))
                roof["useful_step_frac"] = round(uf / (ms_per_step * 1e-3) / 1e12 / peak, 4)  # padded encoder frames not counted
        # RNN-T loss kernels (statistics finalize + alpha/beta lattice + gradient) against the HBM roofline: algorithmic bytes =
        # cells x V x (logit + gradient), time = HIP events around exactly those launches in the timed region
        roof_rnnt = None
        tl_ = model.timers.get("rnnt_loss") or []
        if tl_ and args.model != "contextnet":
            ms = float(np.mean([a.elapsed_time(b) for a, b in tl_]))
            by = float(np.mean(model.timer_work["rnnt_loss"]))
            if by > 0:
                ach = by / (ms * 1e-3) / 1e9
                roof_rnnt = {"kernel": "rnnt_stats_finalize + rnnt_lattice + rnnt_grad (packed lattice)", "bound": "hbm", "achieved": round(ach, 1),
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "ms_per_launch": round(ms, 4),
                             "algorithmic_bytes": by}
        out = {
            "metric": ("audio-hours/sec (train step) ContextNet(alpha=%g) RNN-T" % args.alpha) if args.model == "contextnet"
                      else "audio-hours/sec (train step) Conformer-%s RNN-T" % args.model,
            "value": round(value, 4), "unit": "audio-hours/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "STUB (launch-logic test, not a measurement)" if stub else f"{'ContextNet' if args.model == 'contextnet' else 'Conformer-' + args.model} transducer full train step (fwd+RNN-T loss+bwd+Adam), {size} 16 kHz utterances, "
                                   f"{args.batch}/GPU, padding={args.padding}, SpecAugment {'off' if args.no_specaugment else 'on'}, dropout {cfg.dropout}",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}" + (" (data-parallel hooks on, one-rank RCCL group)" if args.dp_hooks and world == 1 else ""), "params": model.ps.num_trainable()},
            "roofline": roof,
            "roofline_rnnt": roof_rnnt,
            # kernels this library queued per step (host-side count, tfasr_launch_count; torch's own fills / copies are not in it)
            "launches_per_step": None if launches_per_step is None else round(launches_per_step, 1),
            "dp": dp_report,
            # every timed region (each --steps steps between barrier + synchronize pairs); ms_per_step / value are the median region's
            "regions": {"ms_per_step": [round(v, 3) for v in region_ms], "median": round(ms_per_step, 3), "min": round(min(region_ms), 3),
                        "max": round(max(region_ms), 3), "spread_frac": round((max(region_ms) - min(region_ms)) / ms_per_step, 4),
                        "host_enqueue_ms_per_step": [round(v, 3) for v in region_host_ms],
                        "note": "host_enqueue = wall time the host needs to queue one step's launches; when it approaches ms_per_step the host "
                                "(its cores shared with other tenants of the node), not the GPU, sets the step time"},
        }
        # (this leg runs FIRST of the extra legs, straight behind the timed region: behind the reference-padding leg - other shapes, a
        # differently filled allocator - the same ten steps measured 3 ms slower than as a run of their own, behind the 874 MiB of rotating
        # operand sets of the roofline_by_time probe 9 ms slower; profiles/r05_ab/recompute_leg_order.txt)
        if world == 1 and not args.no_extras and args.model in ("M", "S") and not stub and dtype == torch.bfloat16:
            # SURVEY section 8(d) "report both": the joint + loss WITHOUT materialised lattice logits (statistics-only projection, gradient
            # epilogue on a re-computed logit tile; TFASR_JOINT_RECOMPUTE=1) next to the default materialised route, same batches
            try:
                model.joint_recompute = True
                model.timers, model.timer_work = {}, {}
                for i in range(4):
                    one_step(i)
                model.timers, model.timer_work = {}, {}
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nrec = 10
                for i in range(nrec):
                    one_step(i)
                torch.cuda.synchronize()
                dtq = (time.perf_counter() - t1) / nrec
                tj, tr = model.timers.get("joint_vocab_gemm") or [], model.timers.get("rnnt_loss") or []
                by = float(np.mean(model.timer_work["rnnt_loss"])) if tr else 0.0
                msr = float(np.mean([a.elapsed_time(b) for a, b in tr])) if tr else None
                out["joint_recompute_variant"] = {
                    "ms_per_step": round(dtq * 1e3, 3), "steps": nrec,
                    "projection_statistics_only_ms": round(float(np.mean([a.elapsed_time(b) for a, b in tj])), 4) if tj else None,
                    "loss_and_gradient_pass_ms": round(msr, 4) if msr else None,
                    "algorithmic_bytes": by, "achieved_GBps": round(by / (msr * 1e-3) / 1e9, 1) if msr else None,
                    "note": "no [cells, V] logits in HBM: the gradient tensor is written once by the epilogue of a re-computed vocabulary product "
                            "(one more matrix product instead of two passes over the tensor); default = the materialised route of the headline line"}
            except Exception as e:
                out["joint_recompute_variant"] = {"value": None, "error": repr(e)[:200]}
            finally:
                model.joint_recompute = False
                model.timers = None
        if not stub and args.model in ("M", "S") and dtype == torch.bfloat16 and not args.no_extras:
            try:
                rows_list = [args.batch * int(-(-(-(-(-(-int(np.asarray(b["nsamp"], np.int64).max()) // cfg.frame_step)) // 2)) // 2)) for b in batches]
                # in-step duration of the grouped weight gradients: a few more steps with the executor's event probe on
                from tensorflowasr_amd import kernels as _K2

                model.timers = None
                _K2.block_wgrad_probe(True)
                nprobe = 4
                for i in range(nprobe):
                    one_step(i)
                torch.cuda.synchronize()
                ms_in, n_in = _K2.block_wgrad_probe_read()
                _K2.block_wgrad_probe(False)
                fl_in = sum(cfg.num_blocks * sum(2.0 * rows_list[i % nb] * m * n for m, n in wgrad_group_shapes(cfg)) for i in range(nprobe))
                out["roofline_by_time"] = wgrad_group_roofline(cfg, rows_list, dev, in_step=(ms_in, n_in, fl_in))
            except Exception as e:
                out["roofline_by_time"] = {"value": None, "error": repr(e)[:200]}
        if world == 1 and not args.no_extras and args.model == "M" and args.padding == "batch" and size == "LibriSpeech-shaped":
            # BASELINE.md section 2 "report both": the same step with the reference's dataset-maximum padding (every utterance padded
            # to 475 760 samples / 230 labels, datasets.py:342-365).  The packed lattice and the length-aware kernels make the
            # padded LABEL positions free; the padded encoder frames are computed like the reference computes them.
            try:
                model.timers = None
                rb = [make_batch(cfg, args.batch, seed=10 + 13 * i, padding="reference", size=size) for i in range(nb)]
                rd = [to_train_data(b, dev) for b in rb]
                torch.cuda.synchronize()  # (complete in HBM before the first step that reads them: model.prefetched_inputs)
                for i in range(2):
                    model.train_step(rd[i % nb])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nref = 10
                for i in range(nref):
                    model.train_step(rd[i % nb])
                torch.cuda.synchronize()
                dtr = (time.perf_counter() - t1) / nref
                secs_r = float(np.mean([b["seconds"] for b in rb]))
                out["reference_padding"] = {"ms_per_step": round(dtr * 1e3, 3), "value": round(secs_r / 3600.0 / dtr, 4), "unit": "audio-hours/sec",
                                            "steps": nref, "padded_to": "475760 samples / 230 labels (datasets.py:342-365)"}
                del rd, rb
            except Exception as e:  # the headline number must still be reported
                out["reference_padding"] = {"value": None, "error": repr(e)[:200]}
        if world == 1 and not stub and not args.no_extras and not args.dp_hooks and args.model in ("M", "S") and dtype == torch.bfloat16:
            # the data-parallel route at one rank: what a rank of an N-GPU job runs, wire time excluded
            dpm = dp_route_line(["--model", args.model, "--batch", str(args.batch), "--padding", args.padding] + (["--workload", args.workload] if args.workload else []))
            out["dp_route_ms"] = None if dpm is None else round(dpm, 3)
            out["dp_route_over_single"] = None if dpm is None else round(dpm / ms_per_step, 4)
        if box is not None:
            # which box is this?  Clocks / power / temperature at both ends and under load + the two fixed calibration probes
            try:
                box["calibration"] = calibrate(dev, local_rank)
                box["after"] = box_snapshot(local_rank)
                ref_ms = (box["calibration"].get("committed") or {}).get("train_ms_per_step")
                if ref_ms and args.model == "M" and args.padding == "batch" and size == "LibriSpeech-shaped" and dtype == torch.bfloat16:
                    box["step_x_vs_committed"] = round(ms_per_step / ref_ms, 4)
                box["cpu_count"] = os.cpu_count()
                try:
                    box["loadavg"] = [round(v, 2) for v in os.getloadavg()]
                except OSError:
                    pass
            except Exception as e:
                box["calibration"] = {"value": None, "error": repr(e)[:200]}
            out["box"] = box
        if world == 1 and not stub and not args.no_extras and not args.dp_hooks and args.model == "M" and args.mode == "train" and dtype == torch.bfloat16 \
                and args.padding == "batch" and size == "LibriSpeech-shaped":
            # the rest of BASELINE.json's metric and its other single-GPU configurations, as short lines of their own (child processes,
            # this process idle meanwhile): greedy-decode RTF (M and S), configs[1] (Conformer-S, 32 x 10 s), configs[3] (ContextNet-L)
            data.clear()
            torch.cuda.empty_cache()
            out["decode"] = {"M": sub_line(["--mode", "decode", "--model", "M", "--steps", "5", "--warmup", "2"]),
                             "S": sub_line(["--mode", "decode", "--model", "S", "--steps", "5", "--warmup", "2"])}
            out["cfg2_conformer_S_10s"] = sub_line(["--model", "S", "--steps", "20", "--warmup", "5"])
            out["cfg4_contextnet_L"] = sub_line(["--model", "contextnet", "--steps", "10", "--warmup", "3"])
        if not args.no_cpu_baseline and world == 1 and not stub:
            try:
                if args.model != "contextnet":  # the CPU port baseline is the Conformer oracle
                    out["cpu_baseline"] = cpu_baseline(size if args.model.startswith("S") else "M", cfg.vocab_size, args.batch)
            except Exception as e:  # the GPU number must still be reported
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    if dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
