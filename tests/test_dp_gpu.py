"""GPU, world_size 2 on ONE device (gloo carries the collectives; RCCL refuses two ranks on one GPU): the data-parallel
train step - utterance sharding, sync-BN statistics all-reduced between the native block executor's phases A and B (forward
and backward), bucketed gradient all-reduce announced while the backward walks down the encoder - gives the gradients of the
single-process step on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(cfg, lens, ulens, seed=4, N=4000, U=6):
    from tensorflowasr_amd.schemas import TrainData, TrainInput, TrainLabel

    rng = np.random.default_rng(seed)
    B = len(lens)
    sig = np.clip(rng.standard_normal((B, N)) * 0.1, -1, 1).astype(np.float32)
    for b, n in enumerate(lens):
        sig[b, n:] = 0.0
    labels = rng.integers(1, cfg.vocab_size, (B, U)).astype(np.int32)
    for b, u in enumerate(ulens):
        labels[b, u:] = 0
    preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)

    def make(lo, hi):
        return TrainData(
            TrainInput(torch.from_numpy(sig[lo:hi]), torch.tensor(lens[lo:hi], dtype=torch.int32), torch.from_numpy(preds[lo:hi]),
                       torch.tensor([u + 1 for u in ulens[lo:hi]], dtype=torch.int32)),
            TrainLabel(torch.from_numpy(labels[lo:hi]), torch.tensor(ulens[lo:hi], dtype=torch.int32)))

    return make


LENS, ULENS = [4000, 2700, 3300, 3900], [6, 3, 5, 4]


def _worker(rank, world, port, outdir, dtype_name):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tensorflowasr_amd import configs, dp
    from tensorflowasr_amd.conformer import ConformerTransducer

    torch.cuda.set_device(0)
    d = dp.init_from_env(backend="gloo")
    dtype = getattr(torch, dtype_name)
    cfg = configs.conformer_tiny(dropout=0.0)
    model = ConformerTransducer(cfg, torch.device("cuda", 0), dtype=dtype, seed=5, dp=d)
    d.attach(model.ps.grad)
    lo, hi = dp.shard_bounds(len(LENS), world, rank)
    data = _batch(cfg, LENS, ULENS)(lo, hi)
    model.zero_grad()
    costs = model.loss_and_backward(data, True, (None, None))
    torch.cuda.synchronize()
    torch.save(dict(costs=costs.cpu(), grad=model.ps.grad.cpu(), mm=model.ps.state["enc/block1/conv/bn/mm"].cpu()), os.path.join(outdir, f"r{rank}.pt"))
    d.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_two_ranks_match_single_process(dev, tmp_path, dtype_name):
    from tensorflowasr_amd import configs
    from tensorflowasr_amd.conformer import ConformerTransducer

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), dtype_name), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    dtype = getattr(torch, dtype_name)
    cfg = configs.conformer_tiny(dropout=0.0)
    model = ConformerTransducer(cfg, dev, dtype=dtype, seed=5)
    model.zero_grad()
    costs = model.loss_and_backward(_batch(cfg, LENS, ULENS)(0, len(LENS)), True, (None, None))
    torch.cuda.synchronize()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(torch.cat([o["costs"] for o in outs]).numpy(), costs.cpu().numpy(), rtol=tol)
    g = model.ps.grad.cpu().numpy()
    for o in outs:  # every rank holds the all-reduced global-batch gradient
        np.testing.assert_allclose(o["grad"].numpy(), g, rtol=tol, atol=tol * float(np.abs(g).max()))
        np.testing.assert_allclose(o["mm"].numpy(), model.ps.state["enc/block1/conv/bn/mm"].cpu().numpy(), rtol=tol, atol=1e-5)


def _worker_ga(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tensorflowasr_amd import configs, dp
    from tensorflowasr_amd.conformer import ConformerTransducer

    torch.cuda.set_device(0)
    d = dp.init_from_env(backend="gloo")
    cfg = configs.conformer_tiny(dropout=0.0)
    model = ConformerTransducer(cfg, torch.device("cuda", 0), dtype=torch.float32, seed=5, dp=d)
    d.attach(model.ps.grad)
    model.ga_steps = 2
    model.optimizer["schedule"] = 1e-3
    model.optimizer["eps"] = 1e-4
    make = _batch(cfg, LENS, ULENS)
    for micro in range(2):  # micro-batch m = utterances [2m, 2m+2); this rank's shard is one utterance of it
        model.train_step(make(2 * micro + rank, 2 * micro + rank + 1), masks=(None, None))
        if micro == 0:
            local_after_first = model.ps.grad.clone()
    torch.cuda.synchronize()
    torch.save(dict(flat=model.ps.flat.cpu(), grad=model.ps.grad.cpu(), first=local_after_first.cpu(), step=model.step),
               os.path.join(outdir, f"ga{rank}.pt"))
    d.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_with_gradient_accumulation_match_single_process(dev, tmp_path):
    """DP x GA (base_model.py:200-209): micro-gradients accumulate LOCALLY and are all-reduced once, on the apply micro-step.
    Two ranks x ga_steps=2 on one-utterance shards == one process with ga_steps=2 on the two-utterance micro-batches."""
    from tensorflowasr_amd import configs
    from tensorflowasr_amd.conformer import ConformerTransducer

    world = 2
    mp.spawn(_worker_ga, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"ga{r}.pt")) for r in range(world)]
    cfg = configs.conformer_tiny(dropout=0.0)
    model = ConformerTransducer(cfg, dev, dtype=torch.float32, seed=5)
    model.ga_steps = 2
    model.optimizer["schedule"] = 1e-3
    model.optimizer["eps"] = 1e-4
    make = _batch(cfg, LENS, ULENS)
    model.train_step(make(0, 2), masks=(None, None))
    first = model.ps.grad.clone()
    model.train_step(make(2, 4), masks=(None, None))
    torch.cuda.synchronize()
    assert model.step == 1 and all(o["step"] == 1 for o in outs)
    g = model.ps.grad.cpu().numpy()
    tol = 5e-4
    # after micro-step 1 the ranks hold un-reduced LOCAL gradients whose sum is the single-process micro-gradient
    np.testing.assert_allclose((outs[0]["first"] + outs[1]["first"]).numpy(), first.cpu().numpy(), rtol=tol, atol=tol * float(np.abs(g).max()))
    assert float((outs[0]["first"] - outs[1]["first"]).abs().max()) > 0
    for o in outs:
        np.testing.assert_allclose(o["grad"].numpy(), g, rtol=tol, atol=tol * float(np.abs(g).max()))
        np.testing.assert_allclose(o["flat"].numpy(), model.ps.flat.cpu().numpy(), rtol=0, atol=5e-5)


def _worker_gradn(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tensorflowasr_amd import configs, dp
    from tensorflowasr_amd.conformer import ConformerTransducer

    torch.cuda.set_device(0)
    d = dp.init_from_env(backend="gloo")
    cfg = configs.conformer_tiny(dropout=0.0)
    model = ConformerTransducer(cfg, torch.device("cuda", 0), dtype=torch.float32, seed=5, dp=d)
    d.attach(model.ps.grad)
    model.gradn_config = {"step": 0, "stddev": 0.05}
    model.optimizer["schedule"] = 1e-3
    lo, hi = dp.shard_bounds(len(LENS), world, rank)
    make = _batch(cfg, LENS, ULENS)
    for _ in range(2):
        model.train_step(make(lo, hi), masks=(None, None))
    torch.cuda.synchronize()
    torch.save(dict(flat=model.ps.flat.cpu(), grad=model.ps.grad.cpu()), os.path.join(outdir, f"gn{rank}.pt"))
    d.barrier()
    torch.distributed.destroy_process_group()


def test_gradient_noise_keeps_replicas_identical(dev, tmp_path):
    """gradn_config with world > 1: the noise is added after the all-reduce with a rank-free seed (the sum of the replicas'
    independent draws of the reference, base_model.py:185-192, in one N(0, stddev * sqrt(world)) draw), so the parameters of the
    replicas stay BIT-identical - and the noise really is there (the gradient differs from the noise-free one by ~stddev*sqrt(2))."""
    world = 2
    mp.spawn(_worker_gradn, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"gn{r}.pt")) for r in range(world))
    assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["grad"], b["grad"])
    from tensorflowasr_amd import configs
    from tensorflowasr_amd.conformer import ConformerTransducer

    cfg = configs.conformer_tiny(dropout=0.0)
    model = ConformerTransducer(cfg, dev, dtype=torch.float32, seed=5)
    model.optimizer["schedule"] = 1e-3
    make = _batch(cfg, LENS, ULENS)
    for _ in range(2):
        model.train_step(make(0, len(LENS)), masks=(None, None))
    torch.cuda.synchronize()
    diff = (a["grad"] - model.ps.grad.cpu())
    live = model.ps.grad.cpu() != 0
    sd = float(diff[live].std())
    assert 0.05 * 2 ** 0.5 * 0.8 < sd < 0.05 * 2 ** 0.5 * 1.2, sd
