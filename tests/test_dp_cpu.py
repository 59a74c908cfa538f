"""CPU, world_size 2, gloo: the data-parallel hooks (bucketed gradient all-reduce over the flat buffer, sync-BN statistic
sums, metric mean, utterance sharding) behave like a single process on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tensorflowasr_amd import dp

    d = dp.init_from_env(backend="gloo")
    assert d.world == world and d.rank == rank
    n = 10_000
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(n, generator=g)
    d.bucket_bytes = 8_000
    d.attach(grad)
    # backward announces slices from the end of the buffer towards the front, with gaps
    d.grads_ready(9000, 10000)
    d.grads_ready(7000, 9000)
    d.grads_ready(6500, 7000)
    d.grads_ready(1000, 3000)
    d.finish_grads()
    # sync-BN: x rows sharded; global moments from summed (sum, sumsq)
    x = torch.arange(40, dtype=torch.float32).view(10, 4) + 100 * rank
    stats = torch.cat([x.sum(0), (x * x).sum(0)])
    d.allreduce_stats_(stats)
    loss = d.mean_scalar(torch.tensor([float(rank + 1)]))
    lo, hi = dp.shard_bounds(8, world, rank)
    torch.save(dict(grad=grad, stats=stats, loss=loss, shard=(lo, hi)), os.path.join(outdir, f"r{rank}.pt"))
    d.barrier()
    torch.distributed.destroy_process_group()


def test_two_process_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    want = sum(torch.randn(10_000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    for o in outs:
        np.testing.assert_allclose(o["grad"].numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    xs = torch.cat([torch.arange(40, dtype=torch.float32).view(10, 4) + 100 * r for r in range(world)])
    np.testing.assert_allclose(outs[0]["stats"].numpy(), torch.cat([xs.sum(0), (xs * xs).sum(0)]).numpy(), rtol=1e-6)
    assert float(outs[1]["loss"]) == 1.5
    assert outs[0]["shard"] == (0, 4) and outs[1]["shard"] == (4, 8)


def test_shard_bounds_rejects_ragged():
    from tensorflowasr_amd import dp

    with pytest.raises(ValueError):
        dp.shard_bounds(7, 2, 0)


def _worker_ga(rank, world, port, outdir):
    """The train_step_ga contract through the DP hooks: local accumulation, one exchange on the apply micro-step."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tensorflowasr_amd import dp

    d = dp.init_from_env(backend="gloo")
    n, ga = 4096, 3
    grad = torch.zeros(n)
    d.attach(grad)
    d.bucket_bytes = 4096
    snaps = []
    for micro in range(ga):
        d.set_reduce(micro + 1 >= ga)
        g = torch.randn(n, generator=torch.Generator().manual_seed(1000 * micro + rank))
        grad[2048:] += g[2048:]      # "joint / top of the encoder" becomes final first ...
        d.grads_ready(2048, 4096)
        grad[:2048] += g[:2048]      # ... then the bottom
        d.grads_ready(0, 1024)
        d.finish_grads()
        snaps.append(grad.clone())
    torch.save(snaps, os.path.join(outdir, f"ga{rank}.pt"))
    d.barrier()
    torch.distributed.destroy_process_group()


def test_two_process_gloo_gradient_accumulation(tmp_path):
    world, ga, n = 2, 3, 4096
    mp.spawn(_worker_ga, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"ga{r}.pt")) for r in range(world)]
    g = lambda micro, rank: torch.randn(n, generator=torch.Generator().manual_seed(1000 * micro + rank))
    for r in range(world):
        for micro in range(ga - 1):  # before the apply micro-step: purely local sums, nothing exchanged
            want = sum(g(m, r) for m in range(micro + 1))
            np.testing.assert_allclose(outs[r][micro].numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
        want = sum(g(m, rr) for m in range(ga) for rr in range(world))  # every micro-gradient of every rank exactly ONCE
        np.testing.assert_allclose(outs[r][ga - 1].numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


def _worker_wire(rank, world, port, outdir):
    """bf16 gradient wire vs the f32 wire, the dedicated sync-BN communicator and the per-rank accounting."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tensorflowasr_amd import dp

    d = dp.init_from_env(backend="gloo")
    assert d.stats_group is not d.group and d.stats_group is not None  # its own communicator
    d16 = dp.DataParallel(grad_wire="bf16")
    n = 50_000
    # gradient-like values, pre-scaled by 1 / (B_local * world) as the step scales them
    g0 = torch.randn(n, generator=torch.Generator().manual_seed(7 + rank)) * torch.logspace(-6, 0, n) / (4 * world)
    out = {}
    for name, hook in (("f32", d), ("bf16", d16)):
        grad = g0.clone()
        hook.bucket_bytes = 40_000
        hook.attach(grad)
        hook.accounting.enable()
        hook.grads_ready(30_000, 50_000)
        hook.grads_ready(20_000, 30_000)
        hook.grads_ready(0, 5_000)
        st = torch.full((8,), float(rank + 1))
        hook.allreduce_stats_(st)
        hook.finish_grads()
        acc = hook.accounting.summary(1)
        assert acc["syncbn_wait_calls_per_step"] == 1 and acc["grad_allreduce_exposed_calls_per_step"] == 1
        assert acc["syncbn_wait_ms"] >= 0 and acc["grad_allreduce_exposed_ms"] >= 0
        assert float(st[0]) == sum(range(1, world + 1))
        out[name] = grad
    torch.save(out, os.path.join(outdir, f"w{rank}.pt"))
    d.barrier()
    torch.distributed.destroy_process_group()


def test_bf16_gradient_wire_matches_f32_wire_within_bf16_rounding(tmp_path):
    world, n = 2, 50_000
    mp.spawn(_worker_wire, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"w{r}.pt")) for r in range(world)]
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) * torch.logspace(-6, 0, n) / (4 * world) for r in range(world))
    for o in outs:
        np.testing.assert_allclose(o["f32"].numpy(), want.numpy(), rtol=1e-6, atol=1e-9)
        # every summand rounded to bf16 once (2^-9 relative each), one bf16 add: |err| <= 2^-8 * sum|summands| elementwise; whole vector far better
        summ = sum((torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) * torch.logspace(-6, 0, n) / (4 * world)).abs() for r in range(world))
        assert bool(((o["bf16"] - want).abs() <= 2.0 ** -7 * summ + 1e-12).all())
        rel = float((o["bf16"] - want).norm() / want.norm())
        assert rel < 4e-3, rel
    assert torch.equal(outs[0]["bf16"], outs[1]["bf16"])  # replicas stay bit-identical
