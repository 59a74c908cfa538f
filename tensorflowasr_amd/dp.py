"""Single-node data parallelism: one process per GPU, RCCL over xGMI via torch.distributed (backend "nccl").

Replaces tf.distribute.MirroredStrategy (tensorflow_asr/utils/env_util.py:57-70) and the three implicit collectives of
the reference's train step:
  C1  gradient all-reduce (base_model.py:192, keras optimizer.apply under MirroredStrategy)
        -> a few large all-reduces over contiguous slices of the flat gradient buffer, issued as soon as a slice is
           final (joint first, then one per Conformer block while backward walks down the encoder) so they hide under
           the remaining backward; ring all-reduce over xGMI is per-link bound, so few big messages beat many small.
  C2  synchronized BatchNorm statistics (conformer.py:327-333, subsampling.py:197-203)
        -> one small SUM all-reduce of the packed (sum, sumsq) / (sum dz, sum dz*xhat) vector per BN.
  C3  loss metric mean (base_model.py:257-261, keras_util.py:9-26) -> `mean_scalar`.
Utterances are sharded contiguously: global batch = per-replica batch x world (datasets.py:108).  The loss gradient
is pre-scaled by 1/(B_local*world) at its source, so SUM reductions produce the global-batch mean gradient
(base_model.py:166-173).  Works on CPU tensors with the gloo backend too (tests/test_dp_cpu.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return DataParallel()


def shard_bounds(n_global, world, rank):
    """Contiguous utterance shard of rank `rank` (datasets.py:108, base_model.py:86)."""
    per = n_global // world
    if per * world != n_global:
        raise ValueError(f"global batch {n_global} is not divisible by world size {world}")
    return rank * per, (rank + 1) * per


class DataParallel:
    def __init__(self, group=None, bucket_bytes=32 << 20):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.flat_grad = None
        self.bucket_bytes = bucket_bytes
        self._done = []
        self._pending = []
        self._staged = None
        self._reduce = True

    def set_reduce(self, on):
        """Gradient accumulation: micro-steps before the apply step keep their gradients local (the reference all-reduces
        inside optimizer.apply only, base_model.py:200-209); grads_ready / finish_grads are no-ops while this is off."""
        self._reduce = bool(on)

    def attach(self, flat_grad):
        self.flat_grad = flat_grad
        return self

    # -- C2
    def allreduce_stats_(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    # -- C1
    def grads_ready(self, lo, hi):
        """[lo, hi) of the flat gradient is final: start its all-reduce now (async; overlaps the rest of backward).
        Adjacent small slices are coalesced until `bucket_bytes` so each message is large enough for the xGMI ring."""
        if hi <= lo or not self._reduce:
            return
        if self._staged is not None and (self._staged[0] == hi or self._staged[1] == lo):
            self._staged = (min(lo, self._staged[0]), max(hi, self._staged[1]))
        else:
            self._flush()
            self._staged = (lo, hi)
        if (self._staged[1] - self._staged[0]) * 4 >= self.bucket_bytes:
            self._flush()

    def _flush(self):
        if self._staged is None:
            return
        lo, hi = self._staged
        self._staged = None
        work = dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(work)
        self._done.append((lo, hi))

    def finish_grads(self):
        """Reduce whatever has not been announced yet and wait for everything."""
        if not self._reduce:
            return
        self._flush()
        n = self.flat_grad.numel()
        cur = 0
        for lo, hi in sorted(self._done):
            if lo > cur:
                self._pending.append(dist.all_reduce(self.flat_grad[cur:lo], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            cur = max(cur, hi)
        if cur < n:
            self._pending.append(dist.all_reduce(self.flat_grad[cur:n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self._pending:
            w.wait()
        self._pending, self._done = [], []

    # -- C3
    def mean_scalar(self, t):
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world

    def max_scalar(self, x, device):
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        dist.barrier(group=self.group)
