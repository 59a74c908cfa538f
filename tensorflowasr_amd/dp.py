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


class Accounting:
    """Per-rank accounting of where a data-parallel step waits for the wire, read after the timed region:
      syncbn_wait_ms            time of the compute stream between queuing a sync-BN statistics reduction and having its result
                                (RCCL launch + ring latency + waiting for the slowest rank to arrive) - 36 of them per Conformer step;
      grad_allreduce_exposed_ms time the compute stream spends in finish_grads: what is left of the gradient all-reduces after the
                                backward they were overlapped with has ended (0 = fully hidden).
    HIP events on the stream the step runs on (device tensors) or the host clock (gloo / CPU tensors).  Off unless `enable()`d: two
    events per reduction are host work a production step does not need."""

    def __init__(self):
        self.on = False
        self.reset()

    def enable(self, on=True):
        self.on = bool(on)
        self.reset()
        return self

    def reset(self):
        self._ev = {"syncbn_wait_ms": [], "grad_allreduce_exposed_ms": []}
        self._host = {"syncbn_wait_ms": 0.0, "grad_allreduce_exposed_ms": 0.0}
        self.calls = {"syncbn_wait_ms": 0, "grad_allreduce_exposed_ms": 0}

    def begin(self, key, cuda):
        if not self.on:
            return None
        self.calls[key] += 1
        if cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time

        return time.perf_counter()

    def end(self, key, tok, cuda):
        if tok is None:
            return
        if cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev[key].append((tok, e))
        else:
            import time

            self._host[key] += (time.perf_counter() - tok) * 1e3

    def summary(self, steps):
        """-> {key: ms per step, key_calls: calls per step}; synchronises the device."""
        out = {}
        if any(self._ev.values()):
            torch.cuda.synchronize()
        for k in self._ev:
            ms = self._host[k] + sum(a.elapsed_time(b) for a, b in self._ev[k])
            out[k] = round(ms / max(steps, 1), 4)
            out[k.replace("_ms", "_calls_per_step")] = round(self.calls[k] / max(steps, 1), 1)
        return out


class DataParallel:
    """`grad_wire` = "f32" (default) or "bf16": the type the gradient buckets travel in.  bf16 halves the bytes on the xGMI ring (130 MB
    -> 65 MB for Conformer-M); every rank's bucket is rounded to bf16 ONCE before the sum (the local gradients are already pre-scaled by
    1 / (B_local * world), so the sum is the mean and stays in range), the ring adds in bf16, the result is widened back into the f32
    buffer the optimizer reads.  TFASR_DP_GRAD_WIRE overrides.  tests/test_dp_cpu.py bounds the difference against the f32 wire.

    Sync-BN statistics go through their OWN communicator (`stats_group`): torch.distributed gives every process group its own RCCL
    communicator and internal stream, so a 2 KB statistics reduction the forward / backward chain is waiting for never queues behind a
    32 MB gradient bucket that is in flight on the gradient communicator's stream (reference site conformer.py:327-333)."""

    def __init__(self, group=None, bucket_bytes=32 << 20, grad_wire=None, stats_group="own"):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.flat_grad = None
        self.bucket_bytes = bucket_bytes
        self.grad_wire = (grad_wire or os.environ.get("TFASR_DP_GRAD_WIRE", "f32")).lower()
        if self.grad_wire not in ("f32", "bf16"):
            raise ValueError(f"grad_wire must be 'f32' or 'bf16', not {self.grad_wire!r}")
        if stats_group == "own":
            # same ranks, second communicator (a collective call: every rank of `group` constructs its DataParallel at the same point)
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            stats_group = dist.new_group(ranks=ranks)
        self.stats_group = stats_group
        self.accounting = Accounting()
        self._done = []
        self._pending = []
        self._staged = None
        self._staged_events = []
        self._wire = []
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
        tok = self.accounting.begin("syncbn_wait_ms", t.is_cuda)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.stats_group)
        self.accounting.end("syncbn_wait_ms", tok, t.is_cuda)
        return t

    # -- C1
    def grads_ready(self, lo, hi):
        """[lo, hi) of the flat gradient is final ON THE CURRENT STREAM: start its all-reduce (async; overlaps the rest of backward).
        Adjacent small slices are coalesced until `bucket_bytes` so each message is large enough for the xGMI ring.  The release is
        stream-explicit: an event recorded here, on the stream whose launches produced the slice, is what the all-reduce
        waits for - whichever stream happens to be current when the staged bucket is finally flushed."""
        if hi <= lo or not self._reduce:
            return
        if self._staged is not None and (self._staged[0] == hi or self._staged[1] == lo):
            self._staged = (min(lo, self._staged[0]), max(hi, self._staged[1]))
        else:
            self._flush()
            self._staged = (lo, hi)
        if self.flat_grad is not None and self.flat_grad.is_cuda:
            ev = torch.cuda.Event()
            ev.record()  # (current stream)
            self._staged_events.append(ev)
        if (self._staged[1] - self._staged[0]) * 4 >= self.bucket_bytes:
            self._flush()

    def _all_reduce_slice(self, lo, hi):
        g = self.flat_grad[lo:hi]
        if self.grad_wire == "bf16" and g.dtype == torch.float32:
            w = torch.empty(hi - lo, dtype=torch.bfloat16, device=g.device)
            if g.is_cuda:
                from . import kernels as K

                K.cast(g, w)
            else:
                w.copy_(g)
            work = dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._wire.append((work, w, lo, hi))
            return None
        return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _flush(self):
        if self._staged is None:
            return
        lo, hi = self._staged
        self._staged = None
        if self._staged_events:
            cur = torch.cuda.current_stream()
            for ev in self._staged_events:
                cur.wait_event(ev)  # the producers' streams, not merely whatever is current now
            self._staged_events = []
        work = self._all_reduce_slice(lo, hi)
        if work is not None:
            self._pending.append(work)
        self._done.append((lo, hi))

    def finish_grads(self):
        """Reduce whatever has not been announced yet and wait for everything."""
        if not self._reduce:
            return
        cuda = self.flat_grad is not None and self.flat_grad.is_cuda
        self._flush()
        n = self.flat_grad.numel()
        cur = 0
        rest = []
        for lo, hi in sorted(self._done):
            if lo > cur:
                rest.append((cur, lo))
            cur = max(cur, hi)
        if cur < n:
            rest.append((cur, n))
        for lo, hi in rest:
            work = self._all_reduce_slice(lo, hi)
            if work is not None:
                self._pending.append(work)
        tok = self.accounting.begin("grad_allreduce_exposed_ms", cuda)
        for w in self._pending:
            w.wait()
        for work, w, lo, hi in self._wire:
            work.wait()
            g = self.flat_grad[lo:hi]
            if g.is_cuda:
                from . import kernels as K

                K.cast(w, g)
            else:
                g.copy_(w)
        self.accounting.end("grad_allreduce_exposed_ms", tok, cuda)
        self._pending, self._done, self._wire = [], [], []

    # -- C3
    def mean_scalar(self, t):
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world

    def max_scalar(self, x, device):
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def min_scalar(self, x, device):
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return float(t.item())

    def barrier(self):
        dist.barrier(group=self.group)
