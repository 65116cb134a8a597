"""Data-parallel plumbing of the hot path: one process per GPU, batch shards, no data-path collective.

Samples are independent in eval (BatchNorm uses running statistics), so inference needs no exchange step
(SURVEY.md 8(e)); the only collectives are the measurement barrier / max-over-ranks and an optional gather of the
[B, J, 3] poses.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of ``total`` samples for ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: str | None = None, device: torch.device | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); no-op for world 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def barrier(device: torch.device | None = None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: torch.device | None = None) -> float:
    """Slowest rank's time: what whole-job throughput is computed from."""
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device: torch.device | None = None):
    """every rank's value, in rank order (per-rank step times of a measurement)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_poses(pose: torch.Tensor, counts=None) -> torch.Tensor:
    """All ranks' [b_r, J, 3] poses concatenated in rank order (equal shard sizes unless ``counts`` is given)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return pose
    world = dist.get_world_size()
    if counts is None:
        out = [torch.empty_like(pose) for _ in range(world)]
        dist.all_gather(out, pose.contiguous())
        return torch.cat(out, dim=0)
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(pose.shape[1:]), dtype=pose.dtype, device=pose.device)
    pad[: pose.shape[0]] = pose
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


class GradReducer:
    """Gradient averaging OVERLAPPED with the backward: the head's gradients live in one flat fp32 arena laid out
    in the order the backward finishes them (head + encoders first, then ViT layers last to first, patch embedding last); as soon as
    a bucket's kernels are enqueued the training Function calls ``bucket_ready(lo, hi)`` and the bucket's all-reduce starts on the
    process group's own stream (RCCL: torch orders it behind the compute stream's current point) while the backward goes on.
    ``finish()`` makes the compute stream wait for the outstanding buckets (bucket by bucket: on gloo, which has no averaging
    reduction, bucket k is scaled by 1 / world while bucket k+1 is still on the wire; on RCCL the collective itself averages,
    ReduceOp.AVG, and there is no scaling pass) and brackets that wait with two events: ``read_exposed_ms()`` is the EXPOSED
    all-reduce time of the last step.  In place on arena slices: no torch.cat pack, no copy-back.  No-op for one rank unless
    ``force`` is set (tests: a one-rank RCCL group still runs every collective, the side stream and the event hand-off)."""

    def __init__(self):
        self.pending = []
        self.exposed_ms = 0.0        # sum of the exposed waits read so far (read_exposed_ms accumulates)
        self.steps = 0               # finish() calls that had collectives to wait for
        self.force = False           # run the collectives on a one-rank group too
        self.last_buckets = 0        # collectives issued by the last step
        self.last_bytes = 0          # bytes all-reduced by the last step
        self._ev = None
        self._side = None

    def active(self):
        return dist.is_initialized() and (dist.get_world_size() > 1 or self.force)

    def _op(self):
        """RCCL averages inside the collective; gloo has no AVG (sum, then a scaling pass in finish())"""
        return dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM

    def begin(self, arena):
        self.arena = arena
        self.pending = []

    def events(self, n):
        """n reusable events for egotap_lift_backward's bucket hand-off (created by a first record: the raw handle exists afterwards)"""
        if len(getattr(self, "_bucket_events", ())) != n:
            self._bucket_events = [torch.cuda.Event() for _ in range(n)]
            for e in self._bucket_events:
                e.record()
        return self._bucket_events

    def bucket_ready(self, lo, hi, after=None):
        """start the all-reduce of arena[lo:hi]: behind the current point of the compute stream, or (one-call backward: the whole
        backward is already enqueued) behind the event the library recorded when the bucket became final"""
        if not self.active() or hi <= lo:
            return
        if after is None or not self.arena.is_cuda:
            self.pending.append((dist.all_reduce(self.arena[lo:hi], op=self._op(), async_op=True), lo, hi))
            return
        if self._side is None:
            self._side = torch.cuda.Stream()
        with torch.cuda.stream(self._side):
            self._side.wait_event(after)
            self.pending.append((dist.all_reduce(self.arena[lo:hi], op=self._op(), async_op=True), lo, hi))

    def finish(self):
        """Order the caller's stream behind every outstanding bucket.  Returns the number of collectives waited for; the time the
        stream spent waiting is only known once the GPU got there: read_exposed_ms()."""
        if not self.active():
            return 0
        world = dist.get_world_size()
        scale = self._op() == dist.ReduceOp.SUM and world > 1
        cuda = self.arena.is_cuda
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, lo, hi in self.pending:
            work.wait()
            if scale and not cuda:
                self.arena[lo:hi].mul_(1.0 / world)
        if cuda:
            e1.record()
            self._ev = (e0, e1)
            if scale:                                        # (gloo on device tensors: the one-GPU rehearsal)
                for _, lo, hi in self.pending:
                    self.arena[lo:hi].mul_(1.0 / world)
        n = len(self.pending)
        self.last_buckets = n
        self.last_bytes = sum((hi - lo) * self.arena.element_size() for _, lo, hi in self.pending)
        self.pending = []
        self.steps += 1 if n else 0
        return n

    def read_exposed_ms(self):
        """exposed wait of the LAST finish() in milliseconds (synchronises on its closing event); accumulated in ``exposed_ms``"""
        if self._ev is None:
            return 0.0
        self._ev[1].synchronize()
        ms = float(self._ev[0].elapsed_time(self._ev[1]))
        self.exposed_ms += ms
        self._ev = None
        self._last_ms = ms
        return ms
