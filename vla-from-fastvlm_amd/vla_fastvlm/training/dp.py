"""Data-parallel exchange of the head gradient: the ONE collective of the path.

All 12 head gradients live in one flat fp32 buffer (12.2 MB at FastVLM-0.5B), so the exchange is a single all-reduce
(sum); the 1/world average is folded into the fused clip+AdamW kernel (`grad_scale`), and the global-norm clip is
computed on the reduced gradient, identically on every rank.  On GPUs the backend is RCCL over xGMI ("nccl" in torch)
and the collective is issued on a side stream; the same functions run under gloo on CPU for the world-size-2 tests.

Overlap (north_star: "all-reduce overlapped with backward"): the backbone is frozen (reference
model/fastvlm_adapter.py:501), so the frozen forward of batch k+1 does not depend on the parameters batch k is about to
update.  `GradExchange.start()` launches the all-reduce of batch k on the side stream and returns; the caller enqueues
batch k+1's backbone forward on the compute stream; `GradExchange.finish()` makes the compute stream wait for the
collective just before the optimiser kernel.  The collective then runs under ~10-60 ms of tower kernels instead of in
front of the optimiser.
"""
from __future__ import annotations

import itertools
from typing import Optional

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class GradExchange:
    """Two-phase all-reduce of one flat gradient buffer: start() on a side stream, finish() joins it."""

    def __init__(self, device: Optional[torch.device] = None, group=None):
        self.group = group
        self.stream = torch.cuda.Stream(device=device) if device is not None and torch.device(device).type == "cuda" else None
        self._pending = False
        self._ev = None  # (start, end) events of the last collective on the side stream: bench.py's allreduce_ms

    def start(self, flat_grads: torch.Tensor, timed: bool = False) -> float:
        """Launch SUM(flat_grads) across ranks (in place); returns the scale (1/world) the optimiser must apply."""
        world = world_size(self.group)
        if world == 1:
            return 1.0
        if flat_grads.is_cuda and self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(flat_grads.device))
            with torch.cuda.stream(self.stream):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
                if timed:
                    e1.record(self.stream)
                    self._ev = (e0, e1)
            self._pending = True
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / world

    def finish(self, device=None) -> None:
        """The compute stream waits for the collective started last (no host synchronisation)."""
        if self._pending:
            torch.cuda.current_stream(device).wait_stream(self.stream)
            self._pending = False

    def last_ms(self) -> Optional[float]:
        if self._ev is None:
            return None
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])


class BucketedGradExchange:
    """Unfrozen-backbone training: the gradient is ~0.5 G floats, produced bucket by bucket as the backward pass
    walks the decoder from its last layer to its first.  `bucket_ready(bucket, offset, numel)` -- the library's fv_bucket_cb, called on the
    host right after the last kernel writing that slice of the flat gradient has been ENQUEUED -- records an event on the compute stream
    and launches that slice's all-reduce (sum, in place) on the side stream behind it, so layer l's 60 MB travel over xGMI while layers
    l-1 .. 0 are still being differentiated; `finish()` joins the side stream in front of the optimiser and returns 1/world for its
    grad_scale.  Slices are disjoint and each element is reduced exactly once, so the result equals ONE all-reduce of the whole buffer
    bit for bit (tests/test_dp_gloo.py).  `min_numel` coalesces buckets that happen to be adjacent in the buffer (layer l then l-1)
    into fewer, larger collectives: a ring over 8 GPUs is per-link bound (7 x ~153 GB/s point to point), not NVSwitch-shaped."""

    def __init__(self, device: Optional[torch.device] = None, group=None, min_numel: int = 0):
        self.group = group
        self.stream = torch.cuda.Stream(device=device) if device is not None and torch.device(device).type == "cuda" else None
        self.min_numel = int(min_numel)
        self._flat: Optional[torch.Tensor] = None
        self._held: Optional[tuple] = None      # (offset, numel) waiting to be merged with an adjacent bucket
        self.launched: list = []                 # (offset, numel) of every collective of the current step, in launch order
        self.dry_run = False                     # measurement only (bench.py): the bookkeeping of a step without its collectives -- what the exchange exposes

    def begin(self, flat_grads: torch.Tensor) -> None:
        self._flat, self._held, self.launched = flat_grads, None, []

    def _launch(self, off: int, n: int) -> None:
        flat = self._flat
        self.launched.append((off, n))
        if world_size(self.group) == 1 or self.dry_run:
            return
        piece = flat[off: off + n]
        if flat.is_cuda and self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)

    def bucket_ready(self, bucket: int, offset: int, numel: int) -> None:
        if self._flat is None:
            raise RuntimeError("BucketedGradExchange.begin(flat_grads) must be called before the backward pass")
        if self._held is not None:
            ho, hn = self._held
            if offset + numel == ho:            # the new bucket sits right in front of the held one (layers arrive last to first)
                offset, numel = offset, numel + hn
            elif ho + hn == offset:
                offset, numel = ho, hn + numel
            else:
                self._launch(ho, hn)
            self._held = None
        if numel < self.min_numel:
            self._held = (offset, numel)
        else:
            self._launch(offset, numel)

    def finish(self, device=None) -> float:
        if self._held is not None:
            self._launch(*self._held)
            self._held = None
        if self.stream is not None and self._flat is not None and self._flat.is_cuda and world_size(self.group) > 1:
            torch.cuda.current_stream(device).wait_stream(self.stream)
        return 1.0 / world_size(self.group)


def allreduce_flat_grads(flat_grads: torch.Tensor, comm_stream: Optional["torch.cuda.Stream"] = None, group=None) -> float:
    """Sum `flat_grads` across ranks in place and join; returns the scale (1/world) the optimiser must apply.  The
    unpipelined form (start + finish back to back) for callers that have nothing to put in between."""
    world = world_size(group)
    if world == 1:
        return 1.0
    if flat_grads.is_cuda and comm_stream is not None:
        comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm_stream):
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        torch.cuda.current_stream().wait_stream(comm_stream)
    else:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def broadcast_flat(flat: torch.Tensor, src: int = 0, group=None) -> None:
    """Every rank takes rank `src`'s copy of a flat buffer (head parameters at start-up, optimiser moments on resume):
    what DDP/accelerate's prepare() does for the reference (training/trainer.py:68-78).  Only the gradient is exchanged
    afterwards, so replicas that start equal stay equal."""
    if world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def shard_batches(loader, rank: int, world: int):
    """Round-robin batch sharding in lock step: the loader is consumed `world` batches at a time and rank r takes the
    r-th of each group; a ragged tail (fewer than `world` batches left) is dropped, so every rank runs the SAME number
    of steps and the per-step all-reduces always pair up (independent samples, no other exchange)."""
    if world == 1:
        yield from loader
        return
    it = iter(loader)
    while True:
        group = list(itertools.islice(it, world))
        if len(group) < world:
            return
        yield group[rank]
