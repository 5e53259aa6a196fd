"""Data-parallel exchange of the head gradient: the ONE collective of the path.

All 12 head gradients live in one flat fp32 buffer (12.2 MB at FastVLM-0.5B), so the exchange is a single all-reduce
(sum); the 1/world average is folded into the fused clip+AdamW kernel (`grad_scale`), and the global-norm clip is
computed on the reduced gradient, identically on every rank.  On GPUs the backend is RCCL over xGMI ("nccl" in torch)
and the collective is issued on a side stream; the same function runs under gloo on CPU for the world-size-2 tests.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_flat_grads(flat_grads: torch.Tensor, comm_stream: Optional["torch.cuda.Stream"] = None, group=None) -> float:
    """Sum `flat_grads` across ranks in place; returns the scale (1/world) the optimiser must apply."""
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


def shard_batches(loader, rank: int, world: int):
    """Round-robin batch sharding: rank r consumes batches r, r+world, ... (independent samples, no exchange)."""
    import itertools
    return loader if world == 1 else itertools.islice(loader, rank, None, world)
