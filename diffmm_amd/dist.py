"""Data-parallel replica logic (one process per GPU, torch.distributed; backend 'nccl' is RCCL on ROCm, 'gloo' in the CPU tests).

The hot path shards by MINIBATCH at the sizes that fit one GPU: every replica holds the graph and the weights, the minibatches
of a phase are dealt round-robin, and the replicas' gradients are averaged before Adam.  A global step consumes up to `world`
minibatches; in the last global step of a phase some replicas have none -- they join the all-reduce with zeros and the mean is
taken over the contributing replicas only, so the update equals the single-process update on the union of the minibatches' mean
loss.  (InfoNCE negatives stay replica-local.)
"""
from __future__ import annotations

from typing import Iterator, Optional, Sequence, Tuple, TypeVar

import torch

T = TypeVar("T")


def deal(batches: Sequence[T], world: int, rank: int) -> Iterator[Tuple[int, Optional[T]]]:
    """Yield (number of replicas that hold a minibatch in this global step, this replica's minibatch or None)."""
    for gs in range(0, len(batches), world):
        group = batches[gs:gs + world]
        yield len(group), (group[rank] if rank < len(group) else None)


def global_steps(n_batches: int, world: int) -> int:
    return (n_batches + world - 1) // world


def allreduce_mean_(flat: torch.Tensor, contributing: bool, n_contrib: int, group=None) -> torch.Tensor:
    """In-place mean of `flat` over the contributing replicas (non-contributors pass zeros).  One collective per call."""
    import torch.distributed as dist
    if not contributing:
        flat.zero_()
    dist.all_reduce(flat, group=group)
    flat.mul_(1.0 / float(n_contrib))
    return flat


def allgather_disjoint_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """Replicas wrote disjoint slots of a zero-initialised buffer (the rebuild's edge lists): a sum all-reduce gathers them."""
    import torch.distributed as dist
    dist.all_reduce(buf, group=group)
    return buf


def shard_owner(index: int, world: int) -> int:
    """Replica that processes user batch `index` of the rebuild."""
    return index % world
