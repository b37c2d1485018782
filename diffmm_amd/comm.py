"""Collectives of the user-sharded path (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the tests).
One indirection so that ONE process can run a single replica of a larger job with every collective reduced to its local part (`LOCAL` as the
group: tools/scale_epoch.py --emulate-world, the per-GPU work and memory of a job the box has no GPUs for).  Nothing here computes."""
from __future__ import annotations

import os

import torch

LOCAL = "local"  # stand-in group of an emulated replica


def active(world: int) -> bool:
    """Does a run over `world` replicas issue its collectives?  world > 1, or DMM_FORCE_COLLECTIVES=1: a single replica runs the user-sharded engines and
    sends every collective through the process group (of size 1) -- the way the RCCL communicator, its stream ordering and its dtype / contiguity
    rules are exercised on a one-GPU box (tests/test_bench_gpu.py: bench.py --gpus 1 --backend nccl under that variable)."""
    return world > 1 or os.environ.get("DMM_FORCE_COLLECTIVES") == "1"


class _Done:
    def wait(self):
        return True


def all_reduce(t: torch.Tensor, group=None, op=None, async_op: bool = False):
    if group is LOCAL:
        return _Done() if async_op else None
    import torch.distributed as dist
    return dist.all_reduce(t, op=dist.ReduceOp.SUM if op is None else getattr(dist.ReduceOp, op), group=group, async_op=async_op)


def all_gather_into(out: torch.Tensor, t: torch.Tensor, rank: int, group=None) -> None:
    """out [world * n, ...] <- every replica's t [n, ...] in rank order (LOCAL: the own block only)."""
    if group is LOCAL:
        n = t.shape[0]
        out[rank * n:(rank + 1) * n].copy_(t)
        return
    import torch.distributed as dist
    dist.all_gather_into_tensor(out, t, group=group)


def reduce_scatter(out: torch.Tensor, inp: torch.Tensor, rank: int, world: int, group=None) -> None:
    """out [n, ...] <- the rank-th of the `world` equal chunks of the sum of every replica's inp [world * n, ...].  RCCL: one reduce-scatter
    (7/8 of the buffer out per GPU at world 8, against twice that for an all-reduce); gloo (tests) has no reduce-scatter: all-reduce + slice, the same
    values.  LOCAL: the own chunk of the own partial."""
    n = out.shape[0]
    if group is LOCAL:
        out.copy_(inp[rank * n:(rank + 1) * n])
        return
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(out, inp, group=group)
    else:
        dist.all_reduce(inp, group=group)
        out.copy_(inp[rank * n:(rank + 1) * n])


def broadcast(t: torch.Tensor, src: int = 0, group=None) -> None:
    """t <- replica `src`'s t (LOCAL: nothing to do)."""
    if group is LOCAL:
        return
    import torch.distributed as dist
    dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
