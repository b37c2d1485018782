"""N > 1 path on CPU: two processes over gloo exercise the replica scheduling and gradient-averaging logic the GPU path uses
(diffmm_amd/dist.py): round-robin dealing with a ragged last global step, mean over the contributing replicas only, and the
disjoint-slot gather of the rebuild's edge lists.  The reference update on the union of the minibatches is computed in-process."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from diffmm_amd import dist as D


def test_deal_covers_every_batch_once():
    for n in (0, 1, 5, 8, 18):
        for world in (1, 2, 4, 8):
            seen, steps = [], 0
            per_rank = [list(D.deal(list(range(n)), world, r)) for r in range(world)]
            assert all(len(p) == D.global_steps(n, world) for p in per_rank)
            for gs in range(D.global_steps(n, world)):
                sizes = {per_rank[r][gs][0] for r in range(world)}
                assert len(sizes) == 1, "every replica must agree on the number of contributors"
                mine = [per_rank[r][gs][1] for r in range(world) if per_rank[r][gs][1] is not None]
                assert len(mine) == sizes.pop()
                seen += mine
            assert seen == list(range(n))


def _worker(rank, world, port, n_batches, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        w = torch.zeros(64)                                       # replicated "weights"
        grads = [torch.randn(64) for _ in range(n_batches)]      # gradient of minibatch b (same on every rank: seeded)
        for n_contrib, mine in D.deal(list(range(n_batches)), world, rank):
            g = grads[mine].clone() if mine is not None else torch.full((64,), 123.0)  # junk must be ignored when idle
            D.allreduce_mean_(g, mine is not None, n_contrib)
            w -= 0.1 * g
        edges = torch.zeros(10, dtype=torch.int32)
        for i in range(10):
            if D.shard_owner(i, world) == rank:
                edges[i] = 100 + i
        D.allgather_disjoint_(edges)
        out[rank] = (w.numpy().copy(), edges.numpy().copy())
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("n_batches", [4, 5])
def test_two_replicas_match_single_process(n_batches):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_batches, out), nprocs=world, join=True)
    torch.manual_seed(0)
    grads = [torch.randn(64) for _ in range(n_batches)]
    w = torch.zeros(64)
    for gs in range(0, n_batches, world):  # one update per global step with the mean gradient of its minibatches
        w -= 0.1 * torch.stack(grads[gs:gs + world]).mean(0)
    for r in range(world):
        assert np.allclose(out[r][0], w.numpy(), rtol=1e-6, atol=1e-7), f"rank {r} diverged from the single-process update"
        assert np.array_equal(out[r][1], 100 + np.arange(10))
    assert np.array_equal(out[0][0], out[1][0]), "replicas must stay bit-identical"
