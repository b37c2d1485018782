"""User-sharded propagation (diffmm_amd/shard.py).  CPU: the replicas' local blocks reassemble the full normalised
adjacency exactly, the user ranges are a partition balanced by edges, and two gloo processes reproduce the unsharded A X from local
products + the item-row all-reduce (scipy does the local product here: the exchange logic is what is under test).  GPU: the same
through the HIP SpMM."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from diffmm_amd import graph, shard, synth


def _full_dense(g):
    rp, col, val = graph.normalized_bipartite_csr(g.trn_rows, g.trn_cols, g.users, g.items)
    n = g.users + g.items
    return sp.csr_matrix((val, col, rp), shape=(n, n)).toarray()


@pytest.mark.parametrize("name,world", [("toy", 1), ("toy", 2), ("mid", 3), ("mid", 5)])
def test_blocks_reassemble_the_full_matrix(name, world):
    g = synth.named(name)
    U, I = g.users, g.items
    full = _full_dense(g)
    ranges = shard.user_ranges(g.trn_rows, U, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == U and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), "contiguous partition of the users"
    deg = np.bincount(g.trn_rows, minlength=U)
    per = [int(deg[a:b].sum()) for a, b in ranges]
    assert max(per) - min(per) <= 2 * int(deg.max()) + 1, f"edges per replica {per}"
    acc = np.zeros_like(full)
    seen = np.zeros(full.shape, dtype=np.int32)
    for r, (u0, u1) in enumerate(ranges):
        rp, col, val = shard.local_block_csr(g.trn_rows, g.trn_cols, U, I, u0, u1, r, world)
        nu = u1 - u0
        loc = sp.csr_matrix((val, col, rp), shape=(nu + I, nu + I)).tocoo()
        to_global = np.concatenate([np.arange(u0, u1), U + np.arange(I)])
        gr, gc = to_global[loc.row], to_global[loc.col]
        acc[gr, gc] += loc.data
        seen[gr, gc] += 1
        assert (np.diff(rp) >= 0).all() and all((np.diff(col[rp[i]:rp[i + 1]]) > 0).all() for i in range(nu + I)), "columns ascend inside a row"
    assert seen.max() == 1, "every entry of the full matrix lives on exactly one replica"
    assert np.array_equal(acc, full), "same values as the unsharded builder, bit for bit"


def _gloo_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.named("mid")
        U, I, d = g.users, g.items, 8
        rng = np.random.default_rng(3)
        X = rng.standard_normal((U + I, d)).astype(np.float32)  # same on every rank
        u0, u1 = shard.user_ranges(g.trn_rows, U, world)[rank]
        rp, col, val = shard.local_block_csr(g.trn_rows, g.trn_cols, U, I, u0, u1, rank, world)
        nu = u1 - u0
        A = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(nu + I, nu + I))
        y = torch.from_numpy(A @ np.concatenate([X[u0:u1], X[U:]]).astype(np.float64))
        tdist.all_reduce(y[nu:])  # the exchange step of shard.sharded_spmm
        out[rank] = (u0, u1, y.numpy().copy())
    finally:
        tdist.destroy_process_group()


def test_two_replicas_reproduce_the_unsharded_product():
    world, port = 2, 31500 + (os.getpid() % 2000)
    out = mp.Manager().dict()
    mp.spawn(_gloo_worker, args=(world, port, out), nprocs=world, join=True)
    g = synth.named("mid")
    U, I = g.users, g.items
    X = np.random.default_rng(3).standard_normal((U + I, 8)).astype(np.float32)
    ref = _full_dense(g).astype(np.float64) @ X.astype(np.float64)
    for r in range(world):
        u0, u1, y = out[r]
        assert np.allclose(y[: u1 - u0], ref[u0:u1], rtol=1e-12, atol=1e-12), "user rows are complete on their owner"
        assert np.allclose(y[u1 - u0:], ref[U:], rtol=1e-12, atol=1e-12), "item rows after the all-reduce"


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,D", [("mid", 3, 16), ("tiktok", 4, 64)])
def test_sharded_spmm_matches_unsharded_on_the_gpu(name, world, D):
    """Every replica's block through the HIP SpMM (one after the other in this process), item partials summed like the all-reduce."""
    if not torch.cuda.is_available():
        pytest.fail("pytest -m gpu needs a GPU; none visible")
    from diffmm_amd import ops
    dev = torch.device("cuda:0")
    g = synth.named(name)
    U, I = g.users, g.items
    torch.manual_seed(1)
    X = torch.randn(U + I, D, device=dev)
    full = graph.make_adjacency(g.trn_rows, g.trn_cols, U, I, D, dev)
    ref = torch.empty(U + I, D, device=dev)
    ops.spmm_batched([ops.SpmmOp(full, X, y=ref)])
    items_sum = torch.zeros(I, D, device=dev, dtype=torch.float64)
    for r in range(world):
        sh = shard.make_user_shard(g.trn_rows, g.trn_cols, U, I, r, world, D, dev)
        y = torch.empty(sh.n_users + I, D, device=dev)
        shard.sharded_spmm(sh, X[sh.u0:sh.u1].contiguous(), X[U:].contiguous(), y, reduce=False)
        assert torch.allclose(y[: sh.n_users], ref[sh.u0:sh.u1], rtol=1e-5, atol=1e-6), "user rows need no exchange"
        items_sum += y[sh.n_users:].double()
    assert torch.allclose(items_sum.float(), ref[U:], rtol=1e-5, atol=1e-5), "item rows = sum of the replicas' partials"
