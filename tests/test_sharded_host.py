"""Host logic of the user-sharded layout (diffmm_amd/sharded.py) -- runs without a GPU:
  * ShardLayout: contiguous user ranges, ownership lookup (device op == numpy), item slices partition the items;
  * sort_batch_by_owner: a stable permutation of the minibatch whose segments are exactly each replica's triples;
  * the exchange algebra of one sharded SpMM launch + item-row epilogue, emulated with scipy over 2 gloo processes: own user rows need
    no communication, the all-reduced raw item partials followed by the epilogue reproduce the unsharded epilogue;
  * the row-split diffusion step: per-replica sums over disjoint rows, scaled by the GLOBAL batch, add up to the single-process mean."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from diffmm_amd import graph, shard, synth
from diffmm_amd.sharded import ShardLayout, sort_batch_by_owner


@pytest.mark.parametrize("name,world", [("toy", 1), ("toy", 3), ("mid", 2), ("mid", 8), ("tiktok", 8)])
def test_layout_partitions_users_and_items(name, world):
    g = synth.named(name)
    lays = [ShardLayout.make(g.trn_rows, g.users, g.items, world, r) for r in range(world)]
    assert lays[0].u0 == 0 and lays[-1].u1 == g.users
    assert all(a.u1 == b.u0 for a, b in zip(lays, lays[1:])) and sum(l.nu for l in lays) == g.users
    sl = [l.item_slice() for l in lays]
    assert sl[0][0] == 0 and sl[-1][1] == g.items and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
    users = torch.arange(g.users)
    own = lays[0].owner_of(users).numpy()
    assert np.array_equal(own, lays[0].owner_of_np(np.arange(g.users)))
    for r, l in enumerate(lays):
        assert (own[l.u0:l.u1] == r).all()
    x = torch.arange((g.users + g.items) * 2, dtype=torch.float32).view(-1, 2)
    loc = lays[-1].local_rows(x)
    assert loc.shape[0] == lays[-1].n_local and torch.equal(loc[lays[-1].nu:], x[g.users:]) and torch.equal(loc[:lays[-1].nu], x[lays[-1].u0:lays[-1].u1])


@pytest.mark.parametrize("world", [1, 2, 5])
def test_sort_batch_by_owner_is_a_stable_partition(world):
    g = synth.named("mid")
    lay = ShardLayout.make(g.trn_rows, g.users, g.items, world, 0)
    rng = np.random.default_rng(5)
    e = rng.integers(0, g.nnz, size=257)
    users, pos = torch.from_numpy(g.trn_rows[e].astype(np.int32)), torch.from_numpy(g.trn_cols[e].astype(np.int32))
    neg = torch.from_numpy(rng.integers(0, g.items, size=257).astype(np.int32))
    su, sp_, sn, bounds = sort_batch_by_owner(lay, users, pos, neg)
    b = bounds.numpy()
    assert b[0] == 0 and b[-1] == 257 and (np.diff(b) >= 0).all()
    trip = sorted(zip(users.tolist(), pos.tolist(), neg.tolist()))
    assert sorted(zip(su.tolist(), sp_.tolist(), sn.tolist())) == trip, "a permutation of the triples"
    own = lay.owner_of_np(su.numpy())
    for r in range(world):
        assert (own[b[r]:b[r + 1]] == r).all(), "segment r holds exactly replica r's triples"
    # stable: inside a segment the triples keep their order of appearance in the batch
    orig_pos = {}
    for i, t in enumerate(zip(users.tolist(), pos.tolist(), neg.tolist())):
        orig_pos.setdefault(t, []).append(i)
    last = -1
    for r in range(world):
        last = -1
        seen = {}
        for t in zip(su[b[r]:b[r + 1]].tolist(), sp_[b[r]:b[r + 1]].tolist(), sn[b[r]:b[r + 1]].tolist()):
            k = seen.get(t, 0)
            seen[t] = k + 1
            assert orig_pos[t][k] > last
            last = orig_pos[t][k]


def _epilogue(y, c1, noise, alpha, beta, nd):
    """y = alpha acc + beta c1 ; y_pert = y + sign(y) * normalize(noise) * nd  (include/diffmm_hip.h: dmm_spmm_problem)"""
    y = alpha * y + beta * c1
    nrm = np.maximum(np.linalg.norm(noise, axis=1, keepdims=True), 1e-12)
    return y, y + np.sign(y) * noise / nrm * nd


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.named("mid")
        U, I, d = g.users, g.items, 8
        rng = np.random.default_rng(3)
        X, C1, NZ = (rng.standard_normal((U + I, d)) for _ in range(3))  # the same on every rank (seeded)
        lay = ShardLayout.make(g.trn_rows, U, I, world, rank)
        rp, col, val = shard.local_block_csr(g.trn_rows, g.trn_cols, U, I, lay.u0, lay.u1, rank, world)
        nu = lay.nu
        A = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(nu + I, nu + I))
        loc = lambda t: np.concatenate([t[lay.u0:lay.u1], t[U:]])
        acc = A @ loc(X)
        # own user rows: finished in place; item rows: raw partials -> all-reduce -> epilogue
        yu, ypu = _epilogue(acc[:nu], loc(C1)[:nu], loc(NZ)[:nu], 1.5, 0.5, 0.7)
        raw = torch.from_numpy(acc[nu:].copy())
        tdist.all_reduce(raw)
        yi, ypi = _epilogue(raw.numpy(), C1[U:], NZ[U:], 1.5, 0.5, 0.7)
        # row-split diffusion step: loss sums over the own rows of a global batch, gradient scaled by 1 / (global rows * total)
        batch = np.random.default_rng(9).permutation(U)[:64]
        own = batch[(batch >= lay.u0) & (batch < lay.u1)]
        per_row = np.square(X[own]).sum(1)  # stand-in for the per-row loss
        s = torch.tensor([per_row.sum()], dtype=torch.float64)
        tdist.all_reduce(s)
        mean = float(s) / len(batch)
        gpart = torch.from_numpy((2 * X[own]).sum(0) / (len(batch) * mean))
        tdist.all_reduce(gpart)
        out[rank] = (lay.u0, lay.u1, yu, ypu, yi, ypi, mean, gpart.numpy())
    finally:
        tdist.destroy_process_group()


def test_two_replicas_reproduce_epilogue_and_row_split_step():
    world, port = 2, 32500 + (os.getpid() % 2000)
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g = synth.named("mid")
    U, I, d = g.users, g.items, 8
    rng = np.random.default_rng(3)
    X, C1, NZ = (rng.standard_normal((U + I, d)) for _ in range(3))
    rp, col, val = graph.normalized_bipartite_csr(g.trn_rows, g.trn_cols, U, I)
    A = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(U + I, U + I))
    y, yp = _epilogue(A @ X, C1, NZ, 1.5, 0.5, 0.7)
    batch = np.random.default_rng(9).permutation(U)[:64]
    mean = np.square(X[batch]).sum(1).mean()
    grad = (2 * X[batch]).sum(0) / (len(batch) * mean)
    for r in range(world):
        u0, u1, yu, ypu, yi, ypi, m, gsum = out[r]
        assert np.allclose(yu, y[u0:u1], rtol=1e-12, atol=1e-12) and np.allclose(ypu, yp[u0:u1], rtol=1e-12, atol=1e-12)
        assert np.allclose(yi, y[U:], rtol=1e-12, atol=1e-12) and np.allclose(ypi, yp[U:], rtol=1e-12, atol=1e-12)
        assert abs(m - mean) <= 1e-12 * mean and np.allclose(gsum, grad, rtol=1e-12, atol=1e-12)


def test_item_shard_slices_partition_the_items():
    """ItemShard (item-sharded Denoise): slices of S = ceil(I / world) items in rank order, the last one shorter, never empty; item i sits in row i
    of the operands assembled from the replicas' S-row blocks."""
    from diffmm_amd.engine import ItemShard
    for I, W in ((6710, 8), (18357, 2), (1000000, 8), (17, 3), (9, 2)):
        sl = [ItemShard(r, W, I) for r in range(W)]
        assert sl[0].i0 == 0 and sl[-1].i1 == I and all(a.i1 == b.i0 for a, b in zip(sl, sl[1:]))
        assert all(0 < s.items_local <= s.S for s in sl) and all(s.i0 == s.rank * s.S for s in sl)
    with pytest.raises(ValueError):
        ItemShard(0, 8, 9)  # ceil(9 / 8) = 2 items per slice leave the last replicas without items


# ---- Coach.check_replicas: a first mismatch is healed from replica 0, a repeat raises -------------------------------------------------------
def _stub_coach(rank, world):
    """A Coach with just the state check_replicas reads (flat parameter sets on the CPU)."""
    from types import SimpleNamespace
    from diffmm_amd.Main import Coach
    from diffmm_amd.engine import FlatParams
    torch.manual_seed(5)  # the same "replicated" values on every rank
    JP = FlatParams([("emb", (6 + 4, 4)), ("image_layer.weight", (4, 3)), ("modal_weight", (3,))], "cpu")
    DP = [FlatParams([("emb_layer.weight", (3, 3)), ("out_layers.0.bias", (5,)), ("in_layers.0.weight", (8, 8))], "cpu") for _ in range(2)]
    for P in [JP] + DP:
        P.data.normal_(); P.m.normal_(); P.v.uniform_()
    JP.data[:6 * 4] += rank  # the own user rows legitimately differ
    c = Coach.__new__(Coach)
    c.rank, c.world, c.group, c.item_shard, c.epoch_idx = rank, world, None, None, 3
    c.model = SimpleNamespace(engine=SimpleNamespace(nu=6, D=4, P=JP, _emb16_step=7))
    c.diff_engine = SimpleNamespace(P=DP, M=2, n_replicated=0, w16_step=[1, 1], w16_train_step=[1, 1])
    return c, JP, DP


def _heal_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c, JP, DP = _stub_coach(rank, world)
        c.check_replicas()  # identical: nothing happens
        res = {"heals0": getattr(c, "replica_heals", 0)}
        want = (JP.data[24:].clone(), DP[1].v.clone())
        if rank == 1:  # one flipped low bit in an item row and one in a Denoise's Adam state
            JP.data[24 + 5] = torch.nextafter(JP.data[24 + 5], torch.tensor(9.0))
            DP[1].v[3] = torch.nextafter(DP[1].v[3], torch.tensor(9.0))
        c.check_replicas()
        res["heals1"] = c.replica_heals
        res["restored"] = bool(torch.equal(JP.data[24:], want[0]) and torch.equal(DP[1].v, want[1]))
        res["own_rows_kept"] = bool((JP.data[:24] - rank).abs().max() < 10)  # untouched by the broadcast (not a replicated slice)
        res["stale_marks"] = (c.diff_engine.w16_step, c.model.engine._emb16_step)
        if rank == 1:
            DP[0].data[1] += 1.0
        try:
            c.check_replicas()
            res["second"] = "no error"
        except RuntimeError as e:
            res["second"] = str(e)
        out[rank] = res
    finally:
        tdist.destroy_process_group()


def test_replica_mismatch_is_healed_once_then_raises():
    world, port = 2, 34500 + (os.getpid() % 2000)
    out = mp.Manager().dict()
    mp.spawn(_heal_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        res = out[r]
        assert res["heals0"] == 0 and res["heals1"] == 1 and res["restored"] and res["own_rows_kept"]
        assert res["stale_marks"] == ([-1, -1], -1)
        assert "again, right after a re-broadcast" in res["second"], "a mismatch at the very next check raises on every replica"


def test_checkpoint_layout_is_versioned_and_remapped_by_name():
    """the flat parameter layout changed between builds; a checkpoint carries its ordered (name, shape, offset) list and is
    remapped tensor by tensor when the order differs, refused when it has no layout record or other tensors -- never read positionally."""
    from diffmm_amd.Main import restore_flat
    from diffmm_amd.engine import FlatParams
    new = [("emb_layer.weight", (3, 3)), ("gate_layer.weight", (4, 4)), ("out_layers.0.bias", (7,)), ("in_layers.0.weight", (5, 12)), ("out_layers.0.weight", (7, 5))]
    old = [new[0], new[2], new[1], new[4], new[3]]  # the previous order: b2 before the gate layer, W2 before W1
    src = FlatParams(old, "cpu")
    src.data.normal_(); src.m.normal_(); src.v.uniform_(); src.step = 11
    rec = {"data": src.data.clone(), "m": src.m.clone(), "v": src.v.clone(), "step": 11, "layout": src.signature()}
    dst = FlatParams(new, "cpu")
    assert dst.size == src.size, "same element count: a numel check alone would have let the scrambled load through"
    restore_flat(dst, rec, "test")
    for k in src.names:
        for a, b in ((src.data, dst.data), (src.m, dst.m), (src.v, dst.v)):
            assert torch.equal(src.view(k, a), dst.view(k, b)), k
    assert dst.step == 11
    same = FlatParams(old, "cpu")
    restore_flat(same, rec, "test")
    assert torch.equal(same.data, src.data)
    with pytest.raises(ValueError, match="no layout record"):
        restore_flat(dst, {k: v for k, v in rec.items() if k != "layout"}, "test")
    other = FlatParams(new[:-1] + [("out_layers.0.weight", (5, 7))], "cpu")
    with pytest.raises(ValueError, match="do not match"):
        restore_flat(other, rec, "test")


def test_denoise_shard_mode_is_chosen_at_the_documented_threshold(monkeypatch):
    """Coach._denoise_item_shard: `auto` splits the Denoise MLPs by ITEMS once a modality's weight gradient (2 I H fp32) reaches 256 MiB --
    I >= 32768 at H = 1024 -- and by minibatch ROWS below; `rows` / `items` force it; the item split is refused where its kernels do not apply."""
    from diffmm_amd.Conf import Config
    from diffmm_amd.Main import Coach
    from diffmm_amd.engine import ItemShard

    def pick(I, world=8, H=1024, hidden=(1024,), d=64, precision=1, sstep=0, rank=3):
        c = Coach.__new__(Coach)
        c.config = Config()
        c.config.data.item_num, c.config.base.latdim, c.config.hyper.sampling_step = I, d, sstep
        c.world, c.rank, c.group, c.precision = world, rank, None, precision
        return c._denoise_item_shard(H, list(hidden))

    monkeypatch.delenv("DMM_DENOISE_SHARD", raising=False)
    monkeypatch.delenv("DMM_CHAIN_MODE", raising=False)
    monkeypatch.delenv("DMM_FORCE_COLLECTIVES", raising=False)
    assert pick(32767) is None and pick(6710) is None and pick(18357) is None, "every shipped dataset: row split"
    sh = pick(32768)
    assert isinstance(sh, ItemShard) and (sh.rank, sh.world, sh.items_global, sh.S) == (3, 8, 32768, 4096)
    assert isinstance(pick(1_000_000, d=256), ItemShard), "config 5"
    assert pick(1_000_000, world=1) is None, "a single GPU has nothing to split"
    for kw in (dict(hidden=(512, 1024)), dict(d=96), dict(precision=3), dict(sstep=2)):
        assert pick(1_000_000, **kw) is None, f"auto falls back to rows where the item-split step does not apply: {kw}"
    monkeypatch.setenv("DMM_CHAIN_MODE", "stepwise")
    assert pick(1_000_000) is None, "the item-sharded rebuild exists for the latent chain only (ADVICE r3)"
    monkeypatch.delenv("DMM_CHAIN_MODE")
    monkeypatch.setenv("DMM_DENOISE_SHARD", "rows")
    assert pick(1_000_000) is None
    monkeypatch.setenv("DMM_DENOISE_SHARD", "items")
    assert isinstance(pick(6710), ItemShard)
    with pytest.raises(ValueError):
        pick(6710, precision=3)
    with pytest.raises(ValueError):
        pick(9, world=8)  # ceil(9 / 8) = 2 items per slice would leave the last replicas without items
