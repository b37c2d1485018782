"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol of include/diffmm_hip.h,
argument validation works without a GPU, and the SpMM planner's task list reproduces A @ X exactly when executed
by a numpy emulation of the kernel's work decomposition."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from diffmm_amd import _lib as L
    hdr = open(os.path.join(REPO, "include", "diffmm_hip.h")).read()
    declared = set(re.findall(r"\b(dmm_[a-z0-9_]+)\s*\(", hdr))
    lib = L.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libdiffmm_hip.so does not export {name}"
    assert declared == set(L.exported_symbols()), f"ctypes table and header disagree: {declared ^ set(L.exported_symbols())}"
    assert lib.dmm_version() == 400


def test_argument_validation_without_gpu():
    import ctypes as C
    from diffmm_amd import _lib as L
    lib = L.lib()
    assert lib.dmm_gemm(C.byref(L.GemmArgs()), None) == -1 and b"null operand" in lib.dmm_last_error_string()
    assert lib.dmm_spmm_tasks_per_block(64) == 16 and lib.dmm_spmm_tasks_per_block(48) == -1
    p = (L.SpmmProblem * 1)()
    assert lib.dmm_spmm_csr_f32(p, 1, 48, None) == -1
    assert lib.dmm_spmm_csr_f32(p, 9, 64, None) == -1


def test_no_cpu_fallback():
    import torch
    from diffmm_amd import ops, graph, synth
    g = synth.named("toy")
    adj = graph.make_adjacency(g.trn_rows, g.trn_cols, g.users, g.items, 8, "cpu")
    with pytest.raises(RuntimeError, match="device tensors only"):
        ops.spmm_batched([ops.SpmmOp(adj, torch.zeros(adj.n, 8), y=torch.zeros(adj.n, 8))])


def _emulate(plan, col, val, X, G):
    """numpy model of spmm_kernel + spmm_fixup_kernel: per-task partial rows, runs combined in task order by the run head;
    the first n_partial_blocks workgroups park their sums, the fix list adds them per giant row."""
    tasks = plan.tasks
    n, D = X.shape
    Y = np.full((n, D), np.nan, dtype=np.float64)
    part = np.zeros((tasks.shape[0], D))
    slots = np.zeros((plan.n_partial_blocks, D))
    for i, (row, b, e, run) in enumerate(tasks):
        if row >= 0:
            part[i] = (val[b:e, None].astype(np.float64) * X[col[b:e]]).sum(0)
    for i, (row, b, e, run) in enumerate(tasks):
        if row >= 0 and (run >> 16) == 0:
            ln = run & 0xFFFF
            assert (i % G) + ln <= G
            if i // G < plan.n_partial_blocks:
                assert i % G == 0
                slots[i // G] = part[i:i + ln].sum(0)
                continue
            assert np.isnan(Y[row]).all(), "row written twice"
            Y[row] = part[i:i + ln].sum(0)
    for row, first, cnt, _ in plan.fix:
        assert np.isnan(Y[row]).all()
        Y[row] = slots[first:first + cnt].sum(0)
    return Y


@pytest.mark.parametrize("name,D,task_len", [("toy", 8, 2), ("mid", 16, 4), ("mid", 64, 8), ("mid", 256, 4), ("tiktok", 64, 32)])
def test_planner_tiles_every_row(name, D, task_len):
    from diffmm_amd import graph, synth
    g = synth.named(name)
    rows, cols = g.trn_rows, g.trn_cols
    if name == "mid":  # one item every user interacted with: a giant row that must be cut into segments
        rows, cols = np.concatenate([rows, np.arange(g.users)]), np.concatenate([cols, np.zeros(g.users, dtype=cols.dtype)])
    rowptr, col, val = graph.normalized_bipartite_csr(rows, cols, g.users, g.items)
    plan = graph.plan_tasks(rowptr, D, task_len, full=True)
    tasks = plan.tasks
    graph.check_plan(rowptr, tasks, D)
    n_giant = int((np.diff(rowptr) > 2 * (1024 // D) * task_len).sum())
    assert len(plan.fix) == n_giant and (plan.n_partial_blocks >= 2 * n_giant)
    if name == "mid" and D >= 64:
        assert n_giant >= 1, "the all-users item must be a giant row at these task sizes"
    n = len(rowptr) - 1
    X = np.random.default_rng(0).standard_normal((n, 4))
    Y = _emulate(plan, col, val, X, 1024 // D)
    dense = np.zeros((n, n)); dense[np.repeat(np.arange(n), np.diff(rowptr)), col] = val
    assert np.allclose(Y, dense @ X, rtol=1e-12, atol=1e-12)
    if name == "tiktok":
        assert plan.n_partial_blocks == 0
        live = tasks[tasks[:, 0] >= 0]
        assert (live[:, 2] - live[:, 1]).max() <= 64, "no task longer than 2 x task_len on the TikTok degree law"
        assert tasks.shape[0] < 1.4 * n, "padding overhead stays small"


def test_adjacency_matches_oracle_restatement():
    from diffmm_amd import graph, synth
    from oracle import diffmm_oracle as O
    for name in ("toy", "mid", "tiktok"):
        g = synth.named(name)
        a = graph.normalized_bipartite_csr(g.trn_rows, g.trn_cols, g.users, g.items)
        b = O.build_adj_csr(g.trn_rows, g.trn_cols, g.users, g.items)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    # duplicates collapse (the reference binarises)
    r = np.array([0, 0, 1]); c = np.array([1, 1, 0])
    a = graph.normalized_bipartite_csr(r, c, 2, 2)
    assert a[0][-1] == 4 + 2 * 2


def test_schedule_tables_match_reference_vectors():
    from diffmm_amd.engine import schedule_tables
    z = np.load(os.path.join(REPO, "tests", "golden", "sched.npz"))
    for Tn in (5, 40, 200, 1000):
        for sc in (0.1, 0.5):
            s = schedule_tables(Tn, sc, 0.0001, 0.02)
            for k in s:
                assert np.allclose(s[k], z[f"T{Tn}_s{sc}/{k}"], rtol=1e-12, atol=1e-15), (Tn, sc, k)


def test_adjacency_uid_is_never_reused():
    """Launch plans are keyed by Adjacency.uid.  id() of a collected object is handed out again by CPython (that is how a plan
    recorded for the graphs of epoch e was once replayed, stale pointers and task counts included, on the graphs of epoch e+2)."""
    import gc
    import torch
    from diffmm_amd import graph
    z = torch.zeros(1, dtype=torch.int32)
    mk = lambda: graph.Adjacency(n=1, nnz=1, D=64, rowptr=z, col=z, val=z.float(), tasks=torch.zeros((1, 4), dtype=torch.int32))
    seen, ids = set(), set()
    for _ in range(200):
        a = mk()
        assert a.uid not in seen
        seen.add(a.uid); ids.add(id(a))
        del a
        gc.collect()
    assert len(ids) < 200, "this interpreter did reuse id() values, as the docstring says"


def test_shipped_configs_load():
    """conf/*.toml parse with the reference's strict dataclass loader (unknown keys raise, Conf.py:68-77)."""
    from diffmm_amd.Conf import load_config
    for name, M in (("tiktok", 3), ("baby", 2), ("sports", 2)):
        cfg = load_config(os.path.join(REPO, "conf", f"{name}.toml"))
        assert cfg.data.name == name and cfg.base.latdim == 64 and cfg.hyper.steps == 5
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".toml", delete=False) as fh:
        fh.write("[hyper]\ne_loss = 0.01\n")  # a legacy key of the reference's own baby.toml
    with pytest.raises(TypeError):
        load_config(fh.name)
    os.unlink(fh.name)


def test_bundle_round_trip_and_reference_feature_files(tmp_path):
    """the .npz bundle writer / reader (what DataHandler.LoadData picks up) round-trips a data set exactly (fp32 features) or to
    fp16 (the reference's on-disk precision); the reference's own .npy feature files load with allow_pickle=False.  (Its interaction
    matrices are pickles and are never opened.)"""
    import os
    from diffmm_amd import bundle, synth
    from diffmm_amd.Conf import load_config
    from diffmm_amd.DataHandler import DataHandler
    g = synth.named("mid")
    p = bundle.write_bundle(str(tmp_path / "Datasets" / "mid" / "mid.npz"), g.users, g.items, g.trn_rows, g.trn_cols, g.tst_rows, g.tst_cols, g.feats,
                            feat_dtype=np.float32)
    r = bundle.read_bundle(p)
    assert (r.users, r.items) == (g.users, g.items)
    for a, b in ((r.trn_rows, g.trn_rows), (r.trn_cols, g.trn_cols), (r.tst_rows, g.tst_rows), (r.tst_cols, g.tst_cols)):
        assert a.dtype == np.int32 and np.array_equal(a, b)
    assert len(r.feats) == len(g.feats) and all(np.array_equal(a, b) for a, b in zip(r.feats, g.feats))
    p16 = bundle.write_bundle(str(tmp_path / "h.npz"), g.users, g.items, g.trn_rows, g.trn_cols, g.tst_rows, g.tst_cols, g.feats)
    r16 = bundle.read_bundle(p16)
    assert all(np.array_equal(a, b.astype(np.float16).astype(np.float32)) for a, b in zip(r16.feats, g.feats))
    with pytest.raises(ValueError):
        bundle.write_bundle(str(tmp_path / "bad.npz"), g.users, g.items, g.trn_rows + g.users, g.trn_cols, g.tst_rows, g.tst_cols, g.feats)
    # the CLI writes the same file; DataHandler reads it from ./Datasets/<name>/<name>.npz
    bundle.main(["--synth", "toy", "--out", str(tmp_path / "Datasets" / "toy" / "toy.npz")])
    cfg = load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf", "tiktok.toml"))
    cfg.data.name, cfg.train.batch = "toy", 8
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        h = DataHandler(cfg)
        t = synth.named("toy")
        got = h._read()
        assert np.array_equal(got.trn_rows, t.trn_rows) and np.array_equal(got.tst_cols, t.tst_cols) and got.feats[0].shape == t.feats[0].shape
    finally:
        os.chdir(cwd)
    ref = "/root/reference/Datasets/tiktok/image_feat.npy"  # present in the build container only
    if os.path.exists(ref):
        f = bundle.load_feature_npy(ref)
        assert f.dtype == np.float32 and f.shape == (6710, 128) and np.isfinite(f).all()


def test_hub_ordered_plan_is_a_permutation_of_the_row_ordered_plan():
    """graph.hub_keys (a row's highest-degree neighbour) and plan_tasks(row_key=): the same tasks in another order -- every plan the checker accepts
    is valid for the kernel, which reads its row from the task."""
    from diffmm_amd import graph, synth
    g = synth.named("mid")
    rp, col, _ = graph.normalized_bipartite_csr(g.trn_rows, g.trn_cols, g.users, g.items)
    key = graph.hub_keys(rp, col)
    deg = np.diff(rp)
    for r in range(0, len(deg), 5):
        cs = col[rp[r]:rp[r + 1]]
        cs = cs[cs != r]
        want = r if cs.size == 0 else int(cs[deg[cs] == deg[cs].max()].max())
        assert key[r] == want
    for D, tl in ((64, 16), (32, 8)):
        p0, p1 = graph.plan_tasks(rp, D, tl, full=True), graph.plan_tasks(rp, D, tl, full=True, row_key=key)
        graph.check_plan(rp, p1.tasks, D)
        live = lambda p: sorted(map(tuple, p.tasks[p.tasks[:, 0] >= 0].tolist()))
        assert live(p0) == live(p1) and p0.head == p1.head and p0.n_partial_blocks == p1.n_partial_blocks


def test_local_comm_keeps_the_own_part():
    """comm.LOCAL (one process running a single replica of a larger job): reductions are the identity, a gather fills the replica's own block."""
    import torch
    from diffmm_amd import comm
    t = torch.arange(6.0)
    comm.all_reduce(t, group=comm.LOCAL)
    comm.all_reduce(t, group=comm.LOCAL, op="MAX")
    assert torch.equal(t, torch.arange(6.0)) and comm.all_reduce(t, group=comm.LOCAL, async_op=True).wait()
    out = torch.zeros(6, 2)
    comm.all_gather_into(out, torch.ones(2, 2), rank=1, group=comm.LOCAL)
    assert torch.equal(out, torch.tensor([[0, 0], [0, 0], [1, 1], [1, 1], [0, 0], [0, 0]], dtype=torch.float32))


# ---- the reference's positional constructors of the three dataset classes (DataHandler.py:145-228) ------------------------------
def _mid_mats():
    import scipy.sparse as sp
    z = np.load(os.path.join(REPO, "tests", "golden", "epoch_mid.npz"))
    U, I = int(z["users"]), int(z["items"])
    trn = sp.coo_matrix((np.ones(len(z["trn_rows"]), np.float32), (z["trn_rows"], z["trn_cols"])), shape=(U, I))
    tst = sp.coo_matrix((np.ones(len(z["tst_rows"]), np.float32), (z["tst_rows"], z["tst_cols"])), shape=(U, I))
    return z, U, I, trn, tst


def test_dataset_classes_take_the_reference_positional_forms():
    """TrainData(coomat, config), TestData(testMat, trainMat), DiffusionData(data, config) -- the reference's signatures -- build the same
    objects as the array forms LoadData() uses."""
    import torch
    from diffmm_amd.Conf import Config
    from diffmm_amd.DataHandler import DiffusionData, TestData, TrainData
    z, U, I, trn, tst = _mid_mats()
    cfg = Config()
    cfg.data.user_num, cfg.data.item_num = U, I
    a, b = TrainData(trn, cfg), TrainData(z["trn_rows"], z["trn_cols"], cfg, "cpu")
    assert np.array_equal(a.rows, trn.row) and np.array_equal(a.cols, trn.col), "COO order is kept (DataHandler.py:150-151)"
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.cols, b.cols) and len(a) == len(z["trn_rows"])
    assert [list(x) for x in a.user_pos_items] == [list(x) for x in b.user_pos_items]
    a.negSampling()
    keys = set((z["trn_rows"].astype(np.int64) * I + z["trn_cols"]).tolist())
    assert not (set((a.rows.astype(np.int64) * I + a.negs).tolist()) & keys), "no sampled negative is a train positive"
    u, p, n = a[5]
    assert (u, p) == (trn.row[5], trn.col[5]) and n == a.negs[5]
    ta, tb = TestData(tst, trn), TestData(z["tst_rows"], z["tst_cols"], z["trn_rows"], z["trn_cols"], U, I)
    assert np.array_equal(ta.test_users, tb.test_users) and len(ta) == len(np.unique(z["tst_rows"]))
    assert ta.test_user_its == tb.test_user_its
    usr, mask = ta[3]
    want = np.asarray(trn.tocsr()[usr].todense()).reshape(-1)
    assert np.array_equal(mask, want), "TestData[i] = (user, the user's train row) (DataHandler.py:206-208)"
    dense = torch.tensor(trn.toarray(), dtype=torch.float)
    da, db, dc = DiffusionData(dense, cfg), DiffusionData(z["trn_rows"], z["trn_cols"], U, I), DiffusionData(trn, cfg)
    for d in (da, dc):
        assert len(d) == U and torch.equal(d.ptr.cpu(), db.ptr) and torch.equal(d.col.cpu(), db.col)
    if not torch.cuda.is_available():
        assert torch.equal(da.data, dense)
        row, idx = da[7]
        assert idx == 7 and torch.equal(row, dense[7])
    with pytest.raises(TypeError):
        TrainData(z["trn_rows"], z["trn_cols"])
    with pytest.raises(TypeError):
        TestData(tst, z["trn_rows"])
    with pytest.raises(TypeError):
        DiffusionData(z["trn_rows"], z["trn_cols"])


STALE_TOML = """
[base]
latdim = 32
trans = 1          # key of an older code base: Conf.py has no such field
cl_method = 1
[data]
name = "baby"
[hyper]
steps = 7
keepRate = 1
e_loss = 0.01
sampling_steps = 5
rebuild_k = 1
[train]
batch = 256
norm = false
sampling_noise = false
[legacy]
x = 1
"""


def test_tolerant_config_loading(tmp_path, caplog):
    """strict loading (the default) raises TypeError on unknown keys exactly like the reference's Conf.py:69-77 -- which is why
    the reference's own conf/baby.toml and conf/test.toml do not load --; strict=False drops them with ONE logged line."""
    import logging
    from diffmm_amd.Conf import load_config
    p = tmp_path / "stale.toml"
    p.write_text(STALE_TOML)
    with pytest.raises(TypeError, match="trans"):
        load_config(str(p))
    with caplog.at_level(logging.WARNING, logger="diffmm_amd.Conf"):
        cfg = load_config(str(p), strict=False)
    lines = [r.getMessage() for r in caplog.records if r.name == "diffmm_amd.Conf"]
    assert len(lines) == 1
    for k in ("base.trans", "hyper.keepRate", "hyper.e_loss", "hyper.sampling_steps", "hyper.rebuild_k", "train.norm", "train.sampling_noise", "[legacy]"):
        assert k in lines[0]
    assert (cfg.base.latdim, cfg.base.cl_method, cfg.data.name, cfg.hyper.steps, cfg.train.batch) == (32, 1, "baby", 7, 256)
    assert cfg.hyper.sampling_step == 0 and cfg.hyper.sim_weight == 0.1, "nothing is renamed or guessed: dropped keys leave the defaults"
    ref_conf = "/root/reference/conf"  # build container only: the reference's own files (never read on the GPU box -- this is a CPU test)
    if os.path.isdir(ref_conf):
        for f in sorted(os.listdir(ref_conf)):
            cfg = load_config(os.path.join(ref_conf, f), strict=False)
            assert cfg.data.name and cfg.hyper.steps >= 1
        for f in ("baby.toml", "test.toml"):
            with pytest.raises(TypeError):
                load_config(os.path.join(ref_conf, f))


def test_graft_entry_build_runs():
    """__graft_entry__.build() -- what the driver runs as its build check -- compiles (a no-op `make` when the library is current), loads the
    library and agrees with include/diffmm_hip.h on the version (it once kept a stale literal after the header's version moved)."""
    import importlib
    import sys
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_chain_parts_says_which_modalities_a_launch_completes():
    """engine.chain_parts (what DiffusionEngine.rebuild_edges uses to put a finished modality's tail under the next launch of the persistent chain)."""
    from diffmm_amd.engine import chain_parts
    assert chain_parts(64, 9308, 3) == [1, 3]          # the bench shape: 37 groups per modality, 64 + 47
    assert chain_parts(64, 9308, 1) == []              # one launch
    assert chain_parts(0, 9308, 3) == []               # the persistent form cannot run
    assert chain_parts(64, 19445, 2) == [0, 1, 2]      # Baby shape: 76 groups per modality -- nothing complete after the first launch
    assert chain_parts(64, 256 * 64, 2) == [1, 2]      # a boundary: the first launch ends exactly with modality 0
    assert chain_parts(32, 256 * 16, 4) == [2, 4]
    for per, rows, m in ((64, 9308, 3), (40, 5000, 4), (8, 300, 3)):
        cp = chain_parts(per, rows, m)
        assert cp == sorted(cp) and (not cp or cp[-1] == m)
