"""End-to-end parity: diffmm_amd.Main.Coach.trainEpoch()/testEpoch() on the HIP path against full epochs of the
REFERENCE (tests/golden/epoch_*.npz: same graph, same initial weights, same batches, same random draws)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_kernels_gpu import T, close, load, need_gpu  # noqa: E402


def close_weights(a, b, n_steps, lr, what):
    """Weights after n Adam steps.  Adam moves every weight by ~lr per step whatever the gradient's size, so an entry whose
    gradient is ~eps (1e-8) is ill-conditioned: bound those by the total possible travel, and require everything else tight."""
    a = a.detach().double().cpu().numpy(); b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    ok = err <= 3e-4 + 1e-3 * np.abs(b)
    assert ok.mean() >= 0.995, f"{what}: only {ok.mean():.4f} of the entries within tolerance"
    assert err.max() <= 2.0 * lr * n_steps + 1e-6, f"{what}: max abs err {err.max():.3e} exceeds the Adam travel bound"


class _Draws:
    """Replays the recorded / re-generated draws in the reference's call order (same logic as oracle.Draws)."""

    def __init__(self, z):
        self.stored = [z[f"draw/{i}"] for i in range(int(z["n_draws"]))] if "draw/0" in z.files else None
        self.gen = torch.Generator().manual_seed(int(z["draw_seed"]))
        self.i, self.checks = 0, []

    def _next(self, make):
        r = torch.from_numpy(np.asarray(self.stored[self.i])) if self.stored is not None else make()
        self.i += 1
        self.checks.append(float(r.double().sum()))
        return r

    def randint(self, hi, n):
        return self._next(lambda: torch.randint(0, hi, (n,), generator=self.gen))

    def randn(self, shape):
        return self._next(lambda: torch.randn(shape, generator=self.gen, dtype=torch.float32))

    def rand(self, shape):
        return self._next(lambda: torch.rand(shape, generator=self.gen, dtype=torch.float32))


def _coach(z):
    from diffmm_amd import synth
    from diffmm_amd.Conf import Config
    from diffmm_amd.DataHandler import DataHandler
    from diffmm_amd.Main import Coach
    M = int(z["M"])
    cfg = Config()
    cfg.data.name = "tiktok" if M == 3 else "sports"
    cfg.base.latdim, cfg.base.denoise_dim, cfg.base.topk, cfg.base.cl_method = int(z["d"]), f"[{int(z['H'])}]", int(z["topk"]), int(z["cl_method"])
    h_ = cfg.hyper
    h_.steps, h_.noise_scale, h_.noise_degree, h_.cross_cl_rate, h_.cross_cl_temp = int(z["T"]), 0.5, 1.5, 0.5, 0.2
    h_.modal_cl_rate, h_.modal_cl_temp, h_.sim_weight, h_.residual_weight, h_.modal_adj_weight = 0.01, 0.5, 0.01, 0.5, 0.2
    t_ = cfg.train
    t_.batch, t_.test_batch, t_.reg, t_.lr, t_.epoch = int(z["B"]), 64, float(z["reg"]), float(z["lr"]), int(z["train_epochs_cfg"])
    g = synth.SynthGraph(int(z["users"]), int(z["items"]), z["trn_rows"], z["trn_cols"], z["tst_rows"], z["tst_cols"], [z[f"feat{m}"] for m in range(M)])
    h = DataHandler(cfg)
    h.LoadData(graph_data=g)
    h.trainLoader.fixed_perm = T(z["train_perm"].astype(np.int64))
    h.diffusionLoader.fixed_perm = T(z["diff_perm"].astype(np.int64))
    coach = Coach(h, cfg, precision=3)
    coach.prepareModel()
    coach.model.engine.load_reference_state({k[len("init/model/"):]: z[k] for k in z.files if k.startswith("init/model/")})
    for m in range(M):
        coach.diff_engine.load_reference_state(m, {k[len(f"init/den{m}/"):]: z[k] for k in z.files if k.startswith(f"init/den{m}/")})
    return coach, h, M


@pytest.mark.parametrize("fx", ["epoch_toy", "epoch_mid", "epoch_mid_cl1", "epoch_mid_M2"])
def test_epochs_match_reference(fx):
    need_gpu()
    z = load(fx + ".npz")
    coach, h, M = _coach(z)
    draws = _Draws(z)
    for ep in range(int(z["epochs"])):
        h.trainData.set_negs(z[f"ep{ep}/negs"])
        res = coach.trainEpoch(draws=draws)
        # rebuilt modality graphs: the same edge set as the reference's python top-k loop
        same_edges = True
        for m, adj in enumerate([coach.image_adj, coach.text_adj] + ([coach.audio_adj] if M == 3 else [])):
            rp, col = adj.rowptr.cpu().numpy(), adj.col.cpu().numpy()
            row = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
            ref = set(zip(z[f"ep{ep}/adj{m}_row"].tolist(), z[f"ep{ep}/adj{m}_col"].tolist()))
            got = set(zip(row.tolist(), col.tolist()))
            inter = len(ref & got) / max(1, len(ref))
            assert inter >= 0.995, f"epoch {ep} modality {m}: only {inter:.4f} of the rebuilt edges agree"
            same_edges &= ref == got
        tol = dict(rtol=2e-4, atol=1e-6) if same_edges else dict(rtol=5e-3, atol=1e-5)
        for k, v in res.items():
            close(v, z[f"ep{ep}/train/{k}"], what=f"ep{ep} {k}", **tol)
        tres = coach.testEpoch()
        if same_edges:
            for k, v in tres.items():
                close(v, z[f"ep{ep}/test/{k}"], rtol=0.02, atol=2e-3, what=f"ep{ep} test {k}")
            B, lr = int(z["B"]), float(z["lr"])
            n_joint = (ep + 1) * -(-len(z["trn_rows"]) // B)
            n_diff = (ep + 1) * -(-int(z["users"]) // B)
            for k, p in coach.model.state_dict().items():
                close_weights(p, z[f"ep{ep}/model/{k}"], n_joint, lr, f"ep{ep} model {k}")
            for m in range(M):
                for k, p in coach.diff_engine.state_dict(m).items():
                    close_weights(p, z[f"ep{ep}/den{m}/{k}"], n_diff, lr, f"ep{ep} den{m} {k}")
    assert draws.i == int(z["n_draws"])
    close(np.array(draws.checks), z["draw_checksum"], rtol=1e-12, what="the random draws must be the recorded ones")


def test_loader_prefetch_on_a_side_stream_reproduces_the_reference_epoch(monkeypatch):
    """DMM_LOADER_PREFETCH=1 (the joint phase's shuffled epoch gathered on a side stream under the rebuild; off by default, measured neutral) against the
    reference's own epoch: the same permutation from the same generator, handed to the loop through the loader -- losses as in the default path."""
    need_gpu()
    monkeypatch.setenv("DMM_LOADER_PREFETCH", "1")
    z = load("epoch_mid.npz")
    coach, h, M = _coach(z)
    draws = _Draws(z)
    for ep in range(int(z["epochs"])):
        h.trainData.set_negs(z[f"ep{ep}/negs"])
        res = coach.trainEpoch(draws=draws)
        assert getattr(h.trainLoader, "_prefetched", None) is None, "the loop must have consumed the prefetched epoch"
        for k, v in res.items():
            close(v, z[f"ep{ep}/train/{k}"], what=f"ep{ep} {k}", rtol=5e-3, atol=1e-5)


@pytest.mark.parametrize("precision", [1, 3])
def test_epoch_with_noised_chain_start(precision):
    """sampling_step > 0 (every modality's reverse chain starts from its own q_sample(x_0)): the epoch runs on both arithmetic modes
    and the rebuilt graphs keep every user's degree with distinct items (the structural contract of Main.py:215-230)."""
    need_gpu()
    z = load("epoch_mid.npz")
    coach, h, M = _coach(z)
    coach.precision = precision
    coach.config.hyper.sampling_step = 2
    res = coach.trainEpoch()
    assert all(np.isfinite(v) for v in res.values())
    deg = np.bincount(z["trn_rows"], minlength=int(z["users"]))
    U = int(z["users"])
    for adj in [coach.image_adj, coach.text_adj] + ([coach.audio_adj] if M == 3 else []):
        rp, col = adj.rowptr.cpu().numpy(), adj.col.cpu().numpy()
        for u in range(U):
            items = col[rp[u] + 1:rp[u + 1]]  # after the self loop
            assert len(items) == deg[u] and len(set(items.tolist())) == deg[u] and (items >= U).all()


@pytest.mark.parametrize("sampling_step", [0, 1])
def test_checkpoint_resume_continues_the_run(tmp_path, sampling_step):
    """Coach.save_checkpoint / load_checkpoint: a fresh Coach that loads the checkpoint written after epoch 1 produces the same epoch 2
    as the uninterrupted run (up to the summation order of the fp32 atomics in the training kernels).  sampling_step = 1 also exercises
    the Philox position of the noised chain starts (restored on a freshly built engine) and the best-metric bookkeeping."""
    need_gpu()
    z = load("epoch_mid.npz")

    def fresh():
        coach, h, M = _coach(z)
        coach.precision = 1
        coach.config.hyper.sampling_step = sampling_step
        h.trainLoader.fixed_perm = h.diffusionLoader.fixed_perm = None  # real shuffling: the loader generators are part of the state
        return coach

    a = fresh()
    torch.manual_seed(11); torch.cuda.manual_seed_all(11)
    a.trainEpoch()
    a.best, a.his_max = dict(Recall=0.25, NDCG=0.125, Precision=0.0625, epoch=0), [0.25, 0.125, 0.0625]
    ck = os.path.join(tmp_path, "ck.pt")
    a.save_checkpoint(ck)
    res_a = a.trainEpoch()
    b = fresh()
    b.load_checkpoint(ck)
    assert b.epoch_idx == 1 and b.best == a.best and b.his_max == a.his_max
    assert b.diff_engine._xT_calls == (a.M if sampling_step else 0), "Philox position of the noised chain starts"
    res_b = b.trainEpoch()
    # Two runs from one state differ by the summation order of fp32 atomics (measured: weights 1e-8 .. 3e-7, losses 1e-8 relative, identical
    # graphs in 5 runs of 6) -- unless that noise flips a near-tie of the rebuild's top-k, which one run in six showed: then a few EDGES differ
    # and the epoch's weights follow (measured 2.7e-4, losses 3e-5).  The bars admit such a flip, not a lost optimiser / generator state
    # (which moves the losses by per cent and the graphs wholesale).
    for k in res_a:
        assert abs(res_a[k] - res_b[k]) <= 1e-3 * abs(res_a[k]) + 1e-7, (k, res_a[k], res_b[k])
    for (na, pa), (nb, pb) in zip(a._flat_models(), b._flat_models()):
        assert pa.step == pb.step
        d = float((pa.data - pb.data).abs().max())
        assert d <= 2e-3, f"{na}: weights differ by {d:.3e} after the resumed epoch"
    for ea, eb in zip([a.image_adj, a.text_adj, a.audio_adj], [b.image_adj, b.text_adj, b.audio_adj]):
        def edge_keys(adj):  # (row, column) pairs as one integer each: an edge SET (a flipped edge shifts every later position of its row)
            counts = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
            rows = torch.repeat_interleave(torch.arange(counts.numel(), device=adj.col.device), counts)
            return rows * int(counts.numel()) + adj.col.long()
        ka, kb = edge_keys(ea), edge_keys(eb)
        assert ka.numel() == kb.numel()
        same = float(torch.isin(ka, kb).float().mean())
        assert same > 0.99, f"rebuilt graphs share {same:.4f} of their edges"


@pytest.mark.parametrize("fx", ["epoch_mid", "epoch_mid_M2"])
def test_first_epoch_in_bf16_mode_tracks_the_reference(fx):
    """The benchmark's arithmetic (precision 1: bf16 MFMA operands, fp32 accumulation) on the reference's own first epoch -- same
    graph, weights, batches and draws: losses within 1 %, and the rebuilt modality graphs keep >= 97 % of the reference's edges
    (the top-k of a bf16 reverse chain: near-ties at the k-th score flip)."""
    need_gpu()
    z = load(fx + ".npz")
    coach, h, M = _coach(z)
    coach.precision = 1
    h.trainData.set_negs(z["ep0/negs"])
    res = coach.trainEpoch(draws=_Draws(z))
    for k, v in res.items():
        ref = float(z[f"ep0/train/{k}"])
        assert abs(v - ref) <= 1e-2 * abs(ref) + 1e-6, f"{k}: {v} vs reference {ref}"
    for m, adj in enumerate([coach.image_adj, coach.text_adj] + ([coach.audio_adj] if M == 3 else [])):
        rp, col = adj.rowptr.cpu().numpy(), adj.col.cpu().numpy()
        row = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        ref = set(zip(z[f"ep0/adj{m}_row"].tolist(), z[f"ep0/adj{m}_col"].tolist()))
        inter = len(ref & set(zip(row.tolist(), col.tolist()))) / max(1, len(ref))
        print(f"{fx} modality {m}: {inter:.4f} of the reference's rebuilt edges")
        assert inter >= 0.97, f"modality {m}: only {inter:.4f} of the rebuilt edges agree"


@pytest.mark.parametrize("precision", [3, 1])
def test_paired_twelve_epochs_follow_the_reference(precision):
    """The PAIRED long run (tests/golden/epoch_paired.npz, written by make_golden.py importing the reference): 12 epochs of the
    reference at 2000 users x 1200 items (d = 32, H = 64, T = 10, batch 512: 4 diffusion + 11 joint iterations per epoch, rebuild every epoch) with every
    torch draw regenerated from its seed -- 684 draws replayed here in the reference's call order -- and the reference's own negatives per epoch.  No
    statistics over seeds are needed: the HIP path must follow THIS trajectory, epoch by epoch, through 12 rebuilds of the modality graphs.
    precision 3 (~fp32): every epoch's losses within 0.5 % (observed 0.07-0.11 %) and Recall@20 / NDCG@20 at the plateau (mean of epochs 8-11; mean of
    three runs, see below) within 0.003 (observed mean over 14 runs +0.0016 / +0.0003: inside north_star's +-0.002 on average, not run by run).  precision 1 (the benchmark's bf16 GEMMs):
    every epoch's losses within 1 % (north_star: "loss curves ... within a stated tolerance"), plateau Recall@20 within 0.01 -- observed +0.0067: bf16
    products flip near-ties of the per-user top-k in each of the 12 rebuilds, the rebuilt graphs drift apart and the paired trajectories decorrelate;
    on a metric of 1900 test users (0.0005 per user) that is 13 users.  +-0.002 is therefore NOT shown for the bf16 mode by this route;
    every epoch's metrics within 0.02 in both modes."""
    need_gpu()
    z = load("epoch_paired.npz")
    tol = 0.005 if precision == 3 else 0.01

    def one_run():
        coach, h, M = _coach(z)
        coach.precision = precision
        coach.model.precision = precision
        draws = _Draws(z)
        rec, ndcg, rows = [], [], []
        for ep in range(int(z["epochs"])):
            h.trainData.set_negs(z[f"ep{ep}/negs"])
            res = coach.trainEpoch(draws=draws)
            tres = coach.testEpoch()
            rows.append((ep, {k: (float(v), float(z[f"ep{ep}/train/{k}"])) for k, v in res.items()}, {k: (float(v), float(z[f"ep{ep}/test/{k}"])) for k, v in tres.items()}))
            for k in ("Loss", "BPR Loss", "CL loss"):
                got, want = float(res[k]), float(z[f"ep{ep}/train/{k}"])
                assert abs(got - want) <= tol * abs(want), f"precision {precision} epoch {ep} {k}: {got:.5f} vs the reference's {want:.5f}"
            got, want = float(res["reg loss"]), float(z[f"ep{ep}/train/reg loss"])
            assert abs(got - want) <= 2 * tol * abs(want), f"precision {precision} epoch {ep} reg loss: {got:.5f} vs {want:.5f}"
            for k, lim in (("Recall", 0.02), ("NDCG", 0.02)):
                assert abs(float(tres[k]) - float(z[f"ep{ep}/test/{k}"])) <= lim, f"precision {precision} epoch {ep} {k}: {float(tres[k]):.4f} vs {float(z[f'ep{ep}/test/{k}']):.4f}"
            rec.append((float(tres["Recall"]), float(z[f"ep{ep}/test/Recall"])))
            ndcg.append((float(tres["NDCG"]), float(z[f"ep{ep}/test/NDCG"])))
        assert draws.i == int(z["n_draws"])
        close(np.array(draws.checks), z["draw_checksum"], rtol=1e-12, what="the random draws must be the reference's")
        pr = np.mean([a for a, _ in rec[8:]]) - np.mean([b for _, b in rec[8:]])
        pn = np.mean([a for a, _ in ndcg[8:]]) - np.mean([b for _, b in ndcg[8:]])
        worst = max(abs(a - b) / abs(b) for _, tr, _ in rows for k, (a, b) in tr.items() if k in ("Loss", "BPR Loss", "CL loss"))
        print(f"precision {precision}: 12 paired epochs, worst per-epoch loss deviation {worst:.4%}; plateau Recall@20 {np.mean([a for a, _ in rec[8:]]):.4f} vs the reference's "
              f"{np.mean([b for _, b in rec[8:]]):.4f} (difference {pr:+.4f}), NDCG@20 difference {pn:+.4f}; per-epoch Recall differences "
              f"{[round(a - b, 4) for a, b in rec]}")
        return pr, pn

    # The HIP path is not bit-reproducible from run to run (float atomics in the scatter kernels: the order of the gradient sums differs), and 12 graph
    # rebuilds amplify that: tools/paired_spread.py, 6 + 8 runs on two boxes (profiles/r4_paired_run_to_run_spread.txt) -- precision 3: plateau Recall@20
    # difference +0.0003 .. +0.0026 (mean +0.0016, sd 0.0007; one more run in a test log +0.0031), precision 1: +0.0021 .. +0.0080 (mean +0.0050,
    # sd 0.0015).  Every run lands ABOVE the reference's trajectory: once the paired trajectories decorrelate, a run samples the ensemble of runs, whose
    # mean this particular reference run sits slightly under.  So: north_star's +-0.002 holds for the MEAN over runs (+0.0016), not for every single
    # run (a third of them exceed it).  Asserted: the mean of three runs within 0.003 (mean + 3 standard errors), every run within 0.005 and every
    # epoch of every run within the per-epoch bounds above; precision 1 keeps one run and 0.01.
    runs = [one_run() for _ in range(3 if precision == 3 else 1)]
    pr, pn = float(np.mean([r[0] for r in runs])), float(np.mean([r[1] for r in runs]))
    if precision == 3:
        assert all(abs(r[0]) <= 0.005 and abs(r[1]) <= 0.005 for r in runs), f"precision 3: a single run leaves +-0.005: {runs}"
    lim = 0.003 if precision == 3 else 0.01
    assert abs(pr) <= lim and abs(pn) <= lim, f"precision {precision}: plateau Recall@20 {pr:+.4f}, NDCG@20 {pn:+.4f} from the reference's (bound {lim})"
