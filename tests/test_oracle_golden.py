"""Pins oracle/diffmm_oracle.py against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerances are fp32 round-off: the oracle sums in a
different order (coalesced CSR vs the reference's uncoalesced COO) but is otherwise the same math."""
import os

import numpy as np
import pytest
import torch

from oracle import diffmm_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
torch.set_num_threads(4)


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def params(z, prefix):
    return {k[len(prefix):]: T(z[k]).clone() for k in z.files if k.startswith(prefix)}


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    assert np.allclose(a, b, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e} (scale {np.abs(b).max():.3e})"


@pytest.mark.parametrize("name", ["toy", "mid"])
def test_adjacency(name):
    z = load(f"adj_{name}.npz")
    rp, col, val = O.build_adj_csr(z["trn_rows"], z["trn_cols"], int(z["users"]), int(z["items"]))
    row = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    assert np.array_equal(row, z["adj_row"]) and np.array_equal(col, z["adj_col"])
    assert np.array_equal(val, z["adj_val"]), "normalised values must be bit-identical (float64 math, one cast)"


def test_schedule_tables():
    z = load("sched.npz")
    for Tn in (5, 40, 200, 1000):
        for sc in (0.1, 0.5):
            s = O.schedule(Tn, sc, float(z["noise_min"]), float(z["noise_max"]))
            for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                      "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "snr"):
                close(s[k], z[f"T{Tn}_s{sc}/{k}"], rtol=1e-12, atol=1e-15, what=f"T{Tn} s{sc} {k}")


@pytest.mark.parametrize("fx", ["gcn_toy_M3", "gcn_mid_M3", "gcn_mid_M2"])
def test_gcn_mm_outputs_and_grads(fx):
    z = load(fx + ".npz")
    U, I, M = int(z["users"]), int(z["items"]), int(z["M"])
    P = {k: v.requires_grad_() for k, v in params(z, "p/").items()}
    feats = [T(z[f"feat{m}"]) for m in range(M)]
    adj = O.csr_to_torch(*O.build_adj_csr(z["trn_rows"], z["trn_cols"], U, I))
    madj = [O.csr_to_torch(*O.build_adj_csr(z[f"m{m}_rows"], z[f"m{m}_cols"], U, I)) for m in range(M)]
    out = O.gcn_mm(P, feats, adj, madj, float(z["modal_adj_weight"]), float(z["residual_weight"]))
    loss = 0
    for k in [f[4:] for f in z.files if f.startswith("out/")]:
        close(out[k].detach(), z[f"out/{k}"], rtol=2e-5, atol=2e-6, what=k)
        loss = loss + (out[k] * T(z[f"cot/{k}"])).sum()
    loss.backward()
    for k in P:
        close(P[k].grad, z[f"grad/{k}"], rtol=2e-4, atol=2e-5, what="grad " + k)


def test_losses():
    z = load("losses.npz")
    idx = T(z["infonce/idx"])
    for temp in (0.2, 0.5):
        v1 = T(z["infonce/v1"]).clone().requires_grad_(); v2 = T(z["infonce/v2"]).clone().requires_grad_()
        val = O.infonce(v1, v2, idx, temp); val.backward()
        close(val.detach(), z[f"infonce_t{temp}/val"], what="infonce")
        close(v1.grad, z[f"infonce_t{temp}/g1"], atol=1e-7, what="infonce g1")
        close(v2.grad, z[f"infonce_t{temp}/g2"], atol=1e-7, what="infonce g2")
    u, p, n = (T(z[f"bpr/{k}"]).clone().requires_grad_() for k in "upn")
    val = O.bpr(u, p, n); val.backward()
    close(val.detach(), z["bpr/val"]); close(u.grad, z["bpr/gu"], atol=1e-8); close(p.grad, z["bpr/gp"], atol=1e-8); close(n.grad, z["bpr/gn"], atol=1e-8)
    a, b = T(z["l2/a"]).clone().requires_grad_(), T(z["l2/b"]).clone().requires_grad_()
    val = O.l2_reg(float(z["l2/reg"]), [a, b]); val.backward()
    close(val.detach(), z["l2/val"]); close(a.grad, z["l2/ga"]); close(b.grad, z["l2/gb"])


@pytest.mark.parametrize("name", ["toy", "mid", "mid_deep", "mid_deep3"])  # *_deep: denoise_dim with 2 / 3 hidden widths (Model.py:156-160)
def test_diffusion_pieces(name):
    z = load(f"diffusion_{name}.npz")
    s = O.schedule(int(z["T"]), float(z["noise_scale"]), 0.0001, 0.02)
    P = {k: v.requires_grad_() for k, v in params(z, "p/").items()}
    x0, t, noise, feat = T(z["x0"]), T(z["t"]), T(z["noise"]), T(z["feat"])
    xt = O.q_sample(x0, t, noise, s)
    close(xt, z["x_t"], what="q_sample")
    cot = T(z["cot"])
    for tag, mf in (("gate", feat), ("nogate", None)):
        for p in P.values():
            p.grad = None
        xin = T(z["x_t"]).clone().requires_grad_()
        o = O.denoise_forward(P, xin, t, mf)
        close(o.detach(), z[f"{tag}/out"], rtol=2e-5, atol=2e-6, what=tag)
        (o * cot).sum().backward()
        close(xin.grad, z[f"{tag}/gx"], rtol=1e-4, atol=1e-5, what=tag + " gx")
        for k in P:
            if f"{tag}/grad/{k}" in z.files:
                close(P[k].grad, z[f"{tag}/grad/{k}"], rtol=1e-4, atol=1e-5, what=f"{tag} grad {k}")
    with torch.no_grad():
        tt = torch.full((x0.shape[0],), int(z["pmean/t"]), dtype=torch.long)
        mean, logvar = O.p_mean(P, T(z["x_t"]), tt, s)
        close(mean, z["pmean/mean"], rtol=2e-5, atol=2e-6); close(logvar, z["pmean/logvar"])
        close(O.generate_view(P, x0, s, 0), z["genview/out"], rtol=5e-5, atol=5e-6, what="generate_view")
    for p in P.values():
        p.grad = None
    ie = T(z["i_embs"]).clone().requires_grad_()
    tl = O.training_losses(P, x0, ie, feat, t, noise, s, float(z["sim_weight"]), float(z["reg"]))
    assert tl.dtype == torch.float64
    close(tl.detach(), z["tl/loss"], rtol=2e-5, atol=1e-7, what="training_losses")
    tl.mean().backward()
    close(ie.grad, z["tl/g_i_embs"], rtol=1e-4, atol=1e-8, what="tl g i_embs")
    for k in P:
        close(P[k].grad, z[f"tl/grad/{k}"], rtol=2e-4, atol=1e-7, what="tl grad " + k)


def test_calcres_metrics():
    z = load("calcres.npz")
    top, users, k = z["top"], z["users"], int(z["topk"])
    lens = z["its_len"]; flat = z["its_flat"]; off = np.concatenate([[0], np.cumsum(lens)])
    tst_rows = np.concatenate([np.full(lens[u], u) for u in users]); tst_cols = np.concatenate([flat[off[u]:off[u + 1]] for u in users])
    # feed evaluate() embeddings that reproduce exactly these top-k lists: one-hot users, score = rank table
    U, I = len(lens), 30
    score = np.zeros((U, I), np.float32)
    for j, u in enumerate(users):
        score[u, top[j]] = np.arange(k, 0, -1) + 10
    res = O.evaluate(torch.eye(U), T(score).t().contiguous(), np.zeros(0, np.int64), np.zeros(0, np.int64), tst_rows, tst_cols, k)
    n = len(users)
    close(res["Recall"] * n, z["recall"]); close(res["NDCG"] * n, z["ndcg"]); close(res["Precision"] * n, z["precision"])


def _epoch_inputs(z):
    U, I, M, B = int(z["users"]), int(z["items"]), int(z["M"]), int(z["B"])
    feats = [T(z[f"feat{m}"]) for m in range(M)]
    dense = torch.zeros((U, I)); dense[T(z["trn_rows"].astype(np.int64)), T(z["trn_cols"].astype(np.int64))] = 1.0
    data = dict(users=U, items=I, B=B, feats=feats, trn_rows=z["trn_rows"], trn_cols=z["trn_cols"], tst_rows=z["tst_rows"],
                tst_cols=z["tst_cols"], train_perm=z["train_perm"], diff_perm=z["diff_perm"], dense=dense,
                adj=O.csr_to_torch(*O.build_adj_csr(z["trn_rows"], z["trn_cols"], U, I)))
    hp = dict(T=int(z["T"]), sim_weight=0.01, reg=float(z["reg"]), modal_adj_weight=0.2, residual_weight=0.5, noise_degree=1.5,
              cross_cl_temp=0.2, cross_cl_rate=0.5, modal_cl_temp=0.5, modal_cl_rate=0.01, cl_method=int(z["cl_method"]))
    st = O.EpochState(params(z, "init/model/"), [params(z, f"init/den{m}/") for m in range(M)], float(z["lr"]), int(z["train_epochs_cfg"]))
    return data, hp, st, M


@pytest.mark.parametrize("fx", ["epoch_toy", "epoch_mid", "epoch_mid_cl1", "epoch_mid_M2"])
def test_full_epochs_match_reference(fx):
    z = load(fx + ".npz")
    data, hp, st, M = _epoch_inputs(z)
    sched = O.schedule(hp["T"], 0.5, 0.0001, 0.02)
    stored = [z[f"draw/{i}"] for i in range(int(z["n_draws"]))] if "draw/0" in z.files else None
    draws = O.Draws(seed=int(z["draw_seed"]) if stored is None else None, stored=stored)
    for ep in range(int(z["epochs"])):
        data["negs"] = z[f"ep{ep}/negs"]
        golden_adj = [(z[f"ep{ep}/adj{m}_row"], z[f"ep{ep}/adj{m}_col"], z[f"ep{ep}/adj{m}_val"]) for m in range(M)]
        res, csr = O.train_epoch(st, data, draws, sched, hp)
        for m in range(M):  # rebuilt adjacency: same edges, same normalised values
            row = np.repeat(np.arange(len(csr[m][0]) - 1), np.diff(csr[m][0]))
            same = np.array_equal(row, golden_adj[m][0]) and np.array_equal(csr[m][1], golden_adj[m][1])
            assert same, f"epoch {ep} modality {m}: rebuilt edge set differs from the reference"
            close(csr[m][2], golden_adj[m][2], rtol=1e-6)
        for k, v in res.items():
            close(v, z[f"ep{ep}/train/{k}"], rtol=5e-5, atol=1e-7, what=f"ep{ep} {k}")
        st.step_schedulers()
        close(st.opt.lr, z[f"ep{ep}/lr_after"], rtol=1e-9)
        tres = O.test_epoch(st, data, csr, hp, int(z["topk"]))
        for k, v in tres.items():
            close(v, z[f"ep{ep}/test/{k}"], rtol=1e-6, atol=1e-9, what=f"ep{ep} test {k}")
        for k, p in st.model.items():
            close(p.detach(), z[f"ep{ep}/model/{k}"], rtol=2e-4, atol=2e-6, what=f"ep{ep} model {k}")
        for m in range(M):
            for k, p in st.dens[m].items():
                close(p.detach(), z[f"ep{ep}/den{m}/{k}"], rtol=2e-4, atol=2e-6, what=f"ep{ep} den{m} {k}")
    assert draws.i == int(z["n_draws"])
    close(np.array(draws.checks), z["draw_checksum"], rtol=1e-12, what="random draws differ from the recorded ones")
