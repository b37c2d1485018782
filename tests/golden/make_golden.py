#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

Runs only where /root/reference exists (the build container); the GPU box and the test-suite only
read the .npz files this script wrote.  Nothing of the reference's source is copied: the script
imports the reference's modules (Model.py, DataHandler.py, Main.py, Utils/Utils.py) through the
in-process shims of SURVEY.md Appendix A and records inputs, recorded RNG draws and outputs.

The reference's pickled datasets are NOT loaded (torch.load(weights_only=True) refuses them and
pickle is off limits); the reference classes are driven on synthetic graphs from diffmm_amd.synth.

    python tests/golden/make_golden.py [--only NAME]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def _install_shims():
    import tomli
    import torch
    toml = types.ModuleType("toml")
    toml.load = lambda f: tomli.loads(f.read()) if hasattr(f, "read") else tomli.load(open(f, "rb"))
    toml.loads = tomli.loads
    sys.modules["toml"] = toml  # Conf.py:2
    numba = types.ModuleType("numba")
    numba.njit = lambda *a, **k: a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
    numba.prange = range
    sys.modules["numba"] = numba  # Utils/Utils.py:5 (dead code only)
    sys.path.insert(0, "/root/reference")
    if not torch.cuda.is_available():  # Model.py:397, Main.py:88-110,294-296,405-406 hard-code .cuda()
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self


_install_shims()
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

import Conf  # noqa: E402  (reference)
import DataHandler as RDH  # noqa: E402  (reference)
import Main as RMain  # noqa: E402  (reference)
import Model as RModel  # noqa: E402  (reference)
from Utils import Utils as RUtils  # noqa: E402  (reference)

from diffmm_amd import synth  # noqa: E402

torch.set_num_threads(8)
DEV = torch.device("cpu")


class _NullLog:
    def info(self, *_a, **_k):
        pass


RMain.main_log = _NullLog()


def mkcfg(name="tiktok", **kw) -> "Conf.Config":
    cfg = Conf.Config(base=Conf.BaseConfig(), data=Conf.DataConfig(), hyper=Conf.HyperConfig(), train=Conf.TrainConfig())
    cfg.data.name = name
    for k, v in kw.items():
        sec, key = k.split("__")
        setattr(getattr(cfg, sec), key, v)
    RMain.config = cfg
    return cfg


def coo_of(rows, cols, U, I):
    return sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(U, I), dtype=np.float32)


def adj_arrays(t: torch.Tensor):
    c = t.coalesce()
    idx = c.indices().numpy()
    return idx[0].astype(np.int32), idx[1].astype(np.int32), c.values().numpy().astype(np.float32)


def sd_np(prefix, module):
    return {f"{prefix}{k}": v.detach().numpy().copy() for k, v in module.state_dict().items()}


def random_modal_edges(g, rng):
    """Same user degrees as the train graph (rebuild keeps k_u = deg_u), random items."""
    deg = np.bincount(g.trn_rows, minlength=g.users)
    rows, cols = [], []
    for u in range(g.users):
        if deg[u]:
            rows.append(np.full(deg[u], u)); cols.append(rng.choice(g.items, size=deg[u], replace=False))
    return np.concatenate(rows).astype(np.int32), np.concatenate(cols).astype(np.int32)


# ------------------------------------------------------------------------------------------------
def gen_adj(name):
    g = synth.named(name)
    t = RDH.DataHandler.makeTorchAdj(coo_of(g.trn_rows, g.trn_cols, g.users, g.items), g.users, g.items, DEV)
    r, c, v = adj_arrays(t)
    np.savez_compressed(os.path.join(OUT, f"adj_{name}.npz"), users=g.users, items=g.items,
                        trn_rows=g.trn_rows, trn_cols=g.trn_cols, adj_row=r, adj_col=c, adj_val=v)


def gen_sched():
    out = {}
    for T in (5, 40, 200, 1000):
        for sc in (0.1, 0.5):
            cfg = mkcfg(hyper__steps=T, hyper__noise_scale=sc)
            gd = RModel.GaussianDiffusion(cfg)
            tag = f"T{T}_s{sc}"
            for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                      "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                      "posterior_mean_coef1", "posterior_mean_coef2"):
                out[f"{tag}/{k}"] = getattr(gd, k).numpy().astype(np.float64)
            tt = torch.arange(T)
            out[f"{tag}/snr"] = gd.SNR(tt).numpy().astype(np.float64)
    out["noise_min"] = 0.0001
    out["noise_max"] = 0.02
    np.savez_compressed(os.path.join(OUT, "sched.npz"), **out)


def gen_gcn(name, d, M=3):
    g = synth.named(name)
    rng = np.random.default_rng(11)
    cfg = mkcfg("tiktok" if M == 3 else "sports", base__latdim=d, hyper__modal_adj_weight=0.2, hyper__residual_weight=0.5)
    cfg.data.user_num, cfg.data.item_num = g.users, g.items
    cfg.data.image_feat_dim, cfg.data.text_feat_dim = g.feats[0].shape[1], g.feats[1].shape[1]
    if M == 3:
        cfg.data.audio_feat_dim = g.feats[2].shape[1]
    torch.manual_seed(5)
    feats = [torch.from_numpy(f) for f in g.feats[:M]]
    model = RModel.Model(cfg, *feats)
    with torch.no_grad():
        model.modal_weight.copy_(torch.tensor([0.2, 0.5, 0.3][:M]) if M == 3 else torch.tensor([0.35, 0.65]))
    base = RDH.DataHandler.makeTorchAdj(coo_of(g.trn_rows, g.trn_cols, g.users, g.items), g.users, g.items, DEV)
    madj, medges = [], {}
    for m in range(M):
        r, c = random_modal_edges(g, rng)
        medges[f"m{m}_rows"], medges[f"m{m}_cols"] = r, c
        madj.append(RDH.DataHandler.makeTorchAdj(coo_of(r, c, g.users, g.items), g.users, g.items, DEV))
    out = model.gcn_MM(base, *madj)
    names = ["u_final_embs", "i_final_embs", "u_image_embs", "i_image_embs", "u_text_embs", "i_text_embs"]
    if M == 3:
        names += ["u_audio_embs", "i_audio_embs"]
    outs = [getattr(out, n) for n in names]
    cots = [torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32)) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    loss.backward()
    save = dict(users=g.users, items=g.items, d=d, M=M, trn_rows=g.trn_rows, trn_cols=g.trn_cols,
                modal_adj_weight=0.2, residual_weight=0.5, **medges)
    save.update(sd_np("p/", model))
    for m in range(M):
        save[f"feat{m}"] = g.feats[m]
    for n, o, c in zip(names, outs, cots):
        save[f"out/{n}"] = o.detach().numpy()
        save[f"cot/{n}"] = c.numpy()
    for k, p in model.named_parameters():
        save[f"grad/{k}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"gcn_{name}_M{M}.npz"), **save)


def gen_losses():
    rng = np.random.default_rng(3)
    n, d, B = 150, 32, 96
    save = {}
    v1 = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_()
    v2 = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_()
    idx = torch.from_numpy(rng.integers(0, n, size=B))  # duplicates on purpose
    idx[5] = idx[4]
    for temp in (0.2, 0.5):
        v1.grad = v2.grad = None
        val = RUtils.InfoNCE(v1, v2, idx, temp)
        val.backward()
        save[f"infonce_t{temp}/val"] = val.detach().numpy()
        save[f"infonce_t{temp}/g1"] = v1.grad.numpy().copy()
        save[f"infonce_t{temp}/g2"] = v2.grad.numpy().copy()
    save["infonce/v1"], save["infonce/v2"], save["infonce/idx"] = v1.detach().numpy(), v2.detach().numpy(), idx.numpy()
    u = torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32) * 0.3).requires_grad_()
    p = torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32) * 0.3).requires_grad_()
    q = torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32) * 0.3).requires_grad_()
    val = RUtils.bpr_loss(u, p, q)
    val.backward()
    save.update({"bpr/u": u.detach().numpy(), "bpr/p": p.detach().numpy(), "bpr/n": q.detach().numpy(),
                 "bpr/val": val.detach().numpy(), "bpr/gu": u.grad.numpy(), "bpr/gp": p.grad.numpy(), "bpr/gn": q.grad.numpy()})
    a = torch.from_numpy(rng.standard_normal((40, d)).astype(np.float32)).requires_grad_()
    b = torch.from_numpy(rng.standard_normal((30, d)).astype(np.float32)).requires_grad_()
    val = RUtils.l2_reg_loss(1e-4, [a, b], DEV)
    val.backward()
    save.update({"l2/a": a.detach().numpy(), "l2/b": b.detach().numpy(), "l2/val": val.detach().numpy(),
                 "l2/ga": a.grad.numpy(), "l2/gb": b.grad.numpy(), "l2/reg": 1e-4})
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **save)


def gen_diffusion(name, d, H, T, B, dims=None, out_tag=None):
    """dims (optional): the reference's base.denoise_dim list with more than one hidden width, e.g. [48, 64] -> in_layers (I+te -> 64),
    (64 -> 48); out_layers (48 -> 64), (64 -> I) (Model.py:156-160); H is then dims[-1] (width next to the item dimension)."""
    g = synth.named(name)
    rng = np.random.default_rng(17)
    dims = [H] if dims is None else list(dims)
    cfg = mkcfg(base__latdim=d, base__denoise_dim=str(dims), hyper__steps=T, hyper__noise_scale=0.5,
                hyper__sim_weight=0.01, train__reg=1e-4)
    cfg.data.user_num, cfg.data.item_num = g.users, g.items
    torch.manual_seed(9)
    gd = RModel.GaussianDiffusion(cfg)
    out_dims = dims + [g.items]
    den = RModel.Denoise(out_dims[::-1], out_dims, cfg)
    dense = torch.from_numpy(coo_of(g.trn_rows, g.trn_cols, g.users, g.items).toarray())
    rows = torch.from_numpy(rng.choice(g.users, size=B, replace=False))
    x0 = dense[rows]
    t = torch.from_numpy(rng.integers(0, T, size=B))
    t[0] = 0  # exercise the t==0 weight branch (Model.py:412)
    noise = torch.from_numpy(rng.standard_normal((B, g.items)).astype(np.float32))
    feat = torch.from_numpy((rng.standard_normal((g.items, d)) * 0.3).astype(np.float32))
    i_embs = torch.from_numpy((rng.standard_normal((g.items, d)) * 0.1).astype(np.float32)).requires_grad_()
    save = dict(users=g.users, items=g.items, d=d, H=H, T=T, B=B, dims=np.asarray(dims), x0=x0.numpy(), rows=rows.numpy(), t=t.numpy(),
                noise=noise.numpy(), feat=feat.numpy(), i_embs=i_embs.detach().numpy(), noise_scale=0.5,
                sim_weight=0.01, reg=1e-4)
    save.update(sd_np("p/", den))
    xt = gd.forward_cal_xt(x0, t, noise)
    save["x_t"] = xt.numpy()
    cot = torch.from_numpy(rng.standard_normal((B, g.items)).astype(np.float32))
    save["cot"] = cot.numpy()
    for tag, mf in (("gate", feat), ("nogate", None)):
        den.zero_grad()
        xin = xt.clone().requires_grad_()
        o = den.forward(xin, t, mf)
        (o * cot).sum().backward()
        save[f"{tag}/out"] = o.detach().numpy()
        save[f"{tag}/gx"] = xin.grad.numpy().copy()
        for k, p in den.named_parameters():
            if p.grad is not None:
                save[f"{tag}/grad/{k}"] = p.grad.numpy().copy()
    with torch.no_grad():
        tt = torch.full((B,), T - 2, dtype=torch.long)
        mean, logvar = gd.p_mean_variance(den, xt, tt)
        save["pmean/t"] = T - 2
        save["pmean/mean"] = mean.numpy()
        save["pmean/logvar"] = logvar.numpy()
        save["genview/out"] = gd.generate_view(den, x0, 0).numpy()
    # training_losses with recorded draws
    den.zero_grad()
    o_randint, o_randn = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda x, **k: noise.clone()
    try:
        tl = gd.training_losses(den, x0, i_embs, feat)
    finally:
        torch.randint, torch.randn_like = o_randint, o_randn
    tl.mean().backward()
    save["tl/loss"] = tl.detach().numpy()  # float64 (B,)
    save["tl/g_i_embs"] = i_embs.grad.numpy().copy()
    for k, p in den.named_parameters():
        save[f"tl/grad/{k}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"diffusion_{out_tag or name}.npz"), **save)


class _TrainLoader:
    """Stands in for DataLoader(trainData, shuffle=True): fixed, recorded permutation; reads the negs the
    reference's negSampling() wrote just before the loop (Main.py:137)."""

    def __init__(self, data, perm, B):
        self.data, self.perm, self.B = data, perm, B

    def __iter__(self):
        for s in range(0, len(self.perm), self.B):
            i = self.perm[s:s + self.B]
            yield (torch.from_numpy(self.data.rows[i]), torch.from_numpy(self.data.cols[i]), torch.from_numpy(self.data.negs[i]))


class _DiffLoader:
    def __init__(self, dense, perm, B):
        self.dense, self.perm, self.B = dense, perm, B

    def __iter__(self):
        for s in range(0, len(self.perm), self.B):
            i = torch.from_numpy(self.perm[s:s + self.B])
            yield (self.dense[i], i)


def build_handler(cfg, g, B, seed):
    rng = np.random.default_rng(seed)
    h = RDH.DataHandler(cfg)
    trn = coo_of(g.trn_rows, g.trn_cols, g.users, g.items)
    tst = coo_of(g.tst_rows, g.tst_cols, g.users, g.items)
    h.trainMat = trn
    cfg.data.user_num, cfg.data.item_num = g.users, g.items
    h.torchBiAdj = RDH.DataHandler.makeTorchAdj(trn, g.users, g.items, DEV)
    h.trainData = RDH.TrainData(trn, cfg)
    train_perm = rng.permutation(len(h.trainData.rows))
    h.trainLoader = _TrainLoader(h.trainData, train_perm, B)
    h.testData = RDH.TestData(tst, trn)
    h.testLoader = torch.utils.data.DataLoader(h.testData, batch_size=cfg.train.test_batch, shuffle=False, num_workers=0)
    h.image_feats = torch.from_numpy(g.feats[0]); cfg.data.image_feat_dim = g.feats[0].shape[1]
    h.text_feats = torch.from_numpy(g.feats[1]); cfg.data.text_feat_dim = g.feats[1].shape[1]
    if cfg.data.name == "tiktok":
        h.audio_feats = torch.from_numpy(g.feats[2]); cfg.data.audio_feat_dim = g.feats[2].shape[1]
    dense = torch.tensor(trn.toarray(), dtype=torch.float)
    h.diffusionData = RDH.DiffusionData(dense, cfg)
    diff_perm = rng.permutation(g.users)
    h.diffusionLoader = _DiffLoader(dense, diff_perm, B)
    h.user_pos_items = h.trainData.user_pos_items
    return h, train_perm, diff_perm


def gen_epoch(name, d, H, T, B, epochs, store_draws, cl_method=0, M=3, tag=None, slim=False):
    """One or more full Coach.trainEpoch()+testEpoch() of the reference with every random draw pinned.
    slim: a LONG paired run -- only the initial weights, the per-epoch negatives and the per-epoch losses / metrics are kept (no per-epoch weights or
    rebuilt graphs, draws by seed only): what the HIP path must reproduce epoch by epoch when it replays the same draws."""
    g = synth.named(name)
    cfg = mkcfg("tiktok" if M == 3 else "sports", base__latdim=d, base__denoise_dim=f"[{H}]", base__seed=1818,
                base__cl_method=cl_method, base__topk=min(20, g.items // 2),
                hyper__steps=T, hyper__noise_scale=0.5, hyper__noise_degree=1.5, hyper__cross_cl_rate=0.5,
                hyper__cross_cl_temp=0.2, hyper__modal_cl_rate=0.01, hyper__modal_cl_temp=0.5,
                hyper__sim_weight=0.01, hyper__residual_weight=0.5, hyper__modal_adj_weight=0.2,
                train__batch=B, train__test_batch=64, train__reg=1e-4, train__lr=1e-3, train__epoch=max(epochs, 2))
    RMain.seed_it(cfg.base.seed)
    h, train_perm, diff_perm = build_handler(cfg, g, B, seed=23)
    coach = RMain.Coach(h, cfg)
    coach.prepareModel()
    save = dict(users=g.users, items=g.items, d=d, H=H, T=T, B=B, M=M, epochs=epochs, cl_method=cl_method,
                topk=min(20, g.items // 2),
                trn_rows=g.trn_rows, trn_cols=g.trn_cols, tst_rows=g.tst_rows, tst_cols=g.tst_cols,
                train_perm=train_perm, diff_perm=diff_perm, lr=1e-3, reg=1e-4, train_epochs_cfg=max(epochs, 2))
    for m in range(M):
        save[f"feat{m}"] = g.feats[m]
    dens = [coach.image_denoise_model, coach.text_denoise_model] + ([coach.audio_denoise_model] if M == 3 else [])
    save.update(sd_np("init/model/", coach.model))
    for m, dn in enumerate(dens):
        save.update(sd_np(f"init/den{m}/", dn))
    # pin the RNG: every torch draw the hot path makes comes from this generator, in call order
    gen = torch.Generator().manual_seed(4242)
    draws = []
    o = (torch.randint, torch.randn_like, torch.rand_like)

    def _randint(lo, hi, size, **k):
        r = o[0](lo, hi, size, generator=gen); draws.append(("randint", r.numpy().copy())); return r

    def _randn_like(x, **k):
        r = torch.randn(x.shape, generator=gen, dtype=x.dtype); draws.append(("randn", r.numpy().copy())); return r

    def _rand_like(x, **k):
        r = torch.rand(x.shape, generator=gen, dtype=x.dtype); draws.append(("rand", r.numpy().copy())); return r

    torch.randint, torch.randn_like, torch.rand_like = _randint, _randn_like, _rand_like
    np.random.seed(77)
    try:
        for ep in range(epochs):
            res = coach.trainEpoch()
            save[f"ep{ep}/negs"] = h.trainData.negs.copy()
            for k, v in res.items():
                save[f"ep{ep}/train/{k}"] = np.float64(v)
            adjs = [coach.image_adj, coach.text_adj] + ([coach.audio_adj] if M == 3 else [])
            for m, a in enumerate(adjs if not slim else []):
                r, c, v = adj_arrays(a)
                save[f"ep{ep}/adj{m}_row"], save[f"ep{ep}/adj{m}_col"], save[f"ep{ep}/adj{m}_val"] = r, c, v
            if cfg.train.use_lr_scheduler:  # Main.py:59-65
                coach.model_scheduler.step(); coach.image_scheduler.step(); coach.text_scheduler.step()
                if M == 3:
                    coach.audio_scheduler.step()
            with torch.no_grad():
                tres = coach.testEpoch()
            for k, v in tres.items():
                save[f"ep{ep}/test/{k}"] = np.float64(v)
            if not slim:
                save.update(sd_np(f"ep{ep}/model/", coach.model))
                for m, dn in enumerate(dens):
                    save.update(sd_np(f"ep{ep}/den{m}/", dn))
            save[f"ep{ep}/lr_after"] = np.float64(coach.opt.param_groups[0]["lr"])
    finally:
        torch.randint, torch.randn_like, torch.rand_like = o
    save["draw_seed"] = 4242
    save["n_draws"] = len(draws)
    save["draw_kinds"] = np.array([k for k, _ in draws])
    save["draw_checksum"] = np.array([float(np.asarray(a, dtype=np.float64).sum()) for _, a in draws])
    if store_draws:
        for i, (_k, a) in enumerate(draws):
            save[f"draw/{i}"] = a
    np.savez_compressed(os.path.join(OUT, f"epoch_{tag or name}.npz"), **save)
    print(f"epoch_{tag or name}: draws={len(draws)}", {k: float(v) for k, v in save.items() if k.startswith('ep0/t')})


def gen_calcres():
    cfg = mkcfg(base__topk=5)
    coach = RMain.Coach.__new__(RMain.Coach)
    coach.config = cfg
    rng = np.random.default_rng(1)
    top = np.stack([rng.permutation(30)[:5] for _ in range(8)])
    its = [list(rng.choice(30, size=rng.integers(1, 9), replace=False)) for _ in range(12)]
    users = torch.tensor([0, 3, 4, 7, 8, 9, 10, 11])
    r, n, p = coach.calcRes(top, its, users)
    np.savez_compressed(os.path.join(OUT, "calcres.npz"), top=top, users=users.numpy(), topk=5,
                        its_flat=np.concatenate(its), its_len=np.array([len(x) for x in its]),
                        recall=r, ndcg=n, precision=p)


JOBS = {
    "adj": lambda: [gen_adj("toy"), gen_adj("mid")],
    "sched": gen_sched,
    "gcn": lambda: [gen_gcn("toy", 8, 3), gen_gcn("mid", 32, 3), gen_gcn("mid", 32, 2)],
    "losses": gen_losses,
    "diffusion": lambda: [gen_diffusion("toy", 8, 16, 5, 8), gen_diffusion("mid", 32, 64, 5, 64)],
    "diffusion_deep": lambda: [gen_diffusion("mid", 32, 64, 5, 64, dims=[48, 64], out_tag="mid_deep"), gen_diffusion("mid", 32, 32, 4, 64, dims=[24, 40, 32], out_tag="mid_deep3")],
    "epoch": lambda: [gen_epoch("toy", 8, 16, 5, 6, 1, True),
                      gen_epoch("mid", 16, 64, 5, 128, 2, False),
                      gen_epoch("mid", 16, 64, 3, 128, 1, False, cl_method=1, tag="mid_cl1"),
                      gen_epoch("mid", 16, 64, 3, 128, 1, False, M=2, tag="mid_M2")],
    "calcres": gen_calcres,
    # the paired long run: 12 reference epochs at 2000 x 1200, draws regenerated from their seed
    "epoch_paired": lambda: gen_epoch("mid2", 32, 64, 10, 512, 12, False, tag="paired", slim=True),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for k, fn in JOBS.items():
        if a.only and a.only != k:
            continue
        fn()
        print("wrote", k)
