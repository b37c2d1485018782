"""The reference-named Python surface called the way code written against the reference calls it, on the GPU, against the
reference's golden vectors: Utils.InfoNCE / bpr_loss / l2_reg_loss (+ .backward()), Model.gcn_MM, Denoise.forward,
GaussianDiffusion.forward_cal_xt / p_mean_variance / generate_view / training_losses (+ .backward()), TrainData.negSampling."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))

from test_kernels_gpu import T, close, load, need_gpu  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    need_gpu()
    return torch.device("cuda:0")


def _config(**over):
    from diffmm_amd.Conf import load_config
    cfg = load_config(os.path.join(REPO, "conf", "tiktok.toml"))
    for k, v in over.items():
        sec, key = k.split("__")
        setattr(getattr(cfg, sec), key, v)
    return cfg


# ---- Utils/Utils.py:45-98 -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("temp", [0.2, 0.5])
def test_infonce_value_and_backward(dev, temp):
    from diffmm_amd.Utils.Utils import InfoNCE
    z = load("losses.npz")
    v1, v2 = T(z["infonce/v1"]).to(dev).requires_grad_(), T(z["infonce/v2"]).to(dev).requires_grad_()
    idx = T(z["infonce/idx"]).to(dev)  # contains duplicates
    loss = InfoNCE(v1, v2, idx, temp)
    loss.backward()
    close(loss.detach(), z[f"infonce_t{temp}/val"], rtol=2e-5, what="InfoNCE")
    close(v1.grad, z[f"infonce_t{temp}/g1"], rtol=2e-4, atol=2e-7, what="d view1")
    close(v2.grad, z[f"infonce_t{temp}/g2"], rtol=2e-4, atol=2e-7, what="d view2")


def test_infonce_without_cosine_normalisation(dev):
    """b_cos=False (Utils.py:71 skipped).  The reference never calls it this way, so no golden vector exists: checked against the
    oracle's restatement of the same lines (autograd for the gradients)."""
    from diffmm_amd.Utils.Utils import InfoNCE
    from oracle import diffmm_oracle as O
    z = load("losses.npz")
    a, b, idx = T(z["infonce/v1"]) * 0.5, T(z["infonce/v2"]) * 0.5, T(z["infonce/idx"]).long()
    ra, rb = a.clone().requires_grad_(), b.clone().requires_grad_()
    want = O.infonce(ra, rb, idx, 0.7, b_cos=False)
    want.backward()
    v1, v2 = a.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    loss = InfoNCE(v1, v2, idx.to(dev), 0.7, b_cos=False)
    loss.backward()
    close(loss.detach(), want.detach(), rtol=2e-5, what="InfoNCE(b_cos=False)")
    close(v1.grad, ra.grad, rtol=2e-4, atol=2e-7, what="d view1")
    close(v2.grad, rb.grad, rtol=2e-4, atol=2e-7, what="d view2")
    with pytest.raises(ValueError):
        InfoNCE(v1, v2[:, :16], idx.to(dev), 0.7)


def test_bpr_and_l2_losses(dev):
    from diffmm_amd.Utils.Utils import bpr_loss, l2_reg_loss
    z = load("losses.npz")
    u, p, n = (T(z[f"bpr/{k}"]).to(dev).requires_grad_() for k in "upn")
    loss = bpr_loss(u, p, n)
    loss.backward()
    close(loss.detach(), z["bpr/val"], rtol=2e-5, what="bpr_loss")
    for t, k in ((u, "gu"), (p, "gp"), (n, "gn")):
        close(t.grad, z[f"bpr/{k}"], rtol=2e-4, atol=1e-8, what=k)
    a, b = T(z["l2/a"]).to(dev).requires_grad_(), T(z["l2/b"]).to(dev).requires_grad_()
    reg = l2_reg_loss(float(z["l2/reg"]), [a, b], dev)
    reg.backward()
    close(reg.detach(), z["l2/val"], rtol=2e-5, what="l2_reg_loss")
    close(a.grad, z["l2/ga"], rtol=1e-5, what="d a"); close(b.grad, z["l2/gb"], rtol=1e-5, what="d b")


# ---- Model.py:15-134 ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fx", ["gcn_mid_M3", "gcn_mid_M2"])
def test_model_gcn_mm(dev, fx):
    from diffmm_amd import graph
    from diffmm_amd.Model import Model
    z = load(fx + ".npz")
    U, I, M, D = int(z["users"]), int(z["items"]), int(z["M"]), int(z["d"])
    cfg = _config(base__latdim=D, data__user_num=U, data__item_num=I, hyper__modal_adj_weight=0.2, hyper__residual_weight=0.5, train__batch=64)
    feats = [T(z[f"feat{m}"]).to(dev) for m in range(M)]
    model = Model(cfg, *feats, precision=3)
    model.engine.load_reference_state({k[2:]: z[k] for k in z.files if k.startswith("p/")})
    adj = graph.make_adjacency(z["trn_rows"], z["trn_cols"], U, I, D, dev)
    madj = [graph.make_adjacency(z[f"m{m}_rows"], z[f"m{m}_cols"], U, I, D, dev) for m in range(M)]
    out = model.gcn_MM(adj, *madj)
    for k in [f[4:] for f in z.files if f.startswith("out/")]:
        close(getattr(out, k), z[f"out/{k}"], rtol=1e-4, atol=1e-5, what=k)
    if M == 2:
        assert out.u_audio_embs is None and out.i_audio_embs is None
    assert model.getItemEmbs().shape == (I, D) and model.getUserEmbs().shape == (U, D)
    close(model.getImageFeats(), T(z["feat0"]) @ T(z["p/image_layer.weight"]).t() + T(z["p/image_layer.bias"]), rtol=1e-4, atol=5e-5 * 3.2, what="getImageFeats")  # split-bf16 products: 16 mantissa bits per operand


# ---- Model.py:136-428 -----------------------------------------------------------------------------------------------------------
def _diffusion_objects(z, dev, precision=3):
    from diffmm_amd.Model import Denoise, GaussianDiffusion, denoise_dims
    I, D, H, T_, B = int(z["items"]), int(z["d"]), int(z["H"]), int(z["T"]), int(z["B"])
    dims = [int(x) for x in z["dims"]] if "dims" in z.files else [H]  # base.denoise_dim: one or more hidden widths (Model.py:156-160)
    cfg = _config(base__latdim=D, base__denoise_dim=str(dims), data__item_num=I, data__user_num=int(z["users"]), hyper__steps=T_,
                  hyper__noise_scale=float(z["noise_scale"]), hyper__noise_min=0.0001, hyper__noise_max=0.02, hyper__sim_weight=float(z["sim_weight"]),
                  train__reg=float(z["reg"]), train__batch=B)
    in_dims, out_dims = denoise_dims(cfg)
    den = Denoise(in_dims, out_dims, cfg, precision=precision)
    den.engine.load_reference_state(den.slot, {k[2:]: z[k] for k in z.files if k.startswith("p/")})
    return cfg, den, GaussianDiffusion(cfg)


@pytest.mark.parametrize("name", ["toy", "mid", "mid_deep", "mid_deep3"])  # *_deep: 2 / 3 hidden widths
def test_denoise_and_gaussian_diffusion_api(dev, name):
    z = load(f"diffusion_{name}.npz")
    cfg, den, gd = _diffusion_objects(z, dev)
    B, I = int(z["B"]), int(z["items"])
    x0, t, noise, feat, ie = T(z["x0"]).to(dev), T(z["t"]).to(dev), T(z["noise"]).to(dev), T(z["feat"]).to(dev), T(z["i_embs"]).to(dev)
    x_t = gd.forward_cal_xt(x0, t, noise)
    close(x_t, z["x_t"], what="forward_cal_xt")
    close(den.forward(T(z["x_t"]).to(dev), t, feat), z["gate/out"], rtol=1e-4, atol=2e-5, what="Denoise.forward (gate)")
    close(den(T(z["x_t"]).to(dev), t), z["nogate/out"], rtol=1e-4, atol=2e-5, what="Denoise.forward (no gate)")
    tp = torch.full((B,), int(z["pmean/t"]), device=dev)
    mean, logvar = gd.p_mean_variance(den, T(z["x_t"]).to(dev), tp)
    close(mean, z["pmean/mean"], rtol=1e-4, atol=2e-5, what="p_mean_variance mean")
    close(logvar, z["pmean/logvar"], rtol=1e-6, what="p_mean_variance log-variance")
    close(gd.generate_view(den, x0, 0), z["genview/out"], rtol=2e-4, atol=5e-5, what="generate_view")
    close(gd.SNR(t), (gd.alphas_cumprod / (1 - gd.alphas_cumprod + 1e-8))[t.long()], rtol=1e-12, what="SNR")
    # default noise of forward_cal_xt: sign(x_0) * normalize(randn): unit-norm rows restricted to x_0's support
    ts = torch.full((B,), 1, device=dev)
    drawn = gd.forward_cal_xt(x0, ts)
    a, c = float(gd.sqrt_alphas_cumprod[1]), float(gd.sqrt_one_minus_alphas_cumprod[1])
    resid = (drawn - a * x0) / c
    assert float(resid[x0 == 0].abs().max()) == 0.0 and float(resid.norm(dim=1).max()) <= 1.0 + 1e-5
    # training_losses: value, then .mean().backward() -> Denoise gradients + d/d i_embs (Model.py:385-428, Main.py:155-176)
    ie_req = ie.clone().requires_grad_()
    loss = gd.training_losses(den, x0, ie_req, feat, timesteps=t, noise=noise)
    assert loss.dtype == torch.float64 and loss.shape == (B,)
    close(loss.detach(), z["tl/loss"], rtol=1e-4, atol=1e-7, what="training_losses")
    loss.mean().backward()
    g = den.grads()
    keys = [f[len("tl/grad/"):] for f in z.files if f.startswith("tl/grad/")]
    assert set(keys) == set(g), "every parameter of the reference's Denoise has a gradient here, under the same name"
    for k in keys:
        ref = z[f"tl/grad/{k}"]
        close(g[k], ref, rtol=2e-3, atol=2e-4 * float(np.abs(ref).max()) + 1e-9, what="grad " + k)
    ref = z["tl/g_i_embs"]
    close(ie_req.grad, ref, rtol=2e-3, atol=2e-4 * float(np.abs(ref).max()) + 1e-12, what="d i_embs")
    with pytest.raises(NotImplementedError):
        loss2 = gd.training_losses(den, x0, ie.clone().requires_grad_(), feat, timesteps=t, noise=noise)
        (loss2 * torch.arange(B, device=dev)).sum().backward()


# ---- DataHandler.py:159-169 -----------------------------------------------------------------------------------------------------
def test_neg_sampling_rejects_train_positives_and_is_uniform(dev):
    from diffmm_amd import synth
    from diffmm_amd.DataHandler import TrainData
    g = synth.named("mid")
    cfg = _config(data__user_num=g.users, data__item_num=g.items)
    td = TrainData(g.trn_rows, g.trn_cols, cfg, dev)
    pos = set(zip(g.trn_rows.tolist(), g.trn_cols.tolist()))
    counts = np.zeros(g.items)
    prev = None
    for _ in range(20):
        td.negSampling()
        negs = td.negs
        assert negs.shape == g.trn_rows.shape and negs.min() >= 0 and negs.max() < g.items
        assert not any((u, n) in pos for u, n in zip(g.trn_rows.tolist(), negs.tolist())), "a sampled negative is a train positive"
        assert prev is None or not np.array_equal(prev, negs), "every call draws afresh"
        prev = negs.copy()
        counts += np.bincount(negs, minlength=g.items)
        u, p, n = td.batch(torch.arange(5, device=dev))
        assert torch.equal(n.cpu(), torch.from_numpy(negs[:5]))
    # uniform over the items a user has NOT interacted with (DataHandler.py:163-168): interaction e of user u picks each non-positive item
    # with probability 1 / (items - |pos(u)|), so popular items are drawn less often -- compare with that exact expectation
    n_int = np.bincount(g.trn_rows, minlength=g.users).astype(np.float64)
    upos = [set() for _ in range(g.users)]
    for u, i in pos:
        upos[u].add(i)
    w = np.array([n_int[u] / (g.items - len(upos[u])) for u in range(g.users)])
    expect = np.full(g.items, w.sum())
    for u in range(g.users):
        for i in upos[u]:
            expect[i] -= w[u]
    expect *= 20
    assert abs(expect.sum() - counts.sum()) < 1e-6 * counts.sum()
    chi2 = float(((counts - expect) ** 2 / expect).sum())
    assert chi2 < g.items + 6 * np.sqrt(2 * g.items), f"chi-square {chi2:.0f} for {g.items} bins"
    assert abs(counts / expect - 1.0).max() < 0.5


# ---- DataHandler.py:145-228: the three dataset classes built the way the reference's LoadData() builds them ---------------------------------
def test_dataset_classes_in_the_reference_positional_forms_drive_an_epoch(dev):
    """TrainData(coomat, config), TestData(testMat, trainMat), DiffusionData(dense tensor, config) -- DataHandler.py:114-128's calls, verbatim --
    replace the handler's own objects; one epoch + evaluation then reproduces the reference's epoch (tests/golden/epoch_mid.npz, recorded draws)."""
    import scipy.sparse as sp
    from diffmm_amd.DataHandler import BatchLoader, DiffusionData, TestData, TrainData
    from test_epoch_gpu import _Draws, _coach
    z = load("epoch_mid.npz")
    coach, h, M = _coach(z)
    cfg, U, I = coach.config, int(z["users"]), int(z["items"])
    trainMat = sp.coo_matrix((np.ones(len(z["trn_rows"]), np.float32), (z["trn_rows"], z["trn_cols"])), shape=(U, I))
    testMat = sp.coo_matrix((np.ones(len(z["tst_rows"]), np.float32), (z["tst_rows"], z["tst_cols"])), shape=(U, I))
    h.trainData = TrainData(trainMat, cfg)
    h.testData = TestData(testMat, trainMat)
    h.diffusionData = DiffusionData(torch.tensor(trainMat.toarray(), dtype=torch.float, device=dev), cfg)
    assert h.trainData.device.type == "cuda" and h.diffusionData.ptr.is_cuda and h.testData._users_d.is_cuda
    h.trainLoader = BatchLoader(h.trainData, int(cfg.train.batch), True, dev)
    h.diffusionLoader = BatchLoader(h.diffusionData, int(cfg.train.batch), True, dev)
    h.trainLoader.fixed_perm = T(z["train_perm"].astype(np.int64))
    h.diffusionLoader.fixed_perm = T(z["diff_perm"].astype(np.int64))
    h.trainData.set_negs(z["ep0/negs"])
    res = coach.trainEpoch(draws=_Draws(z))
    for k, v in res.items():
        close(v, z[f"ep0/train/{k}"], rtol=5e-3, atol=1e-5, what=f"train {k}")
    for k, v in coach.testEpoch().items():
        close(v, z[f"ep0/test/{k}"], rtol=0.02, atol=2e-3, what=f"test {k}")
    d = h.diffusionData.data
    assert d.shape == (U, I) and float(d.sum()) == len(z["trn_rows"])
