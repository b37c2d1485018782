"""CPU oracle for the DiffMM training hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product (diffmm_amd/) never does: it fails loudly when the HIP library is missing.

What it is: a from-scratch restatement (torch-CPU fp32/fp64 + numpy, autograd for gradients) of the
algorithm the reference runs per training step, one function per row of SURVEY.md section 8(a).
Every function cites the reference lines it follows.  It is written functionally (explicit parameter
dicts, explicit random draws, CSR adjacency) -- nothing here is copied from the reference.

Pinning: tests/test_oracle_golden.py checks every function against golden vectors captured by running
the reference itself in the build container (tests/golden/make_golden.py; the reference publishes no
tests or vectors of its own).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
MODAL = ("image", "text", "audio")


# ================================================================================================
# a1/a2  adjacency:  D^-1/2 (B + I) D^-1/2  of the bipartite block matrix        DataHandler.py:52-93
# ================================================================================================
def build_adj_csr(rows: np.ndarray, cols: np.ndarray, users: int, items: int):
    """Edge list (user, item) -> CSR (rowptr int32 [N+1], col int32 [nnz], val fp32 [nnz]) of the
    symmetric-normalised, self-looped, binarised bipartite adjacency (DataHandler.py:80-85 builds
    [[0,R],[R^T,0]], binarises, adds I; :59-65 scales by deg^-1/2 on both sides in float64; :89 casts fp32).
    Columns are sorted inside each row."""
    n = users + items
    r = np.asarray(rows, dtype=np.int64)
    c = np.asarray(cols, dtype=np.int64) + users
    key = np.unique(np.concatenate([r * n + c, c * n + r, np.arange(n, dtype=np.int64) * (n + 1)]))
    rr, cc = key // n, key % n
    deg = np.bincount(rr, minlength=n).astype(np.float64)  # entries are 1.0 so the row sum is the count
    dinv = np.where(deg > 0, deg ** -0.5, 0.0)
    val = ((dinv[rr] * 1.0) * dinv[cc]).astype(np.float32)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rr, minlength=n), out=rowptr[1:])
    return rowptr.astype(np.int32), cc.astype(np.int32), val


def csr_to_torch(rowptr, col, val) -> Tensor:
    n = len(rowptr) - 1
    row = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr.astype(np.int64)))
    idx = torch.from_numpy(np.stack([row, col.astype(np.int64)]))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(np.asarray(val, dtype=np.float32)), (n, n)).coalesce()


def spmm(adj: Tensor, x: Tensor) -> Tensor:
    """Y = A X (Model.py:90,93,105,111,114,123,130; Main.py:319)."""
    return torch.sparse.mm(adj, x)


# ================================================================================================
# a9/a13  noise schedule tables (float64)                                           Model.py:239-275,380-383
# ================================================================================================
def schedule(T: int, noise_scale: float, noise_min: float, noise_max: float, beta_fixed: bool = True) -> Dict[str, np.ndarray]:
    var = np.linspace(noise_scale * noise_min, noise_scale * noise_max, T, dtype=np.float64)  # :240-242
    keep = 1.0 - var
    betas = np.empty(T, dtype=np.float64)
    betas[0] = 1.0 - keep[0]
    for i in range(1, T):
        betas[i] = min(1.0 - keep[i] / keep[i - 1], 0.999)  # :248
    if beta_fixed:
        betas[0] = 0.0001  # :235 -- note: stored through a torch float64 tensor, exact
    alphas = 1.0 - betas
    ac = np.cumprod(alphas)
    ac_prev = np.concatenate([[1.0], ac[:-1]])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.concatenate([post_var[1:2], post_var[1:]])),  # :270
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),  # :273
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),  # :275
        "snr": ac / (1.0 - ac + 1e-8),  # :383
    }


def loss_weight(sched: Dict[str, np.ndarray], t: np.ndarray) -> np.ndarray:
    """SNR(t-1) - SNR(t), and 1 where t == 0 (Model.py:410-412); float64."""
    t = np.asarray(t, dtype=np.int64)
    w = sched["snr"][np.maximum(t - 1, 0)] - sched["snr"][t]
    return np.where(t == 0, 1.0, w)


# ================================================================================================
# a10  q_sample                                                                          Model.py:324-355
# ================================================================================================
def q_sample(x0: Tensor, t: Tensor, noise: Tensor, sched) -> Tensor:
    a = torch.from_numpy(sched["sqrt_alphas_cumprod"])[t].float().unsqueeze(-1)  # :352 fp64 -> fp32
    b = torch.from_numpy(sched["sqrt_one_minus_alphas_cumprod"])[t].float().unsqueeze(-1)
    return a * x0 + b * noise


def default_noise(x0: Tensor, randn: Tensor) -> Tensor:
    """Noise used when forward_cal_xt is called without one (rebuild with sampling_step > 0; Model.py:337)."""
    return torch.sign(x0) * F.normalize(randn)


# ================================================================================================
# a8  Denoise MLP                                                                        Model.py:183-220
# ================================================================================================
def time_embedding(t: Tensor, te: int) -> Tensor:
    half = te // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)  # :196
    ang = t.unsqueeze(-1).float() * freqs.unsqueeze(0)
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if te % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def denoise_forward(P: Dict[str, Tensor], x_t: Tensor, t: Tensor, feat: Optional[Tensor] = None) -> Tensor:
    te = P["emb_layer.weight"].shape[0]
    temb = F.linear(time_embedding(t, te), P["emb_layer.weight"], P["emb_layer.bias"])  # :202
    if feat is not None:  # modality gate, :204-208
        proj = x_t @ feat
        gate = torch.sigmoid(F.linear(proj, P["gate_layer.weight"], P["gate_layer.bias"]))
        x_t = x_t + (proj * gate) @ feat.t()
    h = torch.cat([x_t, temb], dim=-1)
    n_in = sum(1 for k in P if k.startswith("in_layers.") and k.endswith(".weight"))
    n_out = sum(1 for k in P if k.startswith("out_layers.") and k.endswith(".weight"))
    for j in range(n_in):  # :210-212 -- tanh after every in-layer
        h = torch.tanh(F.linear(h, P[f"in_layers.{j}.weight"], P[f"in_layers.{j}.bias"]))
    for j in range(n_out):  # :213-218 -- tanh on all but the last out-layer
        h = F.linear(h, P[f"out_layers.{j}.weight"], P[f"out_layers.{j}.bias"])
        if j != n_out - 1:
            h = torch.tanh(h)
    return h


# ================================================================================================
# a11/a12  reverse step / reverse chain                                                  Model.py:300-322,357-378
# ================================================================================================
def p_mean(P, x_t: Tensor, t: Tensor, sched) -> Tuple[Tensor, Tensor]:
    pred = denoise_forward(P, x_t, t, None)  # :365 -- no modal_feat => no gate
    c1 = torch.from_numpy(sched["posterior_mean_coef1"])[t].float().unsqueeze(-1)
    c2 = torch.from_numpy(sched["posterior_mean_coef2"])[t].float().unsqueeze(-1)
    logvar = torch.from_numpy(sched["posterior_log_variance_clipped"])[t].float().unsqueeze(-1).expand_as(x_t)
    return c1 * pred + c2 * x_t, logvar


def generate_view(P, x0: Tensor, sched, sampling_step: int = 0, randn: Optional[Tensor] = None) -> Tensor:
    T = len(sched["betas"])
    B = x0.shape[0]
    if sampling_step == 0:
        x_t = x0
    else:
        tt = torch.full((B,), sampling_step - 1, dtype=torch.long)
        x_t = q_sample(x0, tt, default_noise(x0, randn), sched)
    for i in range(T - 1, -1, -1):  # deterministic: the posterior mean is fed back, no noise added (:318-321)
        x_t, _ = p_mean(P, x_t, torch.full((B,), i, dtype=torch.long), sched)
    return x_t


# ================================================================================================
# a17 / a16 / a15  losses                                                              Utils/Utils.py:45-98
# ================================================================================================
def l2_reg(reg: float, embs: Sequence[Tensor]) -> Tensor:
    tot = torch.zeros((), dtype=torch.float32)
    for e in embs:
        tot = tot + (e * e).sum()
    return tot * reg


def bpr(u: Tensor, p: Tensor, n: Tensor) -> Tensor:
    diff = (u * p).sum(1) - (u * n).sum(1)
    return (-torch.log(1e-5 + torch.sigmoid(diff))).mean()  # "10e-6" in the source, Utils.py:96


def infonce(v1: Tensor, v2: Tensor, idx: Tensor, temp: float, b_cos: bool = True) -> Tensor:
    a, b = v1[idx], v2[idx]
    if b_cos:  # Utils.py:71-72
        a, b = F.normalize(a, p=2, dim=1), F.normalize(b, p=2, dim=1)
    logits = (a @ b.t()) / temp
    return -(torch.diagonal(F.log_softmax(logits, dim=1))).mean()


# ================================================================================================
# a14  diffusion training loss                                                           Model.py:385-428
# ================================================================================================
def training_losses(P, x0: Tensor, i_embs: Tensor, feat: Tensor, t: Tensor, noise: Tensor, sched,
                    sim_weight: float, reg: float) -> Tensor:
    x_t = q_sample(x0, t, noise, sched)
    out = denoise_forward(P, x_t, t, feat)
    mse = ((out - x0) ** 2).mean(dim=-1)
    w = torch.from_numpy(loss_weight(sched, t.numpy()))  # float64 -> the result is float64 (:411-413)
    rec = w * mse
    sim = 1.0 - F.cosine_similarity(out @ feat, x0 @ i_embs, dim=-1)  # :416-418
    r = l2_reg(reg, [i_embs]) * reg  # reg applied twice: :421 and :425
    return rec + sim * sim_weight + r


# ================================================================================================
# a5/a6  multi-modal graph propagation                                                   Model.py:47-134
# ================================================================================================
def modality_names(M: int) -> Tuple[str, ...]:
    return MODAL[:M] if M == 3 else ("image", "text")


def modal_feats(P, feats: Sequence[Tensor]) -> List[Tensor]:
    names = modality_names(len(feats))
    return [F.linear(e, P[f"{n}_layer.weight"], P[f"{n}_layer.bias"]) for n, e in zip(names, feats)]


def gcn_mm(P, feats: Sequence[Tensor], adj: Tensor, modal_adj: Sequence[Tensor], modal_adj_weight: float,
           residual_weight: float) -> Dict[str, Tensor]:
    M = len(feats)
    U = P["u_embs"].shape[0]
    names = modality_names(M)
    w = torch.softmax(P["modal_weight"], dim=-1)  # :87
    proj = modal_feats(P, feats)
    z = [spmm(modal_adj[m], torch.cat([P["u_embs"], F.normalize(proj[m])])) for m in range(M)]  # :89-93,104-105
    bx = spmm(adj, torch.cat([P["u_embs"], P["i_embs"]]))  # the reference computes this M times (:110-114,122-123)
    mx = sum(w[m] * (bx + modal_adj_weight * z[m]) for m in range(M))  # :116-127
    # :129-131 -- final_embs aliases modal_embs, so the residual line adds rw * (mx + A mx):
    fin = (1.0 + residual_weight) * (mx + spmm(adj, mx))
    out = {"u_final_embs": fin[:U], "i_final_embs": fin[U:]}
    for m, n in enumerate(names):
        out[f"u_{n}_embs"], out[f"i_{n}_embs"] = z[m][:U], z[m][U:]
    return out


# ================================================================================================
# a20  joint (BPR + contrastive) step loss                                               Main.py:292-373
# ================================================================================================
def joint_loss(P, feats, adj, modal_adj, users: Tensor, pos: Tensor, neg: Tensor, rand3: Sequence[Tensor], hp: dict):
    M = len(feats)
    U = P["u_embs"].shape[0]
    g = gcn_mm(P, feats, adj, modal_adj, hp["modal_adj_weight"], hp["residual_weight"])
    uf, itf = g["u_final_embs"], g["i_final_embs"]
    rec = bpr(uf[users], itf[pos], itf[neg])  # :305-309
    reg = l2_reg(hp["reg"], [P["u_embs"], P["i_embs"]])  # :310
    x = torch.cat([P["u_embs"], P["i_embs"]], dim=0)
    layers = []
    for k in range(3):  # :318-324
        x = spmm(adj, x)
        x = x + torch.sign(x) * F.normalize(rand3[k]) * hp["noise_degree"]
        layers.append(x)
    cl1 = torch.stack(layers).mean(dim=0)
    cl2 = layers[0]
    cl = (infonce(cl1[:U], cl2[:U], users, hp["cross_cl_temp"]) + infonce(cl1[U:], cl2[U:], pos, hp["cross_cl_temp"])) * hp["cross_cl_rate"]
    names = modality_names(M)
    if hp["cl_method"] == 1:  # pairwise modality-vs-modality, :347-349 / :361
        pairs = [(a, b) for i, a in enumerate(names) for b in names[i + 1:]]
        for a, b in pairs:
            cl = cl + (infonce(g[f"u_{a}_embs"], g[f"u_{b}_embs"], users, hp["modal_cl_temp"])
                       + infonce(g[f"i_{a}_embs"], g[f"i_{b}_embs"], pos, hp["modal_cl_temp"])) * hp["modal_cl_rate"]
    else:  # main view as the anchor, :353-355 / :365-366
        for n in names:
            cl = cl + (infonce(uf, g[f"u_{n}_embs"], users, hp["modal_cl_temp"])
                       + infonce(itf, g[f"i_{n}_embs"], pos, hp["modal_cl_temp"])) * hp["modal_cl_rate"]
    return rec + reg + cl, {"bpr": rec, "reg": reg, "cl": cl, "gcn": g}


# ================================================================================================
# a21  Adam + cosine schedule                                                      Main.py:92-110,59-65
# ================================================================================================
class Adam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) restated."""

    def __init__(self, params: Dict[str, Tensor], lr: float):
        self.p, self.lr, self.t = params, lr, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, grads: Dict[str, Tensor]):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        c1, c2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:
                    continue
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(c2)).add_(eps)
                self.p[k].addcdiv_(self.m[k], denom, value=-self.lr / c1)


def cosine_lr(base_lr: float, epoch: int, t_max: int, eta_min: float = 1e-4) -> float:
    """Closed form of CosineAnnealingLR after `epoch` scheduler steps (Main.py:93)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max)) / 2.0


# ================================================================================================
# a19  rebuild: per-user top-k (k = degree) of the denoised rows -> edge list             Main.py:211-253
# ================================================================================================
def topk_edges(denoised: Tensor, user_ids: np.ndarray, deg: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    rows, cols = [], []
    for j, u in enumerate(user_ids):
        k = int(deg[u])
        if k:
            cols.append(torch.topk(denoised[j], k=k).indices.numpy())
            rows.append(np.full(k, u, dtype=np.int64))
    if not rows:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    return np.concatenate(rows).astype(np.int32), np.concatenate(cols).astype(np.int32)


# ================================================================================================
# evaluation                                                                            Main.py:390-448
# ================================================================================================
def evaluate(u_final: Tensor, i_final: Tensor, trn_rows, trn_cols, tst_rows, tst_cols, topk: int, batch: int = 1024):
    U, I = u_final.shape[0], i_final.shape[0]
    test_users = np.unique(tst_rows)
    order = np.argsort(tst_rows, kind="stable")
    sr, sc = np.asarray(tst_rows)[order], np.asarray(tst_cols)[order]
    starts = np.searchsorted(sr, test_users, side="left")
    ends = np.searchsorted(sr, test_users, side="right")
    mask = torch.zeros((U, I), dtype=torch.float32)
    mask[torch.from_numpy(np.asarray(trn_rows, dtype=np.int64)), torch.from_numpy(np.asarray(trn_cols, dtype=np.int64))] = 1.0
    rec = ndcg = prec = 0.0
    for s in range(0, len(test_users), batch):
        usr = torch.from_numpy(test_users[s:s + batch].astype(np.int64))
        mk = mask[usr]
        score = (u_final[usr] @ i_final.t()) * (1 - mk) - mk * 1e8  # :410
        top = torch.topk(score, topk).indices.numpy()
        for j in range(len(usr)):
            k = s + j
            truth = sc[starts[k]:ends[k]]
            n = len(truth)
            ideal = sum(1.0 / math.log2(l + 2) for l in range(min(n, topk)))
            hit_pos = [int(np.flatnonzero(top[j] == it)[0]) for it in truth if it in top[j]]
            rec += len(hit_pos) / n
            ndcg += sum(1.0 / math.log2(p + 2) for p in hit_pos) / ideal
            prec += len(hit_pos) / topk
    n_u = len(test_users)
    return {"Recall": rec / n_u, "NDCG": ndcg / n_u, "Precision": prec / n_u}


# ================================================================================================
# a18-a20  a whole epoch, every random draw supplied by the caller                      Main.py:136-388
# ================================================================================================
class Draws:
    """Source of the hot path's random draws, in the reference's call order: per diffusion batch and
    modality randint(0,T,(B,)) then randn(B,I) (Model.py:397,400); per joint batch 3 x rand(N,d) (Main.py:320)."""

    def __init__(self, seed: Optional[int] = None, stored: Optional[List[np.ndarray]] = None):
        self.gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.stored, self.i = stored, 0
        self.checks: List[float] = []

    def _next(self, make):
        if self.stored is not None:
            r = torch.from_numpy(np.asarray(self.stored[self.i]))
        else:
            r = make()
        self.i += 1
        self.checks.append(float(r.double().sum()))
        return r

    def randint(self, hi: int, n: int) -> Tensor:
        return self._next(lambda: torch.randint(0, hi, (n,), generator=self.gen))

    def randn(self, shape) -> Tensor:
        return self._next(lambda: torch.randn(shape, generator=self.gen, dtype=torch.float32))

    def rand(self, shape) -> Tensor:
        return self._next(lambda: torch.rand(shape, generator=self.gen, dtype=torch.float32))


class EpochState:
    """Parameters + optimiser state of one training run (what Coach.prepareModel builds, Main.py:85-110)."""

    def __init__(self, model: Dict[str, Tensor], dens: List[Dict[str, Tensor]], lr: float, epochs_cfg: int):
        self.model = {k: v.clone().requires_grad_() for k, v in model.items()}
        self.dens = [{k: v.clone().requires_grad_() for k, v in d.items()} for d in dens]
        self.opt = Adam(self.model, lr)
        self.den_opt = [Adam(d, lr) for d in self.dens]
        self.base_lr, self.epochs_cfg, self.epoch = lr, epochs_cfg, 0

    def step_schedulers(self):
        self.epoch += 1
        lr = cosine_lr(self.base_lr, self.epoch, self.epochs_cfg)
        self.opt.lr = lr
        for o in self.den_opt:
            o.lr = lr


def _grads(loss: Tensor, params: Dict[str, Tensor]) -> Dict[str, Tensor]:
    keys = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in keys], allow_unused=True)
    return dict(zip(keys, gs))


def diffusion_step(st: EpochState, feats, x0: Tensor, draws: Draws, sched, hp) -> List[float]:
    """One iteration of Main.py:145-192.  Returns the per-modality mean losses."""
    M = len(feats)
    proj = [p.detach() for p in modal_feats(st.model, feats)]  # :150-151,166-168
    means = []
    for m in range(M):
        t = draws.randint(hp["T"], x0.shape[0]).long()
        noise = draws.randn(tuple(x0.shape))
        means.append(training_losses(st.dens[m], x0, st.model["i_embs"].detach(), proj[m], t, noise, sched,
                                     hp["sim_weight"], hp["reg"]).mean())
    total = sum(float(v.detach()) for v in means)
    # :175-176: batch loss = (sum of the means) / total, value == 1; each denoiser only sees its own term
    for m in range(M):
        st.den_opt[m].step(_grads(means[m] / total, st.dens[m]))
    return [float(v.detach()) for v in means]


def rebuild(st: EpochState, dense: Tensor, batches: Iterable[np.ndarray], deg: np.ndarray, sched, users: int, items: int,
            sampling_step: int = 0):
    """Main.py:195-253: reverse chain over all users, top-k (k = degree), one adjacency per modality."""
    M = len(st.dens)
    edges = [([], []) for _ in range(M)]
    with torch.no_grad():
        for idx in batches:
            x0 = dense[torch.from_numpy(np.asarray(idx, dtype=np.int64))]
            for m in range(M):
                den = generate_view(st.dens[m], x0, sched, sampling_step)
                r, c = topk_edges(den, idx, deg)
                edges[m][0].append(r); edges[m][1].append(c)
    out = []
    for m in range(M):
        r = np.concatenate(edges[m][0]); c = np.concatenate(edges[m][1])
        out.append(build_adj_csr(r, c, users, items))
    return out


def joint_step(st: EpochState, feats, adj, modal_adj, users, pos, neg, draws: Draws, hp):
    """One iteration of Main.py:292-377."""
    N = st.model["u_embs"].shape[0] + st.model["i_embs"].shape[0]
    d = st.model["u_embs"].shape[1]
    rand3 = [draws.rand((N, d)) for _ in range(3)]
    loss, parts = joint_loss(st.model, feats, adj, modal_adj, users, pos, neg, rand3, hp)
    st.opt.step(_grads(loss, st.model))
    return float(loss.detach()), float(parts["bpr"].detach()), float(parts["reg"].detach()), float(parts["cl"].detach())


def train_epoch(st: EpochState, data: dict, draws: Draws, sched, hp, modal_adj_override=None):
    """Coach.trainEpoch restated (Main.py:136-388).  `data`: users, items, trn_rows/cols, feats (list of
    Tensor), negs, train_perm, diff_perm, B.  Returns (result dict, modal adjacency CSR list)."""
    U, I, B = data["users"], data["items"], data["B"]
    feats = data["feats"]
    M = len(feats)
    rows, cols, negs = data["trn_rows"], data["trn_cols"], data["negs"]
    dense = data["dense"]
    deg = np.bincount(rows, minlength=U)
    adj = data["adj"]
    diff_batches = [data["diff_perm"][s:s + B] for s in range(0, U, B)]
    dl = [0.0] * M
    for idx in diff_batches:  # phase 1
        x0 = dense[torch.from_numpy(np.asarray(idx, dtype=np.int64))]
        means = diffusion_step(st, feats, x0, draws, sched, hp)
        tot = sum(means)
        for m in range(M):  # the logging quirk of Main.py:155,177-179: accumulate, then divide the running sum
            dl[m] += means[m]
        for m in range(M):
            dl[m] /= tot
    csr_list = rebuild(st, dense, diff_batches, deg, sched, U, I, hp.get("sampling_step", 0))  # phase 2
    if modal_adj_override is not None:
        csr_list = modal_adj_override
    modal_adj = [csr_to_torch(*c) for c in csr_list]
    perm = data["train_perm"]
    acc = np.zeros(4)
    for s in range(0, len(perm), B):  # phase 3
        i = perm[s:s + B]
        u = torch.from_numpy(rows[i].astype(np.int64)); p = torch.from_numpy(cols[i].astype(np.int64))
        n = torch.from_numpy(negs[i].astype(np.int64))
        acc += np.array(joint_step(st, feats, adj, modal_adj, u, p, n, draws, hp))
    train_steps = len(perm) // B
    diffusion_steps = U // B
    res = {"Loss": acc[0] / train_steps, "BPR Loss": acc[1] / train_steps, "reg loss": acc[2] / train_steps,
           "CL loss": acc[3] / train_steps}
    for m, n in enumerate(modality_names(M)):
        res[f"{n} loss"] = dl[m] / diffusion_steps
    return res, csr_list


def test_epoch(st: EpochState, data: dict, modal_csr, hp, topk: int):
    with torch.no_grad():
        g = gcn_mm(st.model, data["feats"], data["adj"], [csr_to_torch(*c) for c in modal_csr],
                   hp["modal_adj_weight"], hp["residual_weight"])
        return evaluate(g["u_final_embs"], g["i_final_embs"], data["trn_rows"], data["trn_cols"],
                        data["tst_rows"], data["tst_cols"], topk)
