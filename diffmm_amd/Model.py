"""Model-side surface of the reference (Model.py:15-428) on top of the HIP engines.

Model / Denoise / GaussianDiffusion keep the reference's constructor signatures, attribute and method names.
They hold no nn.Parameters: weights live in the engines' flat HBM buffers (engine.FlatParams) and every method
launches HIP kernels.  Parameter initialisation follows the reference's distributions (xavier_uniform embeddings,
nn.Linear default init for the projections and the gate, xavier_normal + N(0, 1e-3) bias for the Denoise MLP).
"""
from __future__ import annotations

import ast
import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor

from . import comm, ops
from .Conf import Config
from .engine import DiffusionEngine, JointEngine, modality_names, schedule_tables
from .graph import Adjacency


def _linear_default_init(out_f: int, in_f: int):
    """torch.nn.Linear's default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
    bound = 1.0 / math.sqrt(in_f)
    return torch.empty(out_f, in_f).uniform_(-bound, bound), torch.empty(out_f).uniform_(-bound, bound)


def hyper_dict(config: Config) -> dict:
    h, t, b = config.hyper, config.train, config.base
    return dict(modal_adj_weight=h.modal_adj_weight, residual_weight=h.residual_weight, noise_degree=h.noise_degree,
                cross_cl_temp=h.cross_cl_temp, cross_cl_rate=h.cross_cl_rate, modal_cl_temp=h.modal_cl_temp, modal_cl_rate=h.modal_cl_rate,
                reg=t.reg, lr=t.lr, cl_method=b.cl_method, seed=b.seed, noise_scale=h.noise_scale, noise_min=h.noise_min,
                noise_max=h.noise_max, sim_weight=h.sim_weight, sampling_step=h.sampling_step)


@dataclass
class GCNOutput:
    u_final_embs: Tensor
    i_final_embs: Tensor
    u_image_embs: Tensor
    i_image_embs: Tensor
    u_text_embs: Tensor
    i_text_embs: Tensor
    u_audio_embs: Optional[Tensor] = None
    i_audio_embs: Optional[Tensor] = None


def joint_shard_mode(world: int, items: int, D: int) -> str:
    """How the joint step is sharded over the replicas (DMM_JOINT_SHARD = 1d | 2d | auto).  "1d" (sharded.py): users sharded, items replicated -- one
    all-reduce of a [I, d] fp32 partial per propagation.  "2d" (sharded2d.py): items sharded as well -- reduce-scatter + all-gather of the gather
    operands, item-side dense passes and Adam state 1 / world.  auto: 2d once a propagation's item partial reaches 64 MiB (I d >= 16M: the exchange
    is then a bandwidth term) and every replica gets a non-empty slice; the small shipped datasets keep the fewer, smaller collectives."""
    import os
    mode = os.environ.get("DMM_JOINT_SHARD", "auto")
    if mode not in ("1d", "2d", "auto"):
        raise ValueError("DMM_JOINT_SHARD must be 1d, 2d or auto")
    fits = (world - 1) * -(-items // world) < items
    if mode == "2d" and not fits:
        raise ValueError("DMM_JOINT_SHARD=2d: a replica's item slice would be empty")
    return "2d" if (mode == "2d" or (mode == "auto" and fits and items * D * 4 >= (64 << 20))) else "1d"


class Model:
    """Multi-modal graph encoder (Model.py:15-134).  `gcn_MM` returns views into the engine's HBM buffers."""

    def __init__(self, config: Config, image_embedding: Tensor, text_embedding: Tensor, audio_embedding: Optional[Tensor] = None,
                 batch_max: Optional[int] = None, precision: int = 1, layout=None, group=None):
        """layout (diffmm_amd.sharded.ShardLayout, world > 1): user-sharded replica -- the engine keeps this replica's user rows and the
        replicated item side; the initial weights are drawn for the FULL model from the (identically seeded) host generator on every
        replica, so the replicas start from the same model the single-GPU run starts from."""
        self.config = config
        self.device = torch.device(f"cuda:{config.base.gpu}" if torch.cuda.is_available() else "cpu")
        feats = [image_embedding, text_embedding] + ([audio_embedding] if audio_embedding is not None else [])
        U, I, D = int(config.data.user_num), int(config.data.item_num), int(config.base.latdim)
        if layout is not None and comm.active(layout.world):
            from .sharded import ShardedJointEngine
            from .sharded2d import Sharded2DJointEngine
            Engine = Sharded2DJointEngine if joint_shard_mode(layout.world, I, D) == "2d" else ShardedJointEngine
            self.engine = Engine(layout, D, feats, hyper_dict(config), self.device, int(batch_max or config.train.batch), group=group)
        else:
            self.engine = JointEngine(U, I, D, feats, hyper_dict(config), self.device, int(batch_max or config.train.batch))
        self.precision = precision
        self.image_embedding, self.text_embedding, self.audio_embedding = image_embedding, text_embedding, audio_embedding
        sd = {"u_embs": torch.nn.init.xavier_uniform_(torch.empty(U, D)), "i_embs": torch.nn.init.xavier_uniform_(torch.empty(I, D))}
        for n, f in zip(modality_names(len(feats)), feats):
            sd[f"{n}_layer.weight"], sd[f"{n}_layer.bias"] = _linear_default_init(D, int(f.shape[1]))
        sd["modal_weight"] = torch.tensor([0.3333, 0.3333, 0.3333]) if len(feats) == 3 else torch.tensor([0.5, 0.5])
        self.engine.load_reference_state(sd)

    def cuda(self, *_a, **_k):
        return self

    @property
    def u_embs(self) -> Tensor:
        return self.engine.u_embs

    @property
    def i_embs(self) -> Tensor:
        return self.engine.i_embs

    @property
    def modal_weight(self) -> Tensor:
        return self.engine.P.view("modal_weight")

    def getItemEmbs(self) -> Tensor:
        return self.engine.i_embs

    def getUserEmbs(self) -> Tensor:
        return self.engine.u_embs

    def _feat(self, m: int) -> Tensor:
        n = modality_names(self.engine.M)[m]
        return ops.gemm(self.engine.feats[m], self.engine.P.view(f"{n}_layer.weight"), trans_b=True, bias=self.engine.P.view(f"{n}_layer.bias"),
                        precision=self.precision)

    def getImageFeats(self) -> Tensor:
        return self._feat(0)

    def getTextFeats(self) -> Tensor:
        return self._feat(1)

    def getAudioFeats(self) -> Optional[Tensor]:
        return self._feat(2) if self.engine.M == 3 else None

    def gcn_MM(self, adj: Adjacency, image_adj: Adjacency, text_adj: Adjacency, audio_adj: Optional[Adjacency] = None) -> GCNOutput:
        madj = [image_adj, text_adj] + ([audio_adj] if audio_adj is not None else [])
        self.engine.set_adjacency(adj, madj)
        self.engine.forward(with_cl=False, precision=self.precision)
        return GCNOutput(**self.engine.gcn_output())

    def state_dict(self):
        return self.engine.state_dict()


class Denoise:
    """Denoise MLP handle (Model.py:136-220): one slot of a DiffusionEngine."""

    def __init__(self, in_dims: List[int], out_dims: List[int], config: Config, dropout: float = 0.5, _engine: Optional[DiffusionEngine] = None,
                 _slot: int = 0, precision: int = 1):
        self.in_dims, self.out_dims = in_dims, out_dims
        self.time_emb_dim = int(config.base.d_emb_size)
        self.device = torch.device(f"cuda:{config.base.gpu}" if torch.cuda.is_available() else "cpu")
        if len(in_dims) < 2 or list(in_dims) != list(out_dims)[::-1]:
            raise ValueError("Denoise: in_dims must be out_dims reversed, [items, H_k, ..., H_1] (Main.py:96-98)")
        I, H, D = int(in_dims[0]), int(in_dims[1]), int(config.base.latdim)
        hidden = [int(x) for x in out_dims[:-1]]  # base.denoise_dim: [H_1, ..., H_k], H = H_k
        self.precision = precision
        if _engine is None:
            _engine = DiffusionEngine(I, D, H, self.time_emb_dim, 1, int(config.hyper.steps), hyper_dict(config), self.device, int(config.train.batch),
                                      hidden=hidden)
        self.engine, self.slot = _engine, _slot
        te = self.time_emb_dim
        xn = torch.nn.init.xavier_normal_
        lin = lambda o, i: (xn(torch.empty(o, i)), torch.empty(o).normal_(0.0, 0.001))  # init_weights(): xavier_normal / N(0, 1e-3)
        sd = {}
        in_t = [I + te] + [int(x) for x in in_dims[1:]]
        for j, (di, do) in enumerate(zip(in_t[:-1], in_t[1:])):     # Model.py:156-158
            sd[f"in_layers.{j}.weight"], sd[f"in_layers.{j}.bias"] = lin(do, di)
        for j, (di, do) in enumerate(zip(out_dims[:-1], out_dims[1:])):  # Model.py:159-160
            sd[f"out_layers.{j}.weight"], sd[f"out_layers.{j}.bias"] = lin(int(do), int(di))
        sd["emb_layer.weight"], sd["emb_layer.bias"] = lin(te, te)
        sd["gate_layer.weight"], sd["gate_layer.bias"] = _linear_default_init(D, D)  # created after init_weights(): default init
        self.engine.load_reference_state(self.slot, sd)

    def cuda(self, *_a, **_k):
        return self

    def forward(self, x_t: Tensor, timesteps: Tensor, modal_feat: Optional[Tensor] = None) -> Tensor:
        return self.engine.denoise_forward(self.slot, x_t.contiguous(), timesteps.to(torch.int32), modal_feat, precision=self.precision).clone()

    __call__ = forward

    def state_dict(self):
        return self.engine.state_dict(self.slot)

    def grads(self):
        """Gradients the last backward pass left for this Denoise (views into the engine's flat buffer, reference parameter names)."""
        P = self.engine.P[self.slot]
        g = {k: P.g(k) for k in P.names}
        g["in_layers.0.weight"] = g["in_layers.0.weight"][:, : self.engine.I + self.engine.te]
        return g


class GaussianDiffusion:
    """Noise schedule + forward / reverse process arithmetic (Model.py:222-428)."""

    def __init__(self, config: Config, beta_fixed: bool = True):
        self.config = config
        self.device = torch.device(f"cuda:{config.base.gpu}" if torch.cuda.is_available() else "cpu")
        self.noise_scale, self.noise_min, self.noise_max = config.hyper.noise_scale, config.hyper.noise_min, config.hyper.noise_max
        self.steps = int(config.hyper.steps)
        self.beta_fixed = beta_fixed
        if self.noise_scale != 0:
            self.betas = torch.tensor(self.get_betas(), dtype=torch.float64, device=self.device)
            if beta_fixed:
                self.betas[0] = 0.0001
            self.calculate_for_diffusion()

    def cuda(self, *_a, **_k):
        return self

    def get_betas(self) -> np.ndarray:
        return schedule_tables(self.steps, self.noise_scale, self.noise_min, self.noise_max, beta_fixed=False)["betas"]

    def calculate_for_diffusion(self) -> None:
        t = schedule_tables(self.steps, self.noise_scale, self.noise_min, self.noise_max, self.beta_fixed)
        for k, v in t.items():
            if k not in ("betas", "snr"):
                setattr(self, k, torch.tensor(v, dtype=torch.float64, device=self.device))
        self._snr = torch.tensor(t["snr"], dtype=torch.float64, device=self.device)

    def _extract_into_tensor(self, var: Tensor, timesteps: Tensor, broadcast_shape) -> Tensor:
        res = var[timesteps.long()].float()
        while res.dim() < len(broadcast_shape):
            res = res.unsqueeze(-1)
        return res.expand(broadcast_shape)

    def SNR(self, t: Tensor) -> Tensor:
        return self._snr[t.long()]

    def _f32_table(self, name: str) -> Tensor:
        """fp32 device copy of a schedule table (the reference extracts with .float(), Model.py:352)."""
        cache = self.__dict__.setdefault("_f32_tables", {})
        if name not in cache:
            cache[name] = getattr(self, name).float().contiguous()
        return cache[name]

    def forward_cal_xt(self, x_0: Tensor, timesteps: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        """x_t = sqrt(alpha_bar_t) x_0 + sqrt(1 - alpha_bar_t) noise (Model.py:324-341) on dmm_rows_axpby; noise=None draws the
        reference's default noise sign(x_0) * normalize(randn) with dmm_noised_start (in-kernel Philox)."""
        x_0 = x_0.contiguous().float()
        t32 = timesteps.to(torch.int32).contiguous()
        if noise is None:
            self._default_noise_calls = getattr(self, "_default_noise_calls", 0) + 1
            noise = torch.empty_like(x_0)
            ops.noised_start(x_0, None, 0.0, 1.0, noise, x_0.shape[1], seed=int(self.config.base.seed) + 2,
                             offset=self._default_noise_calls * x_0.shape[0] * ((x_0.shape[1] + 3) // 4))
        return ops.rows_axpby(x_0, noise.contiguous().float(), t32, self._f32_table("sqrt_alphas_cumprod"), self._f32_table("sqrt_one_minus_alphas_cumprod"))

    def p_mean_variance(self, denoise: Denoise, x_t: Tensor, timesteps: Tensor, noise: Optional[Tensor] = None):
        """(posterior mean, clipped log-variance) of Model.py:357-378: Denoise.forward without modal features, then
        coef1[t] * pred + coef2[t] * x_t (dmm_rows_axpby)."""
        x_t = x_t.contiguous().float()
        t32 = timesteps.to(torch.int32).contiguous()
        pred = denoise.forward(x_t, t32)
        mean = ops.rows_axpby(pred, x_t, t32, self._f32_table("posterior_mean_coef1"), self._f32_table("posterior_mean_coef2"))
        return mean, self._extract_into_tensor(self.posterior_log_variance_clipped, timesteps, x_t.shape)

    def generate_view(self, model: Denoise, x_start: Tensor, sampling_step: int) -> Tensor:
        if sampling_step != 0:
            ts = torch.full((x_start.shape[0],), sampling_step - 1, device=x_start.device)
            x_start = self.forward_cal_xt(x_start, ts)
        return model.engine.generate_view(model.slot, x_start.contiguous(), precision=model.precision).clone()

    def training_losses(self, model: Denoise, x_start: Tensor, i_embs: Tensor, modal_feat: Tensor, timesteps: Optional[Tensor] = None,
                        noise: Optional[Tensor] = None) -> Tensor:
        """Per-row loss vector (float64, like the reference; Model.py:385-428).  Differentiable for a UNIFORM upstream gradient -- the
        reference only ever reduces it with .mean() and scalar factors (Main.py:155-176): `.backward()` then leaves the Denoise
        gradients in `model.grads()` (the engine's flat gradient buffer) and returns d/d i_embs through autograd; a non-uniform
        upstream gradient raises.  (The training loop itself uses DiffusionEngine.train_step, which batches the modalities.)
        `timesteps` / `noise` (optional) replace the random draws."""
        B = x_start.shape[0]
        t = (torch.randint(0, self.steps, (B,), device=x_start.device) if timesteps is None else timesteps).to(torch.int32).contiguous()
        return _TrainingLosses.apply(i_embs, self, model, x_start, modal_feat, t, noise)


class _TrainingLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, i_embs: Tensor, gd: "GaussianDiffusion", model: Denoise, x_start: Tensor, modal_feat: Tensor, t: Tensor, noise):
        E, m, B = model.engine, model.slot, x_start.shape[0]
        ie, feat = i_embs.detach().contiguous().float(), modal_feat.detach().contiguous().float()
        x0p = E._set_x0(x_start.detach().contiguous().float())
        reg_term = E.reg_term(ie)
        E.loss_forward(m, x0p, t, noise.contiguous().float() if noise is not None else None, feat, ie, reg_term, model.precision)
        ctx.args = (E, m, B, ie, feat, reg_term, model.precision)
        return E.act[m]["loss"][:B].clone()

    @staticmethod
    def backward(ctx, gout: Tensor):
        E, m, B, ie, feat, reg_term, precision = ctx.args
        g0 = float(gout[0])
        if not bool((gout == gout[0]).all()):
            raise NotImplementedError("training_losses: only uniform upstream gradients (loss.mean(), scalar factors) are supported")
        x0p = E.x0p[:B]  # still holds x_start: nothing ran on the engine between forward and backward
        E.loss_backward(m, x0p, feat, reg_term, g0, precision)
        # d/d i_embs: through ui = x_0 @ i_embs in the cosine term (symmetric in um / ui: the same kernel with the roles swapped) and
        # through the reg^2 * ||i_embs||^2 term every row carries
        a = E.act[m]
        dui = torch.empty_like(E.ui[:B])
        scratch = torch.empty(B, dtype=torch.float64, device=ie.device)
        ops.diff_loss(a["out"][:B], x0p, a["w"], E.ui[:B], a["um"][:B], E.I, E.hp["sim_weight"], reg_term, g0, scratch, None, dui)
        g_ie = ops.gemm(x0p, dui, trans_a=True, precision=3, M=E.I)
        reg = float(E.hp["reg"])
        g_ie = g_ie + (2.0 * reg * reg * g0 * B) * ie
        return g_ie, None, None, None, None, None, None


def denoise_dims(config: Config):
    out_dims = ast.literal_eval(config.base.denoise_dim) + [int(config.data.item_num)]
    return out_dims[::-1], out_dims
