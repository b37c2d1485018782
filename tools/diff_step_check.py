#!/usr/bin/env python3
"""Diffusion training step at a latent width the dedicated item-space kernels do not cover (d = 256: generic split-K GEMMs): the bf16 batched
step against the split-precision (~fp32) step, called twice (a gradient segment that is accumulated but not zeroed would double)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd.engine import DiffusionEngine
dev = torch.device("cuda:0")
for D in (64, 256):
    torch.manual_seed(3)
    I, H, M, T_, B = 1000, 256, 3, 6, 512
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=11)
    E = DiffusionEngine(I, D, H, 10, M, T_, hp, dev, B)
    for m in range(M):
        E.P[m].data.normal_(0, 0.05)
        E.P[m].view("in_layers.0.weight")[:, I + 10:].zero_()
    x0 = (torch.rand(B, I, device=dev) < 0.02).float()
    feats = [torch.nn.functional.normalize(torch.randn(I, D, device=dev)) for _ in range(M)]
    ie = torch.randn(I, D, device=dev) * 0.1
    t = [torch.randint(0, T_, (B,), device=dev, dtype=torch.int32) for _ in range(M)]
    noise = [torch.randn(B, I, device=dev) for _ in range(M)]
    l3 = E.train_step(x0, feats, ie, t=t, noise=noise, precision=3, update=False)
    g3 = [E.P[m].grad.clone() for m in range(M)]
    for rep in range(2):
        l1 = E.train_step(x0, feats, ie, t=t, noise=noise, precision=1, update=False)
        for m in range(M):
            P = E.P[m]
            worst = max((float((P.g(k) - g3[m][P.offsets[k]:P.offsets[k] + P.g(k).numel()].view_as(P.g(k))).abs().max()) /
                         (float(g3[m][P.offsets[k]:P.offsets[k] + P.g(k).numel()].abs().max()) + 1e-30), k) for k in P.offsets)
            print(f"D={D} rep={rep} m={m} loss p1 {l1[m]:.6f} p3 {l3[m]:.6f}  worst grad err/scale {worst[0]:.3e} ({worst[1]})")
