#!/usr/bin/env python3
"""Hand-off stress of the persistent latent-chain kernel (MI355X_MICROARCH.md: "test every hand-off under UNEVEN load, checking every word"):
N chains at the bench shape (3 modalities x 9308 rows, T = 40), each compared bit for bit with the first, while a second stream keeps part of
the chip busy with copies of varying size (uneven load, other kernels competing for CUs and L2).  Prints the number of mismatching runs and
the error word."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd.engine import DiffusionEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.manual_seed(1)
I, D, H, M, T_, B = 6710, 64, 1024, 3, 40, 9308
hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=5)
E = DiffusionEngine(I, D, H, 10, M, T_, hp, dev, 4096, chain_rows=B)
for m in range(M):
    E.P[m].data.normal_(0, 0.03)
    E.P[m].view("in_layers.0.weight")[:, I + 10:].zero_()
x0 = (torch.rand(B, I, device=dev) < 0.001).float()
ref = [v.clone() for v in E.generate_views_latent(xT=[x0] * M, B=B)]
side = torch.cuda.Stream()
junk_a, junk_b = torch.randn(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
bad = 0
for it in range(N):
    if it % 2 == 0:  # uneven background load on another stream
        with torch.cuda.stream(side):
            n = (1 + it % 7) << 22
            junk_b[:n].copy_(junk_a[:n]); junk_b[:n].mul_(1.0001)
    out = E.generate_views_latent(xT=[x0] * M, B=B)
    if not all(torch.equal(out[m], ref[m]) for m in range(M)):
        bad += 1
        print("run", it, "differs:", [int((out[m] != ref[m]).sum()) for m in range(M)], flush=True)
torch.cuda.synchronize()
print(f"{N} chains, {bad} differing from the first; error word {int(E._lat['flags'][-1])}")
