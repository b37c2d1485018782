#!/usr/bin/env python3
"""The diffusion step's gate projection x_t F ([4096, 6710] x [6710, 64], three modalities): three dmm_item_project launches (256-row workgroups, as the
step issues them) against ONE dmm_item_project_batched launch (64-row workgroups, pipelined stages), fp32 and bf16 x_t."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops
dev = torch.device("cuda:0")
B, I, D, Kp, M = 4096, 6710, 64, 6720, 3
X = [torch.randn(B, Kp, device=dev) for _ in range(M)]
X16 = [x.to(torch.bfloat16) for x in X]
F = [torch.randn(I, D, device=dev) for _ in range(M)]
out = [torch.zeros(B, D, device=dev) for _ in range(M)]
def three():
    for m in range(M):
        ops.zero_(out[m]); ops.item_project(X[m], F[m], out[m], I)
def one(xs):
    ops.zero_regions(out); ops.item_project_batched([dict(X=xs[m], F=F[m], out=out[m], K=I) for m in range(M)], D)
def timed(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
three(); ref = [o.clone() for o in out]
one(X); err = max(float((o - r).abs().max() / r.abs().max()) for o, r in zip(out, ref))
print(f"three launches {timed(three):.1f} us; one batched launch {timed(lambda: one(X)):.1f} us (max rel diff {err:.1e}); DMM_PROJ_WGS={os.environ.get('DMM_PROJ_WGS')}")
try:
    print(f"one batched launch, bf16 x_t: {timed(lambda: one(X16)):.1f} us")
except Exception as e:
    print("bf16 X in this form:", str(e)[:120])
