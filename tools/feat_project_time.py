#!/usr/bin/env python3
"""The joint step's two feature-projection launches alone at the TikTok shape (I = 6710 items, raw widths 128 / 128 / 768, d = 64): forward
Linear_m(E_m) (dmm_item_project_batched, F transposed) and the weight gradient dfeat^T E (out transposed).  Knobs: DMM_PROJ_FWD_SLICES, DMM_PROJ_WGS,
DMM_PROJ_WGS_T (read once per process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops
dev = torch.device("cuda:0")
I, D, Fs = 6710, 64, [128, 128, 768]
X = [torch.randn(I, f, device=dev) for f in Fs]
W = [torch.randn(D, f, device=dev) * 0.1 for f in Fs]
b = [torch.randn(D, device=dev) for _ in Fs]
out = [torch.zeros(I, D, device=dev) for _ in Fs]
Ip = (I + 31) & ~31
XT = [torch.zeros(f, Ip, device=dev) for f in Fs]
for m in range(3): XT[m][:, :I] = X[m].t()
df = [torch.randn(I, D, device=dev) for _ in Fs]
gW = [torch.zeros(D, f, device=dev) for f in Fs]
fwd = lambda: ops.item_project_batched([dict(X=X[m], F=W[m], out=out[m], bias=b[m], K=Fs[m]) for m in range(3)], D, f_transposed=True)
bwd = lambda: ops.item_project_batched([dict(X=XT[m], F=df[m], out=gW[m], K=I) for m in range(3)], D, out_transposed=True)
def timed(f, n=50):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print({k: os.environ.get(k) for k in ("DMM_PROJ_FWD_SLICES", "DMM_PROJ_WGS", "DMM_PROJ_WGS_T")}, f"forward {timed(fwd):.1f} us, weight gradient {timed(bwd):.1f} us")
