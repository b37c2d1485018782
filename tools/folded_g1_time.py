#!/usr/bin/env python3
"""(W1x F)^T [d, H] = F^T [d, I] x W1x^T [I, H] of the folded products (3 modalities) at I items: the item-space kernel (transposed output) against the
generic K-split GEMM.  usage: folded_g1_time.py I D H"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
I, D, H, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 3
Kp = (I + 10 + 63) & ~63
W1 = [torch.randn(H, Kp, device=dev) * 0.02 for _ in range(M)]
F = [torch.randn(I, D, device=dev) for _ in range(M)]
out = torch.zeros(M, D, H, device=dev)
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
res = {"I": I, "D": D, "H": H}
res["item_project_batched(out_transposed) us"] = timed(lambda: (ops.zero_(out), ops.item_project_batched([dict(X=W1[m], F=F[m], out=out[m], K=I) for m in range(M)], D, out_transposed=True)))
ref = out.clone()
for sk in (8, 16, 32, 64):
    res[f"gemm_batched(trans_a, trans_b, split_k={sk}) us"] = timed(lambda: (ops.zero_(out), ops.gemm_batched([dict(A=F[m], B=W1[m], out=out[m], trans_a=True, trans_b=True, precision=1, K=I, split_k=sk) for m in range(M)])))
res["max |diff| / scale"] = float((out - ref).abs().max() / ref.abs().max())
print(json.dumps(res))
