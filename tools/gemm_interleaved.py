#!/usr/bin/env python3
"""The layer-2 GEMM of the scaled diffusion step (8192 x 100000 x 1024, fp32 out) timed alone and alternating with a memory-bound pass
(cast of a [8192, 100032] fp32 tensor): does the GEMM keep its rate between streaming kernels?"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
M, N, K = 8192, 100000, 1024
A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
ld = (N + 63) & ~63
out = torch.empty(M, ld, device=dev); src = torch.randn(M, ld, device=dev); dst = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
def ev(): return torch.cuda.Event(enable_timing=True)
def gemm(): ops.gemm_bf16_nt_batched([dict(A=A, B=B, N=N, out=out)])
def cast(): ops.cast_bf16(src, dst)
gemm(); cast(); torch.cuda.synchronize()
res = {}
for name, seq in (("gemm alone", [gemm]), ("cast alone", [cast]), ("cast, gemm alternating", [cast, gemm]), ("3 casts, gemm", [cast, cast, cast, gemm])):
    tg, tc = [], []
    for it in range(12):
        marks = []
        for f in seq:
            e0, e1 = ev(), ev(); e0.record(); f(); e1.record(); marks.append((f, e0, e1))
        torch.cuda.synchronize()
        for f, e0, e1 in marks:
            (tg if f is gemm else tc).append(e0.elapsed_time(e1) * 1e3)
    res[name] = dict(gemm_us=round(sorted(tg)[len(tg) // 2], 1) if tg else None, cast_us=round(sorted(tc)[len(tc) // 2], 1) if tc else None)
print(json.dumps(res))
