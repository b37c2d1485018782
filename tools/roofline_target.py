#!/usr/bin/env python3
"""The two launches bench.py's `roofline` times, alone: 20 x the reverse chain's batched layer-1 GEMM, then 20 x layer 2 (bench shapes:
3 modalities x 9308 users, I = 6710, Kp = 6720, H = 1024).  Profile with
    rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/roofline_target.py
the first 20 dispatches of gemm_bf16_nt_8ph_kernel are layer 1 (the `roofline.achieved` kernel), the last 20 layer 2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
Mc, I, Kp, H = 9308, 6710, 6720, 1024
torch.manual_seed(0)
w1 = [(torch.randn(H, Kp, device=dev) * 0.02).bfloat16() for _ in range(3)]; b1 = torch.zeros(H, device=dev)
w2 = [(torch.randn(I, H, device=dev) * 0.02).bfloat16() for _ in range(3)]; b2 = torch.zeros(I, device=dev)
x16 = [torch.randn(Mc, Kp, device=dev).bfloat16() for _ in range(3)]
h16 = [torch.zeros(Mc, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
x32 = [torch.randn(Mc, Kp, device=dev) for _ in range(3)]
y32 = [torch.zeros(Mc, Kp, device=dev) for _ in range(3)]
y16 = [torch.zeros(Mc, Kp, dtype=torch.bfloat16, device=dev) for _ in range(3)]
l1 = lambda: ops.gemm_bf16_nt_batched([dict(A=x16[m], B=w1[m], out_bf16=h16[m], bias=b1, act=L.ACT_TANH) for m in range(3)])
l2 = lambda: ops.gemm_bf16_nt_batched([dict(A=h16[m], B=w2[m], out=y32[m], out_bf16=y16[m], bias=b2, residual=x32[m], e1=0.9, e2=0.1, N=I) for m in range(3)])
for fn in (l1, l2):
    fn(); torch.cuda.synchronize()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
