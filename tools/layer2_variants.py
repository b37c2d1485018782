#!/usr/bin/env python3
"""Layer-2 launch of the reverse chain (3 modalities x (9308 x 1024) x (6710 x 1024)^T, batched) under its epilogue variants:
what the posterior-mean epilogue costs on top of the MFMA main loop."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
M, R, I, H = 3, 9308, 6710, 1024
Kp = (I + 10 + 63) & ~63
h16 = [torch.randn(R, H, device=dev).bfloat16() for _ in range(M)]
w2 = [(torch.randn(I, H, device=dev) * 0.03).bfloat16() for _ in range(M)]
b2 = [torch.randn(I, device=dev) for _ in range(M)]
x32 = [torch.randn(R, Kp, device=dev) for _ in range(M)]; y32 = [torch.empty(R, Kp, device=dev) for _ in range(M)]
x16 = [t.bfloat16() for t in x32]; y16 = [torch.empty(R, Kp, device=dev, dtype=torch.bfloat16) for _ in range(M)]
variants = {
    "fp32 state (R fp32 in, C fp32 + Cb bf16 out)": lambda m: dict(out=y32[m], out_bf16=y16[m], residual=x32[m]),
    "bf16 state (Rb in, Cb out)": lambda m: dict(out_bf16=y16[m], residual_bf16=x16[m]),
    "Cb out only, no residual": lambda m: dict(out_bf16=y16[m]),
    "C fp32 out only, no residual": lambda m: dict(out=y32[m]),
    "last step (Rb in, C fp32 out)": lambda m: dict(out=y32[m], residual_bf16=x16[m]),
}
for name, f in variants.items():
    with L.Recorder() as plan:
        ops.gemm_bf16_nt_batched([dict(A=h16[m], B=w2[m], bias=b2[m], e1=0.7, e2=0.3, N=I, **f(m)) for m in range(M)])
    L.replay(plan); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.replay(plan)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(json.dumps({"variant": name, "us": round(t * 1e6, 1), "TFLOPs": round(2.0 * R * M * H * I / t / 1e12, 1)}), flush=True)
