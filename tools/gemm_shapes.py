#!/usr/bin/env python3
"""dmm_gemm_bf16_nt at given shapes (M,N,K[,out]: out = c fp32 | cb bf16 | tanh = fp32 + bf16 with tanh) under the tile / placement knobs
(DMM_GEMM_TILE, DMM_GEMM_PART: read once per process -- run once per setting).  usage: gemm_shapes.py 8192,1024,100032,tanh 8192,100000,1024,c"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    f = spec.split(",")
    M, N, K, kind = int(f[0]), int(f[1]), int(f[2]), (f[3] if len(f) > 3 else "c")
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    ldc = (N + 63) & ~63
    kw = dict(A=A, B=B, N=N, bias=torch.randn(N, device=dev))
    if kind in ("c", "tanh"): kw["out"] = torch.empty(M, ldc, device=dev)
    if kind in ("cb", "tanh"): kw["out_bf16"] = torch.empty(M, ldc, device=dev, dtype=torch.bfloat16)
    if kind == "tanh": kw["act"] = L.ACT_TANH
    with L.Recorder() as plan:
        ops.gemm_bf16_nt_batched([kw])
    L.replay(plan); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.replay(plan)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    print(json.dumps(dict(shape=spec, tile=os.environ.get("DMM_GEMM_TILE", "auto"), part=os.environ.get("DMM_GEMM_PART", "auto"), us=round(t * 1e6, 1),
                          TFLOPs=round(2.0 * M * N * K / t / 1e12, 1))), flush=True)
    del A, B, kw; torch.cuda.empty_cache()
