#!/usr/bin/env python3
"""InfoNCE forward + backward (dmm_infonce_fwd_bwd) at a given batch / width, 8 terms over 10 slots as in the joint step.
usage: infonce_time.py B D   (knobs: DMM_INFONCE_LDS_KSPLIT, read once per process)"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
dev = torch.device("cuda:0")
B, D = int(sys.argv[1]), int(sys.argv[2])
S, N = 12, 20000
src = [torch.randn(N, D, device=dev) for _ in range(S)]
idx = torch.randint(0, N, (B,), device=dev, dtype=torch.int32)
slots = ops.Slots([(src[s], idx, torch.zeros_like(src[s])) for s in range(S)])
terms = ops.make_terms([(8, 10, 0.2, 0.5), (9, 11, 0.2, 0.5)] + [(s % 2, 2 + s, 0.5, 0.01) for s in range(6)])
zhat, znorm = torch.zeros(S, B, D, device=dev), torch.zeros(S, B, device=dev)
pack = ops.infonce_pack(S, B, D, dev)
ops.gather_normalize(slots, B, D, zhat, znorm, pack)
lse, rl, dz = torch.zeros(8, B, device=dev), torch.zeros(8, B, device=dev), torch.zeros(S, B, D, device=dev)
def timed(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
kw = dict(pack=pack, n_slots=S, precision=int(os.environ.get("PREC", "1")))
out = dict(B=B, D=D, ksplit=os.environ.get("DMM_INFONCE_LDS_KSPLIT", "default"), fused=os.environ.get("DMM_INFONCE_FUSED", "0"),
           fwd_us=timed(lambda: ops.infonce_fwd(zhat, terms, 8, B, D, lse, rl, **kw)),
           bwd_us=timed(lambda: ops.infonce_bwd(zhat, lse, terms, 8, B, D, dz, **kw)),
           fwd_bwd_us=timed(lambda: ops.infonce_fwd_bwd(zhat, terms, 8, B, D, lse, rl, dz, **kw)))
# logits per second of the backward (every logit of a term is formed twice: once per pass)
out["bwd_Tlogit_s"] = round(8 * 2 * B * B / out["bwd_us"] / 1e6, 2)
print(json.dumps(out))
