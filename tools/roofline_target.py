#!/usr/bin/env python3
"""The launches bench.py's `roofline` times, alone and as the reverse chain issues them (bench shapes: 3 modalities x 9308 users,
I = 6710, Kp = 6720, H = 1024): N x the latent chain step ([rows, H] x [H, H], fp32 z in / out, bf16 tanh out), N x the chain's last step
(h_0 W2^T -> fp32 view), and the stepwise chain's launches: N x layer 1 (tanh -> bf16), N x layer 2 with the running x_t in bf16, N x layer 2
with the fp32 state of round 1.  Each epilogue kind is its own kernel instantiation, so a profile separates them:
    rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/roofline_target.py [N]
    gemm_bf16_nt_8ph_kernel<9> = latent step, <6> = last step, <1> = layer 1, <3> = layer 2 (bf16 state), <2> = layer 2 (fp32 state);
    latent_chain_8ph_kernel = the persistent chain (T - 1 = 39 steps per call, 2 launches per call at this shape)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import ops, _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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
l2 = lambda: ops.gemm_bf16_nt_batched([dict(A=h16[m], B=w2[m], out_bf16=y16[m], bias=b2, residual_bf16=x16[m], e1=0.9, e2=0.1, N=I) for m in range(3)])
l2_fp32 = lambda: ops.gemm_bf16_nt_batched([dict(A=h16[m], B=w2[m], out=y32[m], out_bf16=y16[m], bias=b2, residual=x32[m], e1=0.9, e2=0.1, N=I) for m in range(3)])
G = [(torch.randn(H, H, device=dev) * 0.02).bfloat16() for _ in range(3)]; g = torch.zeros(H, device=dev); bt = torch.zeros(H, device=dev)
z32 = [torch.randn(Mc, H, device=dev) for _ in range(3)]; z32b = [torch.zeros(Mc, H, device=dev) for _ in range(3)]
hb16 = [torch.zeros(Mc, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
latent = lambda: ops.gemm_bf16_nt_batched([dict(A=h16[m], B=G[m], out=z32b[m], out_bf16=hb16[m], bias=g, residual=z32[m], e1=0.9, e2=0.1, bias2=bt) for m in range(3)])
last = lambda: ops.gemm_bf16_nt_batched([dict(A=h16[m], B=w2[m], out=y32[m], bias=b2, N=I) for m in range(3)])
T = 40
coef = torch.zeros(2 * T, device=dev); coef[0::2] = 0.9; coef[1::2] = 0.5
flags = torch.zeros(514, dtype=torch.int32, device=dev)
chain = lambda: ops.latent_chain([dict(G=G[m], g=g, z=z32[m], h_a=h16[m], h_b=hb16[m], bias_t=bt.expand(T, H).contiguous()) for m in range(3)], rows=Mc, T=T,
                                 coef=coef, s_init=0.5, flags=flags)
for fn in (l1, latent, last, l2, l2_fp32, chain):
    fn(); torch.cuda.synchronize()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
