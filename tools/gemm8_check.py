"""Correctness / race screen / timing of the bf16 NT GEMM tiles (run with DMM_GEMM_TILE=512 for the 256x256 LDS-DMA tile).
The checker is torch's fp32 matmul on the same bf16 operands (test harness only)."""
import os, sys, torch
from diffmm_amd import ops, _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)


def run(M, N, K, nb, mode, reps=3):
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nb)]
    B = [torch.randn(N, K, device=dev).bfloat16() for _ in range(nb)]
    bias = torch.randn(N, device=dev)
    R = [torch.randn(M, N, device=dev) for _ in range(nb)]
    outs = []
    for rep in range(reps):
        C = [torch.full((M, N), float("nan"), device=dev) for _ in range(nb)]
        Cb = [torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nb)]
        if mode == "tanh":
            probs = [dict(A=A[i], B=B[i], out=None, out_bf16=Cb[i], bias=bias, act=L.ACT_TANH, alpha=0.05) for i in range(nb)]
        else:
            probs = [dict(A=A[i], B=B[i], out=C[i], out_bf16=Cb[i], residual=R[i], e1=0.7, e2=-0.3) for i in range(nb)]
        ops.gemm_bf16_nt_batched(probs)
        torch.cuda.synchronize()
        outs.append(([c.clone() for c in C], [c.clone() for c in Cb]))
    worst = 0.0
    for i in range(nb):
        ref = A[i].float() @ B[i].float().t()
        if mode == "tanh":
            ref = torch.tanh(0.05 * ref + bias)
            got = outs[0][1][i].float()
            err = (got - ref).abs().max().item()
            tol = 1e-2
        else:
            ref = 0.7 * ref + (-0.3) * R[i]
            got = outs[0][0][i]
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            tol = 2e-5
            assert torch.equal(outs[0][1][i], got.bfloat16()), "bf16 copy differs from the fp32 output"
        worst = max(worst, err)
        assert err < tol, f"{M}x{N}x{K} nb={nb} {mode}: err {err}"
        for rep in range(1, reps):
            assert torch.equal(outs[rep][1][i], outs[0][1][i]), f"run-to-run difference (bf16) {M}x{N}x{K}"
            if mode != "tanh":
                assert torch.equal(outs[rep][0][i], outs[0][0][i]), f"run-to-run difference {M}x{N}x{K}"
    print(f"ok {M}x{N}x{K} nb={nb} {mode}: max err {worst:.2e}", flush=True)


def timeit(M, N, K, nb, mode, n=20):
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nb)]
    B = [torch.randn(N, K, device=dev).bfloat16() for _ in range(nb)]
    bias = torch.randn(N, device=dev)
    Np = (N + 63) // 64 * 64  # leading dimensions as the engine allocates them (vectorised epilogue path)
    R = [torch.randn(M, Np, device=dev)[:, :N] for _ in range(nb)]
    C = [torch.empty(M, Np, device=dev)[:, :N] for _ in range(nb)]
    Cb = [torch.empty(M, Np, device=dev, dtype=torch.bfloat16)[:, :N] for _ in range(nb)]
    if mode == "tanh":
        probs = [dict(A=A[i], B=B[i], out=None, out_bf16=Cb[i], bias=bias, act=L.ACT_TANH) for i in range(nb)]
    elif mode == "cb":
        probs = [dict(A=A[i], B=B[i], out=None, out_bf16=Cb[i]) for i in range(nb)]
    elif mode == "c":
        probs = [dict(A=A[i], B=B[i], out=C[i], out_bf16=None) for i in range(nb)]
    elif mode == "ccb":
        probs = [dict(A=A[i], B=B[i], out=C[i], out_bf16=Cb[i]) for i in range(nb)]
    elif mode == "rcb":
        probs = [dict(A=A[i], B=B[i], out=None, out_bf16=Cb[i], residual=R[i], e1=0.7, e2=-0.3) for i in range(nb)]
    else:
        probs = [dict(A=A[i], B=B[i], out=C[i], out_bf16=Cb[i], residual=R[i], e1=0.7, e2=-0.3) for i in range(nb)]
    for _ in range(3):
        ops.gemm_bf16_nt_batched(probs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm_bf16_nt_batched(probs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"time {M}x{N}x{K} nb={nb} {mode}: {us:.1f} us  {2.0 * M * N * K * nb / us * 1e-6:.1f} TF/s", flush=True)


if __name__ == "__main__":
    print("DMM_GEMM_TILE =", os.environ.get("DMM_GEMM_TILE"), "DMM_GEMM_8PH =", os.environ.get("DMM_GEMM_8PH"))
    if "check" in sys.argv:
        for K in (64, 128, 192, 256, 1024):
            run(700, 520, K, 1, "res")
        run(256, 256, 512, 1, "res")
        run(1, 5, 64, 1, "res")
        run(257, 255, 320, 3, "tanh")
        run(1000, 777, 6720, 2, "tanh")
        run(9308, 1024, 6720, 3, "tanh", reps=4)
        run(9308, 6710, 1024, 3, "res", reps=4)
    if "ksweep" in sys.argv:
        for K in (64, 256, 512, 1024, 2048, 4096):
            timeit(9308, 6710, K, 3, "cb")
        for K in (64, 1024):
            timeit(9308, 6710, K, 3, "res")
    if "time" in sys.argv:
        timeit(9308, 1024, 6720, 3, "tanh")
        for md in ("res", "cb", "c", "ccb", "rcb"):
            timeit(9308, 6710, 1024, 3, md)
        timeit(4096, 4096, 4096, 1, "tanh")
        timeit(8192, 8192, 8192, 1, "tanh", n=5)
