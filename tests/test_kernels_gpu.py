"""Kernel-level parity: every HIP kernel family (through the C ABI) against the CPU oracle on the same seeded
inputs, and against the golden vectors captured from the reference.  Needs a real MI355X: `pytest -m gpu`.

Tolerances (stated per test): fp32 kernels 1e-5 rel (different summation order only); GEMMs in precision=3
(bf16 hi/lo split, fp32 accumulate) 2e-5 rel of the operand scale; precision=1 (single bf16 product) 1e-2 rel."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("pytest -m gpu needs a GPU; none visible")


@pytest.fixture(scope="module")
def dev():
    need_gpu()
    return torch.device("cuda:0")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def T(a, dev=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dev is not None else t


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b).max() if a.size else 0.0
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    assert np.allclose(a, b, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e} (ref scale {np.abs(b).max():.3e})"


# ------------------------------------------------------------------------------------------------
# SpMM
# ------------------------------------------------------------------------------------------------
def _adj(name, D, dev, seed=1818):
    from diffmm_amd import synth, graph
    g = synth.named(name, seed=seed)
    adj = graph.make_adjacency(g.trn_rows, g.trn_cols, g.users, g.items, D, dev, keep_host=True)
    from oracle import diffmm_oracle as O
    return g, adj, O.csr_to_torch(*adj.host)


@pytest.mark.parametrize("name,D", [("toy", 8), ("mid", 16), ("mid", 32), ("mid", 64), ("mid", 128), ("mid", 256), ("tiktok", 64)])
def test_spmm_plain_and_epilogues(dev, name, D):
    from diffmm_amd import ops
    g, adj, A = _adj(name, D, dev)
    rng = np.random.default_rng(0)
    n = adj.n
    X = T(rng.standard_normal((n, D)).astype(np.float32)); C1 = T(rng.standard_normal((n, D)).astype(np.float32))
    C2 = T(rng.standard_normal((n, D)).astype(np.float32))
    ref = torch.sparse.mm(A, X)
    y = torch.empty((n, D), device=dev)
    ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y=y)])
    close(y, ref, what="plain")
    ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y=y, alpha=1.5, c1=C1.to(dev), beta=-0.5, c2=C2.to(dev), gamma=2.0)])
    close(y, 1.5 * ref - 0.5 * C1 + 2.0 * C2, what="alpha/beta/gamma")
    # split source: rows < U from one tensor, the rest from another
    U = g.users
    ops.spmm_batched([ops.SpmmOp(adj, X[:U].contiguous().to(dev), y=y, x_hi=X[U:].contiguous().to(dev), split=U)])
    close(y, ref, what="split source")


@pytest.mark.parametrize("name,D", [("mid", 16), ("tiktok", 64)])
def test_spmm_perturb_mean_and_batch(dev, name, D):
    from diffmm_amd import ops
    import torch.nn.functional as F
    g, adj, A = _adj(name, D, dev)
    rng = np.random.default_rng(1)
    n = adj.n
    X = T(rng.standard_normal((n, D)).astype(np.float32))
    noise = T(rng.random((n, D)).astype(np.float32))
    m1 = T(rng.standard_normal((n, D)).astype(np.float32)); m2 = T(rng.standard_normal((n, D)).astype(np.float32))
    ref = torch.sparse.mm(A, X)
    refp = ref + torch.sign(ref) * F.normalize(noise) * 1.5
    y, yp, ym = (torch.empty((n, D), device=dev) for _ in range(3))
    ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y=y, y_pert=yp, noise=noise.to(dev), noise_degree=1.5, y_mean=ym, m1=m1.to(dev), m2=m2.to(dev))])
    close(y, ref, what="raw"); close(yp, refp, what="perturbed"); close(ym, (refp + m1 + m2) / 3, what="mean of 3")
    # in-kernel Philox noise: the perturbation is a unit-norm non-negative direction scaled by noise_degree
    ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y=y, y_pert=yp, noise=None, noise_degree=0.7, seed=5, offset=123)])
    delta = (yp - y).cpu() * torch.sign(ref)
    nz = (ref != 0).all(dim=1)
    close(delta[nz].norm(dim=1), torch.full((int(nz.sum()),), 0.7), rtol=1e-4, what="philox perturbation norm")
    assert (delta >= -1e-6).all() and 0.3 < float(delta[nz].mean() / (0.7 / np.sqrt(D))) < 1.2
    yp2 = torch.empty_like(yp)
    ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y_pert=yp2, noise=None, noise_degree=0.7, seed=5, offset=123)])
    assert torch.equal(yp, yp2), "same (seed, offset) must reproduce the same noise"
    # 4 problems in one launch, each with its own operand
    Xs = [T(rng.standard_normal((n, D)).astype(np.float32)) for _ in range(4)]
    ys = [torch.empty((n, D), device=dev) for _ in range(4)]
    ops.spmm_batched([ops.SpmmOp(adj, x.to(dev), y=yy, alpha=float(i + 1)) for i, (x, yy) in enumerate(zip(Xs, ys))])
    for i in range(4):
        close(ys[i], (i + 1) * torch.sparse.mm(A, Xs[i]), atol=2e-5, what=f"batched {i}")


@pytest.mark.parametrize("D", [16, 64, 256])
def test_spmm_giant_rows(dev, D):
    """An item every user interacted with (9k+ non-zeros) and a user with 3000 items: rows cut into multi-workgroup segments, summed
    by the fix-up pass; all epilogues still apply; two uses of one adjacency in a launch get separate scratch."""
    from diffmm_amd import ops, graph, synth
    from oracle import diffmm_oracle as O
    import torch.nn.functional as F
    g = synth.named("tiktok")
    rows = np.concatenate([g.trn_rows, np.arange(g.users), np.full(3000, 5)])
    cols = np.concatenate([g.trn_cols, np.zeros(g.users, dtype=np.int32), np.arange(3000)])
    adj = graph.make_adjacency(rows, cols, g.users, g.items, D, dev, keep_host=True)
    assert adj.n_partial_blocks >= 2 and adj.fix is not None
    A = O.csr_to_torch(*adj.host)
    rng = np.random.default_rng(3)
    X = T(rng.standard_normal((adj.n, D)).astype(np.float32)); C1 = T(rng.standard_normal((adj.n, D)).astype(np.float32))
    noise = T(rng.random((adj.n, D)).astype(np.float32))
    ref = torch.sparse.mm(A, X)
    y, y2, yp = (torch.empty((adj.n, D), device=dev) for _ in range(3))
    ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y=y, alpha=2.0, c1=C1.to(dev), beta=0.5),
                      ops.SpmmOp(adj, C1.to(dev), y=y2, y_pert=yp, noise=noise.to(dev), noise_degree=1.5)])
    close(y, 2.0 * ref + 0.5 * C1, rtol=2e-5, atol=2e-5, what="giant rows, epilogue")
    ref2 = torch.sparse.mm(A, C1)
    close(y2, ref2, rtol=2e-5, atol=2e-5, what="second use of the adjacency")
    close(yp, ref2 + torch.sign(ref2) * F.normalize(noise) * 1.5, rtol=2e-5, atol=2e-5, what="perturbed")
    y3 = torch.empty_like(y)
    ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y=y3, alpha=2.0, c1=C1.to(dev), beta=0.5)])
    assert torch.equal(y, y3), "segmented rows stay bitwise reproducible"
    # the giant rows finished by the last-arriving segment workgroup of the main launch (dmm_spmm_problem.fix_count; the default) against the second
    # launch (spmm_fixup_kernel): the same order and grouping of the sums -> bit-identical, whichever workgroup arrives last, every time
    # (the second launch is the default -- the fold measured slower at TikTok size, ops._SPMM_FOLD -- so `y`, `y2`, `yp` above are the second launch's)
    assert adj.fix_counter(0) is not None
    fold_was = ops._SPMM_FOLD
    ops._SPMM_FOLD = True
    try:
        side = torch.cuda.Stream()
        junk = torch.randn(32 << 20, device=dev)
        for it in range(40):  # under uneven background load (the hand-off between the segment workgroups must hold whoever arrives last)
            if it % 2:
                with torch.cuda.stream(side):
                    junk[: (1 + it % 5) << 22].mul_(1.0001)
            ya, yb, ypb = (torch.empty_like(y) for _ in range(3))
            ops.spmm_batched([ops.SpmmOp(adj, X.to(dev), y=ya, alpha=2.0, c1=C1.to(dev), beta=0.5),
                              ops.SpmmOp(adj, C1.to(dev), y=yb, y_pert=ypb, noise=noise.to(dev), noise_degree=1.5)])
            assert torch.equal(ya, y) and torch.equal(yb, y2) and torch.equal(ypb, yp), f"run {it}: folded fix-up == second-launch fix-up, bit for bit"
    finally:
        ops._SPMM_FOLD = fold_was
    torch.cuda.synchronize()
    assert int(adj.fix_counter(0).abs().sum()) == 0 and int(adj.fix_counter(1).abs().sum()) == 0, "the arrival counters are zero again after every launch"


@pytest.mark.parametrize("name,D,hub", [("toy", 16, False), ("mid", 32, False), ("mid", 64, False), ("mid", 128, False), ("mid", 256, False), ("tiktok", 64, False),
                                        ("tiktok", 64, True), ("tiktok", 256, True)])
def test_spmm_bf16_operands(dev, name, D, hub):
    """dmm_spmm_csr_bf16 (X in bf16, 8 columns per lane, fp32 accumulate / epilogue / outputs + bf16 copies).  Two bars:
      * exactness of the kernel: against torch.sparse.mm on the bf16-ROUNDED operand (only the fp32 summation order differs), 1e-5;
      * the storage tolerance of SURVEY 8c ("bf16 storage: 2e-2 rel. on embeddings"): against the fp32 operand, row by row within
        2^-8 of sum_j |a_ij| |x_j| (every gathered element is off by at most half a bf16 ulp)."""
    from diffmm_amd import ops, graph, synth
    from oracle import diffmm_oracle as O
    import torch.nn.functional as F
    g = synth.named(name)
    rows, cols = g.trn_rows, g.trn_cols
    if hub:  # giant rows: an item every user has, a user with 3000 items -> multi-workgroup segments + fix-up pass
        rows = np.concatenate([rows, np.arange(g.users), np.full(3000, 5)])
        cols = np.concatenate([cols, np.zeros(g.users, dtype=np.int32), np.arange(3000)])
    adj = graph.make_adjacency(rows, cols, g.users, g.items, D, dev, keep_host=True, vec=8)
    assert adj.vec == 8 and (not hub or adj.n_partial_blocks >= 1)
    A = O.csr_to_torch(*adj.host)
    rng = np.random.default_rng(4)
    n, U = adj.n, g.users
    X = T(rng.standard_normal((n, D)).astype(np.float32)); C1 = T(rng.standard_normal((n, D)).astype(np.float32))
    noise = T(rng.random((n, D)).astype(np.float32)); m1 = T(rng.standard_normal((n, D)).astype(np.float32)); m2 = T(rng.standard_normal((n, D)).astype(np.float32))
    X16 = X.bfloat16()
    ref = torch.sparse.mm(A, X16.float())
    y, yp, ym = (torch.empty((n, D), device=dev) for _ in range(3))
    y16, yp16 = (torch.empty((n, D), device=dev, dtype=torch.bfloat16) for _ in range(2))
    ops.spmm_batched([ops.SpmmOp(adj, X16.to(dev), y=y, y16=y16)])
    close(y, ref, rtol=2e-5, atol=2e-5, what="bf16 operand, plain")
    assert torch.equal(y16, y.bfloat16()), "y16 is the bf16 rounding of y"
    absA = torch.sparse_coo_tensor(A.indices(), A.values().abs(), A.shape).coalesce() if A.layout == torch.sparse_coo else None
    full = torch.sparse.mm(A, X)
    bound = torch.sparse.mm(absA if absA is not None else A.abs(), X.abs()) * 2.0 ** -8 + 1e-6
    assert bool(((y.cpu() - full).abs() <= bound).all()), "outside the bf16 storage bound"
    assert float((y.cpu() - full).abs().max()) <= 2e-2 * float(full.abs().max()), "SURVEY 8c: 2e-2 rel. on embeddings"
    # epilogues + split operand + perturbation + mean + bf16 copy of the perturbed output
    ops.spmm_batched([ops.SpmmOp(adj, X16[:U].contiguous().to(dev), x_hi=X16[U:].contiguous().to(dev), split=U, y=y, alpha=1.5, c1=C1.to(dev), beta=-0.5,
                                 y_pert=yp, y_pert16=yp16, noise=noise.to(dev), noise_degree=1.5, y_mean=ym, m1=m1.to(dev), m2=m2.to(dev))])
    r = 1.5 * ref - 0.5 * C1
    rp = r + torch.sign(r) * F.normalize(noise) * 1.5
    close(y, r, rtol=2e-5, atol=3e-5, what="epilogue"); close(yp, rp, rtol=2e-5, atol=3e-5, what="perturbed"); close(ym, (rp + m1 + m2) / 3, rtol=2e-5, atol=3e-5, what="mean")
    assert torch.equal(yp16, yp.bfloat16())
    # the in-kernel Philox draws are the fp32 kernel's (same element -> counter map): same perturbation direction for the same rows
    adj32 = graph.make_adjacency(rows, cols, g.users, g.items, D, dev)
    a32, b16 = torch.empty((n, D), device=dev), torch.empty((n, D), device=dev)
    ops.spmm_batched([ops.SpmmOp(adj32, X16.float().to(dev), y_pert=a32, noise=None, noise_degree=0.7, seed=9, offset=77)])
    ops.spmm_batched([ops.SpmmOp(adj, X16.to(dev), y_pert=b16, noise=None, noise_degree=0.7, seed=9, offset=77)])
    close(b16, a32, rtol=2e-5, atol=3e-5, what="Philox perturbation, bf16 vs fp32 kernel")
    again = torch.empty_like(b16)
    ops.spmm_batched([ops.SpmmOp(adj, X16.to(dev), y_pert=again, noise=None, noise_degree=0.7, seed=9, offset=77)])
    assert torch.equal(again, b16), "bitwise reproducible"
    with pytest.raises(ValueError):
        ops.SpmmOp(adj, X.to(dev), y=y)  # fp32 operand on a bf16-planned adjacency


def test_spmm_is_deterministic_and_symmetric_backward(dev):
    from diffmm_amd import ops
    g, adj, A = _adj("tiktok", 64, dev)
    rng = np.random.default_rng(2)
    X = T(rng.standard_normal((adj.n, 64)).astype(np.float32)).to(dev)
    Y = T(rng.standard_normal((adj.n, 64)).astype(np.float32)).to(dev)
    a, b = torch.empty_like(X), torch.empty_like(X)
    ops.spmm_batched([ops.SpmmOp(adj, X, y=a)]); ops.spmm_batched([ops.SpmmOp(adj, X, y=b)])
    assert torch.equal(a, b), "no atomics: bitwise reproducible"
    ops.spmm_batched([ops.SpmmOp(adj, Y, y=b)])
    lhs, rhs = float((a.double() * Y.double()).sum()), float((X.double() * b.double()).sum())  # <AX, Y> == <X, AY>
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 96, 70), (64, 768, 6710), (300, 6710, 64), (1024, 1000, 517), (5, 3, 2)])
@pytest.mark.parametrize("prec", [3, 1])
def test_gemm_layouts(dev, M, N, K, prec):
    from diffmm_amd import ops, _lib as L
    rng = np.random.default_rng(M + N + K)
    A = T(rng.standard_normal((M, K)).astype(np.float32)); Bm = T(rng.standard_normal((K, N)).astype(np.float32))
    bias = T(rng.standard_normal(N).astype(np.float32)); R = T(rng.standard_normal((M, N)).astype(np.float32))
    ref = A.double() @ Bm.double()
    # hi/lo split keeps 16 mantissa bits per operand: ~1.5e-5 relative per product
    tol = dict(rtol=5e-5, atol=5e-5 * np.sqrt(K)) if prec == 3 else dict(rtol=2e-2, atol=2e-2 * np.sqrt(K))
    Ad, Bd = A.to(dev), Bm.to(dev)
    close(ops.gemm(Ad, Bd, precision=prec), ref, what="NN", **tol)
    close(ops.gemm(Ad, Bd.t().contiguous(), trans_b=True, precision=prec), ref, what="NT", **tol)
    close(ops.gemm(Ad.t().contiguous(), Bd, trans_a=True, precision=prec), ref, what="TN", **tol)
    close(ops.gemm(Ad.t().contiguous(), Bd.t().contiguous(), trans_a=True, trans_b=True, precision=prec), ref, what="TT", **tol)
    out = ops.gemm(Ad, Bd, bias=bias.to(dev), act=L.ACT_TANH, alpha=0.1, residual=R.to(dev), e1=0.5, e2=2.0, precision=prec)
    close(out, 0.5 * torch.tanh(0.1 * ref + bias.double()) + 2.0 * R.double(), what="epilogue", rtol=tol["rtol"], atol=tol["atol"] * 0.1 + 1e-5)
    out = ops.gemm(Ad, Bd, bias=bias.to(dev), act=L.ACT_SIGMOID, alpha=0.05, precision=prec)
    close(out, torch.sigmoid(0.05 * ref + bias.double()), what="sigmoid", rtol=tol["rtol"], atol=tol["atol"] * 0.05 + 1e-5)
    out = torch.zeros((M, N), device=dev)
    ops.gemm(Ad, Bd, out=out, bias=bias.to(dev), split_k=8, precision=prec)
    close(out, ref + bias.double(), what="split-K", **tol)
    acc = R.to(dev).clone()
    ops.gemm(Ad, Bd, out=acc, accumulate=True, precision=prec)
    close(acc, ref + R.double(), what="accumulate", **tol)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 200, 328), (384, 1024, 704), (130, 70, 64), (1024, 6710, 1024), (512, 1024, 6720)])
def test_gemm_bf16_nt(dev, M, N, K):
    """Dedicated bf16 NT kernel (reverse chain): operands exactly representable in bf16, so only fp32 summation order differs."""
    from diffmm_amd import ops, _lib as L
    rng = np.random.default_rng(M * 7 + N)
    A = T(rng.standard_normal((M, K)).astype(np.float32)).bfloat16(); Bw = T(rng.standard_normal((N, K)).astype(np.float32)).bfloat16()
    bias = T(rng.standard_normal(N).astype(np.float32)); R = T(rng.standard_normal((M, N)).astype(np.float32))
    ref = A.double() @ Bw.double().t()
    ld = (N + 7) & ~7
    out, out16 = torch.zeros(M, ld, device=dev), torch.zeros(M, ld, dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16_nt(A.to(dev), Bw.to(dev), out=out, out_bf16=out16, N=N)
    close(out[:, :N], ref, rtol=1e-5, atol=1e-4 * np.sqrt(K / 64), what="plain")
    close(out16[:, :N].float(), ref, rtol=1e-2, atol=1e-2 * np.sqrt(K), what="bf16 copy")
    assert float(out[:, N:].abs().max() if ld > N else 0.0) == 0.0, "columns beyond N must stay untouched"
    ops.gemm_bf16_nt(A.to(dev), Bw.to(dev), out=out, bias=bias.to(dev), act=L.ACT_TANH, alpha=0.05, residual=R.to(dev), e1=0.3, e2=0.9, N=N)
    close(out[:, :N], 0.3 * torch.tanh(0.05 * ref + bias.double()) + 0.9 * R.double(), rtol=1e-5, atol=2e-5, what="epilogue")
    # asymmetric integer data: catches any transposed / permuted fragment or swizzle mistake exactly
    Ai = (torch.arange(M * K).view(M, K) % 13 - 6).float().bfloat16(); Bi = (torch.arange(N * K).view(N, K) % 7 - 3).float().bfloat16()
    ops.gemm_bf16_nt(Ai.to(dev), Bi.to(dev), out=out, N=N)
    assert torch.equal(out[:, :N].cpu().double(), Ai.double() @ Bi.double().t()), "integer GEMM must be exact"


def test_gemm_bf16_big_tile_and_batched(dev):
    """Shapes that select the 256 x 128 / 8-wave tile (>= 768 tiles), ragged in M and N, and 3 problems in one launch."""
    from diffmm_amd import ops, _lib as L
    rng = np.random.default_rng(5)
    M, N, K = 2500, 1030, 520
    probs, refs = [], []
    for z in range(3):
        A = (torch.arange(M * K).view(M, K) % (11 + z) - 5).float().bfloat16()
        Bw = (torch.arange(N * K).view(N, K) % (5 + z) - 2).float().bfloat16()
        bias = T(rng.standard_normal(N).astype(np.float32)); R = T(rng.standard_normal((M, N + 2)).astype(np.float32))
        out = torch.zeros(M, N + 2, device=dev); out16 = torch.zeros(M, N + 2, dtype=torch.bfloat16, device=dev)
        probs.append(dict(A=A.to(dev), B=Bw.to(dev), out=out, out_bf16=out16, bias=bias.to(dev), residual=R.to(dev), e1=0.25, e2=2.0, N=N))
        refs.append(0.25 * (A.double() @ Bw.double().t() + bias.double()) + 2.0 * R[:, :N].double())
    ops.gemm_bf16_nt_batched(probs)   # 3 x 10 x 9 tiles of 256x128 = 270 -> 128x128 kernel; force the big tile through a larger batch below
    for p, r in zip(probs, refs):
        close(p["out"][:, :N], r, rtol=1e-6, atol=1e-3, what="batched")
    M2, N2 = 9308, 1024
    A = (torch.randn(M2, 136)).bfloat16(); Bw = (torch.randn(N2, 136)).bfloat16()  # K % 64 != 0 keeps the register-staged tiles
    outs = [torch.zeros(M2, N2, device=dev) for _ in range(3)]
    ops.gemm_bf16_nt_batched([dict(A=A.to(dev), B=Bw.to(dev), out=o) for o in outs])  # 3 * 37 * 8 = 888 tiles: the 256x128 kernel
    ref = A.double() @ Bw.double().t()
    for o in outs:
        close(o, ref, rtol=1e-5, atol=1e-3, what="big tile")


@pytest.mark.parametrize("K", [64, 128, 192, 448])
def test_gemm_bf16_lds_dma_tile(dev, K):
    """The 256 x 256 LDS-DMA tile (csrc/gemm_bf16_8ph.hip; chosen when K % 64 == 0 and >= 192 tiles): integer data, so every
    output is exact whatever the summation order -- a stale or early-read LDS unit (the kernel's counted waits / one-barrier
    stagger) shows up as a wrong integer.  Ragged M and N (edge tiles take the generic epilogue, interior tiles the fast one),
    3 problems per launch, every epilogue the chain uses, 1..7 K tiles (prologue / steady state / tail), repeated launches."""
    from diffmm_amd import ops, _lib as L
    M, N, nb = 2100, 2330, 3  # 9 x 10 x 3 = 270 tiles
    ld = (N + 63) // 64 * 64
    As, Bs, refs = [], [], []
    for z in range(nb):
        A = ((torch.arange(M * K).view(M, K) * (z + 3)) % 13 - 6).float().bfloat16()
        Bw = ((torch.arange(N * K).view(N, K) * (z + 5)) % 7 - 3).float().bfloat16()
        As.append(A.to(dev)); Bs.append(Bw.to(dev)); refs.append(A.double() @ Bw.double().t())
    bias = (torch.arange(N) % 5 - 2).float()
    R = [((torch.arange(M * ld).view(M, ld) * (z + 1)) % 9 - 4).float().to(dev) for z in range(nb)]
    for rep in range(3):
        C = [torch.full((M, ld), -7.0, device=dev) for _ in range(nb)]
        Cb = [torch.full((M, ld), -7.0, dtype=torch.bfloat16, device=dev) for _ in range(nb)]
        # posterior-mean epilogue of the chain: e1 * (A B^T + b) + e2 * R -> fp32 state + bf16 copy
        ops.gemm_bf16_nt_batched([dict(A=As[z], B=Bs[z], out=C[z], out_bf16=Cb[z], bias=bias.to(dev), residual=R[z], e1=0.5, e2=2.0, N=N) for z in range(nb)])
        for z in range(nb):
            want = 0.5 * (refs[z] + bias.double()) + 2.0 * R[z][:, :N].cpu().double()
            assert torch.equal(C[z][:, :N].cpu().double(), want), f"K={K} rep={rep} problem {z}: residual epilogue"
            assert torch.equal(Cb[z][:, :N].cpu(), want.float().bfloat16()), "bf16 copy"
            assert float((C[z][:, N:] + 7.0).abs().max()) == 0.0 and float((Cb[z][:, N:].float() + 7.0).abs().max()) == 0.0, "pad columns untouched"
        # hidden layer of the chain: tanh -> bf16 only (fast tanh on interior tiles: bf16-level tolerance)
        H16 = [torch.zeros(M, ld, dtype=torch.bfloat16, device=dev) for _ in range(nb)]
        ops.gemm_bf16_nt_batched([dict(A=As[z], B=Bs[z], out=None, out_bf16=H16[z], bias=bias.to(dev), act=L.ACT_TANH, alpha=0.01, N=N) for z in range(nb)])
        for z in range(nb):
            close(H16[z][:, :N].float(), torch.tanh(0.01 * refs[z] + bias.double()), rtol=0, atol=4e-3, what="tanh -> bf16")
    # plain fp32 output and a bf16 residual (generic epilogue)
    out = torch.zeros(M, ld, device=dev)
    ops.gemm_bf16_nt_batched([dict(A=As[0], B=Bs[0], out=out, N=N)] * 3)
    assert torch.equal(out[:, :N].cpu().double(), refs[0])
    Rb = R[1].bfloat16()
    ops.gemm_bf16_nt_batched([dict(A=As[1], B=Bs[1], out=out, residual_bf16=Rb, e1=1.0, e2=1.0, N=N)] * 3)
    assert torch.equal(out[:, :N].cpu().double(), refs[1] + Rb[:, :N].cpu().double())


def test_gemm_asymmetric_identity_and_strided_views(dev):
    from diffmm_amd import ops
    n = 96
    Bm = torch.arange(n * n, dtype=torch.float32).view(n, n) / 64.0  # asymmetric: catches a transposed C-write
    close(ops.gemm(torch.eye(n).to(dev), Bm.to(dev), precision=3), Bm, rtol=1e-6, atol=1e-4, what="I @ B")
    big = torch.randn(40, 200, device=dev)
    close(ops.gemm(big[:, :150], torch.ones(150, 8, device=dev), precision=3, K=150), big[:, :150].double().sum(1, keepdim=True).expand(40, 8),
          rtol=1e-5, atol=1e-4, what="row-strided A, K subset")


# ------------------------------------------------------------------------------------------------
# small kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [8, 32, 64, 256])
def test_rownorm(dev, D):
    from diffmm_amd import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(D)
    x = T(rng.standard_normal((777, D)).astype(np.float32)); x[3] = 0.0
    dy = T(rng.standard_normal((777, D)).astype(np.float32))
    xr = x.clone().requires_grad_(); yr = F.normalize(xr); (yr * dy).sum().backward()
    y, nrm = ops.rownorm_fwd(x.to(dev))
    close(y, yr, what="fwd"); close(nrm, x.norm(dim=1), what="norm")
    dx = ops.rownorm_bwd(dy.to(dev), y, nrm)
    close(dx[4:], xr.grad[4:], rtol=1e-4, atol=1e-5, what="bwd")


@pytest.mark.parametrize("M", [2, 3])
def test_mix_fwd_bwd(dev, M):
    from diffmm_amd import ops
    rng = np.random.default_rng(M)
    n, D, maw = 500, 32, 0.2
    bx = T(rng.standard_normal((n, D)).astype(np.float32)).requires_grad_()
    z = [T(rng.standard_normal((n, D)).astype(np.float32)).requires_grad_() for _ in range(M)]
    w = T(rng.standard_normal(M).astype(np.float32)).requires_grad_()
    cot = T(rng.standard_normal((n, D)).astype(np.float32)); extra = T(rng.standard_normal((n, D)).astype(np.float32))
    dz0 = [T(rng.standard_normal((n, D)).astype(np.float32)) for _ in range(M)]
    sw = torch.softmax(w, -1)
    mx = sum(sw[m] * (bx + maw * z[m]) for m in range(M)); (mx * cot).sum().backward()
    got = ops.gcn_mix_fwd(bx.detach().to(dev), [t.detach().to(dev) for t in z], w.detach().to(dev), maw)
    close(got, mx, what="mix fwd")
    dz = [t.clone().to(dev) for t in dz0]
    dbx, dw = ops.gcn_mix_bwd(cot.to(dev), bx.detach().to(dev), [t.detach().to(dev) for t in z], dz, extra.to(dev), w.detach().to(dev), maw)
    close(dbx, bx.grad + extra, rtol=1e-5, atol=1e-5, what="dbx")
    for m in range(M):
        close(dz[m], dz0[m] + z[m].grad, rtol=1e-5, atol=1e-5, what=f"dz{m}")
    close(dw, w.grad, rtol=1e-4, atol=1e-4, what="dw logits")


def test_losses_against_reference_vectors(dev):
    """InfoNCE / BPR / L2 values and gradients vs tests/golden/losses.npz (captured from Utils/Utils.py)."""
    from diffmm_amd import ops
    z = load("losses.npz")
    v1, v2 = T(z["infonce/v1"]).to(dev), T(z["infonce/v2"]).to(dev)
    idx = T(z["infonce/idx"].astype(np.int32)).to(dev)
    B, D = idx.shape[0], v1.shape[1]
    # precision 0: fp32 VALU kernels; 3: MFMA with hi/lo split operands (~fp32); 1: single bf16 MFMA product
    for temp, prec in [(0.2, 0), (0.5, 0), (0.2, 3), (0.5, 3), (0.2, 1), (0.5, 1)]:
        g1, g2 = torch.zeros_like(v1), torch.zeros_like(v2)
        slots = ops.Slots([(v1, idx, g1), (v2, idx, g2)])
        terms = ops.make_terms([(0, 1, temp, 1.0)])
        zhat, nrm, dz = torch.empty(2, B, D, device=dev), torch.empty(2, B, device=dev), torch.zeros(2, B, D, device=dev)
        lse, rl, out = torch.empty(1, B, device=dev), torch.empty(1, B, device=dev), torch.empty(1, device=dev)
        pack = ops.infonce_pack(2, B, D, dev) if prec else None
        kw = dict(pack=pack, n_slots=2, precision=prec) if prec else {}
        ops.gather_normalize(slots, B, D, zhat, nrm, pack)
        ops.infonce_fwd(zhat, terms, 1, B, D, lse, rl, **kw)
        ops.segment_sum(rl, 1, B, 1.0, out)
        vt = {0: 2e-5, 3: 5e-5, 1: 5e-3}[prec]
        close(out[0], z[f"infonce_t{temp}/val"], rtol=vt, what=f"InfoNCE value prec={prec}")
        ops.infonce_bwd(zhat, lse, terms, 1, B, D, dz, **kw)
        ops.scatter_normalize_bwd(slots, B, D, zhat, nrm, dz)
        gs = float(np.abs(z[f"infonce_t{temp}/g1"]).max())
        gt = {0: dict(rtol=2e-4, atol=2e-7), 3: dict(rtol=5e-4, atol=2e-5 * gs), 1: dict(rtol=5e-2, atol=2e-2 * gs)}[prec]
        close(g1, z[f"infonce_t{temp}/g1"], what=f"InfoNCE g1 prec={prec}", **gt)
        close(g2, z[f"infonce_t{temp}/g2"], what=f"InfoNCE g2 prec={prec}", **gt)
    # BPR: build a [U+I, D] table out of the three golden matrices
    u, p, n = z["bpr/u"], z["bpr/p"], z["bpr/n"]
    Bb = u.shape[0]
    fin = T(np.concatenate([u, p, n])).to(dev)
    ar = torch.arange(Bb, dtype=torch.int32, device=dev)
    dfin, rl, out = torch.zeros_like(fin), torch.empty(Bb, device=dev), torch.empty(1, device=dev)
    ops.bpr_fwd_bwd(fin, ar, ar, ar + Bb, Bb, rl, dfin)
    ops.segment_sum(rl, 1, Bb, 1.0, out)
    close(out[0], z["bpr/val"], what="bpr value")
    close(dfin, np.concatenate([z["bpr/gu"], z["bpr/gp"], z["bpr/gn"]]), rtol=1e-4, atol=1e-8, what="bpr grads")
    a = T(z["l2/a"]).to(dev)
    out = torch.empty(1, device=dev)
    ops.sumsq(a.view(-1), float(z["l2/reg"]), out)
    close(out[0], float(z["l2/reg"]) * float((z["l2/a"].astype(np.float64) ** 2).sum()), what="l2")


@pytest.mark.parametrize("B,D", [(1000, 64), (257, 128), (33, 32)])
def test_infonce_mfma_matches_valu_kernels(dev, B, D):
    """Several terms sharing slots, batch not a multiple of 32: the MFMA kernels (precision 3) vs the fp32 VALU kernels."""
    from diffmm_amd import ops
    rng = np.random.default_rng(B)
    n = 3000
    views = [T(rng.standard_normal((n, D)).astype(np.float32)).to(dev) for _ in range(3)]
    idx = T(rng.integers(0, n, size=B).astype(np.int32)).to(dev)
    terms_l = [(0, 1, 0.2, 0.5), (0, 2, 0.5, 0.01), (1, 2, 0.5, 1.0)]
    res = {}
    for prec in (0, 3):
        grads = [torch.zeros_like(v) for v in views]
        slots = ops.Slots([(v, idx, g) for v, g in zip(views, grads)])
        terms = ops.make_terms(terms_l)
        zhat, nrm, dz = torch.empty(3, B, D, device=dev), torch.empty(3, B, device=dev), torch.zeros(3, B, D, device=dev)
        lse, rl = torch.empty(3, B, device=dev), torch.empty(3, B, device=dev)
        pack = ops.infonce_pack(3, B, D, dev) if prec else None
        kw = dict(pack=pack, n_slots=3, precision=prec) if prec else {}
        ops.gather_normalize(slots, B, D, zhat, nrm, pack)
        ops.infonce_fwd(zhat, terms, 3, B, D, lse, rl, **kw)
        ops.infonce_bwd(zhat, lse, terms, 3, B, D, dz, **kw)
        ops.scatter_normalize_bwd(slots, B, D, zhat, nrm, dz)
        res[prec] = (lse.clone(), rl.clone(), [g.clone() for g in grads])
    close(res[3][0], res[0][0], rtol=2e-5, atol=2e-5, what="lse")
    close(res[3][1].sum(1), res[0][1].sum(1), rtol=5e-5, what="term losses")
    for a, b in zip(res[3][2], res[0][2]):
        close(a, b, rtol=1e-3, atol=2e-5 * float(b.abs().max()), what="grads")


def test_adam_matches_torch(dev):
    from diffmm_amd import ops
    torch.manual_seed(0)
    p0 = torch.randn(10007); ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = p0.clone().to(dev), torch.zeros(10007, device=dev), torch.zeros(10007, device=dev)
    for step in range(1, 6):
        g = torch.randn(10007) * (10.0 ** -step)
        ref.grad = g.clone(); opt.step()
        ops.adam_step(p, g.to(dev), m, v, 1e-3, step)
    close(p, ref, rtol=1e-6, atol=1e-7, what="adam 5 steps")
    # extras + l2
    g, e = torch.randn(10007), torch.randn(4000)
    p2, m2, v2 = p0.clone().to(dev), torch.zeros(10007, device=dev), torch.zeros(10007, device=dev)
    ops.adam_step(p2, g.to(dev), m2, v2, 1e-3, 1, extra=[e.to(dev)], extra_n=[4000], l2_coef=0.02)
    gt = g.clone(); gt[:4000] += e; gt += 0.02 * p0
    r2 = p0.clone().requires_grad_(); o2 = torch.optim.Adam([r2], lr=1e-3); r2.grad = gt; o2.step()
    close(p2, r2, rtol=1e-6, atol=1e-7, what="adam extras")


@pytest.mark.parametrize("cols,kmax", [(9, 4), (200, 40), (6710, 600), (18357, 300), (21000, 50), (100003, 520), (1000003, 300)])  # <= 8192 / <= 20480 columns: register-resident kernels; beyond: streaming kernel
def test_topk_rows(dev, cols, kmax):
    from diffmm_amd import ops
    rng = np.random.default_rng(cols)
    U, rows = (50, 37) if cols < 500000 else (12, 9)
    deg = rng.integers(0, kmax + 1, size=U); deg[0] = 0; deg[1] = min(cols, kmax)
    ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    row_ids = rng.permutation(U)[:rows].astype(np.int32)
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    x[2, : cols // 2] = 0.25  # ties at the threshold
    out = torch.full((int(ptr[-1]) + 1,), -1, dtype=torch.int32, device=dev)
    ld = (cols + 7) & ~7
    xp = torch.zeros(rows, ld); xp[:, :cols] = T(x)
    ops.topk_rows(xp.to(dev), T(row_ids).to(dev), T(ptr).to(dev), out, cols)
    out = out.cpu().numpy()
    for j, u in enumerate(row_ids):
        k = deg[u]
        got = out[ptr[u]:ptr[u + 1]]
        assert len(set(got.tolist())) == k and (np.diff(got) > 0).all(), "distinct ascending ids"
        if k:
            thr = np.sort(x[j])[::-1][k - 1]
            assert (x[j][got] >= thr).all() and (x[j] > thr).sum() == (x[j][got] > thr).sum()
            ties = got[x[j][got] == thr]
            assert np.array_equal(ties, np.flatnonzero(x[j] == thr)[: len(ties)]), "ties go to the smaller column ids"
    untouched = np.setdiff1d(np.arange(U), row_ids)
    for u in untouched:
        assert (out[ptr[u]:ptr[u + 1]] == -1).all()


@pytest.mark.parametrize("cols,k", [(9, 4), (200, 20), (6710, 20), (6710, 64), (18357, 20), (21000, 20), (100003, 64)])
def test_eval_topk(dev, cols, k):
    """Masked top-k of Coach.testEpoch (Main.py:403-411): exact column ids in descending score order; ties -> smaller id."""
    from diffmm_amd import ops
    rng = np.random.default_rng(cols + k)
    U, rows = 60, 41
    deg = rng.integers(0, min(cols - k, 50) + 1, size=U); deg[3] = 0
    ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([rng.choice(cols, size=d, replace=False) for d in deg] + [np.zeros(0, np.int64)]).astype(np.int64)
    users = rng.permutation(U)[:rows].astype(np.int32)
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    x[2, : cols // 2] = 0.75  # a run of equal scores straddling the threshold
    x[5] = np.round(x[5] * 4) / 4  # many ties everywhere
    ld = (cols + 7) & ~7
    xp = torch.zeros(rows, ld); xp[:, :cols] = T(x)
    xd = xp.to(dev)
    out = ops.eval_topk(xd, T(users).to(dev), T(ptr).to(dev), T(col).to(dev), k, cols=cols).cpu().numpy()
    masked = xd.cpu().numpy()[:, :cols]
    for j, u in enumerate(users):
        ref = x[j].copy(); ref[col[ptr[u]:ptr[u + 1]]] = -1e8
        assert np.array_equal(masked[j], ref), "mask applied in place"
        order = np.lexsort((np.arange(cols), -ref))[:k]  # descending score, ascending id on ties
        assert np.array_equal(out[j], order), f"row {j}"


def test_eval_metrics_golden(dev):
    """dmm_eval_metrics against the reference's own calcRes outputs (tests/golden/calcres.npz)."""
    from diffmm_amd import ops
    z = load("calcres.npz")
    top, users, lens, flat = z["top"], z["users"], z["its_len"], z["its_flat"]
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out = ops.eval_metrics(T(top.astype(np.int32), dev), T(users.astype(np.int32), dev), T(ptr, dev), T(flat.astype(np.int64), dev)).sum(0).cpu().numpy()
    close(out[0], z["recall"], rtol=1e-13, atol=0); close(out[1], z["ndcg"], rtol=1e-13, atol=0); close(out[2], z["precision"], rtol=1e-13, atol=0)


@pytest.mark.parametrize("k", [1, 20, 64])
def test_eval_metrics_random(dev, k):
    import math
    from diffmm_amd import ops
    rng = np.random.default_rng(k)
    U, I, rows = 300, 500, 257
    lens = rng.integers(0, 40, size=U); lens[7] = 0
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([rng.choice(I, size=n, replace=False) for n in lens]).astype(np.int64)
    users = rng.permutation(U)[:rows].astype(np.int32)
    top = np.stack([rng.choice(I, size=k, replace=False) for _ in range(rows)]).astype(np.int32)
    out = ops.eval_metrics(T(top, dev), T(users, dev), T(ptr, dev), T(col, dev)).cpu().numpy()
    for j, u in enumerate(users):
        truth = col[ptr[u]:ptr[u + 1]]
        if len(truth) == 0:
            assert (out[j] == 0).all()
            continue
        pos = [int(np.flatnonzero(top[j] == it)[0]) for it in truth if it in top[j]]
        ideal = sum(1.0 / math.log2(l + 2) for l in range(min(len(truth), k)))
        ref = [len(pos) / len(truth), sum(1.0 / math.log2(p + 2) for p in pos) / ideal, len(pos) / k]
        assert np.allclose(out[j], ref, rtol=1e-13, atol=0), (j, out[j], ref)


@pytest.mark.parametrize("name,D,task_len", [("toy", 8, 2), ("mid", 16, 4), ("mid", 256, 4), ("tiktok", 64, 32), ("hub", 64, 32), ("hub", 4, 8),
                                             ("wide18000", 128, 32), ("wide20000", 64, 32)])
def test_device_graph_builder_equals_host_builder(dev, name, D, task_len):
    """dmm_graph_build (rebuilt modality graph + SpMM work list on the device) is bit-identical to the numpy builder and planner.
    'hub': one item chosen by 6000 users -- an item row longer than the in-LDS sort (bitmap path) and giant rows in the plan."""
    from diffmm_amd import graph, synth
    rng = np.random.default_rng(1)
    if name == "hub":
        users, n_items = 6500, 300
        deg = rng.integers(1, 6, size=users); deg[17] = 0
        lists = []
        for u, d in enumerate(deg):
            rest = rng.choice(np.arange(1, n_items), size=max(0, d - 1), replace=False) if d else np.zeros(0, np.int64)
            lists.append(np.sort(np.concatenate([[0], rest])) if (d and u < 6000) else np.sort(rng.choice(np.arange(1, n_items), size=d, replace=False)))
        items = np.concatenate(lists)
    elif name.startswith("wide"):  # item counts around the LDS-histogram limit of the builder (18432): both code paths
        users, n_items = 2500, int(name[4:])
        deg = rng.integers(0, 9, size=users)
        pop = 1.0 / np.arange(1, n_items + 1) ** 0.8; pop /= pop.sum()
        items = np.concatenate([np.sort(rng.choice(n_items, size=d, replace=False, p=pop)) for d in deg if d])
    else:
        g = synth.named(name)
        users, n_items = g.users, g.items
        deg = np.bincount(g.trn_rows, minlength=users)
        pop = 1.0 / np.arange(1, n_items + 1); pop /= pop.sum()  # skewed item choice: a few very long item rows, like a real rebuild
        items = np.concatenate([np.sort(rng.choice(n_items, size=d, replace=False, p=pop)) for d in deg if d])
    ptr = np.concatenate([[0], np.cumsum(deg)])
    rows = np.repeat(np.arange(users), deg)
    rp, c, v = graph.normalized_bipartite_csr(rows, items, users, n_items)
    plan = graph.plan_tasks(rp, D, task_len, full=True)
    gb = graph.DeviceGraphBuilder(T(ptr, dev), users, n_items, D, task_len)
    for rep in range(2):  # the builder's scratch is reused
        adj = gb.build(T(items.astype(np.int32), dev), validate=True)
        assert np.array_equal(adj.rowptr.cpu().numpy(), rp), "rowptr"
        assert np.array_equal(adj.col.cpu().numpy(), c), "col"
        assert np.array_equal(adj.val.cpu().numpy(), v), "val (bitwise)"
        assert np.array_equal(adj.tasks.cpu().numpy(), plan.tasks) and adj.n_head_tasks == plan.head and adj.n_partial_blocks == plan.n_partial_blocks
        if plan.n_partial_blocks:
            assert np.array_equal(adj.fix.cpu().numpy(), plan.fix)
    if name == "hub":
        assert plan.n_partial_blocks > 0 and np.diff(rp).max() > 4096
    if D >= 16:  # planned for bf16 gather operands: the plan of width D / 2, labelled with the full width
        a8 = graph.DeviceGraphBuilder(T(ptr, dev), users, n_items, D, task_len, vec=8).build(T(items.astype(np.int32), dev))
        h8 = graph.adjacency_from_csr(rp, c, v, D, dev, task_len, vec=8)
        assert a8.vec == 8 and a8.D == D and np.array_equal(a8.tasks.cpu().numpy(), h8.tasks.cpu().numpy()) and a8.n_head_tasks == h8.n_head_tasks
        assert a8.n_partial_blocks == h8.n_partial_blocks and np.array_equal(a8.val.cpu().numpy(), v)


# ------------------------------------------------------------------------------------------------
# diffusion pieces vs the reference's vectors
# ------------------------------------------------------------------------------------------------
def _diff_engine(z, dev):
    from diffmm_amd.engine import DiffusionEngine
    hp = dict(noise_scale=float(z["noise_scale"]), noise_min=0.0001, noise_max=0.02, sim_weight=float(z["sim_weight"]), reg=float(z["reg"]), lr=1e-3)
    E = DiffusionEngine(int(z["items"]), int(z["d"]), int(z["H"]), 10, 1, int(z["T"]), hp, dev, int(z["B"]))
    E.load_reference_state(0, {k[2:]: z[k] for k in z.files if k.startswith("p/")})
    return E


@pytest.mark.parametrize("name", ["toy", "mid"])
def test_denoise_and_diffusion_loss_vs_reference(dev, name):
    from diffmm_amd import ops
    z = load(f"diffusion_{name}.npz")
    E = _diff_engine(z, dev)
    B, I = int(z["B"]), int(z["items"])
    x0, t, noise, feat = T(z["x0"]).to(dev), T(z["t"].astype(np.int32)).to(dev), T(z["noise"]).to(dev), T(z["feat"]).to(dev)
    ie = T(z["i_embs"]).to(dev)
    # q_sample + time embedding
    P = E.P[0]
    xcat = torch.zeros(B, E.Kp, device=dev)
    ops.qsample_temb(E._set_x0(x0), noise, t, E.sqrt_ac, E.sqrt_1mac, P.view("emb_layer.weight"), P.view("emb_layer.bias"), xcat, I)
    close(xcat[:, :I], z["x_t"], what="q_sample")
    assert float(xcat[:, I + 10:].abs().max()) == 0.0 if E.Kp > I + 10 else True
    xt = T(z["x_t"]).to(dev)
    close(E.denoise_forward(0, xt, t, feat, precision=3), z["gate/out"], rtol=1e-4, atol=2e-5, what="Denoise (gate)")
    close(E.denoise_forward(0, xt, t, None, precision=3), z["nogate/out"], rtol=1e-4, atol=2e-5, what="Denoise (no gate)")
    got = E.denoise_forward(0, xt, t, None, precision=1)
    scale = float(np.abs(z["nogate/out"]).max())
    assert float((got.cpu() - T(z["nogate/out"])).abs().max()) < 3e-2 * scale, "bf16 single-product mode within 3 % of the output scale"
    close(E.generate_view(0, x0, precision=3), z["genview/out"], rtol=2e-4, atol=5e-5, what="generate_view")
    gv = E.generate_view(0, x0, precision=1).cpu()  # dedicated bf16 chain
    gscale = float(np.abs(z["genview/out"]).max())
    assert float((gv - T(z["genview/out"])).abs().max()) < 3e-2 * gscale, "bf16 reverse chain within 3 % of the output scale"
    # sampling_step > 0: the chain starts from q_sample(x_0, t = sampling_step - 1) (Model.py:310-313); same draws through the oracle
    from oracle import diffmm_oracle as O
    sched = O.schedule(int(z["T"]), float(z["noise_scale"]), 0.0001, 0.02)
    Por = {k[2:]: T(z[k]) for k in z.files if k.startswith("p/")}
    ss = max(1, int(z["T"]) // 2)
    want = O.generate_view(Por, T(z["x0"]), sched, ss, randn=T(z["noise"]))
    close(E.generate_view(0, E.noised_start(0, x0, ss, noise=noise), precision=3), want, rtol=2e-4, atol=5e-5, what="generate_view, noised start")
    drawn = E.noised_start(0, x0, ss).clone()  # in-kernel Philox draw: unit-norm rows, supported on x_0's non-zeros only
    a_, b_ = float(E.sqrt_ac[ss - 1]), float(E.sqrt_1mac[ss - 1])
    resid = (drawn - a_ * x0) / b_
    assert float(resid[x0 == 0].abs().max()) == 0.0, "noise only where x_0 is non-zero"
    nz = (x0 != 0).float().sum(1)
    # E[sum over the row's non-zeros of n_j^2 / ||n||^2] = nnz / I
    assert abs(float((resid ** 2).sum()) / float(nz.sum()) * I - 1.0) < 0.2
    assert not torch.equal(drawn, E.noised_start(0, x0, ss)), "every call draws fresh noise"
    # training_losses with the recorded draws: value and every gradient
    means = E.train_step(x0, [feat], ie, t=[t], noise=[noise], precision=3, update=False)
    close(E.act[0]["loss"][:B], z["tl/loss"], rtol=1e-4, atol=1e-7, what="training_losses")
    close(means[0], float(z["tl/loss"].mean()), rtol=1e-4)
    # train_step scales gradients by 1/total (Main.py:176); the golden grads are of loss.mean() -> rescale
    total = means[0]
    for k in ("emb_layer.weight", "emb_layer.bias", "in_layers.0.bias", "out_layers.0.weight", "out_layers.0.bias", "gate_layer.weight", "gate_layer.bias"):
        ref = z[f"tl/grad/{k}"]
        close(E.P[0].g(k) * total, ref, rtol=2e-3, atol=2e-4 * float(np.abs(ref).max()) + 1e-9, what="grad " + k)
    ref = z["tl/grad/in_layers.0.weight"]
    close(E.P[0].g("in_layers.0.weight")[:, : I + 10] * total, ref, rtol=2e-3, atol=2e-4 * float(np.abs(ref).max()), what="grad W1")
    # the benchmark's mode: the same step with single-product bf16 GEMMs (fast NT kernel + transpose-cast operands)
    means1 = E.train_step(x0, [feat], ie, t=[t], noise=[noise], precision=1, update=False)
    close(means1[0], float(z["tl/loss"].mean()), rtol=2e-2, what="loss (bf16 path)")
    for k in ("emb_layer.weight", "in_layers.0.bias", "out_layers.0.weight", "out_layers.0.bias", "gate_layer.weight", "gate_layer.bias", "in_layers.0.weight"):
        ref = z[f"tl/grad/{k}"]
        got = E.P[0].g(k) * means1[0]
        if k == "in_layers.0.weight":
            got = got[:, : I + 10]
        err = float((got.cpu() - T(ref)).abs().max()) / float(np.abs(ref).max())
        assert err < 6e-2, f"bf16-path grad {k}: max err {err:.3e} of the gradient scale"


@pytest.mark.parametrize("D", [64, 256])
def test_train_step_batched_bf16_matches_split_precision(dev, D):
    """The benchmark's training step (precision 1: the modalities' large GEMMs batched onto the bf16 NT kernels, 256 x 256 LDS-DMA
    tile included at this size; the item-space kernels incl. their d = 256 forms) against the same step in split-bf16 (~fp32) arithmetic,
    which the golden vectors pin."""
    from diffmm_amd.engine import DiffusionEngine
    torch.manual_seed(3)
    items, H, M, T_, B = 2500, 1024, 3, 20, 2048
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=11)
    E = DiffusionEngine(items, D, H, 10, M, T_, hp, dev, B)
    for m in range(M):
        P = E.P[m]
        P.data.normal_(0, 0.02)
        P.view("in_layers.0.weight")[:, items + 10:].zero_()  # pad columns of W1 are zero by construction
    x0 = (torch.rand(B, items, device=dev) < 0.004).float()
    feats = [torch.nn.functional.normalize(torch.randn(items, D, device=dev)) for _ in range(M)]
    ie = torch.randn(items, D, device=dev) * 0.1
    t = [torch.randint(0, T_, (B,), device=dev, dtype=torch.int32) for _ in range(M)]
    noise = [torch.randn(B, items, device=dev) for _ in range(M)]
    m3 = E.train_step(x0, feats, ie, t=t, noise=noise, precision=3, update=False)
    g3 = [{k: E.P[m].g(k).clone() for k in ("in_layers.0.weight", "out_layers.0.weight", "in_layers.0.bias", "out_layers.0.bias", "gate_layer.weight", "emb_layer.weight")}
          for m in range(M)]
    m1 = E.train_step(x0, feats, ie, t=t, noise=noise, precision=1, update=False)
    for m in range(M):
        assert abs(m1[m] - m3[m]) < 2e-2 * abs(m3[m]), f"loss of modality {m}: {m1[m]} vs {m3[m]}"
        for k, ref in g3[m].items():
            got = E.P[m].g(k)
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            assert err < 6e-2, f"modality {m} grad {k}: max err {err:.3e} of the gradient scale"
            cos = float((got * ref).sum() / (got.norm() * ref.norm()))
            assert cos > 0.999, f"modality {m} grad {k}: cosine {cos:.5f}"


def test_train_step_device_side_gradient_scale(dev):
    """train_step(host_means=False): the backward's scale 1 / (minibatch x sum of the loss means) (Main.py:176) reaches the kernels as a device scalar
    and the means come back as a device tensor -- same means and same gradients as the step that reads the means back first."""
    from diffmm_amd.engine import DiffusionEngine
    torch.manual_seed(9)
    items, D, H, M, T_, B = 1500, 64, 1024, 3, 20, 700
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=11)
    E = DiffusionEngine(items, D, H, 10, M, T_, hp, dev, B)
    for m in range(M):
        E.P[m].data.normal_(0, 0.02)
        E.P[m].view("in_layers.0.weight")[:, items + 10:].zero_()
    x0 = (torch.rand(B, items, device=dev) < 0.004).float()
    feats = [torch.nn.functional.normalize(torch.randn(items, D, device=dev)) for _ in range(M)]
    ie = torch.randn(items, D, device=dev) * 0.1
    t = [torch.randint(0, T_, (B,), device=dev, dtype=torch.int32) for _ in range(M)]
    noise = [torch.randn(B, items, device=dev) for _ in range(M)]
    host = E.train_step(x0, feats, ie, t=t, noise=noise, precision=1, update=False)
    g_host = [E.P[m].grad.clone() for m in range(M)]
    devm = E.train_step(x0, feats, ie, t=t, noise=noise, precision=1, update=False, host_means=False)
    assert isinstance(devm, torch.Tensor) and devm.dtype == torch.float64
    close(devm, np.asarray(host), rtol=1e-4, what="loss means")  # (two forwards of the bf16 step differ by the order of their K-slice atomics: ~2e-6)
    for m in range(M):
        scale = float(g_host[m].abs().max())
        assert float((E.P[m].grad - g_host[m]).abs().max()) <= 1e-3 * scale, f"modality {m}: gradients with the device-side scale"  # (two runs of the bf16 step differ by atomics order: 1e-4)



@pytest.mark.parametrize("B,I,D", [(300, 700, 64), (4096, 6710, 64), (1116, 2531, 128), (700, 1900, 256)])
def test_item_project_and_expand(dev, B, I, D):
    """csrc/itemspace.hip against fp64 products of the bf16-rounded operands (only the fp32 summation order differs)."""
    from diffmm_amd import ops
    torch.manual_seed(B + D)
    ld = (I + 10 + 63) // 64 * 64
    X = torch.randn(B, ld, device=dev)  # columns >= I hold (finite) junk, like the time-embedding columns of [x_t | temb | 0]
    F = torch.randn(I, D, device=dev)
    out = torch.zeros(B, D, device=dev)
    ops.item_project(X, F, out, I)
    ref = X[:, :I].bfloat16().double() @ F.bfloat16().double()
    close(out, ref, rtol=1e-5, atol=2e-4 * np.sqrt(I / 64), what="item_project")
    P = torch.randn(B, D, device=dev)
    R = torch.randn(B, ld, device=dev)
    want = R.clone().double()
    want[:, :I] += P.bfloat16().double() @ F.bfloat16().double().t()
    Y = torch.full((B, ld), 5.0, device=dev)
    ops.item_expand(P, F, R, Y, I)
    close(Y[:, :I], want[:, :I], rtol=1e-5, atol=1e-4, what="item_expand")
    assert float((Y[:, I:] - 5.0).abs().max()) == 0.0, "columns beyond I untouched"
    Rc = R.clone()
    ops.item_expand(P, F, Rc, Rc, I)  # in place
    close(Rc[:, :I], want[:, :I], rtol=1e-5, atol=1e-4, what="item_expand in place")
    assert torch.equal(Rc[:, I:], R[:, I:])


@pytest.mark.parametrize("D,B,I", [(64, 300, 1000), (128, 130, 517), (256, 200, 777), (256, 600, 1100)])  # (d = 256: the [256 rows, 64 items] tile)
def test_item_expand_fused_equals_the_separate_passes(dev, D, B, I):
    """dmm_item_expand_fused against the passes it replaces, bit for bit where the arithmetic is the same: forward (item_expand -> cast_bf16,
    transpose_cast_bf16, incl. the time-embedding / pad columns and the zero batch-row padding of the transposed copy) and loss gradient
    (diff_loss(dout) -> item_expand -> cast / transpose-cast; column sums up to summation order)."""
    from diffmm_amd import ops
    torch.manual_seed(B + I)
    Kp = (I + 10 + 63) // 64 * 64
    Bp = (B + 63) // 64 * 64
    P, F = torch.randn(B, D, device=dev), torch.randn(I, D, device=dev)
    R = torch.randn(B, Kp, device=dev)
    R[:, I + 10:] = 0.0
    Y = R.clone()
    ops.item_expand(P, F, R, Y, I)
    want16 = torch.zeros(B, Kp, device=dev, dtype=torch.bfloat16); ops.cast_bf16(Y, want16)
    wantT = torch.full((Kp, Bp), 3.0, device=dev, dtype=torch.bfloat16); ops.transpose_cast_bf16(Y, wantT, rows=B, cols=Kp, rows_pad=Bp)
    got16 = torch.full((B, Kp), 5.0, device=dev, dtype=torch.bfloat16); gotT = torch.full((Kp, Bp), 5.0, device=dev, dtype=torch.bfloat16)
    ops.item_expand_fused(P, F, R, got16, gotT, I, Kp, Kp)
    assert torch.equal(got16, want16), "forward: row-major bf16 operand"
    assert torch.equal(gotT, wantT), "forward: transposed bf16 operand (batch rows >= B zero)"
    # loss gradient
    out, x0 = torch.randn(B, Kp, device=dev), (torch.rand(B, Kp, device=dev) < 0.05).float()
    w = torch.rand(B, device=dev, dtype=torch.float64) + 0.5
    um, ui = torch.randn(B, D, device=dev), torch.randn(B, D, device=dev)
    gscale = 1.0 / (B * 3.7)
    loss = torch.zeros(B, device=dev, dtype=torch.float64)
    dout, dum = torch.zeros(B, Kp, device=dev), torch.zeros(B, D, device=dev)
    ops.diff_loss(out, x0, w, um, ui, I, 0.1, 0.0, gscale, loss, dout, dum)
    ops.item_expand(dum, F, dout, dout, I)
    want16 = torch.zeros(B, Kp, device=dev, dtype=torch.bfloat16); ops.cast_bf16(dout, want16)
    wantT = torch.zeros(I, Bp, device=dev, dtype=torch.bfloat16); ops.transpose_cast_bf16(dout, wantT, rows=B, cols=I, rows_pad=Bp)
    wantc = torch.zeros(I, device=dev); ops.colsum(dout, B, I, wantc)
    dum2 = torch.zeros(B, D, device=dev)
    ops.diff_loss(out, x0, w, um, ui, I, 0.1, 0.0, gscale, None, None, dum2)
    assert torch.equal(dum2, dum), "diff_loss: dum-only mode"
    got16 = torch.full((B, Kp), 5.0, device=dev, dtype=torch.bfloat16); gotT = torch.full((I, Bp), 5.0, device=dev, dtype=torch.bfloat16)
    gotc = torch.zeros(I, device=dev)
    ops.item_expand_fused(dum2, F, out, got16, gotT, I, Kp, I, x0=x0, w=w, gscale=gscale, csum=gotc)
    # (the same fp32 operations up to one fp32 ulp of the sum, which moves a value across a bf16 rounding boundary with probability ~2^-16: a
    # handful of elements in 3e5 differ by one bf16 ulp)
    scale = float(want16.float().abs().max())
    for got, want, what in ((got16, want16, "row-major"), (gotT, wantT, "transposed")):
        diff = (got.float() - want.float()).abs()
        assert float((diff > 0).float().mean()) < 1e-4 and float(diff.max()) <= 2.0 ** -8 * scale, f"loss gradient, {what} bf16 operand"
    assert float(got16[:, I:].float().abs().max()) == 0.0, "columns >= I zero"
    close(gotc, wantc, rtol=1e-4, atol=1e-6 * float(wantc.abs().max()) + 1e-9, what="column sums")


def test_x0_bit_masks_equal_the_dense_rows(dev):
    """dmm_csr_row_bits and the three readers of x_0 that take bit masks instead of dense 0/1 rows: q_sample (explicit noise and the
    in-kernel Philox draw), the forward diffusion loss and the fused loss gradient -- bit for bit what the dense rows give."""
    from diffmm_amd import ops, synth
    from diffmm_amd.DataHandler import DiffusionData
    torch.manual_seed(9)
    g = synth.named("mid")
    U, I, D, B, te = g.users, g.items, 64, 200, 10
    dd = DiffusionData(g.trn_rows, g.trn_cols, U, I, dev)
    users = torch.randperm(U, device=dev)[:B]
    rows = dd.csr_rows(users)
    x0 = dd.rows(users)
    W = (I + 31) // 32
    bits = torch.full((B, W + 3), -1, dtype=torch.int32, device=dev)
    ops.csr_row_bits(rows.ptr, rows.col, rows.users, I, bits)
    cols = torch.arange(I, device=dev)
    unpacked = ((bits[:, cols >> 5] >> (cols & 31)) & 1).float()
    assert torch.equal(unpacked, x0) and int(bits[:, W:].abs().sum()) == 0, "bit masks (whole leading dimension written, spare bits zero)"
    Kp = (I + te + 63) // 64 * 64
    t = torch.randint(0, 6, (B,), device=dev, dtype=torch.int32)
    sa, sb = torch.rand(6, device=dev), torch.rand(6, device=dev)
    we, be = torch.randn(te, te, device=dev), torch.randn(te, device=dev)
    x0p = torch.zeros(B, Kp, device=dev); x0p[:, :I] = x0
    noise = torch.randn(B, I, device=dev)
    for kw in (dict(noise=noise), dict(noise=None, draw_noise=True, seed=5, offset=77)):
        a, b = torch.zeros(B, Kp, device=dev), torch.zeros(B, Kp, device=dev)
        ops.qsample_temb(x0p, kw.get("noise"), t, sa, sb, we, be, a, I, draw_noise=kw.get("draw_noise", False), seed=kw.get("seed", 0), offset=kw.get("offset", 0))
        ops.qsample_temb_bits(bits, kw.get("noise"), t, sa, sb, we, be, b, I, draw_noise=kw.get("draw_noise", False), seed=kw.get("seed", 0), offset=kw.get("offset", 0))
        assert torch.equal(a, b), "q_sample from bit masks"
    out = torch.randn(B, Kp, device=dev)
    w = torch.rand(B, device=dev, dtype=torch.float64) + 0.5
    um, ui = torch.randn(B, D, device=dev), torch.randn(B, D, device=dev)
    l1, l2 = torch.zeros(B, device=dev, dtype=torch.float64), torch.zeros(B, device=dev, dtype=torch.float64)
    ops.diff_loss(out, x0p, w, um, ui, I, 0.1, 0.01, 0.0, l1, None, None)
    ops.diff_loss_bits(out, bits, w, um, ui, I, 0.1, 0.01, l2)
    assert torch.equal(l1, l2), "diffusion loss from bit masks"
    F, dum = torch.randn(I, D, device=dev), torch.randn(B, D, device=dev)
    Bp = (B + 63) // 64 * 64
    res = []
    for kw in (dict(x0=x0p), dict(x0bits=bits)):
        y16, yT, cs = torch.zeros(B, Kp, device=dev, dtype=torch.bfloat16), torch.zeros(I, Bp, device=dev, dtype=torch.bfloat16), torch.zeros(I, device=dev)
        ops.item_expand_fused(dum, F, out, y16, yT, I, Kp, I, w=w, gscale=1e-3, csum=cs, **kw)
        res.append((y16, yT, cs))
    sc = float(res[0][0].float().abs().max())
    for k in (0, 1):
        d_ = (res[0][k].float() - res[1][k].float()).abs()
        assert float((d_ > 0).float().mean()) < 1e-4 and float(d_.max()) <= 2.0 ** -8 * sc, "fused loss gradient from bit masks"
    close(res[1][2], res[0][2], rtol=1e-4, atol=1e-6 * float(res[0][2].abs().max()) + 1e-12, what="column sums")


@pytest.mark.parametrize("D", [64, 128, 256])
def test_item_project_batched_layouts(dev, D):
    """dmm_item_project_batched (the joint step's modality projections and their weight gradients) against fp64 products of the bf16-rounded
    operands: F given as an nn.Linear weight [D, K] with a bias (forward), the result stored transposed (weight gradient), three problems of
    different K / rows in one launch, accumulation into a non-zero output."""
    from diffmm_amd import ops
    torch.manual_seed(D)
    rows, Ks = 700, (128, 768, 96)
    Xs = [torch.randn(rows, k, device=dev) for k in Ks]
    Ws = [torch.randn(D, k, device=dev) * 0.1 for k in Ks]
    bs = [torch.randn(D, device=dev) for _ in Ks]
    outs = [torch.full((rows, D), 0.5, device=dev) for _ in Ks]
    ops.item_project_batched([dict(X=Xs[i], F=Ws[i], out=outs[i], bias=bs[i], K=Ks[i]) for i in range(3)], D, f_transposed=True)
    for i in range(3):
        ref = 0.5 + Xs[i].bfloat16().double() @ Ws[i].bfloat16().double().t() + bs[i].double()
        close(outs[i], ref, rtol=1e-5, atol=2e-4 * np.sqrt(Ks[i] / 64), what=f"projection {i}")
    # weight gradient: E^T [F, I] x dfeat [I, D] -> dW [D, F]
    I = 1000
    Ip = (I + 31) // 32 * 32
    ETs = [torch.zeros(k, Ip, device=dev) for k in Ks]
    for t in ETs:
        t[:, :I].normal_()
        t[:, I:] = 7.0  # readable padding: ignored (the F operand has no rows there)
    dfs = [torch.randn(I, D, device=dev) for _ in Ks]
    dWs = [torch.zeros(D, k, device=dev) for k in Ks]
    ops.item_project_batched([dict(X=ETs[i], F=dfs[i], out=dWs[i], K=I) for i in range(3)], D, out_transposed=True)
    for i in range(3):
        ref = (ETs[i][:, :I].bfloat16().double() @ dfs[i].bfloat16().double()).t()
        close(dWs[i], ref, rtol=1e-5, atol=2e-4 * np.sqrt(I / 64), what=f"weight gradient {i}")


def test_reverse_chains_on_streams_match_single_chain(dev):
    """generate_views_bf16 (one HIP stream per modality, shared x_0 copies, recorded launch plan) against the single-modality chain:
    the same arithmetic on the same operands -- equal up to the summation order of differently tiled GEMMs -- and bit-identical
    from call to call (a chain that read a buffer another chain still owned would not be)."""
    from diffmm_amd.engine import DiffusionEngine
    torch.manual_seed(5)
    items, D, H, M, T_, B = 1500, 64, 1024, 3, 7, 700
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3)
    E = DiffusionEngine(items, D, H, 10, M, T_, hp, dev, B)
    for m in range(M):
        E.P[m].data.normal_(0, 0.03)
        E.P[m].view("in_layers.0.weight")[:, items + 10:].zero_()
    x0 = (torch.rand(B, items, device=dev) < 0.01).float()
    first = [v.clone() for v in E.generate_views_bf16(x0)]
    again = [v.clone() for v in E.generate_views_bf16(x0)]
    for m in range(M):
        assert torch.equal(first[m], again[m]), f"modality {m}: the stream chains are not reproducible"
        single = E.generate_view_bf16(m, x0)
        scale = float(single.abs().max())
        assert float((first[m] - single).abs().max()) < 2e-3 * scale, f"modality {m}: streams vs single chain"
    E.hp["chain_streams"] = False
    batched = E.generate_views_bf16(x0)
    for m in range(M):
        assert float((first[m] - batched[m]).abs().max()) < 2e-3 * float(batched[m].abs().max()), f"modality {m}: streams vs batched launches"
    # the running x_t carried in bf16 (default) against the fp32 state of round 1: what the state's low bits can move is bounded by the
    # bf16 rounding of a value the next layer reads in bf16 anyway (and the last step has c2 = 0: its output is Denoise(x_1) exactly)
    E.hp["chain_streams"], E.hp["chain_state"] = True, "fp32"
    wide = E.generate_views_bf16(x0)
    for m in range(M):
        assert float((first[m] - wide[m]).abs().max()) < 1e-2 * float(wide[m].abs().max()), f"modality {m}: bf16 state vs fp32 state"


def test_split_casts_and_gemv(dev):
    """dmm_cast_bf16_split / dmm_transpose_cast_bf16_split: hi is the plain cast, hi + lo recovers the fp32 value to 2^-16 relative;
    dmm_gemv_rows against fp64."""
    from diffmm_amd import ops
    torch.manual_seed(3)
    R, Cc, ld = 70, 100, 104
    x = torch.randn(R, ld, device=dev)
    hi, lo = torch.full((R, ld), 7.0, device=dev, dtype=torch.bfloat16), torch.full((R, ld), 7.0, device=dev, dtype=torch.bfloat16)
    ops.cast_bf16_split(x, hi, lo, cols=Cc, cols_pad=ld)
    assert torch.equal(hi[:, :Cc], x[:, :Cc].bfloat16()) and float(hi[:, Cc:].abs().max()) == 0.0 and float(lo[:, Cc:].abs().max()) == 0.0
    err = (hi.float() + lo.float() - x)[:, :Cc].abs() / x[:, :Cc].abs().clamp_min(1e-20)
    assert float(err.max()) < 2.0 ** -15
    Rp = 72
    thi, tlo = torch.full((Cc, Rp), 7.0, device=dev, dtype=torch.bfloat16), torch.full((Cc, Rp), 7.0, device=dev, dtype=torch.bfloat16)
    ops.transpose_cast_bf16_split(x, thi, tlo, rows=R, cols=Cc, rows_pad=Rp)
    assert torch.equal(thi[:, :R], x[:, :Cc].t().bfloat16()) and float(thi[:, R:].abs().max()) == 0.0 and float(tlo[:, R:].abs().max()) == 0.0
    assert torch.equal(tlo[:, :R], lo[:, :Cc].t())
    v, out = torch.randn(Cc, device=dev), torch.zeros(R, device=dev)
    ops.gemv_rows(x, v, out, cols=Cc)
    close(out, x[:, :Cc].double() @ v.double(), rtol=1e-5, atol=1e-5, what="gemv_rows")


def fill_named(E, m: int, std: float, seed: int) -> None:
    """Random Denoise weights that do NOT depend on the engine's flat parameter layout: every named tensor is drawn from its own generator (seeded by
    the tensor's name).  Filling the flat buffer in one normal_() call made the weights -- and with them every observed error of the chain tests --
    change whenever the layout was re-ordered (round 3: the small tensors moved in front of b2 / W1 / W2; two bounds below were crossed by that
    re-draw, not by a kernel change)."""
    import zlib
    P = E.P[m]
    for k in P.names:
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(k.encode()) + 17 * m)
        P.view(k).copy_((torch.randn(P.shapes[k], generator=g) * std).to(P.data.device))
    P.view("in_layers.0.weight")[:, E.I + E.te:].zero_()
    E.w16_step = [-1] * E.M
    E.w16_train_step = [-1] * E.M


def test_latent_chain_matches_stepwise_chain(dev):
    """generate_views_latent (the reverse chain carried as z_t = x_t W1x^T: T - 1 GEMMs of [B, H] x [H, H] and one item-sized GEMM)
    against the stepwise chains it replaces and against the reference's vectors:
      * synthetic weights, M = 3: latent vs the ~fp32 stepwise chain (precision 3) -- and no further from it than the stepwise bf16
        chain is (the latent state is fp32; the stepwise state is re-rounded to bf16 every step);
      * CSR start (gather of W1^T rows) vs dense start (layer-1 GEMM): the same products in another summation order;
      * bit-identical from call to call (recorded plan, static buffers);
      * Denoise with 1, 2 and 3 hidden widths against the reference's generate_view output (tests/golden);
      * the rebuilt edges (per-user top-k): the latent chain reproduces the ~fp32 chain's edges at least as well as the stepwise bf16 chain."""
    from diffmm_amd import synth
    from diffmm_amd.DataHandler import DiffusionData
    from diffmm_amd.engine import DiffusionEngine
    torch.manual_seed(21)
    g = synth.named("mid")
    U, I, D, H, M, T_, B = g.users, g.items, 64, 1024, 3, 9, 300
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=5)
    E = DiffusionEngine(I, D, H, 10, M, T_, hp, dev, B)
    assert E._latent_mode() and float(E.c2[0]) == 0.0
    for m in range(M):
        fill_named(E, m, 0.04, 21)
    dd = DiffusionData(g.trn_rows, g.trn_cols, U, I, dev)
    users = torch.randperm(U, device=dev)[:B]
    x0 = dd.rows(users)
    lat = [v.clone() for v in E.generate_views_latent(xT=[x0] * M, B=B)]
    again = E.generate_views_latent(xT=[x0] * M, B=B)
    assert all(torch.equal(lat[m], again[m]) for m in range(M)), "latent chain reproducible"
    sparse = [v.clone() for v in E.generate_views_latent(rows=dd.csr_rows(users))]
    E.check_chain_error()
    # the persistent kernel (all T - 1 steps in one launch sequence, z in registers) against one launch per step (z through memory)
    assert E._latent_fused()
    E.hp["latent_fused"] = False
    E._chain_plans.clear()
    per_step = [v.clone() for v in E.generate_views_latent(xT=[x0] * M, B=B)]
    E.hp["latent_fused"] = True
    E._chain_plans.clear()
    for m in range(M):
        sc = float(per_step[m].abs().max())
        assert float((per_step[m] - lat[m]).abs().max()) < 6e-3 * sc, f"modality {m}: persistent latent kernel vs per-step launches"
    E.hp["chain_state"] = "bf16"
    step16 = [v.clone() for v in E.generate_views_bf16(x0)]
    for m in range(M):
        exact = E.generate_view(m, x0, precision=3)
        scale = float(exact.abs().max())
        e_lat, e_step = float((lat[m] - exact).abs().max()) / scale, float((step16[m] - exact).abs().max()) / scale
        assert e_lat < 2e-2, f"modality {m}: latent chain vs fp32 stepwise chain: {e_lat:.3e} of the output scale"
        assert e_lat < 1.5 * e_step + 1e-3, f"modality {m}: latent {e_lat:.3e} vs stepwise bf16 {e_step:.3e}"
        assert float((sparse[m] - lat[m]).abs().max()) < 2e-3 * scale, f"modality {m}: CSR start vs dense start"
        one = E.generate_view(m, x0, precision=1)  # the single-modality surface takes the latent chain too
        assert float((one - lat[m]).abs().max()) < 2e-3 * scale, f"modality {m}: single-modality latent chain"
    ptr32 = dd.deg_ptr32
    e_lat = E.rebuild_edges(dd.rows, [users], ptr32, int(dd.ptr[-1]), precision=1, csr=(dd.ptr, dd.col32, 0))
    # lean chain buffers (what a large item count switches on by itself): the [rows, I] views are formed 256 rows at a time and consumed by the
    # top-k at once, the bf16 item-sized buffers keep B_max rows -- the same chain, the same GEMM per row: bit-identical edges; the dense start
    # (rows re-chunked to B_max) likewise; the stepwise chain is refused
    El = DiffusionEngine(I, D, H, 10, M, T_, dict(hp, chain_view_bytes=M * 256 * E.Kp * 4), dev, 128, chain_rows=B)
    assert El.lean and El.view_rows == 256 and El.dense_rows_max == 128 and El.cxa[0].shape[0] == 256 and El.cxa16[0].shape[0] == 128
    for m in range(M):
        El.P[m].data.copy_(E.P[m].data)
    e_lean = El.rebuild_edges(dd.rows, [users], ptr32, int(dd.ptr[-1]), precision=1, csr=(dd.ptr, dd.col32, 0))
    e_dense, e_ldense = (X.rebuild_edges(dd.rows, [users], ptr32, int(dd.ptr[-1]), precision=1) for X in (E, El))
    for m in range(M):  # (a GEMM over fewer rows may run on another tile shape: the same products in another summation order -- near-tie flips only)
        s_csr, s_dense = float((e_lean[m] == e_lat[m]).float().mean()), float((e_dense[m] == e_ldense[m]).float().mean())
        print(f"modality {m}: lean vs full chain buffers: {s_csr:.4f} (CSR start), {s_dense:.4f} (dense start in B_max-row chunks) of the edges equal")
        assert s_csr > 0.995 and s_dense > 0.995
    El.check_chain_error()
    with pytest.raises(ValueError):
        El.generate_views_latent(rows=dd.csr_rows(users))
    del El
    E.hp["chain_mode"] = "stepwise"
    assert not E._latent_mode()
    e_step = E.rebuild_edges(dd.rows, [users], ptr32, int(dd.ptr[-1]), precision=1, csr=(dd.ptr, dd.col32, 0))
    e_exact = E.rebuild_edges(dd.rows, [users], ptr32, int(dd.ptr[-1]), precision=3)  # ~fp32 stepwise chain: the reference's arithmetic
    for m in range(M):  # (random weights: scores are dense in near-ties, so a few per cent of the ranked slots differ for either bf16 chain)
        a_lat, a_step = float((e_lat[m] == e_exact[m]).float().mean()), float((e_step[m] == e_exact[m]).float().mean())
        print(f"modality {m}: edges equal to the fp32 chain's: latent {a_lat:.4f}, stepwise bf16 {a_step:.4f}")
        # Each bf16 chain against the fp32 chain's edges with an ABSOLUTE floor -- not against each other: which of the two lands closer
        # varies with the random weights (0.983 / 0.897 and 0.977 / 0.990 for one modality under two draws; the round-3 re-ordering of the flat layout
        # re-drew the weights of this test, which is what moved a result across the old relative bound -- the weights are now pinned by name).
        # The comparison that matters -- against the CPU oracle's chain at the metric's own sizes -- is tests/test_chain_oracle_gpu.py.
        assert a_lat > 0.95, f"modality {m}: latent chain reproduces only {a_lat:.4f} of the fp32 chain's edges"
        assert a_step > 0.85, f"modality {m}: stepwise bf16 chain reproduces only {a_step:.4f} of the fp32 chain's edges"
    for name in ("toy", "mid", "mid_deep", "mid_deep3"):
        z = load(f"diffusion_{name}.npz")
        dims = [int(x) for x in z["dims"]] if "dims" in z.files else [int(z["H"])]
        Eg = DiffusionEngine(int(z["items"]), int(z["d"]), int(z["H"]), 10, 1, int(z["T"]),
                             dict(noise_scale=float(z["noise_scale"]), noise_min=0.0001, noise_max=0.02, sim_weight=float(z["sim_weight"]), reg=float(z["reg"]), lr=1e-3),
                             dev, int(z["B"]), hidden=dims)
        Eg.load_reference_state(0, {k[2:]: z[k] for k in z.files if k.startswith("p/")})
        gv = Eg.generate_views_latent(xT=[T(z["x0"]).to(dev)], B=int(z["B"]))[0]
        gscale = float(np.abs(z["genview/out"]).max())
        err = float((gv.cpu() - T(z["genview/out"])).abs().max()) / gscale
        assert err < 3e-2, f"{name}: latent chain vs the reference's generate_view: {err:.3e} of the output scale"


@pytest.mark.parametrize("rows,T_", [(5700, 4), (1100, 7), (256, 2)])
def test_persistent_latent_kernel_group_barriers(dev, rows, T_):
    """dmm_latent_chain with more groups than one launch holds (3 x 23 row blocks > 64 groups: two launches), an odd and an even number of
    steps (the result lands in either ping-pong buffer), a ragged last row block and the minimal T = 2; against one launch per step.  The error
    word stays clear (no barrier timed out) and the call is bit-reproducible."""
    from diffmm_amd.engine import DiffusionEngine
    torch.manual_seed(rows + T_)
    I, D, H, M = 264, 64, 1024, 3
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=5)
    E = DiffusionEngine(I, D, H, 10, M, T_, hp, dev, rows)
    for m in range(M):
        E.P[m].data.normal_(0, 0.05)
        E.P[m].view("in_layers.0.weight")[:, I + 10:].zero_()
    x0 = (torch.rand(rows, I, device=dev) < 0.03).float()
    fused = [v.clone() for v in E.generate_views_latent(xT=[x0] * M, B=rows)]
    again = E.generate_views_latent(xT=[x0] * M, B=rows)
    assert all(torch.equal(fused[m], again[m]) for m in range(M)), "persistent kernel reproducible"
    E.check_chain_error()
    E.hp["latent_fused"] = False
    E._chain_plans.clear()
    per_step = E.generate_views_latent(xT=[x0] * M, B=rows)
    for m in range(M):
        sc = float(per_step[m].abs().max())
        err = float((per_step[m] - fused[m]).abs().max()) / sc
        # (fp32 sums in another order: where a bf16 h rounds the other way, one bf16 ulp = 4e-3 relative enters the next step)
        assert err < 6e-3, f"modality {m}: persistent kernel vs per-step launches: {err:.3e} of the output scale"


def test_csr_rows_equal_the_dense_path(dev):
    """x_0 as CSR rows (the I = 1e6 configuration cannot hold dense user rows): dmm_csr_dense_rows against the index
    arithmetic it replaces; a training step fed CSR rows against the same step fed the dense rows (the same operand buffer);
    the reverse chain started from CSR rows (layer 1 of the first step = a gather of W1^T rows, its residual a sparse add) against the
    chain started from the dense x_0 -- the same bf16 products and fp32 sums in another order."""
    from diffmm_amd import ops, synth
    from diffmm_amd.DataHandler import DiffusionData
    from diffmm_amd.engine import CsrRows, DiffusionEngine
    torch.manual_seed(7)
    g = synth.named("mid")
    U, I, D, H, M, T_, B = g.users, g.items, 64, 256, 3, 6, 128
    dd = DiffusionData(g.trn_rows, g.trn_cols, U, I, dev)
    users = torch.randperm(U, device=dev)[:B]
    want = torch.zeros(B, I, device=dev)
    r = torch.from_numpy(g.trn_rows.astype(np.int64)).to(dev); c = torch.from_numpy(g.trn_cols.astype(np.int64)).to(dev)
    full = torch.zeros(U, I, device=dev); full[r, c] = 1.0
    want = full[users]
    assert torch.equal(dd.rows(users), want), "dmm_csr_dense_rows"
    padded = torch.full((B, 264), 9.0, device=dev)
    ops.csr_dense_rows(dd.ptr, dd.col32, users, I, padded)
    assert torch.equal(padded[:, :I], want) and float(padded[:, I:].abs().max()) == 0.0, "whole leading dimension written"
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=11)
    E = DiffusionEngine(I, D, H, 10, M, T_, hp, dev, B)
    for m in range(M):
        fill_named(E, m, 0.05, 7)
    feats = [torch.nn.functional.normalize(torch.randn(I, D, device=dev)) for _ in range(M)]
    ie = torch.randn(I, D, device=dev) * 0.1
    t = [torch.randint(0, T_, (B,), device=dev, dtype=torch.int32) for _ in range(M)]
    noise = [torch.randn(B, I, device=dev) for _ in range(M)]
    rows = dd.csr_rows(users)
    assert isinstance(rows, CsrRows) and rows.shape == (B,)
    md = E.train_step(want, feats, ie, t=t, noise=noise, precision=1, update=False)
    gd = [E.P[m].grad.clone() for m in range(M)]
    mc = E.train_step(rows, feats, ie, t=t, noise=noise, precision=1, update=False)
    # the same operand buffer either way; the step itself accumulates a few products with fp32 atomics, so runs agree to round-off only
    for m in range(M):
        assert abs(md[m] - mc[m]) <= 1e-5 * abs(md[m]), "CSR-fed training step == dense-fed training step (loss; K-sliced products: summation order)"
        # (fp32 sums accumulated by K slices and then ROUNDED to a bf16 operand -- (W1x F)^T -- can land on the other side of a rounding
        # boundary from run to run: one bf16 ulp on one of 65k entries)
        # (observed: either ~1e-7 or, when one such flip happens, 2.8e-4 of the largest gradient entry)
        assert float((gd[m] - E.P[m].grad).abs().max()) <= 1e-3 * float(gd[m].abs().max()), "CSR-fed training step == dense-fed training step (gradients)"
    dense = [v.clone() for v in E.generate_views_bf16(want)]
    sparse = E.generate_views_bf16_csr(rows)
    for m in range(M):
        scale = float(dense[m].abs().max())
        # The bound is ONE bf16 ulp of the output scale (2^-8 = 3.9e-3): the two starts form the first step's sums in another order (a gather of
        # W1^T rows vs a GEMM), and where that moves a bf16 rounding of the running state the other way, one ulp of that element enters the chain --
        # nothing larger can.  (Round 2's 2e-3 was an observation under one draw of the weights, not a derived bound: the round-3 re-ordering of
        # the flat layout re-drew this test's weights and gave 2.0e-3.  The weights are now pinned by tensor name: fill_named.)
        assert float((sparse[m] - dense[m]).abs().max()) < 2.0 ** -8 * scale, f"modality {m}: CSR chain start vs dense chain"
    again = [v.clone() for v in sparse]
    sparse2 = E.generate_views_bf16_csr(rows)
    assert all(torch.equal(again[m], sparse2[m]) for m in range(M)), "reproducible (recorded plan, static user buffer)"
    # the rebuild's edges: per-user top-k of both chains agree except at near-ties
    ptr32 = dd.deg_ptr32
    e_d = E.rebuild_edges(dd.rows, [users], ptr32, int(dd.ptr[-1]), precision=1)
    e_s = E.rebuild_edges(dd.rows, [users], ptr32, int(dd.ptr[-1]), precision=1, csr=(dd.ptr, dd.col32, 0))
    for m in range(M):
        same = float((e_d[m] == e_s[m]).float().mean())
        assert same > 0.995, f"modality {m}: {same:.4f} of the rebuilt edges agree"


@pytest.mark.parametrize("name", ["mid_deep", "mid_deep3"])
def test_deep_denoise_engine_paths(dev, name):
    """Denoise with 2 / 3 hidden widths (the reference's ModuleLists, Model.py:156-160,210-218) through every engine path, against the
    reference's vectors: split-precision (~fp32) training step and chain at fp32 round-off; the benchmark's bf16 paths (batched training
    step, stream chains with the mid layers on bf16 GEMMs, CSR chain start) at the bf16 tolerances of the single-hidden-layer tests."""
    from diffmm_amd import synth
    from diffmm_amd.DataHandler import DiffusionData
    from diffmm_amd.engine import DiffusionEngine
    z = load(f"diffusion_{name}.npz")
    I, D, H, T_, B = int(z["items"]), int(z["d"]), int(z["H"]), int(z["T"]), int(z["B"])
    dims = [int(x) for x in z["dims"]]
    hp = dict(noise_scale=float(z["noise_scale"]), noise_min=0.0001, noise_max=0.02, sim_weight=float(z["sim_weight"]), reg=float(z["reg"]), lr=1e-3)
    E = DiffusionEngine(I, D, H, 10, 1, T_, hp, dev, B, hidden=dims)
    assert len(E.mids) == 2 * (len(dims) - 1)
    E.load_reference_state(0, {k[2:]: z[k] for k in z.files if k.startswith("p/")})
    x0, t, noise, feat, ie = T(z["x0"]).to(dev), T(z["t"].astype(np.int32)).to(dev), T(z["noise"]).to(dev), T(z["feat"]).to(dev), T(z["i_embs"]).to(dev)
    xt = T(z["x_t"]).to(dev)
    close(E.denoise_forward(0, xt, t, feat, precision=3), z["gate/out"], rtol=1e-4, atol=2e-5, what="Denoise (gate)")
    close(E.generate_view(0, x0, precision=3), z["genview/out"], rtol=2e-4, atol=5e-5, what="generate_view")
    gscale = float(np.abs(z["genview/out"]).max())
    for state in ("bf16", "fp32"):
        E.hp["chain_state"] = state
        gv = E.generate_views_bf16(x0)[0]
        assert float((gv.cpu() - T(z["genview/out"])).abs().max()) < 3e-2 * gscale, f"bf16 chain ({state} state) within 3 % of the output scale"
    g = synth.named("mid")
    dd = DiffusionData(g.trn_rows, g.trn_cols, g.users, g.items, dev)
    users = T(z["rows"].astype(np.int64)).to(dev)
    assert torch.equal(dd.rows(users), x0)
    E.hp["chain_state"] = "bf16"
    dense = E.generate_views_bf16(x0)[0].clone()
    sparse = E.generate_views_bf16_csr(dd.csr_rows(users))[0]
    # (the same bf16 products summed in another order in layer 1; every further bf16 layer re-rounds the difference)
    assert float((sparse - dense).abs().max()) < 5e-3 * gscale, "CSR chain start vs dense chain"
    for prec, rtol_loss, gtol in ((3, 1e-4, None), (1, 2e-2, 6e-2)):
        means = E.train_step(x0, [feat], ie, t=[t], noise=[noise], precision=prec, update=False)
        close(means[0], float(z["tl/loss"].mean()), rtol=rtol_loss, what=f"loss (precision {prec})")
        for k in [f[len("tl/grad/"):] for f in z.files if f.startswith("tl/grad/")]:
            ref = z[f"tl/grad/{k}"]
            got = E.P[0].g(k) * means[0]  # train_step scales by 1 / total (Main.py:176); the golden grads are of loss.mean()
            if k == "in_layers.0.weight":
                got = got[:, : I + 10]
            if gtol is None:
                close(got, ref, rtol=2e-3, atol=2e-4 * float(np.abs(ref).max()) + 1e-9, what="grad " + k)
            else:
                err = float((got.cpu() - T(ref)).abs().max()) / float(np.abs(ref).max())
                assert err < gtol, f"bf16-path grad {k}: max err {err:.3e} of the gradient scale"


def test_transpose_cast(dev):
    from diffmm_amd import ops
    x = torch.randn(1116, 6710, device=dev)
    dst = torch.full((6710, 1120), 7.0, dtype=torch.bfloat16, device=dev)
    ops.transpose_cast_bf16(x, dst)
    assert torch.equal(dst[:, :1116], x.t().bfloat16()) and float(dst[:, 1116:1120].abs().max()) == 0.0


def test_infonce_one_walk_kernels_match_the_two_kernel_path(dev):
    """dmm_infonce_fwd_bwd with DMM_INFONCE_FUSED=1 (forward + anchor-side gradient in one walk with a fixed exponent reference, then the
    other-view pass) against the default forward + two-pass backward, in a child process (the knob is read once per process)."""
    import subprocess, sys, os
    code = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from diffmm_amd import ops
torch.manual_seed(4)
dev = torch.device("cuda:0")
B, D, S = 300, 64, 4
src = [torch.randn(500, D, device=dev) for _ in range(S)]
idx = torch.randint(0, 500, (B,), device=dev, dtype=torch.int32)
slots = ops.Slots([(src[s], idx, torch.zeros_like(src[s])) for s in range(S)])
terms = ops.make_terms([(0, 1, 0.2, 0.5), (2, 3, 0.5, 0.01), (0, 3, 0.3, 1.0)])
zhat, znorm = torch.zeros(S, B, D, device=dev), torch.zeros(S, B, device=dev)
pack = ops.infonce_pack(S, B, D, dev)
ops.gather_normalize(slots, B, D, zhat, znorm, pack)
lse, rl, dz = torch.zeros(3, B, device=dev), torch.zeros(3, B, device=dev), torch.zeros(S, B, D, device=dev)
ops.infonce_fwd_bwd(zhat, terms, 3, B, D, lse, rl, dz, pack=pack, n_slots=S, precision=int(sys.argv[1]))
torch.save(dict(lse=lse.cpu(), rl=rl.cpu(), dz=dz.cpu()), sys.argv[2])
"""
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for prec, tol in ((1, 2e-2), (3, 2e-4)):
        outs = []
        with tempfile.TemporaryDirectory() as tmp:
            for fused in ("0", "1"):
                path = os.path.join(tmp, f"o{fused}.pt")
                r = subprocess.run([sys.executable, "-c", code, str(prec), path], cwd=repo, env=dict(os.environ, DMM_INFONCE_FUSED=fused), capture_output=True, text=True, timeout=300)
                assert r.returncode == 0, r.stderr[-2000:]
                outs.append(torch.load(path, weights_only=True))
        a, b = outs
        assert float((a["lse"] - b["lse"]).abs().max()) < 1e-4 and float((a["rl"] - b["rl"]).abs().max()) < 1e-6, f"precision {prec}: lse / row loss"
        scale = float(a["dz"].abs().max())
        assert float((a["dz"] - b["dz"]).abs().max()) < tol * scale, f"precision {prec}: gradients {float((a['dz'] - b['dz']).abs().max()) / scale:.2e} of the scale"


@pytest.mark.parametrize("D", [8, 64, 128])
def test_rownorm_bwd_batched_equals_the_separate_launches(dev, D):
    """dmm_rownorm_bwd_batched (three tensors of different row counts in one launch, column sums added to pre-set vectors) against
    dmm_rownorm_bwd + dmm_colsum."""
    from diffmm_amd import ops
    torch.manual_seed(D)
    probs, want = [], []
    for rows in (700, 33, 1):
        x = torch.randn(rows, D, device=dev)
        x[0] = 0.0  # the clamped branch of F.normalize
        y, nrm = ops.rownorm_fwd(x)
        dy = torch.randn(rows, D, device=dev)
        dx_ref = ops.rownorm_bwd(dy, y, nrm)
        cs_ref = torch.zeros(D, device=dev); ops.colsum(dx_ref, rows, D, cs_ref)
        dx, cs = torch.full((rows, D), 7.0, device=dev), torch.full((D,), 0.5, device=dev)
        probs.append(dict(dy=dy, y=y, norm=nrm, dx=dx, csum=cs)); want.append((dx_ref, cs_ref))
    ops.rownorm_bwd_batched(probs)
    for p_, (dx_ref, cs_ref) in zip(probs, want):
        assert torch.equal(p_["dx"], dx_ref), "dx"
        close(p_["csum"] - 0.5, cs_ref, rtol=1e-4, atol=1e-5 * float(cs_ref.abs().max()) + 1e-6, what="column sums")


def test_bf16_gather_copies_written_by_their_producers(dev):
    """dmm_gcn_mix_fwd_x16 / dmm_gcn_mix_bwd_x16 (the mixing kernels also write the bf16 copies of their outputs) equal the plain kernels followed
    by dmm_cast_bf16, bit for bit; dmm_cast_batch_rows_bf16 casts / clears exactly the rows users[b], U + pos[b], U + neg[b]."""
    from diffmm_amd import ops
    torch.manual_seed(5)
    N, D, M, U = 3000, 32, 3, 1800
    bx, z = torch.randn(N, D, device=dev), [torch.randn(N, D, device=dev) for _ in range(M)]
    w = torch.randn(M, device=dev)
    cast = lambda t: (lambda o: (ops.cast_bf16(t.view(-1, D), o.view(-1, D)), o)[1])(torch.zeros(t.shape, dtype=torch.bfloat16, device=dev))
    mx_a = ops.gcn_mix_fwd(bx, z, w, 0.2)
    mx16 = torch.zeros(N, D, dtype=torch.bfloat16, device=dev)
    mx_b = ops.gcn_mix_fwd(bx, z, w, 0.2, mx16=mx16)
    assert torch.equal(mx_a, mx_b) and torch.equal(mx16, cast(mx_a))
    dmx, extra = torch.randn(N, D, device=dev), torch.randn(N, D, device=dev)
    dz0 = [torch.randn(N, D, device=dev) for _ in range(M)]
    dz_a, dz_b = [t.clone() for t in dz0], [t.clone() for t in dz0]
    dbx_a, dw_a = ops.gcn_mix_bwd(dmx, bx, z, dz_a, extra, w, 0.2)
    dbx16, dz16 = torch.zeros(N, D, dtype=torch.bfloat16, device=dev), [torch.zeros(N, D, dtype=torch.bfloat16, device=dev) for _ in range(M)]
    dbx_b, dw_b = ops.gcn_mix_bwd(dmx, bx, z, dz_b, extra, w, 0.2, dbx16=dbx16, dz16=dz16)
    assert torch.equal(dbx_a, dbx_b) and torch.equal(dw_a, dw_b) and all(torch.equal(a, b) for a, b in zip(dz_a, dz_b))
    assert torch.equal(dbx16, cast(dbx_a)) and all(torch.equal(h, cast(f)) for h, f in zip(dz16, dz_a))
    # batch rows
    B = 257
    users = torch.randint(0, U, (B,), device=dev, dtype=torch.int32)
    pos, neg = torch.randint(0, N - U, (B,), device=dev, dtype=torch.int32), torch.randint(0, N - U, (B,), device=dev, dtype=torch.int32)
    src = [torch.randn(N, D, device=dev), torch.randn(N, D, device=dev)]
    dst = [torch.full((N, D), 7.0, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    ops.cast_batch_rows_bf16(src, dst, users, pos, neg, U)
    touched = torch.zeros(N, dtype=torch.bool, device=dev)
    touched[users.long()] = True; touched[U + pos.long()] = True; touched[U + neg.long()] = True
    for s_, d_ in zip(src, dst):
        assert torch.equal(d_[touched], s_[touched].bfloat16()) and bool((d_[~touched] == 7.0).all())
    ops.cast_batch_rows_bf16(None, dst, users, pos, neg, U)
    for d_ in dst:
        assert bool((d_[touched] == 0).all()) and bool((d_[~touched] == 7.0).all())
    # dmm_zero_batch_rows: fp32 and bf16 tensors in one launch
    mixed = [torch.full((N, D), 3.0, device=dev), torch.full((N, D), 3.0, dtype=torch.bfloat16, device=dev), torch.full((N, D), 3.0, device=dev)]
    ops.zero_batch_rows(mixed, users, pos, neg, U)
    for d_ in mixed:
        assert bool((d_[touched] == 0).all()) and bool((d_[~touched] == 3.0).all())
    # dz_overwrite: dz_m = maw w_m dmx whatever dz held
    junk = [torch.full((N, D), 9.0, device=dev) for _ in range(M)]
    ops.gcn_mix_bwd(dmx, bx, z, junk, extra, w, 0.2, dbx16=dbx16, dz16=dz16, dz_overwrite=True)
    zero = [torch.zeros(N, D, device=dev) for _ in range(M)]
    ops.gcn_mix_bwd(dmx, bx, z, zero, extra, w, 0.2)
    assert all(torch.equal(a, b) for a, b in zip(junk, zero)) and all(torch.equal(h, cast(f)) for h, f in zip(dz16, zero))


@pytest.mark.parametrize("B,H,M", [(300, 256, 3), (4096, 1024, 2), (65, 64, 1)])
def test_tanh_bwd_fused_equals_the_four_passes(dev, B, H, M):
    """dmm_tanh_bwd_fused (dpre = dh (1 - h^2) written only as the weight-gradient GEMMs' bf16 operands + the bias gradient, all modalities in one
    launch) against dmm_tanh_bwd + dmm_transpose_cast_bf16 + dmm_cast_bf16 + dmm_colsum_add: the bf16 values bit for bit, the column sums to fp32
    round-off (both are atomic sums)."""
    from diffmm_amd import ops
    torch.manual_seed(B + H)
    Bp = (B + 63) & ~63
    probs, want = [], []
    for m in range(M):
        dh, h = torch.randn(B, H + 8, device=dev)[:, :H], torch.tanh(torch.randn(B, H, device=dev))
        dpre = torch.empty(B, H, device=dev)
        ops.tanh_bwd(dh.contiguous(), h, dpre)
        d16, dT = torch.empty(B, H, device=dev, dtype=torch.bfloat16), torch.full((H, Bp), 7.0, device=dev, dtype=torch.bfloat16)
        ops.cast_bf16(dpre, d16)
        ops.transpose_cast_bf16(dpre, dT, rows=B, cols=H, rows_pad=Bp)
        cs = torch.full((H,), 0.5, device=dev)
        ops.colsum(dpre, B, H, cs, add=True)
        want.append((d16, dT, cs))
        probs.append(dict(dh=dh, h=h, d16=torch.empty(B, H, device=dev, dtype=torch.bfloat16), dT16=torch.full((H, Bp + 8), 7.0, device=dev, dtype=torch.bfloat16),
                          csum=torch.full((H,), 0.5, device=dev)))
    ops.tanh_bwd_fused(probs, B, H, Bp)
    for m in range(M):
        assert torch.equal(probs[m]["d16"], want[m][0]), "bf16 rows"
        assert torch.equal(probs[m]["dT16"][:, :Bp], want[m][1]) and float(probs[m]["dT16"][:, Bp:].float().min()) == 7.0, "bf16 transpose, zero pad, nothing beyond rows_pad"
        close(probs[m]["csum"], want[m][2], rtol=1e-5, atol=1e-4, what="column sums")


@pytest.mark.gpu
@pytest.mark.parametrize("D", [64, 128])
def test_item_project_bf16_x_is_bit_identical(dev, D):
    """dmm_item_project_batched_xb (X cast to bf16 once by the caller) against dmm_item_project_batched (X rounded inside the kernel): the same
    bf16 operand, the same K slicing -- both forms the joint step uses, three problems of unequal shapes as the modalities have them.  The forward
    form has at most two K slices (one commutative addition): bit-identical; the transposed form adds many slices with atomics: 1e-6 of scale."""
    from diffmm_amd import ops
    rng = np.random.default_rng(5)
    I, Fs = 700, [128, 64, 256]
    X = [T(rng.standard_normal((I, f)).astype(np.float32)).to(dev) for f in Fs]
    W = [T(rng.standard_normal((D, f)).astype(np.float32) * 0.1).to(dev) for f in Fs]
    bias = [T(rng.standard_normal(D).astype(np.float32)).to(dev) for _ in Fs]
    outs = []
    for cast in (False, True):
        out = [torch.zeros(I, D, device=dev) for _ in Fs]
        ops.item_project_batched([dict(X=(X[m].to(torch.bfloat16) if cast else X[m]), F=W[m], out=out[m], bias=bias[m], K=Fs[m]) for m in range(3)], D,
                                 f_transposed=True)
        outs.append(out)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    Ip = (I + 31) & ~31
    XT = [torch.zeros(f, Ip, device=dev) for f in Fs]
    for m in range(3):
        XT[m][:, :I] = X[m].t()
    dfeat = [T(rng.standard_normal((I, D)).astype(np.float32)).to(dev) for _ in Fs]
    outs = []
    for cast in (False, True):
        out = [torch.zeros(D, f, device=dev) for f in Fs]
        ops.item_project_batched([dict(X=(XT[m].to(torch.bfloat16) if cast else XT[m]), F=dfeat[m], out=out[m], K=I) for m in range(3)], D, out_transposed=True)
        outs.append(out)
    for a, b in zip(*outs):
        close(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-6 * float(b.abs().max()), what="transposed form")


@pytest.mark.gpu
def test_recorded_spmm_survives_a_rebuild_that_brings_giant_rows(dev):
    """A graph rebuilt in place (DeviceGraphBuilder(reuse=True)) keeps its buffers and its Adjacency object; a launch recorded while the graph had NO giant
    row must still be valid after a rebuild that has some -- engine.patch_spmm_sizes refreshes the sizes, the fix list's address has to be in the
    recorded problem already (Tensor.data_ptr() of the then-empty slice is null: the round-4 failure of test_tiktok_shape_epochs_with_bf16_gather_operands,
    where the first rebuilt graphs of a run had no row above 1024 entries and a later one did)."""
    from diffmm_amd import graph, ops, _lib as L
    from diffmm_amd.engine import plan_spmm_refs, patch_spmm_sizes
    rng = np.random.default_rng(3)
    users, n_items, D, task_len = 3000, 400, 32, 16
    deg = np.full(users, 2)
    ptr = np.concatenate([[0], np.cumsum(deg)])
    flat = lambda lists: T(np.concatenate(lists).astype(np.int32), dev)
    spread = [np.sort(rng.choice(np.arange(1, n_items), size=2, replace=False)) for _ in range(users)]           # no item above a few dozen users
    hub = [np.array([0, int(rng.integers(1, n_items))]) for _ in range(users)]                                    # item 0 chosen by everyone: a giant row
    gb = graph.DeviceGraphBuilder(T(ptr, dev), users, n_items, D, task_len, reuse=True)
    adj = gb.build_all([flat(spread)])[0]
    assert adj.persistent and adj.n_partial_blocks == 0
    x = T(rng.standard_normal((users + n_items, D)).astype(np.float32), dev)
    y = torch.zeros_like(x)
    with L.Recorder() as plan:
        ops.spmm_batched([ops.SpmmOp(adj, x, y=y)])
    refs = plan_spmm_refs(plan, [adj])
    L.replay(plan)
    dense = lambda a: torch.sparse_csr_tensor(a.rowptr.long(), a.col.long(), a.val, (a.n, a.n)).to_dense()
    close(y.cpu().numpy(), (dense(adj) @ x).cpu().numpy(), rtol=1e-5, atol=1e-5, what="before the rebuild")
    adj2 = gb.build_all([flat(hub)])[0]
    assert adj2 is adj and adj.n_partial_blocks > 0 and adj.fix.shape[0] > 0
    patch_spmm_sizes(refs)
    y.zero_()
    L.replay(plan)  # (raised "giant rows need partial + fix buffers" before the fix)
    close(y.cpu().numpy(), (dense(adj) @ x).cpu().numpy(), rtol=1e-5, atol=1e-5, what="after the rebuild")


@pytest.mark.gpu
def test_gather_normalize_hi_only_pack(dev):
    """dmm_gather_normalize_hi writes zhat, the norms and the hi / hiT blocks exactly as dmm_gather_normalize does and leaves the lo / loT blocks alone;
    the precision-1 InfoNCE launches on it give the results of the full pack bit for bit."""
    from diffmm_amd import ops
    torch.manual_seed(2)
    B, D, S = 300, 64, 4
    src = [torch.randn(500, D, device=dev) for _ in range(S)]
    idx = torch.randint(0, 500, (B,), device=dev, dtype=torch.int32)
    slots = ops.Slots([(src[s], idx, torch.zeros_like(src[s])) for s in range(S)])
    terms = ops.make_terms([(0, 1, 0.2, 0.5), (2, 3, 0.5, 0.01), (0, 3, 0.3, 1.0)])
    res = []
    for hi_only in (False, True):
        zhat, znorm = torch.zeros(S, B, D, device=dev), torch.zeros(S, B, device=dev)
        pack = ops.infonce_pack(S, B, D, dev)
        pack.fill_(7.0)
        ops.gather_normalize(slots, B, D, zhat, znorm, pack, hi_only=hi_only)
        lse, rl, dz = torch.zeros(3, B, device=dev), torch.zeros(3, B, device=dev), torch.zeros(S, B, D, device=dev)
        ops.infonce_fwd(zhat, terms, 3, B, D, lse, rl, pack=pack, n_slots=S, precision=1)
        res.append((zhat.clone(), znorm.clone(), pack.clone(), lse.clone(), rl.clone()))
    (z0, n0, p0, l0, r0), (z1, n1, p1, l1, r1) = res
    assert torch.equal(z0, z1) and torch.equal(n0, n1) and torch.equal(l0, l1) and torch.equal(r0, r1)
    q = p0.numel() // 4  # blocks in the pack: hi, lo, hiT, loT (make_pack)
    same = [bool(torch.equal(p0[i * q:(i + 1) * q], p1[i * q:(i + 1) * q])) for i in range(4)]
    untouched = [bool((p1[i * q:(i + 1) * q] == 7.0).all()) for i in range(4)]
    assert sum(same) == 2 and sum(untouched) == 2 and not any(a and b for a, b in zip(same, untouched)), (same, untouched)


@pytest.mark.gpu
def test_loss_reduce_equals_the_separate_reductions(dev):
    """dmm_loss_reduce (row-loss segment sums + the l2 term + the epoch's fp64 running sums in one launch) against dmm_segment_sum and dmm_sumsq: the same
    sums in the same order (bit-identical outputs), the accumulator advanced by exactly the outputs, the arrival counter left at zero -- twice in a row."""
    from diffmm_amd import ops, _lib as L
    torch.manual_seed(0)
    n_seg, B, n = 9, 4096, 16018 * 64 + 3
    rl = torch.randn(n_seg, B, device=dev)
    x = torch.randn(n + 1, device=dev)[:n]  # (n % 4 != 0: the tail path)
    base = torch.zeros(1, device=dev)
    want_seg, want_x = torch.zeros(n_seg, device=dev), torch.zeros(1, device=dev)
    ops.segment_sum(rl, n_seg, B, 0.5, want_seg)
    x16 = torch.zeros(n + 4, device=dev)  # a 16-byte aligned copy (dmm_sumsq / dmm_loss_reduce need it)
    x16[:n] = x
    xa = x16[:n]
    ops.sumsq(xa, 1e-4, want_x)
    scratch = torch.zeros(int(L.lib().dmm_sumsq_scratch(n)) + 1, device=dev)
    acc = torch.zeros(n_seg + 1, dtype=torch.float64, device=dev)
    for rep in range(2):
        got_seg, got_x = torch.zeros(n_seg, device=dev), torch.zeros(1, device=dev)
        ops.loss_reduce(rl, n_seg, B, 0.5, got_seg, xa, 1e-4, got_x, scratch, acc=acc, acc_x=n_seg)
        assert torch.equal(got_seg, want_seg) and torch.equal(got_x, want_x), rep
        assert int(scratch[-1:].view(torch.int32).item()) == 0, "arrival counter must be reset"
    want_acc = 2.0 * torch.cat([want_seg.double(), want_x.double()])
    assert torch.equal(acc, want_acc)


@pytest.mark.gpu
def test_zero_regions(dev):
    from diffmm_amd import ops
    a, b, c = torch.ones(1000, 64, device=dev), torch.ones(12, device=dev), torch.ones(4 * 77, dtype=torch.float64, device=dev)
    big = torch.ones(3 * 1024 * 1024 + 4, device=dev)
    ops.zero_regions([a, b[4:12], c, big[4:]])
    assert float(a.abs().sum()) == 0 and float(c.abs().sum()) == 0 and float(big[4:].abs().sum()) == 0
    assert torch.equal(b, torch.tensor([1.0] * 4 + [0.0] * 8, device=dev)) and float(big[:4].sum()) == 4.0
    with pytest.raises(ValueError):
        ops.zero_regions([b[1:5]])  # not 16-byte aligned


@pytest.mark.gpu
def test_csr_gather_sum_latent_batched_equals_single(dev):
    """The chain's CSR start for several operand tables in one launch == one launch per table (bit-identical z_T and h_{T-1})."""
    from diffmm_amd import ops
    rng = np.random.default_rng(4)
    U, I, H, Bn = 300, 500, 256, 120
    deg = rng.integers(0, 40, size=U); deg[5] = 400
    ptr = T(np.concatenate([[0], np.cumsum(deg)]).astype(np.int64), dev)
    col = T(np.concatenate([np.sort(rng.choice(I, size=d, replace=False)) for d in deg]).astype(np.int32), dev)
    rows = T(rng.choice(U, size=Bn, replace=False).astype(np.int64), dev)
    Wt = [T(rng.standard_normal((I, H)).astype(np.float32), dev).to(torch.bfloat16) for _ in range(3)]
    bias = [T(rng.standard_normal(H).astype(np.float32), dev) for _ in range(3)]
    z1 = [torch.zeros(Bn, H, device=dev) for _ in range(3)]; h1 = [torch.zeros(Bn, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    z2 = [torch.zeros(Bn, H, device=dev) for _ in range(3)]; h2 = [torch.zeros(Bn, H, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    for m in range(3):
        ops.csr_gather_sum_latent(ptr, col, rows, Wt[m], bias[m], z1[m], h1[m])
    ops.csr_gather_sum_latent_batched(ptr, col, rows, [dict(Wt=Wt[m], bias=bias[m], z=z2[m], out16=h2[m]) for m in range(3)])
    for m in range(3):
        assert torch.equal(z1[m], z2[m]) and torch.equal(h1[m], h2[m]), m


@pytest.mark.gpu
@pytest.mark.parametrize("B,D,prec", [(1000, 64, 1), (4160, 64, 1), (257, 128, 1), (300, 256, 1), (300, 256, 3), (97, 64, 1)])
def test_infonce_mfma_every_walk_form_against_the_fp32_kernels(dev, B, D, prec):
    """The round-4 forms of the backward walk against the fp32 VALU kernels: two own tiles per wave (d = 64, precision 1 -- with B = 1000 / 97 the wave's
    second tile is partly or wholly past the view, with B = 4160 the walk is cut into splits of unequal length), one tile pipelined (d = 128, 256) and
    the two-stage ring without the pipeline (d = 256, precision 3).  Precision 1 rounds the operands to bf16: lse within 2e-2, gradients within 3 % of
    their scale; precision 3 as tight as test_infonce_mfma_matches_valu_kernels."""
    from diffmm_amd import ops
    rng = np.random.default_rng(B + D)
    n = 3000
    views = [T(rng.standard_normal((n, D)).astype(np.float32)).to(dev) for _ in range(3)]
    idx = T(rng.integers(0, n, size=B).astype(np.int32)).to(dev)
    terms_l = [(0, 1, 0.2, 0.5), (0, 2, 0.5, 0.01), (1, 2, 0.5, 1.0)]
    res = {}
    for p in (0, prec):
        grads = [torch.zeros_like(v) for v in views]
        slots = ops.Slots([(v, idx, g) for v, g in zip(views, grads)])
        terms = ops.make_terms(terms_l)
        zhat, nrm, dz = torch.empty(3, B, D, device=dev), torch.empty(3, B, device=dev), torch.zeros(3, B, D, device=dev)
        lse, rl = torch.empty(3, B, device=dev), torch.empty(3, B, device=dev)
        pack = ops.infonce_pack(3, B, D, dev) if p else None
        kw = dict(pack=pack, n_slots=3, precision=p) if p else {}
        ops.gather_normalize(slots, B, D, zhat, nrm, pack, hi_only=(p == 1))
        if p:
            ops.infonce_fwd_bwd(zhat, terms, 3, B, D, lse, rl, dz, **kw)
        else:
            ops.infonce_fwd(zhat, terms, 3, B, D, lse, rl)
            ops.infonce_bwd(zhat, lse, terms, 3, B, D, dz)
        ops.scatter_normalize_bwd(slots, B, D, zhat, nrm, dz)
        res[p] = (lse.clone(), rl.clone(), [g.clone() for g in grads])
    tl, tg = (2e-2, 3e-2) if prec == 1 else (2e-5, 1e-3)
    close(res[prec][0], res[0][0], rtol=tl, atol=tl, what="lse")
    close(res[prec][1].sum(1), res[0][1].sum(1), rtol=max(tl, 5e-5), what="term losses")
    for a, b in zip(res[prec][2], res[0][2]):
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= tg * sc, f"gradients: {float((a - b).abs().max()) / sc:.3e} of scale"
