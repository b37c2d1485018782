"""Step-level parity of the HIP engines against the reference's golden vectors and the CPU oracle (pytest -m gpu)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")

from test_kernels_gpu import T, close, load, need_gpu  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    need_gpu()
    return torch.device("cuda:0")


HP = dict(modal_adj_weight=0.2, residual_weight=0.5, noise_degree=1.5, cross_cl_temp=0.2, cross_cl_rate=0.5, modal_cl_temp=0.5,
          modal_cl_rate=0.01, reg=1e-4, lr=1e-3, cl_method=0)


def _engine_from_gcn_fixture(z, dev, B_max=256, **over):
    from diffmm_amd import graph
    from diffmm_amd.engine import JointEngine
    U, I, M, D = int(z["users"]), int(z["items"]), int(z["M"]), int(z["d"])
    hp = dict(HP); hp.update(over)
    E = JointEngine(U, I, D, [T(z[f"feat{m}"]) for m in range(M)], hp, dev, B_max)
    E.load_reference_state({k[2:]: z[k] for k in z.files if k.startswith("p/")})
    adj = graph.make_adjacency(z["trn_rows"], z["trn_cols"], U, I, D, dev)
    madj = [graph.make_adjacency(z[f"m{m}_rows"], z[f"m{m}_cols"], U, I, D, dev) for m in range(M)]
    E.set_adjacency(adj, madj)
    return E


@pytest.mark.parametrize("fx", ["gcn_toy_M3", "gcn_mid_M3", "gcn_mid_M2"])
def test_gcn_mm_forward_vs_reference(dev, fx):
    z = load(fx + ".npz")
    E = _engine_from_gcn_fixture(z, dev)
    E.forward(with_cl=False, precision=3)
    out = E.gcn_output()
    for k in [f[4:] for f in z.files if f.startswith("out/")]:
        close(out[k], z[f"out/{k}"], rtol=1e-4, atol=1e-5, what=k)


@pytest.mark.parametrize("fx,cl_method", [("gcn_mid_M3", 0), ("gcn_mid_M3", 1), ("gcn_mid_M2", 0), ("gcn_mid_M2", 1), ("gcn_toy_M3", 0)])
def test_joint_step_loss_grads_and_update_vs_oracle(dev, fx, cl_method):
    """One full joint step: losses, the gradient of every parameter (vs autograd on the oracle) and the Adam update."""
    from oracle import diffmm_oracle as O
    z = load(fx + ".npz")
    U, I, M, D = int(z["users"]), int(z["items"]), int(z["M"]), int(z["d"])
    E = _engine_from_gcn_fixture(z, dev, cl_method=cl_method)
    rng = np.random.default_rng(7)
    B = 6 if U < 50 else 200
    e = rng.integers(0, len(z["trn_rows"]), size=B)
    users, pos = z["trn_rows"][e].astype(np.int32), z["trn_cols"][e].astype(np.int32)
    neg = rng.integers(0, I, size=B).astype(np.int32)
    noise = [T(rng.random((U + I, D)).astype(np.float32)) for _ in range(3)]
    # oracle
    P = {k[2:]: T(z[k]).clone().requires_grad_() for k in z.files if k.startswith("p/")}
    adj = O.csr_to_torch(*O.build_adj_csr(z["trn_rows"], z["trn_cols"], U, I))
    madj = [O.csr_to_torch(*O.build_adj_csr(z[f"m{m}_rows"], z[f"m{m}_cols"], U, I)) for m in range(M)]
    hp = dict(HP); hp["cl_method"] = cl_method
    loss, parts = O.joint_loss(P, [T(z[f"feat{m}"]) for m in range(M)], adj, madj, T(users.astype(np.int64)), T(pos.astype(np.int64)),
                               T(neg.astype(np.int64)), noise, hp)
    loss.backward()
    # HIP
    E.step(T(users).to(dev), T(pos).to(dev), T(neg).to(dev), noise=[n.to(dev) for n in noise], precision=3, update=False)
    got = E.losses()
    close(got["bpr"], float(parts["bpr"]), rtol=2e-5, what="bpr"); close(got["reg"], float(parts["reg"]), rtol=2e-5, what="reg")
    close(got["cl"], float(parts["cl"]), rtol=5e-5, what="cl"); close(got["loss"], float(loss), rtol=5e-5, what="loss")
    g = E.full_grads()
    for k in P:
        ref = P[k].grad
        close(g[k], ref, rtol=2e-3, atol=2e-4 * float(ref.abs().max()) + 1e-9, what="grad " + k)
    # Adam update
    opt = O.Adam({k: v for k, v in P.items()}, 1e-3)
    opt.step({k: v.grad for k, v in P.items()})
    E.apply_adam()
    sd = E.state_dict()
    for k in P:
        # first Adam step moves every weight by lr * g / (|g| + 1e-8): where |g| is far above eps both sides move by ~lr exactly;
        # where g ~ eps the step is ill-conditioned in g and only bounded by lr
        big = (P[k].grad.abs() > 1e-5).numpy()
        a, b = sd[k].cpu().numpy(), P[k].detach().numpy()
        close(a[big], b[big], rtol=1e-4, atol=2e-5, what="after Adam " + k)
        assert np.abs(a - b).max() <= 2.1e-3, "after Adam (tiny-gradient entries) " + k


@pytest.mark.parametrize("fx,cl_method,nd", [("gcn_mid_M3", 0, 0.0), ("gcn_mid_M3", 0, 1.5), ("gcn_mid_M2", 1, 0.0)])
def test_joint_step_with_bf16_gather_operands(dev, fx, cl_method, nd):
    """The joint step on graphs planned for bf16 gather operands (Adjacency.vec = 8: what graph.spmm_operand_vec picks once the operand
    outgrows the caches) against the fp32-operand step: same launches plus casts, fp32 accumulation and epilogues.  Tolerances are those of a
    bf16-stored operand (2^-9 relative per element, averaged down by the row sums): losses 2e-3, gradients 2 % of each tensor's scale in the
    worst element and 0.5 % in the l2 norm -- with the CL layers' perturbation switched off (noise_degree = 0): sign(y) * noise is discontinuous
    where y crosses zero, so with it on the few elements whose sign the operand rounding flips move by a whole perturbation (measured: 0.1-0.2 %
    of the elements, 3-5 % of the embedding gradients' l2 norm) and only loose bounds hold."""
    from diffmm_amd import graph
    z = load(fx + ".npz")
    U, I, M, D = int(z["users"]), int(z["items"]), int(z["M"]), int(z["d"])
    if D < 16:
        pytest.skip("bf16 operands need d >= 16")
    rng = np.random.default_rng(11)
    B = 200
    e = rng.integers(0, len(z["trn_rows"]), size=B)
    users, pos = T(z["trn_rows"][e].astype(np.int32)).to(dev), T(z["trn_cols"][e].astype(np.int32)).to(dev)
    neg = T(rng.integers(0, I, size=B).astype(np.int32)).to(dev)
    noise = [T(rng.random((U + I, D)).astype(np.float32)).to(dev) for _ in range(3)]
    res = []
    for vec in (4, 8):
        E = _engine_from_gcn_fixture(z, dev, cl_method=cl_method, noise_degree=nd)
        adj = graph.make_adjacency(z["trn_rows"], z["trn_cols"], U, I, D, dev, vec=vec)
        madj = [graph.make_adjacency(z[f"m{m}_rows"], z[f"m{m}_cols"], U, I, D, dev, vec=vec) for m in range(M)]
        E.set_adjacency(adj, madj)
        # a first step on OTHER rows: the bf16-operand step clears its scatter targets row by row, so whatever it leaves behind would show below
        E.step(users.flip(0).roll(7) % U, (pos + 3) % I, (neg + 5) % I, noise=noise, precision=3, update=False)
        E.step(users, pos, neg, noise=noise, precision=3, update=False)
        res.append((E.losses(), {k: v.clone() for k, v in E.full_grads().items()}, {k: v.clone() for k, v in E.gcn_output().items()}))
        assert (E._x16() is not None) == (vec == 8)
        if vec == 8:
            assert float(E.gblock[:3].abs().max()) == 0.0 and float(E._x16()["g"][:2].float().abs().max()) == 0.0, "scatter targets left clean"
    (l4, g4, o4), (l8, g8, o8) = res
    for k in ("bpr", "reg", "cl", "loss"):
        close(l8[k], l4[k], rtol=2e-3, what=k)
    for k in o4:
        close(o8[k], o4[k], rtol=0, atol=8e-3 * float(o4[k].abs().max()), what="forward " + k)
    report = {k: (float((g8[k] - g4[k]).abs().max()) / (float(g4[k].abs().max()) + 1e-30), float((g8[k] - g4[k]).norm()) / (float(g4[k].norm()) + 1e-30)) for k in g4}
    tol = (2e-2, 5e-3) if nd == 0.0 else (0.3, 8e-2)
    for k, (worst, l2) in report.items():
        assert worst <= tol[0] and l2 <= tol[1], f"grad {k}: worst element / scale, l2 error / norm: {report}"
    # mixed operand types in one step are refused
    with pytest.raises(ValueError):
        E.set_adjacency(adj, [graph.make_adjacency(z["m0_rows"], z["m0_cols"], U, I, D, dev, vec=4)] * M)
