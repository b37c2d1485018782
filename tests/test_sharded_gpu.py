"""User-sharded joint engine (diffmm_amd/sharded.py) against the single-GPU engine on the same seeded problem:
  * world = 1 in this process: the raw-row / epilogue split, the batch-table exchange path and the item-slice bookkeeping reproduce
    JointEngine's loss and every gradient;
  * world = 2 and 3 as real processes over gloo (sharing this GPU): every replica's loss equals the single-GPU loss, its gradients
    equal the single-GPU gradients (own user rows; replicated parameters in full) within 1e-5 of the gradient scale, replicated
    parameters stay BIT-identical across the replicas after optimiser steps, and the in-kernel Philox draws (keyed by global row ids)
    give the same trajectory as the single-GPU engine's."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))

from helpers import sharded_worker as W  # noqa: E402
from test_kernels_gpu import need_gpu  # noqa: E402


def _single(fx, cl_method, dev, vec=4):
    """The same sequence of steps as sharded_worker.run_steps on the single-GPU JointEngine."""
    from diffmm_amd import graph
    from diffmm_amd.engine import JointEngine
    z, U, I, M, D, users, pos, neg, noise, hp = W.problem(fx, cl_method)
    E = JointEngine(U, I, D, [torch.from_numpy(z[f"feat{m}"]) for m in range(M)], hp, dev, 256)
    E.load_reference_state({k[2:]: z[k] for k in z.files if k.startswith("p/")})
    E.set_adjacency(graph.make_adjacency(z["trn_rows"], z["trn_cols"], U, I, D, dev, vec=vec),
                    [graph.make_adjacency(z[f"m{m}_rows"], z[f"m{m}_cols"], U, I, D, dev, vec=vec) for m in range(M)])
    t = lambda a: torch.from_numpy(a).to(dev)
    nz = [t(n) for n in noise]
    E.step(t(users), t(pos), t(neg), noise=nz, precision=3, update=False)
    out = {"loss/" + k: v for k, v in E.losses().items()}
    for k, v in E.full_grads().items():
        out["grad/" + k] = v.cpu().numpy()
    E.rng_step = 0
    E.step(t(users), t(pos), t(neg), noise=nz, precision=3, update=True)
    E.step(t(users), t(pos), t(neg), noise=None, precision=3, update=True)
    E.step(t(users[::-1].copy()), t(pos[::-1].copy()), t(neg[::-1].copy()), noise=None, precision=3, update=True)
    for k, v in E.state_dict().items():
        out["param/" + k] = v.cpu().numpy()
    return out, U


def _compare(ref, got, u0, u1, what, vec=4, i0=None, i1=None):
    # vec = 8 (bf16 gather operands on both sides): the engines' fp32 results differ in the last bits (summation order), which now and then moves
    # the bf16 rounding of an operand element by one ulp (2^-9): losses 1e-4, gradients 2e-3 of their scale, a larger share of Adam-step outliers
    tl, tg, frac = (1e-5, 1e-5, 0.01) if vec == 4 else (1e-4, 2e-3, 0.08)
    for k in ("loss", "bpr", "reg", "cl"):
        a, b = float(got["loss/" + k]), float(ref["loss/" + k])
        assert abs(a - b) <= tl * max(1.0, abs(b)), f"{what}: {k} {a} vs {b}"
    for k in [k for k in ref if k.startswith("grad/") or k.startswith("param/")]:
        r = ref[k][u0:u1] if k.endswith("/u_embs") else (ref[k][i0:i1] if (k.endswith("/i_embs") and i0 is not None) else ref[k])
        g = got[k]
        assert g.shape == r.shape, f"{what}: {k} shape {g.shape} vs {r.shape}"
        scale = float(np.abs(ref[k]).max()) + 1e-30
        d = np.abs(g - r)
        err = float(d.max()) if g.size else 0.0
        if k.startswith("grad/"):
            assert err <= tg * scale, f"{what}: {k} max |diff| {err:.3e} > {tg} x scale {scale:.3e}"
        else:
            # three Adam steps of lr = 1e-3: a step is lr * m / (sqrt(v) + 1e-8), ill-conditioned where |g| ~ 1e-8 (bounded by lr per step),
            # so: all but a sliver of the entries agree to 5e-5, none is off by more than the three steps themselves
            assert err <= 6.5e-3 and float((d > 5e-5).mean()) <= frac, f"{what}: {k} max |diff| {err:.3e}, {float((d > 5e-5).mean()):.4f} of entries off"


@pytest.mark.parametrize("fx,cl_method,vec", [("gcn_mid_M3", 0, 4), ("gcn_mid_M2", 1, 4), ("gcn_toy_M3", 0, 4), ("gcn_mid_M3", 0, 8)])
def test_world1_sharded_engine_equals_joint_engine(fx, cl_method, vec):
    """vec = 8: both engines on graphs planned for bf16 gather operands -- the same bf16 copies on both sides, so the comparison keeps its fp32 tolerance."""
    need_gpu()
    dev = torch.device("cuda:0")
    ref, U = _single(fx, cl_method, dev, vec)
    z, U, I, M, D, users, pos, neg, noise, hp = W.problem(fx, cl_method)
    lay, E = W.build_sharded(z, U, I, M, D, hp, dev, 0, 1, vec=vec)
    assert (E._x16() is not None) == (vec == 8)
    got = W.run_steps(E, lay, users, pos, neg, noise, dev)
    _compare(ref, got, 0, U, f"{fx} world=1 vec={vec}", vec)


@pytest.mark.parametrize("fx,cl_method,world,vec", [("gcn_mid_M3", 0, 2, 4), ("gcn_mid_M2", 1, 3, 4), ("gcn_mid_M3", 0, 2, 8)])
def test_replicas_over_gloo_equal_joint_engine(fx, cl_method, world, vec):
    need_gpu()
    dev = torch.device("cuda:0")
    ref, U = _single(fx, cl_method, dev, vec)
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DMM_TEST_SPMM_VEC=str(vec))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + os.getpid() % 250), os.path.join(REPO, "tests", "helpers", "sharded_worker.py"), fx, str(cl_method), tmp]
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-4000:]
        outs = [dict(np.load(os.path.join(tmp, f"rank{r}.npz"))) for r in range(world)]
    covered = 0
    for r, got in enumerate(outs):
        u0, u1 = int(got["u0"]), int(got["u1"])
        covered += u1 - u0
        _compare(ref, got, u0, u1, f"{fx} rank {r}/{world}", vec)
    assert covered == U, "the replicas' user ranges partition the users"
    for k in [k for k in outs[0] if k.startswith("param/") and not k.endswith("/u_embs")]:
        for r in range(1, world):
            assert np.array_equal(outs[0][k], outs[r][k]), f"replicated parameter {k} differs between replica 0 and {r}"
    for r in range(1, world):
        assert np.array_equal(outs[0]["loss3"][:-2], outs[r]["loss3"][:-2]) or np.allclose(outs[0]["loss3"][:-2], outs[r]["loss3"][:-2], rtol=1e-6)


# ---- users AND items sharded (diffmm_amd/sharded2d.py): reduce-scatter of the item partials, epilogue on the own slice, all-gather of the gather operands ----
@pytest.mark.parametrize("fx,cl_method,vec", [("gcn_mid_M3", 0, 4), ("gcn_mid_M2", 1, 4), ("gcn_toy_M3", 0, 4), ("gcn_mid_M3", 0, 8)])
def test_world1_2d_engine_equals_joint_engine(fx, cl_method, vec):
    need_gpu()
    dev = torch.device("cuda:0")
    ref, U = _single(fx, cl_method, dev, vec)
    z, U, I, M, D, users, pos, neg, noise, hp = W.problem(fx, cl_method)
    lay, E = W.build_sharded(z, U, I, M, D, hp, dev, 0, 1, vec=vec, two_d=True)
    got = W.run_steps(E, lay, users, pos, neg, noise, dev)
    _compare(ref, got, 0, U, f"{fx} 2D world=1 vec={vec}", vec, 0, I)


@pytest.mark.parametrize("fx,cl_method,world,vec", [("gcn_mid_M3", 0, 2, 4), ("gcn_mid_M2", 1, 3, 4), ("gcn_mid_M3", 0, 2, 8)])
def test_2d_replicas_over_gloo_equal_joint_engine(fx, cl_method, world, vec):
    """Every replica's loss equals the single-GPU loss; its gradients and its parameters after optimiser steps equal the single-GPU engine's
    restricted to the rows it owns (own users, own item slice) and in full for the replicated projections / modal_weight, within 1e-5 of the
    gradient scale (fp32 gather operands; the bf16-operand tolerance otherwise); the replicated tensors stay bit-identical across the replicas; the
    user ranges and the item slices partition the users and the items."""
    need_gpu()
    dev = torch.device("cuda:0")
    ref, U = _single(fx, cl_method, dev, vec)
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DMM_TEST_SPMM_VEC=str(vec), DMM_TEST_JOINT_SHARD="2d")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(29400 + os.getpid() % 250), os.path.join(REPO, "tests", "helpers", "sharded_worker.py"), fx, str(cl_method), tmp]
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-4000:]
        outs = [dict(np.load(os.path.join(tmp, f"rank{r}.npz"))) for r in range(world)]
    cu = ci = 0
    I = ref["param/i_embs"].shape[0]
    for r, got in enumerate(outs):
        u0, u1, i0, i1 = int(got["u0"]), int(got["u1"]), int(got["i0"]), int(got["i1"])
        cu, ci = cu + u1 - u0, ci + i1 - i0
        _compare(ref, got, u0, u1, f"{fx} 2D rank {r}/{world}", vec, i0, i1)
    assert cu == U and ci == I, "the user ranges and the item slices partition the users and the items"
    for k in [k for k in outs[0] if k.startswith("param/") and not k.endswith("/u_embs") and not k.endswith("/i_embs")]:
        for r in range(1, world):
            assert np.array_equal(outs[0][k], outs[r][k]), f"replicated parameter {k} differs between replica 0 and {r}"
