"""BASELINE.json configurations 3, 4 and 5 at their full shapes, live on the GPU (they used to be run by hand only):
  * config 5 -- 10M users x 1M items, 3 modalities, d = 256, T = 1000, batch 8192 on 8 GPUs: ONE replica of the 8 at the full shape on this GPU
    (collectives reduced to their local part, comm.LOCAL) through the code tools/scale_epoch.py --emulate-world 8 drives -- see the test;
  * config 3 -- Amazon-Baby shape (19445 x 7050, image + text, d = 64, T = 40, conf/baby.toml: sampling_step = 0 as the reference's
    Conf.py would run it, the modality-vs-modality contrastive branch cl_method = 1): two epochs;
  * config 4 -- Sports shape (35598 x 18357, image + text, d = 128, T = 200, batch 8192): one epoch on one GPU, and the same epoch
    user-sharded over two replicas (gloo on this GPU; the 8-GPU RCCL run is the driver's) -- the same minibatches and the same number
    of optimiser steps, so the epoch's losses must agree (the replicas draw their Gaussian / timestep noise from other streams).
Checked: finite losses of the expected magnitude that decrease, rebuilt modality graphs with exactly 2 E + N non-zeros (k_u = deg_u),
sane metrics.  (Value-level parity of these code paths is pinned at fixture size by the M = 2 / cl_method = 1 goldens: test_epoch_gpu.py.)"""
import json
import math
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))

from helpers.config_worker import run_config  # noqa: E402
from test_kernels_gpu import need_gpu  # noqa: E402


def _check(out, epochs):
    assert len(out["epochs"]) == epochs
    N = out["users"] + out["items"]
    for e in out["epochs"]:
        for k, v in {**e["train"], **e["test"]}.items():
            assert math.isfinite(v), (k, v)
        assert 0.3 < e["train"]["BPR Loss"] < 0.75 and e["train"]["CL loss"] > 0 and e["train"]["reg loss"] > 0
        assert 0.0 < e["test"]["Recall"] < 1.0 and 0.0 < e["test"]["NDCG"] < 1.0 and 0.0 < e["test"]["Precision"] < 1.0


_SPORTS = {}


def _single_sports():
    if "out" not in _SPORTS:  # (one single-GPU epoch serves both parametrisations)
        _SPORTS["out"] = run_config("sports", 128, 200, 8192, 1)
    return _SPORTS["out"]


def test_baby_shape_two_epochs():
    need_gpu()
    out = run_config("baby", 64, 40, 4096, 2)
    assert (out["users"], out["items"], out["M"]) == (19445, 7050, 2) and out["sampling_step"] == 0
    _check(out, 2)
    N = out["users"] + out["items"]
    for e in out["epochs"]:
        assert e["modal_nnz"] == [2 * out["E"] + N] * 2, "k_u = deg_u: every rebuilt graph has the base graph's non-zero count"
    a, b = out["epochs"][0], out["epochs"][1]
    assert b["train"]["Loss"] < a["train"]["Loss"] and b["test"]["Recall"] > 0.5 * a["test"]["Recall"]


@pytest.mark.parametrize("denoise_shard", ["rows", "items"])
def test_sports_shape_single_and_user_sharded(denoise_shard):
    """denoise_shard: how the replicas split the Denoise MLPs (Main.Coach._denoise_item_shard): "rows" -- replicated weights, own users' minibatch rows,
    gradient all-reduce; "items" -- every replica holds the W1 columns / W2 rows of its item slice and exchanges [B, H] partial activations."""
    need_gpu()
    single = _single_sports()
    assert (single["users"], single["items"], single["M"]) == (35598, 18357, 2)
    _check(single, 1)
    N = single["users"] + single["items"]
    assert single["epochs"][0]["modal_nnz"] == [2 * single["E"] + N] * 2
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "out.json")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DMM_DENOISE_SHARD=denoise_shard)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(29950 + os.getpid() % 40), os.path.join(REPO, "tests", "helpers", "config_worker.py"), "sports", "128", "200", "8192", "1", path]
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=1500)
        assert p.returncode == 0, p.stderr[-4000:]
        shard = json.load(open(path))
    assert shard["denoise_shard"] == denoise_shard
    _check(shard, 1)
    assert shard["modal_nnz_sum_over_replicas"] == [2 * single["E"] + N] * 2, "the replicas' blocks hold every entry of the rebuilt graphs exactly once"
    s, d = single["epochs"][0], shard["epochs"][0]
    for k, tol in (("Loss", 0.02), ("BPR Loss", 0.01), ("CL loss", 0.02), ("reg loss", 0.08)):
        # (the L2 term is almost all first-epoch embedding GROWTH, which follows the replicas' own Philox noise streams through the rebuilt graphs:
        # 1.5 % and 4 % apart in two measured runs; the loss terms that are means over the minibatches agree to about 1 %: the CL term -- whose
        # perturbation sign(y) * noise flips with the streams -- was 1.05 % apart in one run of many, hence 2 %)
        assert abs(s["train"][k] - d["train"][k]) <= tol * abs(s["train"][k]), f"{k}: single {s['train'][k]:.5f} vs user-sharded {d['train'][k]:.5f}"
    assert abs(s["test"]["Recall"] - d["test"]["Recall"]) <= 0.15 * s["test"]["Recall"] + 0.005


def test_tiktok_shape_epochs_with_bf16_gather_operands():
    """The whole-epoch plumbing of the bf16 gather operands (DataHandler -> Adjacency.vec = 8 with task length 64, DeviceGraphBuilder planning the rebuilt
    graphs the same way, JointEngine._forward_x16 and the backward's producer-made copies), forced on at TikTok shape with DMM_SPMM_BF16=1 in a child
    process (the knob is read when the graphs are planned): three epochs against the fp32-operand run in this process -- same minibatches, same Philox
    streams -- must agree like two arithmetic modes of one run: first-epoch losses within 1 % (measured 0.04-0.15 %, tools/bf16_operand_epochs.py);
    later epochs, whose rebuilt graphs differ by top-k near-ties: Loss 2 %, CL 3 %, reg 8 %, BPR 10 % (measured 0.2 / 0.8 / 2.9 / 4.1 %), Recall
    within 0.03 (measured 0.005)."""
    need_gpu()
    ref = run_config("tiktok", 64, 5, 1024, 3)
    code = ("import json, sys; sys.path.insert(0, %r); from helpers.config_worker import run_config; "
            "json.dump(run_config('tiktok', 64, 5, 1024, 3), open(sys.argv[1], 'w'))") % os.path.join(REPO, "tests")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "out.json")
        p = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, cwd=REPO, env=dict(os.environ, DMM_SPMM_BF16="1"), timeout=900)
        assert p.returncode == 0, p.stderr[-4000:]
        got = json.load(open(path))
    _check(got, 3)
    N = ref["users"] + ref["items"]
    for i, (a, b) in enumerate(zip(ref["epochs"], got["epochs"])):
        assert b["modal_nnz"] == [2 * ref["E"] + N] * 3
        # (first epoch: two RUNS of this configuration differ by more than the arithmetic modes do -- 30 single-GPU runs in round 4, fp32 operands: BPR 0.6595 .. 0.6635
        # in most, 0.650 .. 0.655 in one run of five (the rebuilt graphs after ten Denoise steps rank near-uniform scores: ties flip with the order of the float
        # atomics), reg 0.0333 .. 0.0338, CL 3.478 .. 3.493, total 4.173 .. 4.181 -- so BPR and reg are held to 2.5 % / 3 %, the others to 1 %)
        tols = {"Loss": 0.01, "BPR Loss": 0.025, "CL loss": 0.01, "reg loss": 0.03} if i == 0 else {"Loss": 0.02, "BPR Loss": 0.10, "CL loss": 0.03, "reg loss": 0.08}
        for k, tol in tols.items():
            assert abs(a["train"][k] - b["train"][k]) <= tol * abs(a["train"][k]), f"epoch {i} {k}: fp32 operands {a['train'][k]:.5f} vs bf16 {b['train'][k]:.5f}"
        assert abs(a["test"]["Recall"] - b["test"]["Recall"]) <= 0.03


def test_config5_one_replica_of_eight_at_full_shape():
    """BASELINE config 5 (10M users x 1M items, M = 3, d = 256, T = 1000, batch 8192, 8 GPUs): replica 0 of 8 at the FULL shape on this GPU -- its
    1.25M own users' graph rows, its local adjacency blocks over [own users ; 1M items] planned for bf16 gather operands, its 125k-item slice of the three
    Denoise MLPs (item-sharded), lean chain buffers -- with every collective reduced to its local part (comm.LOCAL; the exchanges themselves need 8 GPUs).
    A bounded sample of an epoch: 2 diffusion iterations, the T = 1000 reverse chain + per-user top-k over 1M columns + sharded graph build for 2048 own
    users (the others keep their observed interactions), 2 joint iterations, the evaluation of the own test users.  Asserted: finite losses of the expected
    magnitude, rebuilt blocks with exactly the base block's non-zero count (k_u = deg_u; 2 E_local + own users + this replica's item self loops), the
    rebuilt rows keep their degree with distinct items, clear chain error word, peak HBM under 200 GiB (measured 175), inside the time limit."""
    need_gpu()
    import torch
    torch.cuda.empty_cache()
    cmd = [sys.executable, os.path.join(REPO, "tools", "scale_epoch.py"), "--users", "10000000", "--items", "1000000", "--latdim", "256", "--T", "1000", "--batch", "8192",
           "--emulate-world", "8", "--diff-iters", "2", "--joint-iters", "2", "--rebuild-users", "2048"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=850)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print({k: out[k] for k in ("epoch_s", "phase_s", "test_s", "peak_alloc_GiB", "base_block_nnz")})
    assert out["emulated_replica"] == [0, 8] and out["denoise_shard"] == "items" and (out["users"], out["items"], out["d"], out["T"], out["batch"]) == (10_000_000, 1_000_000, 256, 1000, 8192)
    assert out["own_users"] == 1_250_000 and out["diff_iters_run"] == 2 and out["joint_iters_run"] == 2
    for k, v in {**out["train"], **out["test"]}.items():
        assert math.isfinite(v), (k, v)
    # (Loss terms are divided by the FULL epoch's step count, Main.py:379-383, of which 2 ran: only their sign / finiteness is meaningful here)
    assert out["train"]["BPR Loss"] > 0 and out["train"]["CL loss"] > 0 and out["train"]["reg loss"] > 0
    assert 0.0 < out["test"]["Recall"] < 1.0 and 0.0 < out["test"]["NDCG"] < 1.0
    want = 2 * out["E"] + out["own_users"] + 1_000_000 // 8  # item i's self loop lives on replica i % 8
    assert out["base_block_nnz"] == want and out["modal_block_nnz"] == [want] * 3
    assert out["rebuilt_rows_keep_degree"] and out["rebuilt_rows_distinct"] and out["chain_error_word_clear"]
    assert out["peak_alloc_GiB"] < 200.0


@pytest.mark.parametrize("denoise_shard", ["rows", "items"])
def test_tiktok_shape_epochs_with_users_and_items_sharded(denoise_shard):
    """DMM_JOINT_SHARD=2d (diffmm_amd/sharded2d.py) through whole epochs: two replicas (gloo on this GPU) against the single-GPU run of the same
    configuration -- the same minibatches and optimiser steps, other noise streams: first-epoch losses within 1-2 %, the replicas' rebuilt blocks
    hold every entry of the rebuilt graphs exactly once, sane metrics; both Denoise splits (the item-sharded Denoise reads the replica's own slice
    of the projected features, the row split an all-gather of them)."""
    need_gpu()
    single = run_config("tiktok", 64, 5, 1024, 2)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "out.json")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DMM_JOINT_SHARD="2d", DMM_DENOISE_SHARD=denoise_shard)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(29900 + os.getpid() % 40), os.path.join(REPO, "tests", "helpers", "config_worker.py"), "tiktok", "64", "5", "1024", "2", path]
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-4000:]
        shard = json.load(open(path))
    assert shard["joint_shard"] == "2d" and shard["denoise_shard"] == denoise_shard
    _check(shard, 2)
    N = single["users"] + single["items"]
    assert shard["modal_nnz_sum_over_replicas"] == [2 * single["E"] + N] * 3
    s, d = single["epochs"][0], shard["epochs"][0]
    # (BPR: 2.5 % -- the run-to-run spread of this configuration's first epoch, see test_tiktok_shape_epochs_with_bf16_gather_operands; at 1 % this test failed
    # in one run of three with a single-GPU value of 0.650 .. 0.652 against 0.660 on the replicas, and so did two single-GPU runs against each other)
    for k, tol in (("Loss", 0.02), ("BPR Loss", 0.025), ("CL loss", 0.02), ("reg loss", 0.08)):
        assert abs(s["train"][k] - d["train"][k]) <= tol * abs(s["train"][k]), f"{k}: single {s['train'][k]:.5f} vs 2D-sharded {d['train'][k]:.5f}"
    assert abs(single["epochs"][1]["test"]["Recall"] - shard["epochs"][1]["test"]["Recall"]) <= 0.03
