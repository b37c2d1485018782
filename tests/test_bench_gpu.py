"""bench.py end to end on a small synthetic graph: the JSON contract of the driver, and the N = 2 path (two user-sharded replicas sharing
this GPU over gloo: SpMM item partials all-reduced, minibatch rows split by owner, own-user rebuild) ending with bit-identical replicated
parameters."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--workload", "mid", "--batch", "64", "--T", "5", "--latdim", "64", "--steps", "2", "--warmup", "1"]


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}:\\n{out[-2000:]}"
    return json.loads(lines[0])


def test_bench_line_contract():
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *SMALL, "--no-cpu-baseline"], capture_output=True, text=True, cwd=REPO, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["value"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_two_replicas_on_one_gpu_stay_identical():
    """`python bench.py --gpus 2` the way the driver starts it -- NO torchrun around it: the parent launches the two ranks itself (before any GPU
    call), relays rank 0's JSON line and returns the children's status."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", *SMALL, "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "user-shard2" and j["scaling"] == "strong"
    assert j["replicated_params_identical"] is True


def test_self_launch_refuses_more_rccl_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n + 1), *SMALL], capture_output=True, text=True, cwd=REPO, env=env, timeout=300)
    assert p.returncode == 2 and "needs" in p.stderr and not p.stdout.strip(), "a clear refusal instead of an RCCL hang"
    env["WORLD_SIZE"] = "3"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", *SMALL], capture_output=True, text=True, cwd=REPO, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


def test_two_replicas_with_the_items_sharded_as_well():
    """DMM_JOINT_SHARD=2d (diffmm_amd/sharded2d.py): the joint step's item partials are reduce-scattered, the gather operands all-gathered, i_embs and the
    item-side passes split over the replicas -- whole epochs over two replicas (gloo: all-reduce + slice stands in for the reduce-scatter)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", DMM_JOINT_SHARD="2d")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", *SMALL, "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "user-shard2+item-shard" and j["replicated_params_identical"] is True and j["value"] > 0


@pytest.mark.parametrize("denoise_shard", ["rows", "items", "items+2d"])
def test_every_collective_of_an_epoch_runs_through_rccl(denoise_shard):
    """A one-GPU box cannot hold two RCCL ranks, but it can hold ONE: under DMM_FORCE_COLLECTIVES=1 a single replica runs the user-sharded engines and
    sends every collective of the epoch (item partials on the side stream under the next SpMM launch, batch table, gradient tails, Denoise partial
    activations / weight gradients, item counts of the rebuild, replica checksums in int64 MIN / MAX, metric sums) through an nccl process group of
    size 1: RCCL's communicator, stream ordering, dtype and contiguity rules all see the calls the 8-GPU job makes."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    two_d = denoise_shard.endswith("+2d")  # ... and the 2D joint step's reduce_scatter_tensor / all_gather_into_tensor
    denoise_shard = denoise_shard.split("+")[0]
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", DMM_FORCE_COLLECTIVES="1", DMM_DENOISE_SHARD=denoise_shard, DMM_JOINT_SHARD="2d" if two_d else "1d")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--backend", "nccl", *SMALL, "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p.stdout)
    assert j["config"]["parallelism"] == "user-shard1" + ("+item-shard" if two_d else "") and j["config"]["backend"] == "nccl" and j["config"]["denoise_shard"] == denoise_shard
    assert j["replicated_params_identical"] is True and j["value"] > 0


def test_scale_workload_runs_a_bounded_sample():
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", "scale", "--users", "20000", "--items", "3000", "--latdim", "64", "--T", "6", "--batch", "1024",
           "--diff-iters", "2", "--joint-iters", "3", "--rebuild-users", "700", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p.stdout)
    assert "2 diffusion iters" in j["config"]["workload"] and "3 joint iters" in j["config"]["workload"] and "700 users" in j["config"]["workload"]
    assert j["scaling"] == "strong" and abs(j["value"] - 5 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    # the weak form over two replicas (gloo, sharing this GPU): users are per GPU, value counts shard-iterations
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd + ["--gpus", "2", "--backend", "gloo", "--scale-mode", "weak"], capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p.stdout)
    assert j["scaling"] == "weak" and "U=40000" in j["config"]["workload"] and j["replicated_params_identical"] is True
    assert abs(j["value"] - 2 * 5 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
