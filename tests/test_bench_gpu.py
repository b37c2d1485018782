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
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(29600 + os.getpid() % 300), os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", *SMALL, "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "user-shard2" and j["scaling"] == "strong"
    assert j["replicated_params_identical"] is True
