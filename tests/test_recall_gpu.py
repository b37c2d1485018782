"""End-to-end accuracy, live on the GPU (north_star: "Recall@20 within +-0.002 of reference"): the HIP path trains 30 epochs at the bench
configuration (synthetic TikTok-shape graph, d = 64, T = 40, batch 4096, bf16 GEMMs -- exactly what bench.py times) for three seeds and
its plateau (epochs 10-29) Recall@20 / NDCG@20 / losses are compared with the REFERENCE's own runs for the same three seeds on the same
data (baselines/ref_cpu_synth_tiktok_T40_B4096_seed*.csv, written by tests/golden/make_curve.py importing the reference).

RNG streams cannot be shared between torch-CPU and the in-kernel Philox draws, so a run is one sample of the training noise.  The bar is
therefore stated against the reference's measured seed-to-seed spread:
  * mean over the three seeds:  |HIP - reference| <= 0.002 in Recall@20 (the north_star tolerance) and 0.002 in NDCG@20;
  * every single seed: |HIP - reference(same seed)| <= the reference's own spread between seeds (max - min of its three plateau means),
    floored at 0.002 -- a single run cannot be pinned tighter than the reference pins itself;
  * every loss term within 1 % of the reference's plateau mean, seed by seed.
The user-sharded path (two replicas over gloo on this GPU) is held to the same per-seed bar for one seed: the same algorithm, the same
number of optimiser steps."""
import csv
import os
import statistics as st
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
SEEDS = (1818, 7, 42)
EPOCHS, LO, HI = 30, 10, 30

from helpers.curve_worker import train_curve  # noqa: E402
from test_kernels_gpu import need_gpu  # noqa: E402


def _ref(seed):
    return list(csv.DictReader(open(os.path.join(REPO, "baselines", f"ref_cpu_synth_tiktok_T40_B4096_seed{seed}.csv"))))


def _plateau(rows, key):
    return st.mean(float(r[key]) for r in rows[LO:HI])


def _spread(key):
    m = [_plateau(_ref(s), key) for s in SEEDS]
    return max(m) - min(m), m


@pytest.fixture(scope="module")
def hip_runs():
    need_gpu()
    return {s: train_curve(s, EPOCHS) for s in SEEDS}


def test_plateau_recall_and_ndcg_match_the_reference_over_three_seeds(hip_runs):
    for key, tol in (("Recall", 0.002), ("NDCG", 0.002)):
        spread, ref_means = _spread(key)
        hip_means = [_plateau(hip_runs[s], key) for s in SEEDS]
        d_mean = st.mean(hip_means) - st.mean(ref_means)
        print(f"{key}@20 plateau: reference {['%.4f' % x for x in ref_means]} (spread {spread:.4f}), HIP {['%.4f' % x for x in hip_means]}, mean diff {d_mean:+.4f}")
        assert abs(d_mean) <= tol, f"{key}: mean over seeds differs by {d_mean:+.4f}"
        for s, h, r in zip(SEEDS, hip_means, ref_means):
            assert abs(h - r) <= max(tol, spread), f"{key} seed {s}: HIP {h:.4f} vs reference {r:.4f} (reference's own seed spread {spread:.4f})"


def test_plateau_losses_match_the_reference(hip_runs):
    for s in SEEDS:
        ref = _ref(s)
        assert len(ref) == EPOCHS
        for key in ("Loss", "BPR Loss", "CL loss", "reg loss"):
            r, h = _plateau(ref, key), _plateau(hip_runs[s], key)
            assert abs(h - r) <= 0.01 * abs(r), f"seed {s} {key}: reference {r:.5f} vs HIP {h:.5f}"
        rows = hip_runs[s]
        assert rows[-1]["Loss"] < rows[5]["Loss"] < rows[0]["Loss"], "the loss keeps decreasing like the reference's"


def test_user_sharded_replicas_reach_the_same_plateau():
    need_gpu()
    seed = SEEDS[0]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "curve.csv")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(29850 + os.getpid() % 100), os.path.join(REPO, "tests", "helpers", "curve_worker.py"), str(seed), str(EPOCHS), out]
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=1500)
        assert p.returncode == 0, p.stderr[-4000:]
        rows = list(csv.DictReader(open(out)))
    ref = _ref(seed)
    for key, tol in (("Recall", 0.002), ("NDCG", 0.002)):
        spread, _ = _spread(key)
        h, r = _plateau(rows, key), _plateau(ref, key)
        print(f"user-shard2 {key}@20 plateau {h:.4f} vs reference {r:.4f}")
        assert abs(h - r) <= max(tol, spread), f"user-sharded {key}: {h:.4f} vs reference {r:.4f}"
    for key in ("Loss", "BPR Loss", "CL loss", "reg loss"):
        r, h = _plateau(ref, key), _plateau(rows, key)
        assert abs(h - r) <= 0.01 * abs(r), f"user-sharded {key}: reference {r:.5f} vs HIP {h:.5f}"
