#!/usr/bin/env python3
"""Three TikTok-shape epochs with fp32 and (child process, DMM_SPMM_BF16=1) bf16 gather operands: relative differences of the epoch losses and Recall
(the margins behind tests/test_configs_gpu.py::test_tiktok_shape_epochs_with_bf16_gather_operands)."""
import json, os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
if len(sys.argv) > 1:
    from helpers.config_worker import run_config
    json.dump(run_config("tiktok", 64, 5, 1024, 3), open(sys.argv[1], "w")); sys.exit(0)
from helpers.config_worker import run_config
ref = run_config("tiktok", 64, 5, 1024, 3)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "o.json")
    subprocess.run([sys.executable, os.path.abspath(__file__), path], check=True, env=dict(os.environ, DMM_SPMM_BF16="1"))
    got = json.load(open(path))
for i, (a, b) in enumerate(zip(ref["epochs"], got["epochs"])):
    print(i, {k: round(abs(a["train"][k] - b["train"][k]) / abs(a["train"][k]), 5) for k in ("Loss", "BPR Loss", "CL loss", "reg loss")},
          "Recall", round(a["test"]["Recall"], 4), round(b["test"]["Recall"], 4))
