#!/usr/bin/env python3
"""Summarise the separate rocprofv3 --pmc passes of tools/roofline_target.py (tools/final_profile.sh) into profiles/<name>.json: per
epilogue kind of the 256 x 256 GEMM the fabric-side bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
coalesced reads on gfx950; WRITE_SIZE exact; both reported in KiB), the L2 hit rate and the MFMA-pipe busy share."""
import collections, csv, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
NAMES = {"1": "layer1_tanh_bf16", "3": "layer2_bf16_state", "2": "layer2_fp32_state", "9": "latent_step", "6": "last_step_fp32_out"}
ALG = {  # operands read once + outputs written once (+ residual read once), bytes per batched launch of 3 modalities
    "layer1_tanh_bf16": 3 * (9308 * 6720 * 2 + 1024 * 6720 * 2 + 9308 * 1024 * 2),
    "layer2_bf16_state": 3 * (9308 * 1024 * 2 + 6710 * 1024 * 2 + 2 * 9308 * 6710 * 2),
    "layer2_fp32_state": 3 * (9308 * 1024 * 2 + 6710 * 1024 * 2 + 9308 * 6710 * (4 + 4 + 2)),
    "latent_step": 3 * (9308 * 1024 * (4 + 4 + 2 + 2) + 1024 * 1024 * 2),
    # per LAUNCH of the persistent chain (two per call: 64 + 47 row-block groups; the counters are per launch, so is this figure: half a call):
    # z_T and G read once, and per step every h element written once and read once (the exchange between the 4 workgroups of a row block)
    "latent_chain": (3 * 9308 * 1024 * 4 + 39 * 3 * 9308 * 1024 * (2 + 2) + 3 * 1024 * 1024 * 2 + 3 * 9308 * 1024 * 2) // 2,
    "last_step_fp32_out": 3 * (9308 * 1024 * 2 + 6710 * 1024 * 2 + 9308 * 6710 * 4),
}


def kind(name):
    if "latent_chain_8ph_kernel" in name:
        return "latent_chain"
    if "gemm_bf16_nt_8ph_kernel<" not in name:
        return None
    return NAMES.get(name.split("gemm_bf16_nt_8ph_kernel<", 1)[1].split(">")[0].strip())


def means(path):
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        k = kind(r["Kernel_Name"])
        if k:
            d[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in d.items()}


out = {"source": "separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) of "
                 "tools/roofline_target.py on an MI355X; per-launch means; fabric-side counters (Infinity-Cache hits included)"}
acc = collections.defaultdict(dict)
for sub in ("fetch", "write", "tcc", "sq"):
    f = [os.path.join(dp, x) for dp, _, fs in os.walk(os.path.join(src, sub)) for x in fs if x.endswith("counter_collection.csv")]
    if f:
        for k, cs in means(f[0]).items():
            acc[k].update(cs)
dur = {}
for dp, _, fs in os.walk(os.path.join(src, "trace")):
    for x in fs:
        if x.endswith("kernel_stats.csv"):
            for r in csv.DictReader(open(os.path.join(dp, x))):
                k = kind(r["Name"])
                if k:
                    dur[k] = float(r["AverageNs"]) / 1e3
for k, c in acc.items():
    fetch, write = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    e = {"traffic_bytes": int(2 * fetch + write), "fetch_bytes_corrected_x2": 2 * fetch, "write_bytes": write, "algorithmic_bytes": ALG[k],
         "avg_duration_us_kernel_trace": dur.get(k)}
    if "TCC_HIT_sum" in c:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)  # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
    if e["avg_duration_us_kernel_trace"]:
        e["fabric_TBps"] = e["traffic_bytes"] / e["avg_duration_us_kernel_trace"] / 1e6
    out[k] = e
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
