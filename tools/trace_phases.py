"""Phase-resolved kernel time of the LAST epoch in a rocprofv3 kernel trace of bench.py (csv)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
bpr = [i for i, e in enumerate(ev) if "bpr_kernel" in e[2]]
n_joint = int(sys.argv[2]) if len(sys.argv) > 2 else 18
last_start = bpr[-(n_joint + 1)] + 1 if len(bpr) > n_joint else 0  # (a single-epoch trace: from the start)
i0 = [i for i, e in enumerate(ev) if "qsample" in e[2] and i > last_start][0]
tk = [i for i, e in enumerate(ev) if "topk_rows" in e[2] and i > i0]
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2] and i0 < i < tk[0]]
gb = [i for i, e in enumerate(ev) if "gb_finish_item_rows" in e[2] and i > tk[-1]]
cuts = [("diffusion", i0, adam[-1]), ("rebuild", adam[-1] + 1, gb[-1] if gb else tk[-1]), ("joint", (gb[-1] if gb else tk[-1]) + 1, len(ev) - 1)]
for label, a, b in cuts:
    d = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in ev[a:b + 1]:
        k = n.split("(")[0][:64]
        d[k][0] += 1; d[k][1] += (e - s) / 1e3
    span = (ev[b][1] - ev[a][0]) / 1e3
    tot = sum(v[1] for v in d.values())
    print(f"== {label}: span {span / 1e3:.2f} ms, sum of kernel time {tot / 1e3:.2f} ms, {b - a + 1} kernels")
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
        print(f"   {k:64s} n={v[0]:4d} {v[1] / 1e3:8.3f} ms  avg {v[1] / v[0]:7.1f} us")
