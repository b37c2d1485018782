"""Idle gaps between consecutive kernels of the joint phase in a rocprofv3 kernel trace (csv): total, and the largest ones with the kernels on either side,
aggregated by (before, after) pair.  usage: trace_gaps.py <trace.csv>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
bpr = [i for i, e in enumerate(ev) if "bpr_kernel" in e[2]]
a, b = bpr[1], bpr[-1]  # from the second joint iteration on (the first records the launch plan)
pairs = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
end = ev[a][1]
for i in range(a + 1, b + 1):
    gap = (ev[i][0] - end) / 1e3
    if gap > 2.0:
        k = (ev[i - 1][2], ev[i][2]); pairs[k][0] += 1; pairs[k][1] += gap; tot += gap
    end = max(end, ev[i][1])
print(f"joint iterations {len(bpr) - 2}: span {(ev[b][1] - ev[a][0]) / 1e6:.2f} ms, idle (gaps > 2 us) {tot / 1e3:.2f} ms")
for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {v[1] / 1e3:8.2f} ms  n={v[0]:4d}  avg {v[1] / v[0]:8.1f} us   {k[0]}  ->  {k[1]}")
