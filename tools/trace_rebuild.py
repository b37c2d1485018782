#!/usr/bin/env python3
"""Timeline of the LAST rebuild phase in a rocprofv3 kernel trace (csv): every launch between the last epoch's two latent-chain launches' start and
the first joint-phase SpMM behind them -- start, end (us from the first chain launch), queue, kernel.  usage: trace_rebuild.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void dmm::", "").replace("dmm::", "")[:60], r.get("Queue_Id", "?")) for r in rows)
chains = [i for i, e in enumerate(ev) if "latent_chain" in e[2]]
i0 = chains[-2]
t0 = ev[i0][0]
for s, e, name, q in ev[max(0, i0 - 14):]:
    if "spmm_kernel" in name and s > ev[chains[-1]][1]:
        print(f"{(s - t0) / 1e3:9.1f}  first joint SpMM")
        break
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  q{q:>3s}  {name}")
