"""Kernel sequence of one diffusion-training step out of a rocprofv3 kernel trace (csv): the launches between the N-th and (N+1)-th
qsample_temb launch group, in start order, with duration and grid size.  usage: trace_sequence.py <trace.csv> [step index, default 5]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
per = 3  # Adam launches per diffusion step (one per modality)
a, b = adam[n * per - 1] + 1, adam[(n + 1) * per - 1]
t0 = ev[a][0]
for s, e, name, gx, wx in ev[a:b + 1]:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  grid {gx:>10s}/{wx:>4s}  {name.split('(')[0][:90]}")
print(f"step span {(ev[b][1] - t0) / 1e3:.1f} us, {b - a + 1} launches")
