"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (kernel name prefix -> counter -> mean per dispatch)."""
import csv, collections, sys
for path in sys.argv[1:]:
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        d[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, cs in d.items():
        line = f"{k:50s}"
        for c, disp in sorted(cs.items()):
            line += f" {c}={sum(disp.values()) / len(disp):.4g}"
        print(line, f"(n={len(next(iter(cs.values())))})")
