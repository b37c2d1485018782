"""Per-launch means of rocprofv3 --pmc counter_collection.csv files for the 256x256 GEMM kernel (L1 / L2 launches alternate)."""
import csv, collections, sys
for path in sys.argv[1:]:
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "8ph" in r["Kernel_Name"]:
            d[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, disp in d.items():
        vals = [disp[i] for i in sorted(disp)]
        l1, l2 = vals[0::2], vals[1::2]
        print(f"{path.split('/')[-2]:8s} {c:26s} L1 {sum(l1) / len(l1):.4g}   L2 {sum(l2) / len(l2):.4g}   (n={len(vals)})")
