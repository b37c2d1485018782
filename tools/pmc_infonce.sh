# PMC passes over the InfoNCE kernels at the bench shape (tools/infonce_time.py 4096 64): issue / wait split, MFMA busy, LDS, L2.
# usage (GPU box): bash tools/pmc_infonce.sh [tag]   -> gpurun_out/pmc_infonce_<tag>/{sq,sq2,tc}/...counter_collection.csv + summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_infonce_${1:-a}
mkdir -p $O
R="--kernel-include-regex infonce --output-format csv"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE $R -d $O/sq -o s -- python3 tools/infonce_time.py 4096 64 > $O/log_s.txt 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES $R -d $O/sq2 -o s -- python3 tools/infonce_time.py 4096 64 > $O/log_s2.txt 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum $R -d $O/tc -o t -- python3 tools/infonce_time.py 4096 64 > $O/log_t.txt 2>&1
python3 - $O <<'PY' | tee $O/summary.txt
import csv, collections, glob, sys
for path in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv")):
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void dmm::", "")[:40]
        d[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k in d:
        for c, disp in d[k].items():
            v = list(disp.values())
            print(f"{k:42s} {c:28s} mean/launch {sum(v) / len(v):14.5g}  (n={len(v)})")
PY
