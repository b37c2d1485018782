cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  O=gpurun_out/abprof_x16_$v; mkdir -p $O
  DMM_FEAT_X16=$v timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-shard-roofline --no-fp32-like --no-roofline > $O/log.txt 2>&1
done
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(d + "/**/p_kernel_stats.csv", recursive=True)[0]
    return {r["Name"].split("(")[0][:70]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("gpurun_out/abprof_x16_0"), load("gpurun_out/abprof_x16_1")
rows = sorted(((b.get(k, (0, 0))[1] - v[1], k, v, b.get(k)) for k, v in a.items()), reverse=True)
for d, k, va, vb in rows[:14]: print(f"{d:+8.3f} ms  {k:70s} {va} -> {vb}")
print("...")
for d, k, va, vb in rows[-6:]: print(f"{d:+8.3f} ms  {k:70s} {va} -> {vb}")
PY
