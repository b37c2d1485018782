# Round-end refresh of the recorded runs (writes under gpurun_out/; copy the CSVs to baselines/ afterwards).
set -e
export PYTHONPATH=.
O=gpurun_out/refresh; mkdir -p $O
python tools/train_curve.py --precision 1 --seed 1818 --out $O/hip_synth_tiktok_T5_B1024_seed1818_prec1.csv > $O/curve_p1.log 2>&1
python tools/train_curve.py --precision 3 --seed 1818 --out $O/hip_synth_tiktok_T5_B1024_seed1818_prec3.csv > $O/curve_p3.log 2>&1
python tools/train_curve.py --precision 1 --seed 7 --out $O/hip_synth_tiktok_T5_B1024_seed7_prec1.csv > $O/curve_s7.log 2>&1
python bench.py --workload baby --latdim 64 --T 40 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_baby.log 2>&1
python bench.py --workload sports --latdim 128 --T 200 --batch 8192 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_sports.log 2>&1
python tools/eval_time.py tiktok > $O/eval_time.log 2>&1
tail -1 $O/bench_baby.log | cut -c1-200; tail -1 $O/bench_baby.log | grep -o "phase_ms_per_epoch.*" | cut -c1-120
tail -1 $O/bench_sports.log | cut -c1-200; tail -1 $O/bench_sports.log | grep -o "phase_ms_per_epoch.*" | cut -c1-120
tail -1 $O/eval_time.log; tail -2 $O/curve_p1.log | cut -c1-300
