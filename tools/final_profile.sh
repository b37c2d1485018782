# Round-end artefacts: full bench line (with cpu_baseline), rocprofv3 stats of the bench and of the isolated roofline launches.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_full.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r1g -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roofline -o r1g_roofline -- python3 tools/roofline_target.py > $O/prof_roofline.log 2>&1 || exit 1
tail -1 $O/bench_full.log
