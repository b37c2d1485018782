#!/bin/bash
# Round-end artefacts (run on the GPU box: bash tools/final_profile.sh <tag>, e.g. r2): the full bench line (cpu_baseline, shard-scale SpMM
# roofline, fp32-like rate), rocprofv3 stats of the bench, stats + separate PMC passes of the isolated roofline launches.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r2c}; O=gpurun_out/final_$TAG; mkdir -p $O
timeout -k 10 700 python3 bench.py --steps 20 --warmup 5 > $O/bench_full.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o ${TAG}_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard-roofline --no-fp32-like --no-roofline > $O/prof_bench.log 2>&1 || exit 1
T=$(find $O/prof_bench -name "${TAG}_bench_kernel_trace.csv" | head -1)
python3 tools/trace_phases.py $T 18 > $O/phases.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain/trace -o t -- python3 tools/roofline_target.py > $O/log_trace.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "8ph" --output-format csv -d $O/chain/fetch -o f -- python3 tools/roofline_target.py 4 > $O/log_f.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "8ph" --output-format csv -d $O/chain/write -o w -- python3 tools/roofline_target.py 4 > $O/log_w.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "8ph" --output-format csv -d $O/chain/tcc -o c -- python3 tools/roofline_target.py 4 > $O/log_c.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "8ph" --output-format csv -d $O/chain/sq -o s -- python3 tools/roofline_target.py 4 > $O/log_s.txt 2>&1 || exit 1
python3 tools/pmc_chain_sum.py $O/chain $O/${TAG}_pmc_chain_gemm.json > $O/pmc_chain.txt 2>&1
tail -1 $O/bench_full.log | cut -c1-600
cat $O/phases.txt | head -60
