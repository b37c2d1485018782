#!/bin/bash
# SpMM at shard scale (1M users x 500k items) : wall-clock run, kernel trace, then separate PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit).
# usage (GPU box): bash tools/spmm_scale_pmc.sh <out-dir-under-gpurun_out> [extra spmm_scale.py args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
python3 tools/spmm_scale.py --dims 64 "$@" > $O/plain.txt 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 tools/spmm_scale.py --dims 64 "$@" > $O/log_trace.txt 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "spmm" --output-format csv -d $O/fetch -o f -- python3 tools/spmm_scale.py --dims 64 "$@" > $O/log_f.txt 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "spmm" --output-format csv -d $O/write -o w -- python3 tools/spmm_scale.py --dims 64 "$@" > $O/log_w.txt 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "spmm" --output-format csv -d $O/tcc -o c -- python3 tools/spmm_scale.py --dims 64 "$@" > $O/log_c.txt 2>&1
cat $O/plain.txt
find $O -name "*.csv" | head -20
