#!/bin/bash
# rocprofv3 kernel trace + stats of the bench (3 timed epochs), then the per-phase breakdown of the last epoch.
# usage (GPU box): bash tools/profile_bench.sh <out-dir-under-gpurun_out> <name>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; N=$2; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $N -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard-roofline --no-fp32-like --no-roofline > $O/bench.log 2>&1 || exit 1
T=$(find $O/prof -name "${N}_kernel_trace.csv" | head -1)
python3 tools/trace_phases.py $T 18 40 > $O/phases.txt 2>&1
cat $O/phases.txt
