#!/bin/bash
# kernel sequence of one diffusion step of the scaled epoch.  usage (GPU box): bash tools/profile_scale_seq.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 800 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -o scale -- python3 tools/scale_epoch.py --joint-iters 2 --no-test --users ${2:-200000} > $O/seq.log 2>&1 || { tail -5 $O/seq.log; exit 1; }
T=$(find /tmp/prof_seq -name "scale_kernel_trace.csv" | head -1)
python3 tools/trace_sequence.py $T 5 > $O/sequence.txt 2>&1
tail -3 $O/sequence.txt
