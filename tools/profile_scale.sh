#!/bin/bash
# rocprofv3 kernel stats of a scaled epoch (1M x 100k, d = 256, T = 40) with a bounded joint phase.  usage (GPU box): bash tools/profile_scale.sh <tag> [joint iters]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; J=${2:-40}; mkdir -p $O
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_scale -o scale -- python3 tools/scale_epoch.py --joint-iters $J --no-test > $O/scale.log 2>&1 || { tail -5 $O/scale.log; exit 1; }
S=$(find /tmp/prof_scale -name "scale_kernel_stats.csv" | head -1)
cp $S $O/scale_kernel_stats.csv
T=$(find /tmp/prof_scale -name "scale_kernel_trace.csv" | head -1)
python3 tools/trace_phases.py $T 1000 45 > $O/phases.txt 2>&1; python3 tools/trace_gaps.py $T > $O/gaps.txt 2>&1
tail -2 $O/scale.log | cut -c1-600; head -70 $O/phases.txt
