#!/bin/bash
# rocprofv3 kernel stats of one replica of BASELINE config 5 (10M users x 1M items, d = 256, 8 replicas; collectives reduced to their local part) with bounded
# phases.  usage (GPU box): bash tools/profile_config5.sh <tag> [joint iters] [diffusion iters] [T]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; J=${2:-12}; DI=${3:-3}; T=${4:-40}; mkdir -p $O
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python3 tools/scale_epoch.py --users 10000000 --items 1000000 --latdim 256 \
  --T $T --emulate-world 8 --joint-iters $J --diff-iters $DI --no-test > $O/scale.log 2>&1 || { tail -5 $O/scale.log; exit 1; }
S=$(find /tmp/prof_c5 -name "c5_kernel_stats.csv" | head -1)
cp $S $O/c5_kernel_stats.csv
TR=$(find /tmp/prof_c5 -name "c5_kernel_trace.csv" | head -1)
python3 tools/trace_phases.py $TR 1000 45 > $O/phases.txt 2>&1
tail -1 $O/scale.log | cut -c1-700; head -120 $O/phases.txt
