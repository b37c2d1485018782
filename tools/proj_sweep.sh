#!/bin/bash
# weight-gradient projection kernel time against the K-slice target of dmm_item_project_batched (DMM_PROJ_WGS_T), from a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 256 1024 4096; do
  export DMM_PROJ_WGS_T=$t
  bash tools/profile_bench.sh sw_$t sw_$t > gpurun_out/sw_$t.txt 2>&1
  echo "WGS_T=$t $(grep 'item_project_kernel<64, 1, false, true>' gpurun_out/sw_$t.txt)"
done
