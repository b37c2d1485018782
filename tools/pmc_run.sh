cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc4
mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "infonce|spmm_kernel|topk_rows|item_" --output-format csv -d $O/sq -o s -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/log_s.txt 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-include-regex "infonce|spmm_kernel|topk_rows|item_" --output-format csv -d $O/tc -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/log_t.txt 2>&1
ls $O/*
