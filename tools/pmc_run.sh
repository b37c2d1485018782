cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc3
mkdir -p $O
export DMM_GEMM_PART=2
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 tools/pmc_chain2.py > $O/log_f.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit -o h -- python3 tools/pmc_chain2.py > $O/log_h.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 tools/pmc_chain2.py > $O/log_w.txt 2>&1
