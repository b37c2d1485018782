export PYTHONPATH=.
for part in 1 2; do
  echo "== part $part"
  DMM_GEMM_PART=$part timeout -k 10 120 python tools/gemm8_check.py time 2>&1 | grep -E "time" | grep -E "res|tanh|cb:"
done
