#!/bin/bash
# Debug build of the library with per-workgroup timestamps (-DP8_TIMING) -> build_timing/libdiffmm_hip_timing.so (use with DMM_LIB=...).
set -e
cd "$(dirname "$0")/../diffmm_amd/csrc"
O=../../build_timing; mkdir -p $O
for f in spmm elementwise losses gemm gemm_bf16 gemm_bf16_8ph diffusion graph_build itemspace; do
  if [ $f = gemm_bf16_8ph ] || [ ! -f $O/$f.o ] || [ $f.hip -nt $O/$f.o ]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -DP8_TIMING -c $f.hip -o $O/$f.o &
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdiffmm_hip_timing.so $O/*.o
echo built $O/libdiffmm_hip_timing.so
