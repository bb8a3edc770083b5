#!/bin/bash
cd "$(dirname "$0")/.." && export VD_QUIET=1
B=$PWD/versatile-diffusion_amd/build
for lib in r01 cur; do
  if [ $lib = cur ]; then L=""; else L="VD_HIP_LIB=$B/libvd_hip_$lib.so"; fi
  echo "== $lib"; env VD_AB_KSWEEP=1 $L VD_GEMM_TUNE=0 timeout 300 python tools/gemm_ab.py 2>&1 | grep "M="
done
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
for rep in 1 2; do
run "r01 kernels" "VD_HIP_LIB=$B/libvd_hip_r01.so VD_GEMM_TUNE=0 VD_LN_FOLD=0"
run "current (fold)" ""
done
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or tile" 2>&1 | tail -2
