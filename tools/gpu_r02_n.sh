#!/bin/bash
cd "$(dirname "$0")/.." && export VD_QUIET=1
B=$PWD/versatile-diffusion_amd/build
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
for rep in 1 2; do
run "r01 kernels" "VD_HIP_LIB=$B/libvd_hip_r01.so VD_GEMM_TUNE=0 VD_LN_FOLD=0"
run "r02 v0 nofold" "VD_GEMM_VARIANT=0 VD_LN_FOLD=0"
run "r02 v0 nofold, LN compiled out" "VD_HIP_LIB=$B/libvd_hip_noln.so VD_GEMM_VARIANT=0 VD_LN_FOLD=0"
run "r02 v0 fold" "VD_GEMM_VARIANT=0"
done
