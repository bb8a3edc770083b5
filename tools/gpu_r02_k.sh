#!/bin/bash
cd "$(dirname "$0")/.." && export VD_QUIET=1
R01=$PWD/versatile-diffusion_amd/build/libvd_hip_r01.so
for rep in 1 2; do
echo "== r01 kernels (LN kernels, old GEMM loop)"; VD_HIP_LIB=$R01 VD_GEMM_TUNE=0 VD_LN_FOLD=0 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
echo "== r02 kernels, model only"; VD_GEMM_TUNE=0 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
echo "== r02 kernels, tuned table"; timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
echo "== r02 kernels, model only, no LN fold"; VD_GEMM_TUNE=0 VD_LN_FOLD=0 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
done
