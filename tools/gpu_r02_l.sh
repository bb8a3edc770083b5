#!/bin/bash
cd "$(dirname "$0")/.." && export VD_QUIET=1
R01=$PWD/versatile-diffusion_amd/build/libvd_hip_r01.so
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
for rep in 1 2; do
run "r01 kernels" "VD_HIP_LIB=$R01 VD_GEMM_TUNE=0 VD_LN_FOLD=0"
run "r02 burst, planner rules" "VD_X=1"
run "r02 burst, VARIANT=0 (64-deep tiles only)" "VD_GEMM_VARIANT=0"
run "r02 burst, no LN fold" "VD_LN_FOLD=0"
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" 2>&1 | tail -2
