#!/bin/bash
# session O: LN fold as compile-time variant, 2-load GN shift, variant rules off -- A/B against the round-1 kernels
cd "$(dirname "$0")/.." && export VD_QUIET=1
B=$PWD/versatile-diffusion_amd/build
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv or groupnorm or gn or layernorm or ln or tile" 2>&1 | tail -3
for rep in 1 2; do
run "r01 kernels" "VD_HIP_LIB=$B/libvd_hip_r01.so VD_GEMM_TUNE=0 VD_LN_FOLD=0"
run "r02 nofold" "VD_LN_FOLD=0"
run "r02 fold" ""
done
