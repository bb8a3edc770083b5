#!/bin/bash
# session W: GroupNorm own-shift + recentring; A/B
cd "$(dirname "$0")/.." && export VD_QUIET=1
B=$PWD/versatile-diffusion_amd/build
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or gn" 2>&1 | tail -3
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
for rep in 1 2; do
run "r01 kernels" "VD_HIP_LIB=$B/libvd_hip_r01.so VD_GEMM_TUNE=0 VD_LN_FOLD=0"
run "current (fold)" ""
done
timeout 600 python tools/gn_bench.py 2>&1 | tail -12
