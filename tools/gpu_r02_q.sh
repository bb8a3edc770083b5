#!/bin/bash
# session Q: in-kernel split-K fix-up -- tests + A/B against the previous build and the reduce-kernel path
cd "$(dirname "$0")/.." && export VD_QUIET=1
B=$PWD/versatile-diffusion_amd/build
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv or tile" 2>&1 | tail -5
for rep in 1 2; do
run "previous build" "VD_HIP_LIB=$B/libvd_hip_pre.so"
run "fix-up" ""
run "fix-up build, reduce kernel" "VD_GEMM_FIXUP=0"
done
