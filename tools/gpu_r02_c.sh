#!/bin/bash
# round-2 GPU session C: in-forward A/B of output-store policy (nt vs cached) and 32-deep / 4-stage K pipeline variants
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "every_tile or layernorm_fold" > $O/c_kernels.log 2>&1; echo "kernels rc=$?"
run() { echo "== $1"; env $1 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -2; }
run "VD_X=0"
run "VD_GEMM_NT=0"
run "VD_GEMM_VARIANT=q"
run "VD_GEMM_NT=0 VD_GEMM_VARIANT=q"
run "VD_GEMM_NT=0 VD_LN_FOLD=0"
run "VD_GEMM_NT=0 VD_GEMM_VARIANT=q VD_LN_FOLD=0"
VD_GEMM_NT=0 timeout 300 python tools/shape_profile.py > $O/c_shapes_nt0.txt 2>&1
VD_GEMM_NT=0 VD_GEMM_VARIANT=q timeout 300 python tools/shape_profile.py > $O/c_shapes_nt0_q.txt 2>&1
head -4 $O/c_shapes_nt0.txt $O/c_shapes_nt0_q.txt; tail -4 $O/c_kernels.log
