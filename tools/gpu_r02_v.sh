#!/bin/bash
# session V: LayerNorm fold with precomputed row statistics -- tests, forward correctness, A/B vs in-loop statistics and no fold
cd "$(dirname "$0")/.." && export VD_QUIET=1
B=$PWD/versatile-diffusion_amd/build
D=/tmp/fd; mkdir -p $D
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or row_stats or tile" 2>&1 | tail -3
env VD_LN_FOLD=0 VD_FWD_OVERRIDE=0 python tools/fwd_dump.py $D/ref.pt | tail -1
python tools/fwd_dump.py $D/cur.pt | tail -1
python tools/fwd_dump.py --cmp $D/cur.pt $D/ref.pt
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
for rep in 1 2; do
run "r01 kernels" "VD_HIP_LIB=$B/libvd_hip_r01.so VD_GEMM_TUNE=0 VD_LN_FOLD=0"
run "fold, statistics kernel" ""
run "no fold" "VD_LN_FOLD=0"
done
timeout 600 python tools/shape_profile.py > gpurun_out/v_shapes.txt 2>&1; grep -n "cls=2\|cls=3\|row_stats\|layernorm\|total" gpurun_out/v_shapes.txt | head -20
