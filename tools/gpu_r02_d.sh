#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
run() { echo "== $1"; env $1 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -2; }
run "VD_X=0"
run "VD_GEMM_VARIANT=h"
run "VD_GEMM_VARIANT=0"
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu --durations=12 -k "50_step or c3_shape or i2i_partial or injected_noise or rng_consumption or tiny_ddim or eta_and" > $O/d_parity.log 2>&1; echo "parity rc=$?"
tail -25 $O/d_parity.log
