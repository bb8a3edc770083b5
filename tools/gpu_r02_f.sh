#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 300 python tools/rng_debug.py > $O/f_rng.log 2>&1; tail -14 $O/f_rng.log
timeout 900 python -m pytest tests/test_optimus_gpu.py -x -q -m gpu > $O/f_optimus.log 2>&1; echo "optimus rc=$?"; tail -12 $O/f_optimus.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "adjust_rank or mask_patch or color_adjust or attention" > $O/f_kernels.log 2>&1; echo "kernels rc=$?"; tail -12 $O/f_kernels.log
