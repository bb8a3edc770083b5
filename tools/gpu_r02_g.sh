#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" > $O/g_kernels.log 2>&1; echo "kernels rc=$?"; tail -5 $O/g_kernels.log
timeout 900 python tools/gemm_sweep.py $O/r02_sweep_g.json > $O/r02_sweep_g.txt 2>&1; echo "sweep rc=$?"; tail -3 $O/r02_sweep_g.txt | cut -c1-300
