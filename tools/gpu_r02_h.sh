#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 1500 python tools/tune_forward.py --workload t2i --reps 4 --out $O/gemm_tune_t2i.json > $O/h_tune_t2i.log 2>&1; echo "tune rc=$?"; tail -4 $O/h_tune_t2i.log | cut -c1-200
echo "== model only"; VD_GEMM_TUNE=0 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
echo "== tuned"; VD_GEMM_TUNE=$O/gemm_tune_t2i.json timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
