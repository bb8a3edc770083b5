#!/bin/bash
# session Q (round 3): XCD tile order (which operand an XCD's L2 keeps): 0 = n fastest always, auto, 1 = m fastest always
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or conv or halo" > $O/q_pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/q_pytest.txt
for rep in 1 2; do
echo "mfast 0:    $(VD_GEMM_MFAST=0 timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"
echo "mfast auto: $(timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"
echo "mfast 1:    $(VD_GEMM_MFAST=1 timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"
done
