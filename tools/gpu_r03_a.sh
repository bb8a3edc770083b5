#!/bin/bash
# session A (round 3): conv3x3_halo_kernel -- correctness of every variant, per-shape timing against gemm_f16_kernel,
# graph-replayed UNet forward per setting, the conv / gemm kernel tests
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python tools/halo_check.py check > $O/a_check.txt 2>&1; echo "check rc=$?"; tail -3 $O/a_check.txt
timeout 900 python tools/halo_check.py time > $O/a_time.txt 2>&1; echo "time rc=$?"; tail -4 $O/a_time.txt
timeout 900 python tools/halo_forward.py > $O/a_forward.txt 2>&1; echo "forward rc=$?"; cat $O/a_forward.txt | tail -20
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or every_tile or split_k" > $O/a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/a_pytest.txt
