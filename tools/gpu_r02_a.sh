#!/bin/bash
# round-2 GPU session A: new GEMM main loop / tile configs / LayerNorm fold -- correctness, sweep, forward timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv or mfma" > $O/a_kernels.log 2>&1; echo "kernels rc=$?" | tee -a $O/a_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/a_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/a_summary.txt
timeout 600 python tools/gemm_sweep.py $O/r02_sweep_a.json > $O/r02_sweep_a.txt 2>&1; echo "sweep rc=$?" | tee -a $O/a_summary.txt
timeout 300 python tools/unet_forward.py 3 time > $O/a_fwd_fold.log 2>&1; echo "fwd fold rc=$?" | tee -a $O/a_summary.txt
VD_LN_FOLD=0 timeout 300 python tools/unet_forward.py 3 time > $O/a_fwd_nofold.log 2>&1; echo "fwd nofold rc=$?" | tee -a $O/a_summary.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "full_unet_forward or bench_shape or tiny" > $O/a_parity.log 2>&1; echo "parity rc=$?" | tee -a $O/a_summary.txt
tail -3 $O/a_kernels.log $O/a_smoke.log $O/a_fwd_fold.log $O/a_fwd_nofold.log $O/a_parity.log; tail -5 $O/r02_sweep_a.txt
