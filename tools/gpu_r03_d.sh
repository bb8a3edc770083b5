#!/bin/bash
# session D (round 3): fused feed-forward kernel -- correctness, timing against the chain, forward A/B
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 300 python tools/ff_check.py > $O/d_ff_check.txt 2>&1; echo "ff rc=$?"; cat $O/d_ff_check.txt | tail -8
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "ff_geglu or halo" > $O/d_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/d_pytest.txt
VD_FF_FUSED=0 timeout 600 python tools/unet_forward.py 3 graph > $O/d_fwd_chain.txt 2>&1; echo "chain: $(tail -1 $O/d_fwd_chain.txt)"
VD_FF_FUSED=1 timeout 600 python tools/unet_forward.py 3 graph > $O/d_fwd_fused.txt 2>&1; echo "fused: $(tail -1 $O/d_fwd_fused.txt)"
VD_FF_FUSED=0 timeout 600 python tools/unet_forward.py 3 graph > $O/d_fwd_chain2.txt 2>&1; echo "chain: $(tail -1 $O/d_fwd_chain2.txt)"
VD_FF_FUSED=1 timeout 600 python tools/unet_forward.py 3 graph > $O/d_fwd_fused2.txt 2>&1; echo "fused: $(tail -1 $O/d_fwd_fused2.txt)"
