#!/bin/bash
# session P (round 3): cross-attention block mapping (heads of one query block on one XCD)
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/p_pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/p_pytest.txt
for rep in 1 2; do for f in 0 1; do echo "ctxmap $f: $(VD_ATTN_CTXMAP=$f timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"; done; done
for f in 0 1; do VD_ATTN_CTXMAP=$f timeout 300 python tools/shape_profile.py 2>/dev/null | grep "attn_fwd" | sed "s/^/ctxmap $f /"; done
