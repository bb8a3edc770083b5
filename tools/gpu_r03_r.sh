#!/bin/bash
# session R (round 3): threshold of the XCD tile-order choice
cd "$(dirname "$0")/.." && export VD_QUIET=1
for rep in 1 2; do for t in 0.5 0.8 1.0 1.3; do echo "thr $t: $(VD_GEMM_MFAST_THR=$t timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"; done; done
