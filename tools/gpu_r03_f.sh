#!/bin/bash
# session F (round 3): ff_geglu_kernel VER 0 / 1
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
for v in 0 1 0 1; do VD_FF_VER=$v timeout 300 python tools/ff_check.py > $O/f_ff_v$v.txt 2>&1; echo "ver $v rc=$?"; tail -6 $O/f_ff_v$v.txt | cut -c1-150; done
for v in 0 1; do VD_FF_VER=$v timeout 600 python tools/unet_forward.py 3 graph > $O/f_fwd_v$v.txt 2>&1; echo "ver $v: $(tail -1 $O/f_fwd_v$v.txt)"; done
