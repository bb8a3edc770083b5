#!/bin/bash
# session J (round 3): MODE 3 (LDS-DMA requests spread between the MFMAs) -- correctness, forward A/B
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python tools/halo_check.py check > $O/j_check.txt 2>&1; echo "check rc=$?"; tail -1 $O/j_check.txt
timeout 600 python tools/halo_forward.py "3 9 3 9 10" > $O/j_forward.txt 2>&1; echo "forward rc=$?"; grep round $O/j_forward.txt
for s in 3 9; do VD_CONV_HALO=$s timeout 600 python tools/shape_profile.py > $O/j_per_shape_$s.txt 2>&1; head -2 $O/j_per_shape_$s.txt | tail -1; done
