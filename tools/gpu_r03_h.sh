#!/bin/bash
# session H (round 3): ff_geglu_kernel ablations (timing only; build with -DVD_FF_ABLATIONS)
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
export VD_HIP_LIB=$R/versatile-diffusion_amd/build/libvd_hip_abl.so VD_FF_VER=0
for a in 0 1 2 3 4 5; do VD_FF_ABL=$a timeout 300 python tools/ff_check.py > $O/h_ff_abl$a.txt 2>&1; echo "abl $a: $(tail -1 $O/h_ff_abl$a.txt | cut -c1-60)"; done
