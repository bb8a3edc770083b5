#!/bin/bash
# session I (round 3): conv3x3_halo_kernel ablations (timing only; -DVD_HALO_ABLATIONS build): 0 none, 1 epilogue without
# global traffic, 2 no main loop (prologue + epilogue only), 3 no LDS-DMA inside the loop
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
export VD_HIP_LIB=$R/versatile-diffusion_amd/build/libvd_hip_abl.so
for a in 0 1 2 3; do echo "== abl $a"; VD_HALO_ABL=$a timeout 300 python tools/halo_abl.py 2>&1 | grep -v amdgpu.ids | tee $O/i_halo_abl$a.txt; done
