#!/bin/bash
# session X: kernel traces of the graph-replayed forward, round-1 library vs current, same box
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; B=$R/versatile-diffusion_amd/build; O=$R/gpurun_out/x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env VD_HIP_LIB=$B/libvd_hip_r01.so VD_GEMM_TUNE=0 VD_LN_FOLD=0 timeout 600 rocprofv3 --kernel-trace -d $O/a -o a -- python $R/tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
timeout 600 rocprofv3 --kernel-trace -d $O/b -o b -- python $R/tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
env VD_LN_FOLD=0 timeout 600 rocprofv3 --kernel-trace -d $O/c -o c -- python $R/tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
cd $R
A=$(find $O/a -name "*.db" | head -1); Bd=$(find $O/b -name "*.db" | head -1); C=$(find $O/c -name "*.db" | head -1)
echo "=== r01 (A) vs current fold (B)"; python tools/kernel_diff.py $A $Bd 63
echo "=== r01 (A) vs current nofold (B)"; python tools/kernel_diff.py $A $C 63
rm -rf $O
