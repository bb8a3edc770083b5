#!/bin/bash
cd "$(dirname "$0")/.." && export VD_QUIET=1 && mkdir -p gpurun_out
R01=$PWD/versatile-diffusion_amd/build/libvd_hip_r01.so
VD_HIP_LIB=$R01 VD_GEMM_TUNE=0 VD_LN_FOLD=0 timeout 300 python tools/shape_profile.py > gpurun_out/m_shapes_r01.txt 2>&1
VD_GEMM_VARIANT=0 VD_LN_FOLD=0 timeout 300 python tools/shape_profile.py > gpurun_out/m_shapes_r02_v0_nofold.txt 2>&1
VD_LN_FOLD=0 timeout 300 python tools/shape_profile.py > gpurun_out/m_shapes_r02_nofold.txt 2>&1
timeout 300 python tools/shape_profile.py > gpurun_out/m_shapes_r02.txt 2>&1
head -3 gpurun_out/m_shapes_*.txt
