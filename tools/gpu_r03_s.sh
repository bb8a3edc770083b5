#!/bin/bash
# session S (round 3): wide single-head attention (VAE mid block): kernel test, VAE parity, decode timing
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/s_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/s_pytest.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "vae" > $O/s_pytest2.txt 2>&1; echo "pytest2 rc=$?"; tail -3 $O/s_pytest2.txt
timeout 300 python tools/vae_profile.py > $O/s_vae_profile.txt 2>&1; tail -15 $O/s_vae_profile.txt
