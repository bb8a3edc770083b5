#!/bin/bash
# session Z: small-M halo variant (conv3x3_halo_kernel<128,32,32,32,256,4>): correctness, forward A/B, per-shape times
mkdir -p gpurun_out/r03z; O=gpurun_out/r03z
export VD_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "halo or conv" -x > $O/pytest_conv.txt 2>&1; tail -5 $O/pytest_conv.txt
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_CONV_SMALLM=$w"; VD_CONV_SMALLM=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1
done; done
for w in 1 0; do VD_CONV_SMALLM=$w timeout 300 python tools/shape_profile.py 2>/dev/null | grep -i "M=512 .*ks=3\|total\|splitk" | head -8; done
