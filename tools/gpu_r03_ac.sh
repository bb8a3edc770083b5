#!/bin/bash
# session AC: 128-pixel / four-wave halo variant (11) instead of the channel-chunk split at the 32x32 / 16x16 levels
mkdir -p gpurun_out/r03ac; O=gpurun_out/r03ac
export VD_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "halo" -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_CONV_HALO128=$w"; VD_CONV_HALO128=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1
done; done
for w in 1 0; do VD_CONV_HALO128=$w timeout 300 python tools/shape_profile.py 2>/dev/null | grep -i "conv3x3_halo\|total" | head -14; done
