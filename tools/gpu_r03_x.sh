#!/bin/bash
# session X: vd_xattn_f16 variants (VD_XATTN_VAR: 0 = stages aliased by both K/V tiles, 1 = early K/V tile 0, 2 = D=40 at 3 blocks/CU)
mkdir -p gpurun_out/r03x; O=gpurun_out/r03x
export VD_QUIET=1
for v in 0 1 2; do
  VD_XATTN_VAR=$v timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "xattn" -x 2>&1 | tail -1
  for rep in 1 2; do echo "== forward VD_XATTN_VAR=$v"; VD_XATTN_VAR=$v timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1; done
  VD_XATTN_VAR=$v timeout 300 python tools/shape_profile.py 2>/dev/null | grep -i "xattn" | head
done
