#!/bin/bash
# session W: vd_xattn_f16 (LayerNorm + to_q + cross-attention in one launch): correctness, forward A/B, output A/B
mkdir -p gpurun_out/r03w; O=gpurun_out/r03w
export VD_QUIET=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "xattn" -x > $O/pytest_xattn.txt 2>&1; tail -15 $O/pytest_xattn.txt
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_XATTN_FUSED=$w"; VD_XATTN_FUSED=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1
done; done
VD_XATTN_FUSED=0 timeout 300 python tools/fwd_dump.py $O/a.pt > /dev/null 2>&1
VD_XATTN_FUSED=1 timeout 300 python tools/fwd_dump.py $O/b.pt > /dev/null 2>&1
timeout 120 python tools/fwd_dump.py --cmp $O/a.pt $O/b.pt 2>&1 | tail -3
rm -f $O/a.pt $O/b.pt
VD_XATTN_FUSED=1 timeout 300 python tools/shape_profile.py 2>/dev/null | grep -i "xattn\|Nk=77\|total" | head
