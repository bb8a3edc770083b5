#!/bin/bash
# session AD: SpatialTransformer entry chain (GroupNorm affine -> proj_in -> LN -> q|k|v in one launch): correctness, forward A/B
mkdir -p gpurun_out/r03ad; O=gpurun_out/r03ad
export VD_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "row320" -x > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_ST_CHAIN=$w"; VD_ST_CHAIN=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1
done; done
for w in 1 0; do VD_ST_CHAIN=$w timeout 300 python tools/shape_profile.py 2>/dev/null | grep -i "rowchain\|rowgemm\|affine\|M=32768 N=320 K=320\|groupnorm\|total" | head -8; done
VD_ST_CHAIN=0 timeout 300 python tools/fwd_dump.py $O/a.pt > /dev/null 2>&1
VD_ST_CHAIN=1 timeout 300 python tools/fwd_dump.py $O/b.pt > /dev/null 2>&1
timeout 120 python tools/fwd_dump.py --cmp $O/a.pt $O/b.pt 2>&1 | tail -2
rm -f $O/a.pt $O/b.pt
