#!/bin/bash
# session Y: LayerNorm statistics inside the K loop of the LN-folded GEMMs (v_dot2): correctness, forward A/B, output A/B
mkdir -p gpurun_out/r03y; O=gpurun_out/r03y
export VD_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "layernorm_fold or every_tile or ff_geglu or xattn" -x > $O/pytest_ln.txt 2>&1; tail -5 $O/pytest_ln.txt
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_LN_INLOOP=$w"; VD_LN_INLOOP=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1
done; done
VD_LN_INLOOP=0 timeout 300 python tools/fwd_dump.py $O/a.pt > /dev/null 2>&1
VD_LN_INLOOP=1 timeout 300 python tools/fwd_dump.py $O/b.pt > /dev/null 2>&1
timeout 120 python tools/fwd_dump.py --cmp $O/a.pt $O/b.pt 2>&1 | tail -3
rm -f $O/a.pt $O/b.pt
for w in 1 0; do VD_LN_INLOOP=$w timeout 300 python tools/shape_profile.py 2>/dev/null | grep -i "cls=2\|cls=3\|row_stats\|total" | head -12; done
