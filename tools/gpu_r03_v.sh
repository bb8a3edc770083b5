#!/bin/bash
# session V: 8-wave D=40 self-attention blocks: correctness, isolated timing and the in-forward A/B
mkdir -p gpurun_out/r03v; O=gpurun_out/r03v
export VD_QUIET=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" -x > $O/pytest_attn.txt 2>&1; tail -3 $O/pytest_attn.txt
for w in 1 0; do
  echo "== VD_ATTN_W8=$w"; VD_ATTN_W8=$w timeout 300 python tools/attn_bench.py attn 2>&1 | grep -v amdgpu.ids | head -3
done
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_ATTN_W8=$w"; VD_ATTN_W8=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -2
done; done
