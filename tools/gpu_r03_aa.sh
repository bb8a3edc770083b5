#!/bin/bash
# session AA: vd_gemm_row320_f16 (rows of x in registers, K = 320): correctness, forward A/B, per-shape times
mkdir -p gpurun_out/r03aa; O=gpurun_out/r03aa
export VD_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "row320" -x > $O/pytest_row.txt 2>&1; tail -8 $O/pytest_row.txt
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_GEMM_ROW320=$w"; VD_GEMM_ROW320=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1
done; done
for w in 1 0; do VD_GEMM_ROW320=$w timeout 300 python tools/shape_profile.py 2>/dev/null | grep -i "rowgemm\|M=32768 N=320 K=320\|M=32768 N=960\|total" | head -8; done
VD_GEMM_ROW320=0 timeout 300 python tools/fwd_dump.py $O/a.pt > /dev/null 2>&1
VD_GEMM_ROW320=1 timeout 300 python tools/fwd_dump.py $O/b.pt > /dev/null 2>&1
timeout 120 python tools/fwd_dump.py --cmp $O/a.pt $O/b.pt 2>&1 | tail -2
rm -f $O/a.pt $O/b.pt
