#!/bin/bash
# round-2 GPU session B: true (graph-replayed) forward time, LayerNorm fold A/B, per-shape profile, bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 300 python tools/unet_forward.py 3 graph > $O/b_fwd_fold.log 2>&1; echo "fwd fold rc=$?"
VD_LN_FOLD=0 timeout 300 python tools/unet_forward.py 3 graph > $O/b_fwd_nofold.log 2>&1; echo "fwd nofold rc=$?"
timeout 300 python tools/shape_profile.py > $O/b_shapes_fold.txt 2>&1; echo "shapes rc=$?"
VD_LN_FOLD=0 timeout 300 python tools/shape_profile.py > $O/b_shapes_nofold.txt 2>&1; echo "shapes nofold rc=$?"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b_bench.log 2>&1; echo "bench rc=$?"
grep -h "forward ms" $O/b_fwd_fold.log $O/b_fwd_nofold.log; head -3 $O/b_shapes_fold.txt; head -3 $O/b_shapes_nofold.txt; tail -2 $O/b_bench.log | cut -c1-400
