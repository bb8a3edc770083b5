#!/bin/bash
# session C (round 3): per-shape in-forward profile for halo settings 0 / 3 / 6 (planner rule by shape)
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
for s in 0 3 6 8; do
  VD_CONV_HALO=$s timeout 600 python tools/shape_profile.py > $O/c_per_shape_$s.txt 2>&1; echo "shape $s rc=$?"; head -2 $O/c_per_shape_$s.txt | tail -1
done
