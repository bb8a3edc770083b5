#!/bin/bash
# session B (round 3): planner defaults for the halo conv -- forward A/B, per-shape profile, whole GPU suite
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python tools/halo_forward.py "0 -1 3 6" > $O/b_forward.txt 2>&1; echo "forward rc=$?"; tail -8 $O/b_forward.txt
timeout 600 python tools/shape_profile.py > $O/b_per_shape.txt 2>&1; echo "shape rc=$?"; head -40 $O/b_per_shape.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/b_pytest.txt
