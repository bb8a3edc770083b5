#!/bin/bash
# session K (round 3): whole GPU suite with the new parity tests (durations)
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > $O/k_pytest.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/k_pytest.txt
