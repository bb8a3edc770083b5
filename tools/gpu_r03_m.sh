#!/bin/bash
# session M (round 3): the C x C projections (80 launches / forward) pinned to other instantiations, graph-replayed forward
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
run() { # $1 = label, $2 = tune string
  VD_FWD_TUNE="$2" timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1 | sed "s/^/$1: /"
}
echo "base: $(timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"
for cfg in 15 14 1 4 18 17 0 13 2; do
  T=""
  for key in "32768,320,320" "8192,640,640" "2048,1280,1280"; do for cls in 0 2; do T="$T$key,1,$cls,$cfg,1;"; done; done
  run "all three shapes cfg $cfg" "${T%;}"
done
for cfg in 15 14 0 13; do
  T=""; for cls in 0 2; do T="${T}2048,1280,1280,1,$cls,$cfg,1;"; done
  run "only 2048x1280x1280 cfg $cfg" "${T%;}"
done
echo "base: $(timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"
