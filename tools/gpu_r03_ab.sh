#!/bin/bash
# session AB: ResBlock skip 1x1 convolution on a side stream (VD_RES_FORK=1) vs in line: forward A/B + output A/B
mkdir -p gpurun_out/r03ab; O=gpurun_out/r03ab
export VD_QUIET=1
for rep in 1 2; do for w in 1 0; do
  echo "== forward VD_RES_FORK=$w"; VD_RES_FORK=$w timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep -v amdgpu.ids | tail -1
done; done
VD_RES_FORK=0 timeout 300 python tools/fwd_dump.py $O/a.pt > /dev/null 2>&1
VD_RES_FORK=1 timeout 300 python tools/fwd_dump.py $O/b.pt > /dev/null 2>&1
timeout 120 python tools/fwd_dump.py --cmp $O/a.pt $O/b.pt 2>&1 | tail -2
rm -f $O/a.pt $O/b.pt
