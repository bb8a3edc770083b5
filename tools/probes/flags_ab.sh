#!/bin/bash
# Compiler-flag variants of libvd_hip.so (tools/probes/build_variant.py) against the product library: graph-replayed forward
# time (tools/unet_forward.py 3 graph) and the forward's output (tools/fwd_dump.py) per library, one gpurun call.
#   bash tools/probes/flags_ab.sh <variant> [<variant> ...]
cd /root/repo; export VD_QUIET=1
O=gpurun_out/flags; mkdir -p $O
B=versatile-diffusion_amd/build
run() { VD_HIP_LIB=$2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "graph forward" | sed 's/graph forward ms://' | tr '\n' ' '; echo " <- $1"; }
run base ""
for v in "$@"; do run $v $B/$v/libvd_hip_$v.so; done
run base ""
for v in "$@"; do run $v $B/$v/libvd_hip_$v.so; done
timeout 300 python tools/fwd_dump.py $O/base.pt | tail -1
for v in "$@"; do
    VD_HIP_LIB=$B/$v/libvd_hip_$v.so timeout 300 python tools/fwd_dump.py $O/$v.pt | tail -1
    python tools/fwd_dump.py --cmp $O/$v.pt $O/base.pt
done
rm -f $O/*.pt
