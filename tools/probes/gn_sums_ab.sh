cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "gn_apply_sums or gemm_out_stats or ff_chain or groupnorm or gn_fused" 2>&1 | tail -3
run() { echo "$*"; env "$@" python tools/unet_forward.py 3 graph 2>&1 | grep "graph forward" | tail -2 | tr '\n' ' '; echo; }
run VD_GN_SUMS=0
run VD_GN_SUMS=1
run VD_GN_SUMS_CHUNK=16384
run VD_GN_SUMS_CHUNK=24576
run VD_GN_SUMS_MINROWS=16
run VD_GN_SUMS_MINROWS=32
run VD_GN_SUMS_CHUNK=16384 VD_GN_SUMS_MINROWS=24
run VD_GN_SUMS=0
run VD_GN_SUMS=1
