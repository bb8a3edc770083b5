#!/bin/bash
# session T: GEGLU epilogue fix for 4-tile-wide waves -- tests, forward correctness, 256x256 vs 128x128w8 for GEGLU
cd "$(dirname "$0")/.." && export VD_QUIET=1
D=/tmp/fd; mkdir -p $D
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv or tile" 2>&1 | tail -3
env VD_LN_FOLD=0 VD_FWD_OVERRIDE=0 python tools/fwd_dump.py $D/ref.pt | tail -1
python tools/fwd_dump.py $D/cur.pt | tail -1
env VD_LN_FOLD=0 python tools/fwd_dump.py $D/cur_nofold.pt | tail -1
python tools/fwd_dump.py --cmp $D/cur.pt $D/ref.pt
python tools/fwd_dump.py --cmp $D/cur_nofold.pt $D/ref.pt
run() { echo "== $1"; env $2 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1; }
G="32768,2560,320,1,3,3,1;8192,5120,640,1,3,3,1;32768,2560,320,1,1,3,1;8192,5120,640,1,1,3,1"
for rep in 1 2; do
run "GEGLU 256x256" ""
run "GEGLU 128x128w8" "VD_FWD_TUNE=$G"
done
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "bench_shape or c3_shape" 2>&1 | tail -3
