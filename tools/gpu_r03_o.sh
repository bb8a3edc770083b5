#!/bin/bash
# session O (round 3): gemm_f16_kernel with the mid-barrier loop (cfg 25 = 256x128, 26 = 128x128 on 8 waves): tests, forward A/B per shape
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "every_tile or split_k or gemm_plain or geglu or layernorm_fold" > $O/o_pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/o_pytest.txt
run() { VD_FWD_TUNE="$2" timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1 | sed "s/^/$1: /"; }
echo "base: $(timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"
for cfg in 25 26; do
run "geglu 32x32 cfg $cfg" "8192,5120,640,1,3,$cfg,1"
run "geglu 16x16 cfg $cfg" "2048,10240,1280,1,3,$cfg,1"
run "ffout 32x32 cfg $cfg" "8192,640,2560,1,0,$cfg,1"
run "ffout 16x16 cfg $cfg split1" "2048,1280,5120,1,0,$cfg,1"
run "ffout 16x16 cfg $cfg split3" "2048,1280,5120,1,0,$cfg,3"
run "qkv 32x32 cfg $cfg" "8192,1920,640,1,2,$cfg,1"
run "qkv 16x16 cfg $cfg" "2048,3840,1280,1,2,$cfg,1"
run "qkv 64x64 cfg $cfg" "32768,960,320,1,2,$cfg,1"
run "all of them cfg $cfg" "8192,5120,640,1,3,$cfg,1;2048,10240,1280,1,3,$cfg,1;8192,640,2560,1,0,$cfg,1;8192,1920,640,1,2,$cfg,1;2048,3840,1280,1,2,$cfg,1"
done
echo "base: $(timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"
