#!/bin/bash
# session N (round 3): ticketed in-kernel split-K reduction of the halo conv -- correctness, forward A/B
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "halo or conv" > $O/n_pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/n_pytest.txt
for rep in 1 2; do for f in 0 1; do echo "fixup $f: $(VD_HALO_FIXUP=$f timeout 300 python tools/unet_forward.py 3 graph 2>/dev/null | tail -1)"; done; done
VD_HALO_FIXUP=1 timeout 300 python tools/halo_abl.py 2>&1 | grep -v amdgpu | sed 's/^/fixup 1 /'
VD_HALO_FIXUP=0 timeout 300 python tools/halo_abl.py 2>&1 | grep -v amdgpu | sed 's/^/fixup 0 /'
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "bench_shape or c4_per or tiny_ddim or graph_reuse or full_vae" > $O/n_pytest2.txt 2>&1; echo "pytest2 rc=$?"; tail -2 $O/n_pytest2.txt
