#!/bin/bash
# round-2 GPU session I: tune the launch table inside the forward of every BASELINE workload, then bench each
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
T=versatile-diffusion_amd/configs/gemm_tune_gfx950.json
rm -f $T
for w in t2i i2v dual triple; do
  timeout 1500 python tools/tune_forward.py --workload $w --reps 4 --merge > $O/i_tune_$w.log 2>&1; echo "tune $w rc=$?"; tail -2 $O/i_tune_$w.log | cut -c1-160
done
cp $T $O/gemm_tune_gfx950.json
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or gemm" > $O/i_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $O/i_kernels.log
echo "== forward tuned"; timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
echo "== forward model"; VD_GEMM_TUNE=0 timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
for w in t2i i2v dual triple; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/i_bench_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/i_bench_$w.log | cut -c1-200
done
