#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/e_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/e_kernels.log
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "rng_consumption or tiny_clip or tiny_ddim" > $O/e_parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/e_parity.log
echo "== forward"; timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
for w in t2i i2v dual triple; do
  timeout 900 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline > $O/e_bench_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/e_bench_$w.log | cut -c1-330
done
# thread count of the cpu_baseline leg: the oracle forward on 8 .. 128 pinned threads (measured 5.1 / 4.2 / 4.6 / 6.3 / 11.2 s)
for th in 8 16 32 64 128; do VD_CPU_THREADS=$th timeout 300 python bench.py --cpu-baseline-only 2>/dev/null | tail -1; done > gpurun_out/e_cpu_threads.log 2>&1
cat gpurun_out/e_cpu_threads.log | tail -8
