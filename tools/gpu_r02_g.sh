#!/bin/bash
# session G: captured DDIM step kept across sample() calls -- tests + bench A/B
cd "$(dirname "$0")/.." && export VD_QUIET=1
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "graph_reuse or tiny_ddim or eta_and_unguided or rng_consumption or injected_noise or tiny_text_latent or i2i_partial" 2>&1 | tail -4
for rep in 1 2; do
echo "== graph re-captured per call"; env VD_DDIM_GRAPH_CACHE=0 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-130
echo "== graph kept across calls"; timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-130
done
