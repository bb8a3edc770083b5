#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export VD_QUIET=1
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/e_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/e_kernels.log
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "rng_consumption or tiny_clip or tiny_ddim" > $O/e_parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/e_parity.log
echo "== forward"; timeout 300 python tools/unet_forward.py 3 graph 2>&1 | grep "forward ms" | tail -1
for w in t2i i2v dual triple; do
  timeout 900 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline > $O/e_bench_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/e_bench_$w.log | cut -c1-330
done
python - <<'PY' > gpurun_out/e_cpu_threads.log 2>&1
import os, sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "versatile-diffusion_amd")
import bench
from oracle import vd_oracle as O
print("cpu_count", os.cpu_count())
from lib.cfg_helper import model_cfg_bank
from lib.model_zoo import get_model
net = get_model()(model_cfg_bank()("openai_unet_2d_v1"), verbose=False)
sd = {"diffuser.image." + k: v.detach().float() for k, v in net.state_dict().items()}
from lib.model_zoo import get_model as gm
net0 = get_model()(model_cfg_bank()("openai_unet_0d_v1_c"), verbose=False)
sd.update({"diffuser.text." + k: v.detach().float() for k, v in net0.state_dict().items()})
g = torch.Generator().manual_seed(0)
x = torch.randn((2, 4, 64, 64), generator=g); c = torch.randn((2, 77, 768), generator=g) * 0.5; t = torch.tensor([501, 501])
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        O.apply_model(sd, O.unet_plan(), x, t, c, c_type="text", global_ptr="image")
        t0 = time.time(); O.apply_model(sd, O.unet_plan(), x, t, c, c_type="text", global_ptr="image"); dt = time.time() - t0
    print("threads %d: forward %.2f s" % (th, dt)); sys.stdout.flush()
PY
cat gpurun_out/e_cpu_threads.log | tail -8
