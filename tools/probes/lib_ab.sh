# A/B of the product library against development / older builds under versatile-diffusion_amd/build/*/ (VD_HIP_LIB): graph-replayed
# forward at the bench shape, two processes per library interleaved; first the GPU suite of the product library ($1 = "tests").
cd /root/repo; export VD_QUIET=1
if [ "$1" = "tests" ]; then timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4; fi
VARS="VD_QUIET=1"
for f in versatile-diffusion_amd/build/*/libvd_hip_*.so; do [ -f "$f" ] && VARS="$VARS VD_HIP_LIB=/root/repo/$f"; done
for rep in 1 2; do for v in $VARS; do
  echo "== forward $v: $(env $v python tools/unet_forward.py 3 graph 2>&1 | grep 'graph forward' | tail -2 | tr '\n' ' ')"
done; done
