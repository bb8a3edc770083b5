cd /root/repo; export VD_QUIET=1
VARS="VD_ATTN_PIPE=0 VD_ATTN_PIPE=1"
for f in versatile-diffusion_amd/build/abl*/libvd_hip_*.so; do VARS="$VARS VD_HIP_LIB=/root/repo/$f"; done
for v in $VARS; do
  echo "== $v"; env $v python tools/attn_bench.py attn 2>&1 | grep "Nq=4096 Nk=4096"
done
