# A/B of the software-pipelined 64-queries-per-wave D = 40 self-attention (round 6) against the serial 8-wave loop:
# isolated launches and the graph-replayed forward; development variants built by tools/probes/build_variant.py ride along.
cd /root/repo; export VD_QUIET=1
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3
VARS="VD_ATTN_PIPE=0 VD_ATTN_PIPE=1"
for f in versatile-diffusion_amd/build/*/libvd_hip_*.so; do [ -f "$f" ] && VARS="$VARS VD_HIP_LIB=/root/repo/$f"; done
for rep in 1 2; do
for v in $VARS; do
  echo "== $v"; env $v python tools/attn_bench.py attn 2>&1 | grep "Nq=4096 Nk=4096"
done; done
for v in $VARS; do
  echo "== forward $v"; env $v python tools/unet_forward.py 3 graph 2>&1 | grep "graph forward" | tail -2 | tr '\n' ' '; echo
done
