cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -2
OLD=VD_HIP_LIB=/root/repo/versatile-diffusion_amd/libvd_hip_old.so
for v in "$OLD" "VD_ATTN_STAG=0" "$OLD" "VD_ATTN_STAG=0"; do
  echo "== $v"; env $v python tools/attn_bench.py attn 2>&1 | grep "Nq=4096 Nk=4096\|Nq=1024 Nk=1024\|Nq=256 Nk=256"
done
for v in "$OLD" "VD_ATTN_STAG=0" "$OLD" "VD_ATTN_STAG=0"; do
  echo "== forward $v"; env $v python tools/unet_forward.py 3 graph 2>&1 | grep "graph forward" | tail -2 | tr '\n' ' '; echo
done
