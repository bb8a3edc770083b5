cd /root/repo
for v in 1 0 1 0; do echo "VD_ATTN_W8=$v"; VD_ATTN_W8=$v python tools/attn_bench.py attn 2>&1 | grep "Nq=4096 Nk=4096"; VD_ATTN_W8=$v python tools/unet_forward.py 3 graph 2>&1 | grep "graph forward" | tail -2 | tr '\n' ' '; echo; done
