cd /root/repo
for v in 0 1 0 1; do echo "VD_ST_CENTER=$v"; VD_ST_CENTER=$v python tools/unet_forward.py 3 graph 2>&1 | grep "graph forward" | tail -2 | tr '\n' ' '; echo; done
