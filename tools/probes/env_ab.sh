# A/B of environment settings on the graph-replayed forward: env_ab.sh "A=1" "A=0" ... (two processes per setting, interleaved)
cd /root/repo; export VD_QUIET=1
for rep in 1 2; do for v in "$@"; do
  echo "== forward $v: $(env $v python tools/unet_forward.py 3 graph 2>&1 | grep 'graph forward' | tail -2 | tr '\n' ' ')"
done; done
