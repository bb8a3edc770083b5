# A/B of the forked context-type branches (round 6): dual / triple workloads with VD_CTX_FORK=0 / 1, plus the parity tests
cd /root/repo; export VD_QUIET=1
python -m pytest tests -m gpu -x -q -k "dual or c4_ or c5_ or multicontext or triple" 2>&1 | tail -3
for rep in 1 2; do
for v in 0 1; do
  for w in dual triple; do
    echo "== VD_CTX_FORK=$v $w"; VD_CTX_FORK=$v python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('unet_forward_ms_per_ddim_step_bs4'), d.get('config',{}).get('c_abi_calls_per_step'))"
  done
done; done
