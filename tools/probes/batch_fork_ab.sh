# A/B of the half-batch branches over the low-resolution UNet levels (round 6, VD_BATCH_FORK): its parity test, then the workloads
# ($WL, default: t2i i2v) with the fork off (0) / default (auto: batches >= 16 with one context type) / forced (1).
cd /root/repo; export VD_QUIET=1
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "half_batch or c3_shape_forward" 2>&1 | tail -3
line() { python bench.py --workload $1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-workloads 2>>gpurun_out/batch_fork_err.log | tail -1 | python -c "import sys,json; t=sys.stdin.read(); d=json.loads(t) if t.strip().startswith('{') else {'value':'FAILED','ms_per_step':None}; print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for w in ${WL:-t2i i2v}; do
  echo "== $w fork 0: $(VD_BATCH_FORK=0 line $w)"
  echo "== $w fork auto: $(line $w)"
  echo "== $w fork 1: $(VD_BATCH_FORK=1 line $w)"
done; done
