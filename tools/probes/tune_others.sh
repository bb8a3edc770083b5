cd /root/repo
for w in dual i2v triple; do
  echo "===== $w"
  timeout 400 python tools/tune_graph.py --workload $w --cfgs 0,1,2,3,4,13,14,15 --top 8 --reps 10 2>&1 | grep -v amdgpu.ids
done
