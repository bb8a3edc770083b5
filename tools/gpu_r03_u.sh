#!/bin/bash
# session U (round 3): evidence -- PMC traffic passes (stamped with the library digest), bench lines for all four workloads,
# kernel trace of the bench command, per-shape table, graph-replayed forward, torchrun launch on one rank
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/tools/unet_forward.py 3 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/tools/unet_forward.py 3 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python tools/pmc_traffic.py $F $W > $O/r03_pmc_traffic.json 2> $O/pmc_traffic.err; echo "traffic rc=$? $(wc -c < $O/r03_pmc_traffic.json) bytes"
cp $O/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
timeout 900 python bench.py --steps 3 --warmup 1 --dump-kernel-table $O/r03_forward_kernel_table.json > $O/r03_bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/r03_bench.json | cut -c1-1200
for w in i2v dual triple; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"; tail -1 $O/r03_bench_$w.json | cut -c1-300
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o final -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof.log 2>&1; echo "prof rc=$?"
cd $R
DB=$(find $O/prof_bench -name "*.db" | head -1)
python tools/kernel_stats.py $DB > $O/r03_kernel_stats.csv 2> $O/kernel_stats.err; head -8 $O/r03_kernel_stats.csv | cut -c1-200
NF=$(python - <<PY
import sqlite3,sys
con=sqlite3.connect("$DB"); cur=con.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
t=[x for x in tabs if x=="kernels"] or [x for x in tabs if "kernel_dispatch" in x]
cols=[c[1] for c in cur.execute("pragma table_info('%s')"%t[0])]
nc="name" if "name" in cols else [c for c in cols if "name" in c][0]
print(sum(1 for r in cur.execute("select %s from %s"%(nc,t[0])) if "timestep_embedding" in r[0]))
PY
)
echo "forwards in the traced run: $NF"
python tools/kernel_breakdown.py $DB $NF > $O/r03_kernel_breakdown.txt 2>&1; head -30 $O/r03_kernel_breakdown.txt
rm -rf $O/prof_bench
timeout 600 python tools/shape_profile.py > $O/r03_forward_per_shape.txt 2>&1; head -3 $O/r03_forward_per_shape.txt
timeout 300 python tools/unet_forward.py 3 graph > $O/r03_graph_forward.txt 2>&1; tail -3 $O/r03_graph_forward.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/r03_bench_torchrun1.json 2> $O/torchrun.err; echo "torchrun rc=$?"; tail -1 $O/r03_bench_torchrun1.json | cut -c1-300
