#!/bin/bash
# final evidence of a round: full GPU suite, forward / counters / traces / per-shape tables, then the bench lines (the traffic and
# trace files must sit in profiles/ before bench.py runs: it reads them when their library digest matches)
cd /root/repo
D=$1
R=${2:-r06}   # round prefix of the files bench.py reads back (PMC_TRAFFIC_FILE / TRACE_FILE)
bash tools/gpu_session.sh $D tests:all fwd pmc trace gtrace shape sq > gpurun_out/${D}_console.log 2>&1
cp gpurun_out/$D/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp gpurun_out/$D/trace_dominant.json profiles/${R}_trace_dominant.json
python bench.py --steps 5 --warmup 1 --dump-kernel-table gpurun_out/$D/forward_kernel_table.json > gpurun_out/$D/bench.json 2> gpurun_out/$D/bench.err
tail -c 600 gpurun_out/$D/bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/$D/bench_torchrun1.json 2> gpurun_out/$D/bench_torchrun1.err
tail -c 300 gpurun_out/$D/bench_torchrun1.json
python tools/shape_profile.py 2 dual > gpurun_out/$D/dual_per_shape.txt 2>&1
python tools/shape_profile.py 8 i2v > gpurun_out/$D/i2v_per_shape.txt 2>&1
python tools/shape_profile.py 4 triple > gpurun_out/$D/triple_per_shape.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$D/smoke.log 2>&1; tail -1 gpurun_out/$D/smoke.log
