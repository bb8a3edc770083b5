#!/bin/bash
# One parametrised gpurun session script (round 4 on; replaces the per-session gpu_r0*_*.sh files).
#   tools/gpu_session.sh <out-subdir> <step> [<step> ...]
# steps:
#   tests:<pytest -k expression>   pytest -m gpu restricted to the expression ("all" = the whole GPU suite)
#   fwd[:ENV=V,ENV=V]              graph-replayed UNet forward at the bench shape (tools/unet_forward.py 3 graph) under the env
#   shape[:ENV=V,...]              per-shape table of one instrumented forward (tools/shape_profile.py)
#   bench[:args]                   python bench.py <args>   (commas separate arguments)
#   gtrace[:ENV=V,...]             the same of the graph-replayed forward (in-graph kernel durations, gaps between kernels)
#   ktrace[:ENV=V,...]             rocprofv3 --kernel-trace --stats of 6 eager forwards: per-kernel average durations
#   trace                          rocprofv3 --kernel-trace --stats of a short bench run -> kernel_stats.csv / breakdown
#   pmc                            FETCH_SIZE / WRITE_SIZE passes over the forward -> pmc_traffic.json
#   sq                             two SQ counter passes over 3 eager forwards -> pmc_sq.txt
#   py:<script and args>           python tools/<script> (commas separate arguments)
#   rtrace:<script,args>           rocprofv3 --kernel-trace --stats of a tool -> per-kernel averages
#   rpmc:<CTR+CTR>;<script,args>   one rocprofv3 --pmc pass over a tool -> per-kernel counter means
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
envrun() {   # envrun "A=1,B=2" cmd...
    local e="$1"; shift
    if [ -n "$e" ]; then env $(echo "$e" | tr ',' ' ') "$@"; else "$@"; fi
}
n=0
for step in "$@"; do
    n=$((n+1)); kind=${step%%:*}; arg=""; [ "$step" != "$kind" ] && arg=${step#*:}
    tag=$(printf "%02d_%s" $n "$(echo "$step" | tr -c 'A-Za-z0-9_=.-' '_' | cut -c1-60)")
    echo "=== [$n] $step"
    case $kind in
    tests)
        if [ "$arg" = "all" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $O/$tag.log 2>&1
        else timeout 1800 python -m pytest tests -m gpu -x -q -k "$arg" > $O/$tag.log 2>&1; fi
        echo "rc=$?"; tail -6 $O/$tag.log ;;
    fwd)
        envrun "$arg" timeout 600 python tools/unet_forward.py 3 graph > $O/$tag.log 2>&1; echo "rc=$?"; tail -3 $O/$tag.log ;;
    shape)
        envrun "$arg" timeout 600 python tools/shape_profile.py > $O/$tag.txt 2>&1; echo "rc=$?"; head -40 $O/$tag.txt ;;
    bench)    # bench:args   or   bench:ENV=V,ENV=V;args
        benv=""; bargs="$arg"
        case "$arg" in *\;*) benv="${arg%%;*}"; bargs="${arg#*;}" ;; esac
        envrun "$benv" timeout 1500 python bench.py $(echo "$bargs" | tr ',' ' ') > $O/$tag.json 2> $O/$tag.err; echo "rc=$?"
        tail -1 $O/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ms_per_step','unet_forward_ms_per_ddim_step_bs4','whole_path_frac_of_mfma_peak')}, {k: v.get('value') for k, v in (d.get('other_workloads') or {}).items()}, (d.get('roofline') or {}).get('frac'))" ;;
    trace)
        (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/$tag.log 2>&1); echo "rc=$?"
        DB=$(find $O/prof_bench -name "*.db" | head -1)
        python tools/kernel_stats.py $DB > $O/kernel_stats.csv 2> $O/kernel_stats.err; head -12 $O/kernel_stats.csv | cut -c1-220
        NF=$(python tools/count_forwards.py $DB); echo "forwards in the traced run: $NF"
        python tools/kernel_breakdown.py $DB $NF > $O/kernel_breakdown.txt 2>&1; head -30 $O/kernel_breakdown.txt
        python tools/trace_dominant.py $DB > $O/trace_dominant.json 2> $O/trace_dominant.err; cat $O/trace_dominant.json
        rm -rf $O/prof_bench ;;
    gtrace)   # kernel trace of the GRAPH-replayed forward (3 eager + 60 replays: the averages are the in-graph durations)
        (cd /tmp && export TMPDIR=/tmp && envrun "$arg" timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o t -- python $R/tools/unet_forward.py 3 graph > $O/$tag.log 2>&1); echo "rc=$?"; tail -3 $O/$tag.log
        DB=$(find $O/prof_$n -name "*.db" | head -1)
        python tools/kernel_stats.py $DB > $O/$tag.csv 2> $O/$tag.err; head -40 $O/$tag.csv | cut -c1-150
        python tools/graph_gaps.py $DB > $O/$tag.gaps.txt 2>&1; tail -8 $O/$tag.gaps.txt
        rm -rf $O/prof_$n ;;
    ktrace)   # kernel trace of a few eager forwards (tools/unet_forward.py) under the env: per-kernel average durations
        (cd /tmp && export TMPDIR=/tmp && envrun "$arg" timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o t -- python $R/tools/unet_forward.py 6 > $O/$tag.log 2>&1); echo "rc=$?"
        DB=$(find $O/prof_$n -name "*.db" | head -1)
        python tools/kernel_stats.py $DB > $O/$tag.csv 2> $O/$tag.err; head -40 $O/$tag.csv | cut -c1-150
        rm -rf $O/prof_$n ;;
    pmc)
        (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/tools/unet_forward.py 3 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
         timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/tools/unet_forward.py 3 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?")
        F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
        python tools/pmc_traffic.py $F $W > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "traffic rc=$? $(wc -c < $O/pmc_traffic.json) bytes"
        rm -rf $O/pmc_fetch $O/pmc_write ;;
    rtrace)   # rtrace:<script,args>   rocprofv3 --kernel-trace --stats of python tools/<script> -> per-kernel averages
        (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o t -- python $R/tools/$(echo "$arg" | tr ',' ' ') > $O/$tag.log 2>&1); echo "rc=$?"
        DB=$(find $O/prof_$n -name "*.db" | head -1)
        python tools/kernel_stats.py $DB > $O/$tag.csv 2> $O/$tag.err; head -40 $O/$tag.csv | cut -c1-200
        rm -rf $O/prof_$n ;;
    rpmc)     # rpmc:<CTR+CTR+...>;<script,args>   one counter pass (no trace domains besides the kernel trace) -> per-kernel means
        ctrs="${arg%%;*}"; scr="${arg#*;}"
        (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc $(echo "$ctrs" | tr '+' ' ') -d $O/prof_$n -o c -- python $R/tools/$(echo "$scr" | tr ',' ' ') > $O/$tag.log 2>&1); echo "rc=$?"
        DB=$(find $O/prof_$n -name "*.db" | head -1)
        python tools/pmc_summary.py $DB > $O/$tag.txt 2> $O/$tag.err; head -60 $O/$tag.txt | cut -c1-160
        rm -rf $O/prof_$n ;;
    sq)       # two SQ counter passes over 3 eager forwards -> pmc_sq.txt (issue / stall split, matrix-pipe busy, LDS conflicts per kernel)
        (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $O/sq_a -o a -- python $R/tools/unet_forward.py 3 > $O/sq_a.log 2>&1; echo "sq a rc=$?"
         timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/sq_b -o b -- python $R/tools/unet_forward.py 3 > $O/sq_b.log 2>&1; echo "sq b rc=$?")
        A=$(find $O/sq_a -name "*.db" | head -1); B=$(find $O/sq_b -name "*.db" | head -1)
        python tools/pmc_sq_table.py $A $B > $O/pmc_sq.txt 2> $O/pmc_sq.err; echo "sq table rc=$?"; head -30 $O/pmc_sq.txt | cut -c1-170
        rm -rf $O/sq_a $O/sq_b ;;
    py)
        timeout 900 python tools/$(echo "$arg" | tr ',' ' ') > $O/$tag.log 2>&1; echo "rc=$?"; tail -25 $O/$tag.log ;;
    *) echo "unknown step $step" ;;
    esac
done
