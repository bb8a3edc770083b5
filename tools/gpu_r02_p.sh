#!/bin/bash
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fwd -o fwd -- python $R/tools/unet_forward.py 3 graph 2>&1 | grep "forward ms"
cd $R
DB=$(find gpurun_out/prof_fwd -name "*.db" | head -1)
echo "db: $DB"
python tools/kernel_breakdown.py $DB 63
