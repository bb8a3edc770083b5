#!/bin/bash
# session AE: SQ counters per kernel over 3 eager UNet forwards (MFMA busy, waits, LDS) -> profiles/r03_pmc_sq.txt
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03ae; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_a -o a -- python $R/tools/unet_forward.py 3 > $O/pmc_a.log 2>&1; echo "pmc a rc=$?"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_b -o b -- python $R/tools/unet_forward.py 3 > $O/pmc_b.log 2>&1; echo "pmc b rc=$?"
cd $R
A=$(find $O/pmc_a -name "*.db" | head -1); B=$(find $O/pmc_b -name "*.db" | head -1)
python tools/pmc_summary.py $A > $O/r03_pmc_sq_a.txt 2>&1; python tools/pmc_summary.py $B > $O/r03_pmc_sq_b.txt 2>&1
wc -l $O/r03_pmc_sq_a.txt $O/r03_pmc_sq_b.txt
rm -rf $O/pmc_a $O/pmc_b
