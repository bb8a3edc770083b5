#!/bin/bash
# session E (round 3): PMC counters of ff_geglu_kernel and conv3x3_halo_kernel (two SQ passes each)
cd "$(dirname "$0")/.." && export VD_QUIET=1
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_WAVES"
for t in ff conv; do
  if [ $t = ff ]; then CMD="python $R/tools/one_ff.py"; else CMD="python $R/tools/one_conv.py 8 64 64 320 320"; fi
  timeout 300 rocprofv3 --pmc $P1 -d $O/pmc_${t}_1 -o p -- $CMD > $O/e_pmc_${t}_1.log 2>&1; echo "$t pass1 rc=$?"
  timeout 300 rocprofv3 --pmc $P2 -d $O/pmc_${t}_2 -o p -- $CMD > $O/e_pmc_${t}_2.log 2>&1; echo "$t pass2 rc=$?"
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc_${t}_3 -o p -- $CMD > $O/e_pmc_${t}_3.log 2>&1; echo "$t pass3 rc=$?"
  for i in 1 2 3; do D=$(find $O/pmc_${t}_$i -name "*.db" | head -1); python $R/tools/pmc_summary.py $D > $O/e_pmc_${t}_$i.txt 2>&1; done
  rm -rf $O/pmc_${t}_1 $O/pmc_${t}_2 $O/pmc_${t}_3
done
cat $O/e_pmc_ff_1.txt $O/e_pmc_ff_2.txt $O/e_pmc_ff_3.txt | grep -v "^   .*n=.*mean=0$" | head -60
