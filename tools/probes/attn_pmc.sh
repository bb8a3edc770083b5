# SQ counter passes over the isolated D = 40 self-attention launch, serial loop vs pipelined loop (round 6)
cd /root/repo; export VD_QUIET=1
for v in ${1:-0 1}; do
  VD_ATTN_PIPE=$v bash tools/gpu_session.sh attn_pmc_$v \
    "rpmc:SQ_WAVE_CYCLES+SQ_BUSY_CYCLES+SQ_WAIT_ANY+SQ_WAIT_INST_ANY+SQ_ACTIVE_INST_ANY+SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE;one_attn.py,8,8,4096,40" \
    "rpmc:SQ_INSTS_VALU+SQ_ACTIVE_INST_VALU+SQ_ACTIVE_INST_LDS+SQ_WAIT_INST_LDS+SQ_ACTIVE_INST_MISC+SQ_ACTIVE_INST_SCA+SQ_INSTS_MFMA+SQ_INSTS_LDS;one_attn.py,8,8,4096,40" \
    "rpmc:SQ_LDS_BANK_CONFLICT+SQ_LDS_IDX_ACTIVE+SQ_INSTS_SALU+SQ_INSTS_VMEM+SQ_ACTIVE_INST_VMEM+SQ_LDS_ADDR_CONFLICT+SQ_INST_CYCLES_SALU+SQ_WAVES;one_attn.py,8,8,4096,40" 2>&1 | grep -v "^==="
done
