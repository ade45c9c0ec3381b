#!/bin/bash
# usage (GPU box): [H=128 FILT=gru_seq4] tools/probe/gru_pmc.sh  -> MFMA busy, L2 hit/miss, HBM read requests of the H = 768 GRU step kernels (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
i=0
H=${H:-768}; FILT=${FILT:-gru_step}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr"; do
  rm -rf /tmp/gq_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/gq_$i -o p -- python $GRAFT_REPO_ROOT/tools/probe/gru_host.py $H > /tmp/gq_$i.out 2>&1 \
    && python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/gq_$i/p_counter_collection.csv | grep -E "$FILT" | cut -c1-300 \
    || echo "pass '$grp' failed: $(tail -2 /tmp/gq_$i.out)"
  i=$((i+1))
done
