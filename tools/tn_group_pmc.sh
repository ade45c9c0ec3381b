#!/bin/bash
# usage (GPU box): [ARITH=6] tools/tn_group_pmc.sh -> kernel durations + PMC passes (one per counter group, kernel trace only) of tools/tn_group_bench.py
cd /tmp && export TMPDIR=/tmp
A=${ARITH:-6}
rm -rf /tmp/tq_k; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tq_k -o p -- python $GRAFT_REPO_ROOT/tools/tn_group_bench.py $A 30 > /tmp/tq_k.out 2>&1
tail -1 /tmp/tq_k.out
python - <<PY
import csv
for r in csv.DictReader(open("/tmp/tq_k/p_kernel_stats.csv")):
    if "ur::" in r["Name"]: print("%8.1f us avg x %4d  %s" % (float(r["AverageNs"])/1000, int(r["Calls"]), r["Name"][:100]))
PY
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/tq_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/tq_$i -o p -- python $GRAFT_REPO_ROOT/tools/tn_group_bench.py $A 10 > /tmp/tq_$i.out 2>&1 \
    && python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/tq_$i/p_counter_collection.csv | grep -E "gemm_tn" | cut -c1-300 \
    || echo "pass '$grp' failed: $(tail -2 /tmp/tq_$i.out)"
  i=$((i+1))
done
