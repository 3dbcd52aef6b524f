#!/bin/bash
# PMC passes on the SHIPPED global-attention launches only (the launch plan picks 512-row q tiles / 8 waves at 64 views,
# 256-row / 4 waves at 8 views). Counters in their own runs (only --kernel-trace next to --pmc), one SQ / TCC set per pass.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_attn
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ATTN="python $R/tests/bench_kernels.py attn --modes global --views 8 64 --variants 0 --rounds 1 --target-ms 60"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/attn_pmc$i" -- $ATTN > "$O/last.log" 2>&1 || { echo "pass $i FAILED"; tail -5 "$O/last.log"; }
done
python "$R/tools/pmc_summary.py" "$O"/attn_pmc* > "$R/gpurun_out/r02_pmc_attention_final.txt" 2>&1
find "$O" -name "*.csv" -size +1M -delete
tail -40 "$R/gpurun_out/r02_pmc_attention_final.txt"
