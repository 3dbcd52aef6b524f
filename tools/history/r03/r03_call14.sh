#!/bin/bash
# round 3, call 14: clocks of the attention bodies: rocprofv3 --pmc GRBM_GUI_ACTIVE (+ matrix-pipe busy, wave cycles, issue stalls) of the 64-view
# launch of the product body, the order-pinned body and the 32x32x16 body, each on random and on zero-filled data (one build per pass: --solo)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; P=$O/prof_r03lab
mkdir -p "$P"; cd /tmp; export TMPDIR=/tmp
C="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
run() { # name variant tag [--zero]
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$P/$3" -- python "$R/tools/lab/run_attn_lab.py" --solo --views 64 --variants $2 --rounds 1 --names $1 $4 > "$P/$3.log" 2>&1 || { echo "$3 FAILED"; tail -3 "$P/$3.log"; }
}
run control 57 control_random
run control 57 control_zero --zero
run pipe_v2 57 pipe_v2_random
run pipe_v2 57 pipe_v2_zero --zero
run attn32_w1 80 attn32_random
run attn32_w1 80 attn32_zero --zero
cd "$R"
for t in control_random control_zero pipe_v2_random pipe_v2_zero attn32_random attn32_zero; do
  echo "==== $t"; python tools/pmc_summary.py "$P/$t" 2>/dev/null | grep -A14 "attn16_kernelIDF16bLi4ELi8\|attn32_kernel" | grep -v "^--" | head -16
done > "$O/r03_pmc_attn_lab_clocks.txt" 2>&1
find "$P" -name "*.csv" -size +1M -delete
cat "$O/r03_pmc_attn_lab_clocks.txt"
