#!/bin/bash
# Round-3 extra profiling on the final tree: PMC passes on the GEMMs at M = 10 992 (8 views: the 128 x 128 kernels the launch heuristic picks
# there, and the shape every rank of the 8-GPU run works on) and rocprofv3 kernel stats of the whole OmniVGGT.forward at 8 views (all heads on HIP).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r03x
mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*"; "$@" > "$P/last.log" 2>&1 || { echo "   FAILED rc=$?"; tail -5 "$P/last.log"; }; }
GEMM="python $R/tests/bench_kernels.py gemm --views 8 --tiles 0 --rounds 1 --target-ms 5"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i + 1))
  run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/gemm8_pmc$i" -- $GEMM
done
python "$R/tools/pmc_summary.py" "$P"/gemm8_pmc* > "$O/r03_pmc_gemm_S8.txt" 2>&1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$P/e2e8" -- python "$R/bench.py" --views 8 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --e2e
f=$(find "$P/e2e8" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r03_bench_e2e_S8_kernel_stats.csv"
find "$P" -name "*.csv" -size +1M -delete
grep -v "^    [A-Z]" "$O/r03_pmc_gemm_S8.txt" | grep -A8 "linear_kernel\|qkv_kernel" | head -60
