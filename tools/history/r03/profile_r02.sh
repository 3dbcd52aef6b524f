#!/bin/bash
# Round-2 profiling pass (run on the GPU box through gpurun): per-shape rocprofv3 kernel stats of bench.py (64 views and
# 8 views in SEPARATE runs, so per-kernel averages do not mix shapes) and PMC passes on the shipped attention / GEMM kernels.
# Counters are collected in their own runs (only --kernel-trace next to --pmc), one SQ / TCC set per pass.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*"; "$@" > "$O/last.log" 2>&1 || { echo "   FAILED rc=$?"; tail -5 "$O/last.log"; }; }

run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_s64" -- python "$R/bench.py" --views 64 --steps 3 --warmup 1 --no-cpu-baseline --no-parity
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_s8" -- python "$R/bench.py" --views 8 --steps 5 --warmup 2 --no-cpu-baseline --no-parity

ATTN="python $R/tests/bench_kernels.py attn --modes global --views 8 64 --variants 0 --rounds 1 --target-ms 60"
GEMM="python $R/tests/bench_kernels.py gemm --views 64 --tiles 2 --rounds 1 --target-ms 5"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i + 1))
  run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/attn_pmc$i" -- $ATTN
  run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/gemm_pmc$i" -- $GEMM
done
python "$R/tools/pmc_summary.py" "$O"/attn_pmc* > "$R/gpurun_out/r02_pmc_attention.txt" 2>&1
python "$R/tools/pmc_summary.py" "$O"/gemm_pmc* > "$R/gpurun_out/r02_pmc_gemm.txt" 2>&1
for d in bench_s64 bench_s8; do
  f=$(find "$O/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$R/gpurun_out/r02_${d}_kernel_stats.csv"
done
# keep the merged-back payload small: the raw traces stay on the box
find "$O" -name "*.csv" -size +2M -delete
du -sh "$O" | tail -1
