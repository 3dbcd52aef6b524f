#!/bin/bash
# RECORD of an intermediate pass: run on tree 21591a0; provenance of profiles/r03_gemm_staged_stores_ab.txt (tile flag 16 = OVG_TILE_R02_EPILOGUE still exists).
# Round-3 GPU pass 3: whole-line stores staged through the idle LDS (both GEMM tile sizes, QK head rows): parity, isolated A/B against the
# r02 epilogue forms (tile flag 16), in-situ A/B on the whole forward at 64 / 16 / 8 views.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
(time python -m pytest tests/test_gpu_kernels.py tests/test_gpu_aggregator.py -m gpu -q -k "gemm256 or linear_epilogues or qkv or block_forward or baseline_view_counts or modality_combos or headline" 2>&1 | tail -15) > "$O/r03_call3_tests.log" 2>&1
cat "$O/r03_call3_tests.log"
python tests/bench_kernels.py gemm --views 8 16 64 --tiles 17 1 18 2 --rounds 5 > "$O/r03_gemm_staged_stores_ab.txt" 2>&1
cat "$O/r03_gemm_staged_stores_ab.txt"
for v in 64 16 8; do
  for t in 16 0 16 0; do
    python bench.py --views $v --gemm-tile $t --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-situ views=$v gemm_tile=$t frames/s', d['value'], 'ms', d['ms_per_step'], 'attn ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
  done
done > "$O/r03_gemm_staged_stores_insitu.txt" 2>&1
cat "$O/r03_gemm_staged_stores_insitu.txt"
