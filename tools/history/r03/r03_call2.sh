#!/bin/bash
# RECORD of an intermediate pass: run on tree dc56e19 (tile 4 = the persistent-stream kernels, since removed); provenance of profiles/r03_gemm_persistent_ab.txt, r03_attention_tail256_sweep.txt.
# Round-3 GPU pass 2: the persistent-stream 256 x 256 GEMMs (OVG_TILE_256P) -- parity, isolated A/B against the one-tile-per-workgroup
# kernels at 8 / 16 / 32 / 64 views, in-situ A/B on the whole forward; finer sweep of the 256-row attention tail split.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
(time python -m pytest tests/test_gpu_kernels.py tests/test_gpu_aggregator.py -m gpu -q -k "gemm256 or linear_epilogues or block_forward or outlier or global_attention_at_bench or baseline_view_counts" 2>&1 | tail -15) > "$O/r03_call2_tests.log" 2>&1
cat "$O/r03_call2_tests.log"
python tests/bench_kernels.py gemm --views 8 16 32 64 --tiles 1 2 4 --rounds 5 > "$O/r03_gemm_persistent_ab.txt" 2>&1
cat "$O/r03_gemm_persistent_ab.txt"
python tests/bench_kernels.py attn --views 9 10 11 13 14 --variants 50 72 --modes global --rounds 7 > "$O/r03_attn_tail256_sweep.txt" 2>&1
cat "$O/r03_attn_tail256_sweep.txt"
for v in 64 16; do
  for t in 0 4 0 4; do
    python bench.py --views $v --gemm-tile $t --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-situ views=$v gemm_tile=$t frames/s', d['value'], 'ms', d['ms_per_step'], 'attn ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
  done
done > "$O/r03_gemm_persistent_insitu.txt" 2>&1
cat "$O/r03_gemm_persistent_insitu.txt"
