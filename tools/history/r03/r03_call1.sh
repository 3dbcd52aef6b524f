#!/bin/bash
# RECORD of an intermediate pass: run on tree a744f67 (tile 3 = the first r03 epilogue forms, since replaced); kept as the provenance of profiles/r03_gemm_epilogue_forms_ab.txt, r03_attention_tail256_ab.txt, r03_multirank_one_gpu_gloo.txt.
# Round-3 GPU pass 1: the -m gpu suite (new: headline / stress configs vs the oracle, 8 emulated ranks, camera tables, f16 outliers),
# A/B of the r03 GEMM epilogue forms and of the 256-row attention tail split, the default bench line, the N = 2/4/8 control flow on one GPU.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
(time python -m pytest tests -m gpu -q -rA 2>&1 | grep -v "^PASSED\|^$" ) > "$O/r03_gputest.log" 2>&1
tail -25 "$O/r03_gputest.log"
python tests/bench_kernels.py gemm --views 8 64 --tiles 1 2 3 --rounds 5 > "$O/r03_gemm_epilogue_ab.txt" 2>&1
cat "$O/r03_gemm_epilogue_ab.txt"
python tests/bench_kernels.py attn --views 8 10 12 --variants 50 72 0 --modes global --rounds 7 > "$O/r03_attn_tail256_ab.txt" 2>&1
cat "$O/r03_attn_tail256_ab.txt"
python bench.py > "$O/r03_bench_default_call1.json" 2> "$O/r03_bench_default_call1.err"
tail -c 1500 "$O/r03_bench_default_call1.json"; tail -5 "$O/r03_bench_default_call1.err"
bash tools/multirank_one_gpu.sh > "$O/r03_multirank_one_gpu.txt" 2>&1
cat "$O/r03_multirank_one_gpu.txt"
