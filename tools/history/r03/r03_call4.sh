#!/bin/bash
# RECORD of an intermediate pass: run on a throw-away build of 21591a0 + a block-level flag 32 (fc1 / fc2 forced onto 256 x 256), removed in 8714e52; provenance of profiles/r03_gemm_mlp256_insitu.txt.
# in-situ A/B: MLP GEMMs (fc1 / fc2) on the 256 x 256 kernels below the 20 000-row threshold (gemm_tile flag 32)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd "$R"
for v in 8 12 4; do
  for t in 0 32 0 32; do
    python bench.py --views $v --gemm-tile $t --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-situ views=$v gemm_tile=$t frames/s', d['value'], 'ms', d['ms_per_step'], 'attn ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
  done
done > "$O/r03_gemm_mlp256_insitu.txt" 2>&1
cat "$O/r03_gemm_mlp256_insitu.txt"
