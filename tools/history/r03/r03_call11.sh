#!/bin/bash
# round 3, call 11: 32x32x16-MFMA attention (lab): operand layout probe, then the lab kernels (80 / 82: 512-row pinned / plain HIP body, 81 / 83: 256-row)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
( timeout 60 tools/probes/mfma32_layout ) > $O/r03_attn32_lab.txt 2>&1
N="control pipe_v2 attn32_w1 attn32_w2"
timeout 300 python tools/lab/run_attn_lab.py --views 8 --variants 0 81 83 --rounds 3 --names $N >> $O/r03_attn32_lab.txt 2>&1
timeout 300 python tools/lab/run_attn_lab.py --views 64 --variants 0 80 82 81 --rounds 3 --names $N >> $O/r03_attn32_lab.txt 2>&1
echo "lab rc=$?" >> $O/r03_attn32_lab.txt
grep -v "amdgpu.ids" $O/r03_attn32_lab.txt
