#!/bin/bash
# round 3, call 12: probe with dependent-accumulator cases (same accumulator every 1 / 2 / 4 MFMAs) and the 32x32x16 lab kernels with three QK^T issue orders
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
( timeout 300 tools/probes/mfma_valu_mix ) > $O/r03_probe_mfma_valu_mix.txt 2>&1
N="control attn32_w1 attn32_x1 attn32_x2"
timeout 300 python tools/lab/run_attn_lab.py --views 16 --variants 0 80 81 --rounds 3 --names $N > $O/r03_attn32_lab2.txt 2>&1
echo "lab rc=$?" >> $O/r03_attn32_lab2.txt
sed -n '/dependent MFMAs/,$p' $O/r03_probe_mfma_valu_mix.txt; grep -v "amdgpu.ids" $O/r03_attn32_lab2.txt
