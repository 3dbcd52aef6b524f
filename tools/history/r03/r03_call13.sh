#!/bin/bash
# round 3, call 13: is the attention launch bounded by the chip's power budget? the same kernels on random and on zero-filled q / k / v
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
N="control pipe_v2 attn32_w1"
timeout 300 python tools/lab/run_attn_lab.py --views 64 --variants 0 80 --rounds 3 --names $N > $O/r03_attn_lab_power.txt 2>&1
timeout 300 python tools/lab/run_attn_lab.py --views 64 --variants 0 80 --rounds 3 --names $N --zero >> $O/r03_attn_lab_power.txt 2>&1
echo "lab rc=$?" >> $O/r03_attn_lab_power.txt
grep -v "amdgpu.ids" $O/r03_attn_lab_power.txt
