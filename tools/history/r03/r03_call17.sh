#!/bin/bash
# round 3, call 17: the pinned body on the 128-row kernels too (QB = 2: tail launch of the 64-view plan, frame-local attention at 64 views)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
N="control pipe_v2 pipe_v2_q2"
timeout 100 python tools/lab/run_attn_lab.py --views 64 --variants 0 --rounds 4 --names $N > $O/r03_attn_lab_q2.txt 2>&1
timeout 100 python tools/lab/run_attn_lab.py --mode frame --views 64 8 --variants 0 --rounds 4 --names $N >> $O/r03_attn_lab_q2.txt 2>&1
grep -v "amdgpu.ids" $O/r03_attn_lab_q2.txt
