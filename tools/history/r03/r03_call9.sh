#!/bin/bash
# round 3, call 9: attention lab, second batch: clustered order-pinned bodies (v4, v5), antiphase skew behind the barrier of the 8-wave kernel,
# and the 256-row kernel (variant 50: 2 workgroups per CU, SIMD partners never synchronised) at 64 views with the pinned bodies
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
N="control pipe_v2 pipe_v4 pipe_v5 skew1 skew2 pipe_v4_skew1 pipe_v4_skew2 pipe_v2_skew2"
timeout 300 python tools/lab/run_attn_lab.py --views 64 --variants 0 50 --rounds 5 --names $N > $O/r03_attn_lab2.txt 2>&1
timeout 200 python tools/lab/run_attn_lab.py --views 8 16 --variants 0 --rounds 5 --names $N >> $O/r03_attn_lab2.txt 2>&1
echo "lab rc=$?" >> $O/r03_attn_lab2.txt
grep -v "amdgpu.ids" $O/r03_attn_lab2.txt
