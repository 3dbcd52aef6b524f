#!/bin/bash
# round 3, call 15: what do the 8 row-sum MFMAs per tile cost under the power cap? pinned body v2 with and without them (the second one computes garbage: timing only)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/lab/run_attn_lab.py --views 64 8 --variants 0 --rounds 4 --names control pipe_v2 pipe_v2_norowsum > $O/r03_attn_lab_norowsum.txt 2>&1
grep -v "amdgpu.ids" $O/r03_attn_lab_norowsum.txt
