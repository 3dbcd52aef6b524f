#!/bin/bash
# round 3, call 8: micro-probes for the order-pinned attention body (MFMA / VALU mix in one wave and in two waves per SIMD) and the
# attention lab: experimental builds (tools/lab/build_lab.py: static wave priority, order-pinned tile bodies v1-v3) A/B'd in one process
# against the product build, outputs compared bit for bit. Nothing here touches the product library.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
( timeout 120 tools/probes/mfma_valu_mix ) > $O/r03_probe_mfma_valu_mix.txt 2>&1
( timeout 60 tools/probes/valu_rate ) > $O/r03_probe_valu_rate.txt 2>&1
timeout 400 python tools/lab/run_attn_lab.py --views 64 8 --rounds 5 > $O/r03_attn_lab.txt 2>&1
echo "lab rc=$?" >> $O/r03_attn_lab.txt
cat $O/r03_probe_mfma_valu_mix.txt; echo; cat $O/r03_probe_valu_rate.txt; echo; grep -v "amdgpu.ids" $O/r03_attn_lab.txt
