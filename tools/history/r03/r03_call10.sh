#!/bin/bash
# round 3, call 10: the MFMA / VALU mix probe with v_mfma_f32_32x32x16_bf16 added, and the in-situ effect of the order-pinned attention body
# (lab build pipe_v2) on the whole aggregator forward at 8 and 64 views (control = the product sources through the same build path)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
( timeout 200 tools/probes/mfma_valu_mix ) > $O/r03_probe_mfma_valu_mix.txt 2>&1
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['views'], 'views:', d['value'], 'frames/s', d['ms_per_step'], 'ms  roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; }
B="--no-cpu-baseline --no-parity --no-secondary"
: > $O/r03_lab_insitu.txt
for n in control pipe_v2 control pipe_v2; do
  timeout 200 python tools/lab/bench_with_lab.py $n --views 8 --steps 20 --warmup 3 $B 2>/dev/null | line $n >> $O/r03_lab_insitu.txt
done
for n in control pipe_v2; do
  timeout 300 python tools/lab/bench_with_lab.py $n --views 64 --steps 5 --warmup 1 $B 2>/dev/null | line $n >> $O/r03_lab_insitu.txt
done
sed -n '/32x32x16/,$p' $O/r03_probe_mfma_valu_mix.txt; cat $O/r03_lab_insitu.txt
