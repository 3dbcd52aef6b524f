#!/bin/bash
# round 3, call 18: in-situ effect of the promotion candidate (lab build pipe_v2_q2: pinned order on the 512- / 256- / 128-row kernels) at 64 views
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['views'], 'views:', d['value'], 'frames/s', d['ms_per_step'], 'ms  roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; }
B="--no-cpu-baseline --no-parity --no-secondary"
: > $O/r03_lab_insitu_q2.txt
for n in control pipe_v2_q2; do
  timeout 60 python tools/lab/bench_with_lab.py $n --views 64 --steps 4 --warmup 1 $B 2>/dev/null | line $n >> $O/r03_lab_insitu_q2.txt
done
cat $O/r03_lab_insitu_q2.txt
