#!/bin/bash
# round 3, call 7: camera head timing (hip bf16 / hip f32 / pytorch f32 at 8 views) and the f32 end-to-end line with all three heads on HIP
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 300 python tests/gpu_selftest.py --only camera 2>&1 | grep -i "camera head S=8\|FAIL\|SELFTEST" > $O/r03_camhead_f32_timing.txt
timeout 300 python bench.py --dtype f32 --views 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 8 views, all three heads on HIP f32:', d['value'], d['ms_per_step'], d.get('e2e'), d.get('e2e_error'))" > $O/r03_f32_e2e_all_hip.txt 2>&1
cat $O/r03_camhead_f32_timing.txt $O/r03_f32_e2e_all_hip.txt
