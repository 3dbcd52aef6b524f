#!/bin/bash
# Run on tree 07381af; provenance of profiles/r03_f32_e2e_hip_vs_pytorch_heads.txt.
# Round-3 GPU pass 5: the DPT heads of the f32 parity mode on the HIP f32 kernels -- kernel / whole-head parity, the f32 end-to-end tests
# against the oracle and the reference goldens, and the f32 end-to-end forward timed with HIP and with PyTorch (MIOpen) heads.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
(time python -m pytest tests/test_gpu_kernels.py tests/test_gpu_aggregator.py tests/test_real_inputs.py -m gpu -q -rA -k "dpt_head or end_to_end or predictions_vs_reference or other_resolutions or real_inputs or integration" 2>&1 | grep -v "^PASSED\|^$" | tail -40) > "$O/r03_call5_tests.log" 2>&1
cat "$O/r03_call5_tests.log"
python bench.py --dtype f32 --views 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 8 views HIP f32 DPT heads:', d['value'], d['ms_per_step'], d.get('e2e'), d.get('e2e_error'))" > "$O/r03_f32_e2e.txt" 2>&1
python bench.py --dtype f32 --views 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --e2e --torch-heads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 8 views PyTorch heads      :', d['value'], d['ms_per_step'], d.get('e2e'), d.get('e2e_error'))" >> "$O/r03_f32_e2e.txt" 2>&1
cat "$O/r03_f32_e2e.txt"
