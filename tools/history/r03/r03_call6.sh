#!/bin/bash
# round 3, call 6: f32 camera head on ovg_camera_head (exact-f32 MFMA weight streams) -- kernel test + timing first, then the whole GPU suite
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k camera_head -s > gpurun_out/r03_camhead_f32_test.txt 2>&1
echo "camera_head test rc=$?" | tee -a gpurun_out/r03_camhead_f32_test.txt
timeout 200 python tests/gpu_selftest.py camera 2>&1 | grep -i "camera head S=8\|FAIL\|Error" > gpurun_out/r03_camhead_f32_timing.txt
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r03_gpu_suite_call6.txt 2>&1
echo "suite rc=$?" >> gpurun_out/r03_gpu_suite_call6.txt
tail -15 gpurun_out/r03_gpu_suite_call6.txt
grep -c PASS gpurun_out/r03_camhead_f32_test.txt; grep "f32" gpurun_out/r03_camhead_f32_test.txt | tail -12; cat gpurun_out/r03_camhead_f32_timing.txt
