#!/bin/bash
# round 3, call 16: pre-validation of the order-pinned attention body (lab build pipe_v2) on the attention-related GPU tests
# (kernel selftests incl. ragged segments / rescale / forced fallback / split-KV / LSE merge, block forward, emulated 2- and 8-rank sharding, forced split-KV)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 215 python tools/lab/pytest_with_lab.py pipe_v2 tests/test_gpu_kernels.py tests/test_gpu_sharded.py tests/test_gpu_aggregator.py -m gpu -q -x \
  -k "flash_attention or global_attention or lse_output or block_forward or two_uneven or two_even or eight_ranks or forced_split" -p no:cacheprovider > $O/r03_lab_pipe_v2_gpu_tests.txt 2>&1
echo "rc=$?" >> $O/r03_lab_pipe_v2_gpu_tests.txt
tail -8 $O/r03_lab_pipe_v2_gpu_tests.txt
