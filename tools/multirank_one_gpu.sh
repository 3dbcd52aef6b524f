#!/bin/bash
# The N > 1 path of bench.py on ONE GPU: N processes sharing the device (OVG_FORCE_DEVICE=0), gloo backend (RCCL rejects duplicate GPUs),
# collectives staged through the host by ViewSharding. Throughput is meaningless; it checks the multi-process control flow of both exchange
# forms (pre-flight comparison, pipelined head groups, local-first all-gather + merge, split-KV per-rank launches, watchdog heartbeat) on the
# HIP kernels at 2, 4 and 8 ranks, even and uneven view counts.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
export OVG_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in ${OVG_MULTIRANK_CFGS:-"2:8" "4:16" "8:16" "8:20"}; do
  n=${cfg%%:*}; v=${cfg##*:}
  # `python bench.py --gpus N` by itself: the script re-executes under torch.distributed.run (bench.self_launch, r04)
  out=$(env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout 900 python bench.py --gpus $n --backend gloo --views $v --steps 2 --warmup 1 ${OVG_MULTIRANK_ARGS:-} 2>/tmp/multirank_${n}_${v}.err | tail -1)
  rc=$?
  echo "ranks=$n views=$v rc=$rc $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], d['config']['parallelism'], '| frames/s', d['value'], '| comm', d.get('comm'), '| preflight', d.get('preflight'), '| second_form', (d.get('second_form') or {}).get('parallelism', (d.get('second_form') or {}).get('skipped')))" 2>&1)"
  grep -c "rank" /tmp/multirank_${n}_${v}.err | sed "s/^/  rank-tagged stderr lines: /"
  grep -i "watchdog\|error\|Traceback" /tmp/multirank_${n}_${v}.err | head -5
done
