#!/bin/bash
# The N > 1 path of bench.py on ONE GPU: N processes sharing the device (OVG_FORCE_DEVICE=0), gloo backend (RCCL rejects duplicate GPUs),
# collectives staged through the host by ViewSharding. Throughput is meaningless; it checks the multi-process control flow of both exchange
# forms (incl. the pipelined head groups, the local-first all-gather + merge, split-KV per-rank launches) on the HIP kernels.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
export OVG_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "2 8" "2 7" "4 16"; do
  set -- $cfg
  out=$(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $1 --backend gloo --views $2 --steps 2 --warmup 1 2>/dev/null | tail -1)
  rc=$?
  echo "ranks=$1 views=$2 rc=$rc $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['parallelism'], '| frames/s', d['value'], '| comm', d.get('comm',{}).get('exchange_form'), '| second_form', (d.get('second_form') or {}).get('parallelism'), '| forms_agree', (d.get('second_form') or {}).get('forms_agree'))" 2>&1)"
done
