#!/bin/bash
# After a change to the attention sources: re-check the attention launches and re-take the two traffic passes that profiles/traffic.json is built from.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r03t
mkdir -p "$P"
cd "$R"
(python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sharded.py tests/test_gpu_aggregator.py -m gpu -q -k "attention or eight_ranks or headline or forced_split or baseline_view_counts" 2>&1 | tail -4) > "$O/r03_retake_tests.log" 2>&1
cat "$O/r03_retake_tests.log"
cd /tmp && export TMPDIR=/tmp
i=2
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i + 1))
  for v in 8 64; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/attn_S${v}_pmc$i" -- python "$R/tests/bench_kernels.py" attn --modes global --views $v --variants 0 --rounds 1 --target-ms 60 > "$P/last.log" 2>&1 || { echo "pass FAILED"; tail -5 "$P/last.log"; }
  done
done
cd "$R" && python tools/traffic_json.py --views 8 "$P"/attn_S8_pmc3 "$P"/attn_S8_pmc4 --views 64 "$P"/attn_S64_pmc3 "$P"/attn_S64_pmc4 --out "$O/traffic.json" \
  --source "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum / WRITE_SIZE TCC_MISS_sum passes (tools/retake_traffic_r03.sh) of the shipped global-attention launches; PMC counters cannot be read from inside bench.py, so the figure is not re-measured in the bench run" > "$O/traffic_json.log" 2>&1
python -c "import json; d=json.load(open('$O/traffic.json')); print(d['attention_source_digest'][:12], d['global_attn_S64_bytes_per_launch'], d['global_attn_S8_bytes_per_launch'], d['global_attn_S64_dispatches'])"
python bench.py --views 64 --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['kernel'][:260])"
find "$P" -name "*.csv" -size +1M -delete
