#!/bin/bash
# Round-2 validation pass on the GPU box (through gpurun): the -m gpu suite, the default bench line, and per-shape
# rocprofv3 kernel stats of bench.py (64 views and 8 views in SEPARATE runs) + of the end-to-end forward with the HIP heads.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O/prof"
cd "$R"
(time python -m pytest tests -m gpu -q) > "$O/gputest.log" 2>&1
tail -4 "$O/gputest.log"
python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
tail -c 600 "$O/bench_default.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof/s64" -- python "$R/bench.py" --views 64 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > "$O/prof_s64.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof/s8" -- python "$R/bench.py" --views 8 --steps 5 --warmup 2 --no-cpu-baseline --no-parity > "$O/prof_s8.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof/e2e8" -- python "$R/bench.py" --views 8 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --e2e > "$O/prof_e2e8.log" 2>&1
for d in s64 s8 e2e8; do
  f=$(find "$O/prof/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$O/r02_bench_${d}_kernel_stats.csv"
done
find "$O/prof" -name "*.csv" -size +1M -delete
ls -la "$O" | tail -15
