#!/bin/bash
# Round-4 GPU passes, one script with selectable stages (run through gpurun; everything lands under gpurun_out/r04/):
#   tools/validate_r04.sh [tests] [attn_tests] [bench] [multirank] [traffic] [prof64] [prof8] [f32x] ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p "$O"
cd "$R"
for stage in "$@"; do
  echo "=== stage $stage ($(date +%H:%M:%S))"
  case $stage in
    tests)      (timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) | tee "$O/gpu_tests.log" ;;
    attn_tests) (timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sharded.py tests/test_gpu_aggregator.py -m gpu -q -x \
                   -k "attention or attn or eight_ranks or headline or forced_split or baseline_view_counts or block" 2>&1 | tail -8) | tee "$O/attn_tests.log" ;;
    smoke)      (timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3) | tee "$O/smoke.log" ;;
    bench)      timeout 900 python bench.py 2>"$O/bench_default.err" | tail -1 | tee "$O/bench_default_line.json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames/s', d['value'], 'frac', d['roofline']['frac'], 'fallback_wgs', d['roofline'].get('fallback_workgroups'), 'traffic', d['roofline'].get('traffic'), '| S8', d.get('secondary',{}).get('frames_per_s'), d.get('secondary',{}).get('roofline',{}).get('frac'))" ;;
    bench_quick) timeout 600 python bench.py --no-cpu-baseline --no-parity --steps 6 --warmup 2 2>"$O/bench_quick.err" | tail -1 | tee "$O/bench_quick_line.json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames/s', d['value'], 'frac', d['roofline']['frac'], 'fallback_wgs', d['roofline'].get('fallback_workgroups'), '| S8', d.get('secondary',{}).get('frames_per_s'), d.get('secondary',{}).get('roofline',{}).get('frac'))" ;;
    multirank)  OVG_MULTIRANK_CFGS="${OVG_MULTIRANK_CFGS:-2:8 8:16}" bash tools/multirank_one_gpu.sh 2>&1 | tee "$O/multirank_one_gpu_gloo.txt" ;;
    traffic)    bash tools/retake_traffic_r03.sh 2>&1 | tail -6 | tee "$O/traffic_retake.log"; cp "$R/gpurun_out/traffic.json" "$O/traffic.json" 2>/dev/null ;;
    prof64|prof8)
      v=${stage#prof}
      (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_S$v" -- python "$R/bench.py" --views $v --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > "$O/prof_S$v.log" 2>&1)
      f=$(find "$O/prof_S$v" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/bench_S${v}_kernel_stats.csv" && head -12 "$f" | cut -c1-200
      find "$O/prof_S$v" -name "*.csv" -size +1M -delete ;;
    *) echo "unknown stage $stage" ;;
  esac
done
