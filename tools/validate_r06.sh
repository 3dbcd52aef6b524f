#!/bin/bash
# Round-6 GPU passes, one script with selectable stages (run through gpurun; everything lands under gpurun_out/r05/):
#   tools/validate_r06.sh [tests] [attn_tests] [bench] [multirank] [traffic] [prof64] [prof8] [f32x] ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06v
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
    f32x_kernels) (timeout 900 python tests/gpu_selftest.py --only f32x 2>&1 | tail -120) | tee "$O/f32x_selftest.log" | grep -E "FAIL|SELFTEST|Error|error" | head -40 ;;
    f32x_agg)   (timeout 1500 python -m pytest tests/test_gpu_aggregator.py -m gpu -q -s -k "f32x" 2>&1 | grep -E "max-rel|passed|failed|Error|error|f32x full" | tail -40) | tee "$O/f32x_agg.log" ;;
    bench_f32x) timeout 1200 python bench.py --dtype f32x --steps 3 --warmup 1 --no-cpu-baseline 2>"$O/bench_f32x.err" | tail -1 | tee "$O/bench_f32x_line.json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32x frames/s', d['value'], 'ms', d['ms_per_step'], 'attn frac', d['roofline']['frac'], 'achieved', d['roofline']['achieved'], '| S8', d.get('secondary',{}).get('frames_per_s'), '| parity', {k: v.get('max_rel') for k, v in d.get('parity',{}).items() if isinstance(v, dict)})" || tail -20 "$O/bench_f32x.err" ;;
    bench_f32)  timeout 1200 python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>"$O/bench_f32.err" | tail -1 | tee "$O/bench_f32_line.json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 frames/s', d['value'], 'ms', d['ms_per_step'], 'attn frac', d['roofline']['frac'])" ;;
    rank_probe) (timeout 600 python tools/probes/attn_rank_shape_probe.py 2>&1 | tail -9) | tee "$O/attention_rank_shapes.txt" ;;
    traffic_only)
      P=$O/prof_traffic; mkdir -p "$P"
      (cd /tmp && export TMPDIR=/tmp && i=2 && for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do i=$((i + 1)); for v in 8 64; do
          rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/attn_S${v}_pmc$i" -- python "$R/tests/bench_kernels.py" attn --modes global --views $v --variants 0 --kv-splits 0 --rounds 1 --target-ms 60 > "$P/last.log" 2>&1 || { echo "pass FAILED"; tail -5 "$P/last.log"; }
        done; done)
      python tools/traffic_json.py --views 8 "$P"/attn_S8_pmc3 "$P"/attn_S8_pmc4 --views 64 "$P"/attn_S64_pmc3 "$P"/attn_S64_pmc4 --out "$O/traffic.json" \
        --source "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum / WRITE_SIZE TCC_MISS_sum passes (tools/validate_r06.sh traffic_only) of the shipped global-attention launches; PMC counters cannot be read from inside bench.py, so the figure is not re-measured in the bench run" > "$O/traffic_json.log" 2>&1
      python -c "import json; d=json.load(open('$O/traffic.json')); print('traffic', d['attention_source_digest'][:12], d['global_attn_S64_bytes_per_launch'], d['global_attn_S8_bytes_per_launch'])"
      cp "$O/traffic.json" "$R/profiles/traffic.json"       # the bench stage of the same pass reads the record taken on THIS tree
      find "$P" -name "*.csv" -size +1M -delete ;;
    attn_ab)    (timeout 900 python tools/probes/attn_ab_probe.py ${OVG_AB_ARGS:-} 2>&1 | grep -v amdgpu.ids | tail -30) | tee "$O/attn_ab.txt" ;;
    configs)    # the other BASELINE configs + end-to-end lines (aggregator + three heads), one JSON line each
      run() { name=$1; shift; timeout 900 python bench.py "$@" --no-cpu-baseline 2>"$O/$name.err" | tail -1 > "$O/$name.json"; python -c "import sys,json; d=json.load(open('$O/$name.json')); print('$name', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'], 'e2e', d.get('e2e'), 'parity', {k: max(v['max_rel']) for k, v in d.get('parity', {}).items() if isinstance(v, dict) and 'max_rel' in v})" || tail -5 "$O/$name.err"; }
      run bench_config2_S16_aux --views 16 --aux --steps 10 --warmup 2
      run bench_config4_S128_f16_partial_aux --views 128 --dtype f16 --partial-aux --steps 3 --warmup 1
      run bench_e2e_S8 --views 8 --steps 10 --warmup 2 --no-parity --e2e --e2e-views 8
      run bench_e2e_S64 --views 64 --steps 4 --warmup 1 --no-parity --e2e --e2e-views 64
      run bench_f32x_e2e_S8 --dtype f32x --views 8 --steps 5 --warmup 1 --no-parity --e2e --e2e-views 8
      run bench_f32_S64 --dtype f32 --views 64 --steps 2 --warmup 1 --no-parity ;;
    printed)    (timeout 1500 python -m pytest tests/test_gpu_aggregator.py tests/test_gpu_sharded.py tests/test_gpu_kernels.py -m gpu -q -s \
                   -k "full_depth_8_views or full_depth_16_views or batch_of_two_scenes_full or attention_sinks or eight_ranks_allgather or headline_64 or stress_128" 2>&1 | grep -E "vs oracle|re-ran|emulated ranks|passed|failed|Error" | cut -c1-400) | tee "$O/printed_parity_numbers.txt" ;;
    multirank_f32x) OVG_MULTIRANK_CFGS="2:8" OVG_MULTIRANK_ARGS="--dtype f32x --no-second-form" bash tools/multirank_one_gpu.sh 2>&1 | tee "$O/multirank_one_gpu_gloo_f32x.txt" ;;
    heads_dtype) (timeout 900 python tools/probes/heads_dtype_probe.py 2>&1 | grep -v amdgpu.ids | tail -20) | tee "$O/heads_dtype_probe.txt" ;;
    pmc)        # SQ / GRBM counter passes (own runs, only --kernel-trace next to --pmc): shipped bf16 attention and GEMM launches, split-f16 forward
      P=$O/prof_pmc; mkdir -p "$P"
      (cd /tmp && export TMPDIR=/tmp
       run() { "$@" > "$P/last.log" 2>&1 || { echo "   FAILED: $*"; tail -4 "$P/last.log"; }; }
       i=0
       for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
                  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
         i=$((i + 1))
         for v in 8 64; do
           run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/attn_S${v}_pmc$i" -- python "$R/tests/bench_kernels.py" attn --modes global --views $v --variants 0 --kv-splits 0 --rounds 1 --target-ms 60
           run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/gemm_S${v}_pmc$i" -- python "$R/tests/bench_kernels.py" gemm --views $v --tiles 0 --rounds 1 --target-ms 5
           run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/f32x_S${v}_pmc$i" -- python "$R/bench.py" --dtype f32x --views $v --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-secondary
         done
       done)
      python tools/pmc_summary.py "$P"/attn_S* > "$O/pmc_attention.txt" 2>&1
      python tools/pmc_summary.py "$P"/gemm_S64_* > "$O/pmc_gemm.txt" 2>&1
      python tools/pmc_summary.py "$P"/gemm_S8_* > "$O/pmc_gemm_S8.txt" 2>&1
      python tools/pmc_summary.py "$P"/f32x_S64_* --only attn16_kernel --only linear --only qkv > "$O/pmc_f32x_S64.txt" 2>&1
      python tools/pmc_summary.py "$P"/f32x_S8_* --only attn16_kernel --only linear --only qkv > "$O/pmc_f32x_S8.txt" 2>&1
      grep -h "grid=\|matrix pipe\|effective clock" "$O/pmc_attention.txt" "$O/pmc_f32x_S64.txt" | cut -c1-170 | head -40
      find "$P" -name "*.csv" -size +1M -delete ;;
    prof_f32x)  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_f32x_S64" -- python "$R/bench.py" --dtype f32x --views 64 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > "$O/prof_f32x_S64.log" 2>&1)
      f=$(find "$O/prof_f32x_S64" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/bench_f32x_S64_kernel_stats.csv" && head -9 "$f" | cut -c1-200
      find "$O/prof_f32x_S64" -name "*.csv" -size +1M -delete ;;
    sweep)      for v in 4 12 16 24 32 48; do
        timeout 600 python bench.py --views $v --steps 6 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views', d['config']['views'], 'frames/s', d['value'], 'ms', d['ms_per_step'], 'attention ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'fallback', d['roofline']['fallback_workgroups'])"
      done 2>&1 | tee "$O/bench_view_sweep.txt" ;;
    attn_st)    (timeout 900 python tests/gpu_selftest.py --only attn,attn_big,lse_merge,fallback 2>&1 | grep -v amdgpu.ids | grep -E "FAIL|SELFTEST|Error|error|Traceback|raise|fault|core|keytail|\(attn" | head -80) | tee "$O/attn_selftest.txt" ;;
    attn_ab8)   (timeout 900 python tests/bench_kernels.py attn --modes global --views ${OVG_AB_VIEWS:-8 9 10 12 13} --variants ${OVG_AB_VARIANTS:-0 50} --kv-splits ${OVG_AB_SPLITS:-0 1} --rounds 4 --target-ms 30 2>&1 | grep -v amdgpu.ids | tail -40) | tee "$O/attn_keytail_ab.txt" ;;
    ckpt)       (timeout 1500 python tools/validate_checkpoint.py --synthetic /tmp/ovg_synth_ckpt.safetensors --views 2 8 --aux --out "$O/checkpoint_rehearsal.json" 2>&1 | grep -v amdgpu.ids | tail -60) | tee "$O/checkpoint_rehearsal.txt"; rm -f /tmp/ovg_synth_ckpt.safetensors ;;
    heads_st)   (timeout 900 python tests/gpu_selftest.py --only heads 2>&1 | grep -v amdgpu.ids | grep -E "FAIL|SELFTEST|Error|error|Traceback|raise|fault|core|c256|dpt_tail|dpt_head|\(heads" | head -90) | tee "$O/heads_selftest.txt" ;;
    e2e)        for v in 8 64; do timeout 900 python bench.py --views $v --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views', d['config']['views'], 'aggregator frames/s', d['value'], 'e2e', d.get('e2e'), d.get('e2e_error'))"; done 2>&1 | tee "$O/bench_e2e.txt" ;;
    prof_e2e8|prof_e2e64)
      ev=${stage#prof_e2e}
      (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_e2e_S$ev" -- python "$R/bench.py" --views $ev --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > "$O/prof_e2e_S$ev.log" 2>&1)
      f=$(find "$O/prof_e2e_S$ev" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/bench_e2e_S${ev}_kernel_stats.csv" && head -16 "$f" | cut -c1-200
      t=$(find "$O/prof_e2e_S$ev" -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python - "$t" <<'PY' | tee "$O/conv_launches_S$ev.txt"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "conv" in n or "upsample" in n or "dpt_out" in n or "dpt_tail" in n or "head_layernorm" in n:
        agg[(n.split("(")[0][-60:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s grid %-9s calls %3d  avg %9.1f us  total %9.1f us  %5.1f %%" % (k[0], k[1], len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
PY
      find "$O/prof_e2e_S$ev" -name "*.csv" -size +1M -delete ;;
    pmc_lite)   # SQ / GRBM counter passes on the shipped bf16 attention and GEMM launches (own runs, only --kernel-trace next to --pmc)
      P=$O/prof_pmc; mkdir -p "$P"
      (cd /tmp && export TMPDIR=/tmp
       run() { "$@" > "$P/last.log" 2>&1 || { echo "   FAILED: $*"; tail -4 "$P/last.log"; }; }
       i=0
       for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
                  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
         i=$((i + 1))
         for v in 8 64; do
           run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/attn_S${v}_pmc$i" -- python "$R/tests/bench_kernels.py" attn --modes global --views $v --variants 0 --kv-splits 0 --rounds 1 --target-ms 60
           run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/gemm_S${v}_pmc$i" -- python "$R/tests/bench_kernels.py" gemm --views $v --tiles 0 --rounds 1 --target-ms 5
         done
       done)
      python tools/pmc_summary.py "$P"/attn_S* > "$O/pmc_attention.txt" 2>&1
      python tools/pmc_summary.py "$P"/gemm_S64_* > "$O/pmc_gemm.txt" 2>&1
      python tools/pmc_summary.py "$P"/gemm_S8_* > "$O/pmc_gemm_S8.txt" 2>&1
      grep -h "grid=\|matrix pipe\|effective clock" "$O/pmc_attention.txt" "$O/pmc_gemm_S8.txt" | cut -c1-170 | head -40
      find "$P" -name "*.csv" -size +1M -delete ;;
    gemm_tl)    (timeout 900 python tools/probes/gemm_timeline.py ${OVG_TL_ARGS:-} 2>&1 | grep -v amdgpu.ids | tail -80) | tee "$O/gemm_timeline.txt" ;;
    gemm_ab)    (timeout 900 python tests/bench_kernels.py gemm ${OVG_GEMM_AB_ARGS:---views 8 16 64 --tiles 1 2 --rounds 3} 2>&1 | grep -v amdgpu.ids | tail -80) | tee "$O/gemm_ab.txt" ;;
    *) echo "unknown stage $stage" ;;
  esac
done
