#!/bin/bash
# Round-3 validation + profiling pass on the GPU box (through gpurun): the -m gpu suite, the default bench line, the other BASELINE configs,
# per-shape rocprofv3 kernel stats of bench.py (64 and 8 views in SEPARATE runs) and PMC passes on the shipped attention / GEMM launches
# (counters in their own runs, only --kernel-trace next to --pmc, one SQ / TCC set per pass), from which profiles/traffic.json is rewritten
# with the digest of the attention sources it was measured on.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r03
mkdir -p "$P"
cd "$R"
(time python -m pytest tests -m gpu -q -rA 2>&1 | grep -v "^PASSED\|^$") > "$O/r03_gputest_final.log" 2>&1
tail -6 "$O/r03_gputest_final.log"
python bench.py > "$O/r03_bench_default_line.json" 2> "$O/r03_bench_default.err"
tail -c 400 "$O/r03_bench_default_line.json"; echo
python bench.py --views 16 --aux --steps 10 --warmup 3 --no-cpu-baseline > "$O/r03_bench_config2_S16_aux.json" 2>/dev/null
python bench.py --views 128 --dtype f16 --partial-aux --steps 3 --warmup 1 --no-cpu-baseline > "$O/r03_bench_config4_S128_f16_partial_aux.json" 2>/dev/null
python bench.py --dtype f32 --views 64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$O/r03_bench_f32_S64.json" 2>/dev/null
python bench.py --views 8 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --e2e > "$O/r03_bench_e2e_S8.json" 2>/dev/null
python bench.py --views 64 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --e2e --e2e-views 64 > "$O/r03_bench_e2e_S64.json" 2>/dev/null
for f in r03_bench_config2_S16_aux r03_bench_config4_S128_f16_partial_aux r03_bench_f32_S64 r03_bench_e2e_S8 r03_bench_e2e_S64; do
  python - "$O/$f.json" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    par = d.get("parity", {})
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("e2e"), {k: (v.get("max_rel") if isinstance(v, dict) else None) for k, v in par.items() if k.startswith("S")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*"; "$@" > "$P/last.log" 2>&1 || { echo "   FAILED rc=$?"; tail -5 "$P/last.log"; }; }
run rocprofv3 --kernel-trace --stats --output-format csv -d "$P/bench_s64" -- python "$R/bench.py" --views 64 --steps 3 --warmup 1 --no-cpu-baseline --no-parity
run rocprofv3 --kernel-trace --stats --output-format csv -d "$P/bench_s8" -- python "$R/bench.py" --views 8 --steps 5 --warmup 2 --no-cpu-baseline --no-parity
GEMM="python $R/tests/bench_kernels.py gemm --views 64 --tiles 0 --rounds 1 --target-ms 5"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i + 1))
  for v in 8 64; do
    run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/attn_S${v}_pmc$i" -- python "$R/tests/bench_kernels.py" attn --modes global --views $v --variants 0 --rounds 1 --target-ms 60
  done
  run rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/gemm_pmc$i" -- $GEMM
done
python "$R/tools/pmc_summary.py" "$P"/attn_S* > "$O/r03_pmc_attention.txt" 2>&1
python "$R/tools/pmc_summary.py" "$P"/gemm_pmc* > "$O/r03_pmc_gemm.txt" 2>&1
cd "$R" && python tools/traffic_json.py --views 8 "$P"/attn_S8_pmc3 "$P"/attn_S8_pmc4 --views 64 "$P"/attn_S64_pmc3 "$P"/attn_S64_pmc4 --out "$O/traffic.json" \
  --source "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum / WRITE_SIZE TCC_MISS_sum passes (tools/validate_r03.sh) of the shipped global-attention launches; PMC counters cannot be read from inside bench.py, so the figure is not re-measured in the bench run" > "$O/traffic_json.log" 2>&1
tail -3 "$O/traffic_json.log"
for d in bench_s64 bench_s8; do
  f=$(find "$P/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$O/r03_${d}_kernel_stats.csv"
done
find "$P" -name "*.csv" -size +1M -delete
du -sh "$P" | tail -1
