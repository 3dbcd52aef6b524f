#!/bin/bash
# Round-6 lab passes on the GPU box (run through gpurun; everything lands under gpurun_out/r06/):
#   tools/lab_r06.sh [gemm_parity] [gemm_ab] ...      env: OVG_LAB_TILES="10,42,74", OVG_AB_VIEWS="8 64", OVG_AB_SQUARE="4096 8192"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p "$O"
cd "$R"
TILES=${OVG_LAB_TILES:-10,42,74}
for stage in "$@"; do
  echo "=== stage $stage ($(date +%H:%M:%S))"
  case $stage in
    gemm_parity) (timeout 900 python tests/gpu_selftest.py --quick --only gemm256,linear,qkv,f32x 2>&1 | grep -v amdgpu.ids | tail -400) > "$O/gemm_lab_parity.log"; grep -E "FAIL|SELFTEST|Error|error" "$O/gemm_lab_parity.log" | head -20 ;;
    gemm_ab)     (timeout 1200 python tests/bench_kernels.py gemm --alt-lib lab --views ${OVG_AB_VIEWS:-8 64} --square ${OVG_AB_SQUARE:-4096 8192} --tiles 1 2 ${TILES//,/ } --rounds ${OVG_AB_ROUNDS:-5} 2>&1 | grep -v amdgpu.ids) | tee "$O/gemm_lab_ab.txt" | tail -80 ;;
    insitu)      # in-situ A/B of alternate builds: OVG_INSITU_LIBS="product fr1" OVG_INSITU_VIEWS="64 8" OVG_INSITU_REPS=2
      for v in ${OVG_INSITU_VIEWS:-64 8}; do for rep in $(seq 1 ${OVG_INSITU_REPS:-2}); do for lib in ${OVG_INSITU_LIBS:-product fr1}; do
        steps=$([ "$v" -ge 32 ] && echo 6 || echo 20)
        timeout 600 python tools/probes/run_with_lib.py $lib bench.py --views $v --steps $steps --warmup 3 --no-cpu-baseline --no-parity --no-e2e --no-secondary ${OVG_INSITU_ARGS:-} 2>"$O/insitu.err" | tail -1 \
          | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('insitu views $v lib $lib rep $rep: %.2f frames/s  %.3f ms/step  attn %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms', 0), d['roofline']['frac']))" || tail -5 "$O/insitu.err"
      done; done; done 2>&1 | tee -a "$O/insitu_ab.txt" ;;
    attn_tail_ab) # 512-row kernel: 128-row tail (variant 71) vs the plan's key-split tail (variant 0 with a workspace) vs forced key-range counts (74)
      (timeout 1200 python tests/bench_kernels.py attn --modes global --views ${OVG_AB_VIEWS:-16 24 32 48 64} --variants 71 0 --kv-splits 0 --rounds 3 --target-ms 60 2>&1 | grep -v amdgpu.ids
       timeout 900 python tests/bench_kernels.py attn --modes global --views ${OVG_TAIL_FORCED_VIEWS:-16 64} --variants 74 --kv-splits 2 3 4 5 6 7 8 --rounds 3 --target-ms 60 2>&1 | grep -v amdgpu.ids) | tee "$O/attn_tail_ab.txt" | tail -60 ;;
    bench_quick) timeout 900 python bench.py --no-cpu-baseline --no-parity --steps 6 --warmup 2 2>"$O/bench_quick.err" | tail -1 | tee "$O/bench_quick_line.json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames/s', d['value'], 'ms', d['ms_per_step'], 'attn ms', d['roofline'].get('avg_launch_ms'), 'frac', d['roofline']['frac'], 'fallback_wgs', d['roofline'].get('fallback_workgroups'), '| e2e', d.get('e2e',{}).get('frames_per_s'), '| S8', d.get('secondary',{}).get('frames_per_s'), d.get('secondary',{}).get('roofline',{}).get('frac'), 'e2e8', d.get('secondary',{}).get('e2e',{}).get('frames_per_s'))" || tail -20 "$O/bench_quick.err" ;;
    tests)      (timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) | tee "$O/gpu_tests.log" ;;
    x3_ab)       # split-f16 mode with fewer PV products (lab builds x3pv2 / x3pv1): distance from the f32 mode at layers 0/4/11/17/23 and its own rate
      for lib in ${OVG_X3_LIBS:-product x3pv2 x3pv1}; do
        timeout 900 python tools/probes/run_with_lib.py $lib bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>"$O/x3_ab.err" | tail -1 \
          | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['parity']; print('x3_ab lib $lib:', {k: (p[k]['f32x_mode']['frames_per_s'], p[k]['f32x_mode']['max_rel_vs_f32_mode']) for k in p if isinstance(p[k], dict) and 'f32x_mode' in p[k]})" || tail -5 "$O/x3_ab.err"
      done 2>&1 | tee "$O/x3_ab.txt" ;;
    selftest)    (timeout 1500 python tests/gpu_selftest.py --only ${OVG_SELFTEST_ONLY:-heads,f32x} 2>&1 | grep -v amdgpu.ids | tail -400) > "$O/selftest_${OVG_SELFTEST_TAG:-x}.log"; grep -E "FAIL|SELFTEST|Error|error" "$O/selftest_${OVG_SELFTEST_TAG:-x}.log" | head -20 ;;
    gemm_rotate) # 128^2 vs 256^2 at 8 views with the weights resident (the usual microbench) and streamed from HBM (48 copies in turn: the in-situ condition)
      for rot in 1 48; do echo "--- rotate $rot"; timeout 900 python tests/bench_kernels.py gemm --views ${OVG_AB_VIEWS:-8} --tiles 1 2 --rotate $rot --rounds 5 2>&1 | grep -v amdgpu.ids; done | tee "$O/gemm_rotate_ab.txt" ;;
    rank_probe)  (for cfg in "8 2 8" "4 4 16" "2 4 32"; do echo "--- ranks / heads per launch / views per rank: $cfg"; timeout 600 python tools/probes/attn_rank_shape_probe.py $cfg 2>&1 | grep -v amdgpu.ids | tail -10; done) | tee "$O/attention_rank_shapes.txt" ;;
    insitu_e2e)  # end-to-end (aggregator + three heads) A/B of alternate builds: OVG_INSITU_LIBS="product cvA" OVG_INSITU_VIEWS="8 64"
      for v in ${OVG_INSITU_VIEWS:-8}; do for rep in $(seq 1 ${OVG_INSITU_REPS:-2}); do for lib in ${OVG_INSITU_LIBS:-product cvA}; do
        steps=$([ "$v" -ge 32 ] && echo 4 || echo 12)
        timeout 600 python tools/probes/run_with_lib.py $lib bench.py --views $v --steps $steps --warmup 2 --no-cpu-baseline --no-parity --no-secondary 2>"$O/insitu.err" | tail -1 \
          | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d.get('e2e',{}); print('e2e views $v lib $lib rep $rep: aggregator %.2f frames/s | e2e %.2f frames/s  %.3f ms/forward' % (d['value'], e.get('frames_per_s', 0), e.get('ms_per_forward', 0)))" || tail -5 "$O/insitu.err"
      done; done; done 2>&1 | tee -a "$O/insitu_e2e_ab.txt" ;;
    # (gemm128f_ab -- tile 33, the 128 x 128 free-running lab kernel -- ran from commit 'Review item 1c measured'; its kernel left the tree with that commit)
    *) echo "unknown stage $stage" ;;
  esac
done
