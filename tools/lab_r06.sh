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
    *) echo "unknown stage $stage" ;;
  esac
done
