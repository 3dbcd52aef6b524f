#!/bin/bash
# Round-3 closing pass on the FINAL tree: the complete -m gpu suite, smoke(), the default bench line, a view-count sweep and the per-rank
# attention shapes of the 2 / 4 / 8-GPU runs timed on one GPU (inputs of the projection in DESIGN section 6).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
(time python -m pytest tests -m gpu -q 2>&1 | tail -6) > "$O/r03_gputest_closing.log" 2>&1
cat "$O/r03_gputest_closing.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/r03_smoke.log" 2>&1; tail -3 "$O/r03_smoke.log"
python bench.py > "$O/r03_bench_default_line_closing.json" 2> "$O/r03_bench_default_closing.err"
python - "$O/r03_bench_default_line_closing.json" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("default bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], "| secondary", d["secondary"]["frames_per_s"], d["secondary"]["roofline"]["frac"], "| cpu", d["cpu_baseline"]["value"])
PY
{
echo "# Single-GPU sweep over the view count, final tree of round 3: python bench.py --views S --steps 5 --warmup 2 --no-cpu-baseline --no-parity (bf16, images-only, one box)"
echo "#   views   frames/s   ms/forward   whole-forward TFLOP/s   global attention ms (HIP events)   fraction of the 2.5 PFLOP/s MFMA peak"
for v in 4 8 9 10 12 16 24 32 48 64; do
  python bench.py --views $v --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('views $v', d['value'], d['ms_per_step'], d['tflops_per_gpu'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
} > "$O/r03_bench_view_sweep.txt" 2>&1
cat "$O/r03_bench_view_sweep.txt"
{
echo "# Per-rank global-attention launch of the view-sharded 64-view run (head-parallel form), timed on ONE GPU (tools/probes/attn_rank_shape_probe.py):"
echo "# == 8 ranks (16 entries = 8 sources x 2 heads, 8 views of queries, 8 segments of 8 views of keys)"
python tools/probes/attn_rank_shape_probe.py 8 2 8
echo "# == 4 ranks (16 entries = 4 sources x 4 heads, 16 views of queries)"
python tools/probes/attn_rank_shape_probe.py 4 4 16
echo "# == 2 ranks, one of two head groups (8 entries = 2 sources x 4 heads, 32 views of queries)"
python tools/probes/attn_rank_shape_probe.py 2 4 32
} > "$O/r03_attention_rank_shapes.txt" 2>&1
grep -v amdgpu "$O/r03_attention_rank_shapes.txt"
