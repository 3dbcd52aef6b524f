#!/bin/bash
# Bench lines of the BASELINE configs other than the default one, final tree of round 2 (through gpurun).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
python bench.py --views 16 --aux --steps 10 --warmup 3 --no-cpu-baseline --no-parity > "$O/r02_bench_config2_S16_aux.json" 2>/dev/null
python bench.py --views 128 --dtype f16 --partial-aux --steps 3 --warmup 1 --no-cpu-baseline --no-parity > "$O/r02_bench_config4_S128_f16_partial_aux.json" 2>/dev/null
python bench.py --views 64 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --e2e --e2e-views 64 > "$O/r02_bench_e2e_S64.json" 2>/dev/null
python bench.py --views 8 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --e2e > "$O/r02_bench_e2e_S8.json" 2>/dev/null
for f in r02_bench_config2_S16_aux r02_bench_config4_S128_f16_partial_aux r02_bench_e2e_S64 r02_bench_e2e_S8; do
  python - "$O/$f.json" <<'EOF'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("e2e"))
EOF
done
