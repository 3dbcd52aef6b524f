#!/bin/bash
# Last pass of round 3 on the final tree (after the launch-plan refactor): complete -m gpu suite, smoke(), default bench line.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd "$R"
(time python -m pytest tests -m gpu -q 2>&1 | tail -5) > "$O/r03_gputest_closing2.log" 2>&1
cat "$O/r03_gputest_closing2.log"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > "$O/r03_bench_default_line_closing2.json" 2>/dev/null
python - "$O/r03_bench_default_line_closing2.json" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("default bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_source", "")[:60], "| secondary", d["secondary"]["frames_per_s"], d["secondary"]["roofline"]["frac"], d["secondary"]["roofline"]["traffic"], "| cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
print(d["roofline"]["kernel"])
PY
