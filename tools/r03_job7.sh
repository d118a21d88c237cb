#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03g"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1800 python3 -m pytest tests -m gpu -x -q > "$O/pytest_all.log" 2>&1; tail -6 "$O/pytest_all.log"
timeout 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python3 examples/verify_synthetic.py 2>&1 | tail -3
timeout 600 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc $?"; python3 -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
print(d['secondary'][-1])
print(d['cpu_baseline'])"
timeout 600 python3 bench.py --workload train > "$O/train.json" 2> "$O/train.err"; tail -c 1500 "$O/train.json"
