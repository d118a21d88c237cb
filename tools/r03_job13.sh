#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03m"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trunk or uint8 or golden or embed or variants" > "$O/pytest.log" 2>&1; tail -5 "$O/pytest.log"
FFR_BENCH_LIVE_PMC=0 timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$O/bench.json" 2> "$O/bench.err"; python3 -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['parity_checked'], d['roofline']['kernel_ms_per_step'])"
