#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03i"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd or trunk_stage or golden or knobs" > "$O/pytest.log" 2>&1; tail -4 "$O/pytest.log"
FFR_BENCH_LIVE_PMC=0 timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc $?"
timeout 300 python3 tools/wf_trace.py 2>&1 | grep "wf trace" | grep "transform in" > "$O/wf_trace.txt"
python3 - <<PY
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['parity_checked']['max_rel_err_vs_reference_golden_G1'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
PY
cut -c1-300 "$O/wf_trace.txt" | head -14
