#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03h"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd or trunk_stage or golden" > "$O/pytest.log" 2>&1; tail -4 "$O/pytest.log"
for o in 1 0; do
  timeout 300 python3 bench.py --opt wf_wide=$o --no-cpu-baseline --no-secondary > "$O/bench_w$o.json" 2> "$O/bench_w$o.err"; echo "bench w$o rc $?"
  FFR_OPT_WF_WIDE=$o timeout 300 python3 tools/wf_trace.py 2>&1 | grep "wf trace" | grep "transform in" > "$O/wf_trace_w$o.txt"
done
python3 - <<PY
import json
for o in (1,0):
    try:
        d=json.loads([l for l in open('$O/bench_w%d.json'%o) if l.startswith('{')][-1])
        print('wide',o, d['value'], d['ms_per_step'], d['parity_checked']['max_rel_err_vs_reference_golden_G1'], d['roofline']['kernel_ms_per_step']['wino_fused'], d['roofline']['frac'])
    except Exception as e: print(o,'ERR',e)
PY
cut -c1-300 "$O/wf_trace_w1.txt" | head -14; echo; cut -c1-300 "$O/wf_trace_w0.txt" | head -3
