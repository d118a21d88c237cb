#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03e"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trunk or batch_independence_full or knobs or embed or golden" > "$O/pytest.log" 2>&1; tail -8 "$O/pytest.log"
for o in 1 0; do
  timeout 300 python3 bench.py --opt combine_v=$o --no-cpu-baseline --no-secondary > "$O/bench_cv$o.json" 2> "$O/bench_cv$o.err"; echo "bench cv$o rc $?"
done
python3 - <<PY
import json
for o in (1,0):
    try:
        d=json.loads([l for l in open('$O/bench_cv%d.json'%o) if l.startswith('{')][-1])
        print('combine_v',o, d['value'], d['ms_per_step'], d['parity_checked']['max_rel_err_vs_reference_golden_G1'], d['roofline']['kernel_ms_per_step'])
    except Exception as e: print(o,'ERR',e)
PY
