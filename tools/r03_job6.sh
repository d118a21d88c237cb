#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03f"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trunk or batch_independence or knobs or embed or golden or lfw_protocol" > "$O/pytest.log" 2>&1; tail -8 "$O/pytest.log"
for o in 1 0; do
  timeout 300 python3 bench.py --opt epi_v=$o --no-cpu-baseline --no-secondary > "$O/bench_ev$o.json" 2> "$O/bench_ev$o.err"; echo "bench ev$o rc $?"
done
timeout 300 python3 bench.py --batch 128 --no-cpu-baseline --no-secondary > "$O/bench128.json" 2> "$O/bench128.err"
python3 - <<PY
import json
for o in ('_ev1','_ev0','128'):
    try:
        d=json.loads([l for l in open('$O/bench%s.json'%o) if l.startswith('{')][-1])
        print(o, d['value'], d['ms_per_step'], d['parity_checked']['max_rel_err_vs_reference_golden_G1'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(o,'ERR',e)
PY
