#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03b"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 60 ./build/probe_dma > "$O/probe_dma.txt" 2>&1; cat "$O/probe_dma.txt"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd or conv_operator" > "$O/pytest_wino.log" 2>&1; tail -15 "$O/pytest_wino.log"
for o in 1 0; do
  timeout 300 python3 bench.py --opt wf_dma=$o --no-cpu-baseline --no-secondary > "$O/bench_dma$o.json" 2> "$O/bench_dma$o.err"; echo "bench dma$o rc $?"
done
python3 - <<PY
import json
for o in (1,0):
    try:
        d=json.loads([l for l in open('$O/bench_dma%d.json'%o) if l.startswith('{')][-1])
        print('dma',o, d['value'], d['ms_per_step'], d['parity_checked'], d['roofline']['kernel_ms_per_step'])
    except Exception as e: print(o,'ERR',e)
PY
timeout 1500 python3 -m pytest tests -m gpu -x -q > "$O/pytest_all.log" 2>&1; tail -15 "$O/pytest_all.log"
