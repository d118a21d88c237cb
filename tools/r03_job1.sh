#!/bin/bash
# round-3 GPU job 1: full GPU test suite, bench at batch 256 / 128 / 64 with per-layer tables
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03a"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests -m gpu -x -q -s > "$O/pytest.log" 2>&1; echo "pytest rc $?" >> "$O/pytest.log"
tail -5 "$O/pytest.log"
timeout 600 python3 bench.py > "$O/bench256.json" 2> "$O/bench256.err"; echo "bench rc $?"
for b in 128 64; do
  timeout 300 python3 bench.py --batch $b --no-cpu-baseline --no-secondary > "$O/bench$b.json" 2> "$O/bench$b.err"; echo "bench$b rc $?"
done
cd /tmp && export TMPDIR=/tmp
for b in 256 128 64; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof$b" -o p --output-format csv -- python3 "$R/bench.py" --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2> "$O/prof$b.err"
  python3 "$R/tools/layer_times.py" "$O/prof$b/p_kernel_trace.csv" > "$O/layers$b.txt" 2>&1
  cp "$O/prof$b/p_kernel_stats.csv" "$O/kernel_stats$b.csv" 2>/dev/null
  rm -rf "$O/prof$b"
done
python3 - <<PY
import json
for b in (256,128,64):
    try:
        d=json.loads([l for l in open('$O/bench%d.json'%b) if l.startswith('{')][-1])
        print(b, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'] if d.get('roofline') else None)
    except Exception as e: print(b,'ERR',e)
PY
