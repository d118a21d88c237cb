#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03j"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$O/prof" -o p --output-format csv -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > "$O/line.json" 2>/dev/null
python3 "$R/tools/layer_times.py" "$O/prof/p_kernel_trace.csv" > "$O/layers.txt"; tail -6 "$O/layers.txt"
python3 "$R/tools/graph_vs_eager.py" 2>&1 | tail -5
rm -rf "$O/prof"
