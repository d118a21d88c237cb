#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03c"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
for o in 1 0; do
  FFR_OPT_WF_DMA=$o timeout 300 python3 tools/wf_trace.py 2>&1 | grep "wf trace" > "$O/wf_trace_dma$o.txt"
done
grep "transform in the kernel" "$O/wf_trace_dma1.txt" | cut -c1-330
echo ----
grep "transform in the kernel" "$O/wf_trace_dma0.txt" | cut -c1-330
