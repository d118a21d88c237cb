#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03k"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
sha256sum ffr-net_amd/libffrnet_hip.so | cut -c1-16
timeout 1800 python3 -m pytest tests -m gpu -q > "$O/pytest_all.log" 2>&1; tail -6 "$O/pytest_all.log"
timeout 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
