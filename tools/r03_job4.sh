#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03d"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_gpu_train.py -m gpu -x -q > "$O/pytest_train.log" 2>&1; tail -8 "$O/pytest_train.log"
timeout 600 python3 tools/bench_train.py > "$O/train_step.json" 2> "$O/train_step.err"; tail -3 "$O/train_step.json"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_independence or two_ranks or knobs" > "$O/pytest_par.log" 2>&1; tail -8 "$O/pytest_par.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/proft" -o p --output-format csv -- python3 "$R/tools/bench_train.py" --steps 5 --warmup 2 > "$O/train_under_rocprof.json" 2>/dev/null
cp "$O/proft/p_kernel_stats.csv" "$O/train_kernel_stats.csv"; rm -rf "$O/proft"
head -25 "$O/train_kernel_stats.csv" | cut -c1-150
