#!/bin/bash
# Measurement artefacts of one round (run via gpurun from the repo root, ONCE, after the kernels are frozen):  bash tools/make_profiles.sh r06
# Writes gpurun_out/<round>/<round>_*; the PMC summary is installed under profiles/ ONLY when every counter pass succeeded
# and its so_sha256 is the hash of the library in the tree.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the root of the snapshot)}"
RD="${1:-r06}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/$RD"
rm -rf -- "$O"; mkdir -p -- "$O"
cd -- "$R"
SHA=$(sha256sum ffr-net_amd/libffrnet_hip.so | cut -d' ' -f1)
# PMC passes first: bench.py reads profiles/<round>_pmc_hbm_traffic.json (hash-locked to this build) for roofline.traffic
if bash tools/pmc_bench.sh > "$O/pmc_bench.log" 2>&1 && grep -q "\"so_sha256\": \"$SHA\"" "$R/gpurun_out/pmc_bench/summary.json"; then
  cp -- "$R/gpurun_out/pmc_bench/summary.json" "$O/${RD}_pmc_hbm_traffic.json"
  mkdir -p -- "$R/profiles" && cp -- "$O/${RD}_pmc_hbm_traffic.json" "$R/profiles/${RD}_pmc_hbm_traffic.json"
else
  echo "PMC pass failed or measured another build: profiles/${RD}_pmc_hbm_traffic.json NOT updated" | tee "$O/pmc_FAILED.txt"
fi
if FFR_PMC_WORKLOAD=train bash tools/pmc_bench.sh > "$O/pmc_train.log" 2>&1 && grep -q "\"so_sha256\": \"$SHA\"" "$R/gpurun_out/pmc_bench/summary.json"; then
  cp -- "$R/gpurun_out/pmc_bench/summary.json" "$O/${RD}_pmc_train_traffic.json"
  cp -- "$O/${RD}_pmc_train_traffic.json" "$R/profiles/${RD}_pmc_train_traffic.json"
else
  echo "training PMC pass failed or measured another build: profiles/${RD}_pmc_train_traffic.json NOT updated" | tee "$O/pmc_train_FAILED.txt"
fi
python3 bench.py > "$O/${RD}_bench.json" 2> "$O/bench.err" || echo "bench rc $?"
for b in 128 64; do
  python3 bench.py --batch $b --no-cpu-baseline --no-secondary > "$O/${RD}_bench_batch$b.json" 2>> "$O/bench.err" || echo "bench$b rc $?"
done
python3 bench.py --pairs-per-step 512 --no-cpu-baseline --no-secondary > "$O/${RD}_bench_pairs512_1gpu.json" 2>> "$O/bench.err" || true
# the driver's flag-less N-GPU call on ONE device over gloo: the weak-scaling line + its strong-scaling `secondary` entry (timings mean nothing)
FFR_BENCH_BACKEND=gloo FFR_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline --no-cpu-baseline 2>> "$O/bench.err" | grep '^{' > "$O/${RD}_bench_selflaunch_2ranks_gloo_one_device.json" || true
python3 bench.py --workload train > "$O/${RD}_train_step.json" 2>> "$O/bench.err" || true
python3 tools/wf_trace.py 2>&1 | grep "wf trace" > "$O/wf_trace_all.txt" || true
grep -E "mixed|type \(|CUs ran" "$O/wf_trace_all.txt" > "$O/${RD}_wino_mixed_phase_trace.txt" || true
grep "k_channel_path" "$O/wf_trace_all.txt" > "$O/${RD}_channel_path_phase_trace.txt" || true
grep -v -E "mixed|type \(|CUs ran|k_channel_path" "$O/wf_trace_all.txt" > "$O/${RD}_wino_fused_phase_trace.txt" || true
B=64 python3 tools/wf_trace.py 2>&1 | grep "k_channel_path" >> "$O/${RD}_channel_path_phase_trace.txt" || true
TRACE=igemm_trace python3 tools/wf_trace.py 2>&1 | grep "igemm trace" > "$O/${RD}_igemm_trace.txt" || true
cd /tmp && export TMPDIR=/tmp
export FFR_BENCH_LIVE_PMC=0      # runs under rocprofv3 never start profiler passes of their own (bench.py also detects the profiler)
for b in 256 128 64; do
  rocprofv3 --kernel-trace --stats -d "$O/prof$b" -o p --output-format csv -- python3 "$R/bench.py" --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > "$O/${RD}_bench_batch${b}_under_rocprof.json" 2>/dev/null || true
  cp -- "$O/prof$b/p_kernel_stats.csv" "$O/${RD}_bench_batch${b}_kernel_stats.csv" || true
  python3 "$R/tools/layer_times.py" "$O/prof$b/p_kernel_trace.csv" > "$O/${RD}_bench_batch${b}_layer_times.txt" || true
  rm -rf -- "$O/prof$b"
done
python3 "$R/tools/roofline_check.py" "$O/${RD}_bench_batch256_kernel_stats.csv" "$O/${RD}_bench.json" > "$O/${RD}_roofline_check.txt" 2>&1 || echo "roofline_check: DISAGREEMENT (see ${RD}_roofline_check.txt)"
rocprofv3 --kernel-trace --stats -d "$O/proft" -o p --output-format csv -- python3 "$R/bench.py" --workload train --steps 5 --warmup 2 --no-roofline > "$O/${RD}_train_step_under_rocprof.json" 2>/dev/null || true
cp -- "$O/proft/p_kernel_stats.csv" "$O/${RD}_train_step_kernel_stats.csv" || true
python3 "$R/tools/roofline_check.py" --train "$O/${RD}_train_step_kernel_stats.csv" "$O/${RD}_train_step.json" >> "$O/${RD}_roofline_check.txt" 2>&1 || echo "roofline_check --train: DISAGREEMENT (see ${RD}_roofline_check.txt)"
rm -rf -- "$O/proft"
ls -la -- "$O"
