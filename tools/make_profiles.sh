#!/bin/bash
# Round-2 measurement artefacts (run via gpurun from the repo root; copy gpurun_out/r02/* into profiles/).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd $R
# PMC passes first: bench.py reads profiles/r02_pmc_hbm_traffic.json (hash-locked to this build) for roofline.traffic
bash tools/pmc_bench.sh > /dev/null 2>&1
cp $R/gpurun_out/pmc_bench/summary.json $O/r02_pmc_hbm_traffic.json
mkdir -p $R/profiles && cp $O/r02_pmc_hbm_traffic.json $R/profiles/r02_pmc_hbm_traffic.json
python3 bench.py > $O/r02_bench.json 2> $O/bench.err
python3 bench.py --pairs-per-step 512 --no-cpu-baseline --no-secondary > $O/r02_bench_pairs512_1gpu.json 2>> $O/bench.err
FFR_BENCH_BACKEND=gloo FFR_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --pairs-per-step 512 --no-roofline --no-cpu-baseline > $O/r02_bench_selflaunch_2ranks_one_device.json 2>> $O/bench.err
python3 tools/bench_train.py --cpu-baseline > $O/r02_train_step.json 2>> $O/bench.err
FFR_WF_TRACE=1 python3 tools/wf_trace.py 2>&1 | grep "wf trace" > $O/r02_wino_fused_phase_trace.txt
FFR_IGEMM_TRACE=1 python3 tools/wf_trace.py 2>&1 | grep "igemm trace" > $O/r02_igemm_trace.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/r02_bench_under_rocprof.json 2>/dev/null
cp $O/prof/p_kernel_stats.csv $O/r02_bench_kernel_stats.csv
python3 $R/tools/layer_times.py $O/prof/p_kernel_trace.csv > $O/r02_bench_layer_times.txt
rocprofv3 --kernel-trace --stats -d $O/proft -o p --output-format csv -- python3 $R/tools/bench_train.py --steps 5 --warmup 2 > $O/r02_train_step_under_rocprof.json 2>/dev/null
cp $O/proft/p_kernel_stats.csv $O/r02_train_step_kernel_stats.csv
rm -rf $O/prof $O/proft
ls -la $O
