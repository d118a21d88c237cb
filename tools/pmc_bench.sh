#!/bin/bash
# HBM traffic of k_igemm over a whole forward (separate --pmc passes, MI355X_MICROARCH.md: FETCH_SIZE
# reads half of a wide coalesced stream on gfx950 -> doubled below; WRITE_SIZE exact). Run via gpurun.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_bench
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bench/$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>$R/gpurun_out/pmc_bench/$c.err
done
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(lambda: [0.0, 0])
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('$R/gpurun_out/pmc_bench/%s/*/*counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            k = 'igemm' if ('k_igemm' in r['Kernel_Name'] or 'k_gemm_stream' in r['Kernel_Name']) else r['Kernel_Name'].split('(')[0][-40:]
            if r['Counter_Name'] == c:
                tot[(k, c)][0] += float(r['Counter_Value']); tot[(k, c)][1] += 1
out = {}
for (k, c), (v, n) in sorted(tot.items()):
    kb = v / n
    out.setdefault(k, {})[c + '_KB_per_launch'] = kb
    out[k]['launches'] = n
ig = out.get('igemm', {})
if ig:
    ig['hbm_bytes_per_launch_corrected'] = (2 * ig.get('FETCH_SIZE_KB_per_launch', 0) + ig.get('WRITE_SIZE_KB_per_launch', 0)) * 1024
print(json.dumps(out, indent=1))
json.dump(out, open('$R/gpurun_out/pmc_bench/summary.json', 'w'), indent=1)
PY
