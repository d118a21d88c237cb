#!/bin/bash
# PMC passes over tools/conv_one.py (run on the GPU box through gpurun). Output: gpurun_out/pmc_<tag>/
tag=${1:-conv}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0; mkdir -p $R/gpurun_out/pmc_$tag
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -- python3 $R/tools/conv_one.py > /dev/null 2>$R/gpurun_out/pmc_$tag/p$i.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/pmc_$tag/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'igemm' not in r['Kernel_Name']: continue
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%d mean=%.4g' % (c, len(v), sum(v)/len(v)))
PY
