#!/bin/bash
# Vector-memory front-end counters (texture addresser TA, L1 TCP) per kernel of one forward at batch 256 -- the hardware
# evidence for EXPERIMENTS.md 3.2 "cost model".  Separate --pmc passes; run via gpurun, copy the summary into profiles/.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"
OUT="$R/gpurun_out/pmc_vmem"
rm -rf -- "$OUT"; mkdir -p -- "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
# (a pass with TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum made rocprofv3
# abort and hang on this pool: not collected; every pass runs under its own timeout)
for c in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>"$OUT/p$i.err" || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json, hashlib
R = '$R'
sha = hashlib.sha256(open(R + '/ffr-net_amd/libffrnet_hip.so', 'rb').read()).hexdigest()
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    stems = sorted({int(r['Dispatch_Id']) for r in rows if 'k_stem' in r['Kernel_Name']})
    first = stems[1] if len(stems) > 3 else stems[0]
    for r in rows:
        if int(r['Dispatch_Id']) < first:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not k.startswith('ffr::'):
            continue
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and cnt[k]['GRBM_GUI_ACTIVE'] and f.find('/p1/') < 0:
            continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
out = {'so_sha256': sha, 'how': 'tools/pmc_vmem.sh: rocprofv3 --pmc, four passes, the 3 batch-256 forwards of bench.py --steps 2 --warmup 1; '
       'values per LAUNCH (averages); *_sum counters add all instances of the chip (256 TAs / TCPs), GRBM_GUI_ACTIVE the 8 XCDs', 'kernels': {}}
for k in sorted(tot):
    e = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
    e['launches'] = max(cnt[k].values())
    ga = e.get('GRBM_GUI_ACTIVE', 0) / 8.0
    if ga:
        if 'TA_TA_BUSY_sum' in e: e['ta_busy_frac'] = round(e['TA_TA_BUSY_sum'] / 256.0 / ga, 4)
        if 'TA_ADDR_STALLED_BY_TC_CYCLES_sum' in e: e['ta_addr_stalled_by_tc_frac'] = round(e['TA_ADDR_STALLED_BY_TC_CYCLES_sum'] / 256.0 / ga, 4)
        if 'TA_DATA_STALLED_BY_TC_CYCLES_sum' in e: e['ta_data_stalled_by_tc_frac'] = round(e['TA_DATA_STALLED_BY_TC_CYCLES_sum'] / 256.0 / ga, 4)
        if 'TA_TOTAL_WAVEFRONTS_sum' in e and e.get('TA_TA_BUSY_sum'): e['ta_busy_cycles_per_wave_instruction'] = round(e['TA_TA_BUSY_sum'] / e['TA_TOTAL_WAVEFRONTS_sum'], 2)
        if 'TCP_PENDING_STALL_CYCLES_sum' in e: e['tcp_pending_stall_frac'] = round(e['TCP_PENDING_STALL_CYCLES_sum'] / 256.0 / ga, 4)
    out['kernels'][k] = e
json.dump(out, open('$OUT/summary.json', 'w'), indent=1)
for k, e in out['kernels'].items():
    if any(s in k for s in ('wino_fused', 'igemm<128', 'wino_in_c', 'combine', 'stem')):
        print(k[:44], {x: e[x] for x in e if x.endswith('_frac') or x.startswith('ta_busy_cycles')})
PY
