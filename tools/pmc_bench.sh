#!/bin/bash
# HBM traffic and matrix-core counters of one forward at batch 256 (run via gpurun; copy the summary into profiles/).
# Separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2); FETCH_SIZE reads half of a
# wide coalesced stream on gfx950 -> doubled below, WRITE_SIZE is exact.  The program follows `--` directly.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"
OUT="$R/gpurun_out/pmc_bench"
rm -rf -- "$OUT"; mkdir -p -- "$OUT"
cd /tmp && export TMPDIR=/tmp
export FFR_BENCH_LIVE_PMC=0      # the profiled bench never starts profiler passes of its own
WL="${FFR_PMC_WORKLOAD:-embed}"      # embed (bench.py's headline workload) | train (bench.py --workload train: one k_stem launch per iteration)
if [ "$WL" = train ]; then B="${FFR_PMC_BATCH:-128}"; else B="${FFR_PMC_BATCH:-256}"; fi
PT="${FFR_PMC_PASS_TIMEOUT:-200}"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  d=$(echo $c | cut -d' ' -f1)
  timeout "$PT" rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$d" -- python3 "$R/bench.py" --workload "$WL" --batch "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>"$OUT/$d.err"
done
python3 - <<PY
import csv, glob, collections, json, hashlib
R = '$R'
BATCH = int('$B')
WL = '$WL'
sha = hashlib.sha256(open(R + '/ffr-net_amd/libffrnet_hip.so', 'rb').read()).hexdigest()
STEPS = 3          # warmup 1 + steps 2 forwards at batch BATCH; the batch-8 parity forward in front of them is cut off
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    stems = sorted({int(r['Dispatch_Id']) for r in rows if 'k_stem' in r['Kernel_Name']})
    # dispatch id of the first counted forward: embed = the batch-8 parity forward + STEPS forwards (one k_stem each), train =
    # STEPS iterations (one k_stem each: clean + occluded images in one encoder pass).  Anything else means bench.py changed
    # what it launches and the per-step figures below would be skewed: fail instead of guessing.
    want = STEPS + 1 if WL == 'embed' else STEPS
    if len(stems) != want:
        raise SystemExit('pmc_bench: %d k_stem launches in %s, expected %d for workload %s' % (len(stems), f, want, WL))
    first = stems[1] if WL == 'embed' else stems[0]
    for r in rows:
        if int(r['Dispatch_Id']) < first:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not k.startswith('ffr::'):
            continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
out = {'so_sha256': sha, 'batch': BATCH, 'workload': WL, 'how': 'tools/pmc_bench.sh: rocprofv3 --pmc, one pass per counter group, bench.py --steps 2 --warmup 1: the 3 '
       'forwards at batch %d (the batch-8 parity forward is excluded); FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE exact. '
       'FETCH_SIZE counts L2 misses, Infinity-Cache hits included (MI355X_MICROARCH.md): re-reads of a 151 MB V by the other XCDs '
       'appear here although they need not reach HBM.' % BATCH, 'kernels': {}}
gb = 0.0
for k in sorted(tot):
    e = {'launches_per_step': max(cnt[k].values()) / STEPS}
    fk, wk = tot[k].get('FETCH_SIZE', 0.0), tot[k].get('WRITE_SIZE', 0.0)
    e['hbm_gb_per_step_corrected'] = round((2 * fk + wk) * 1024 / STEPS / 1e9, 3)
    e['fetch_kb_per_launch'] = fk / max(1, cnt[k].get('FETCH_SIZE', 1))
    e['write_kb_per_launch'] = wk / max(1, cnt[k].get('WRITE_SIZE', 1))
    gb += (2 * fk + wk) * 1024 / STEPS
    for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY',
              'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE'):
        if c in tot[k]:
            e[c + '_per_launch'] = tot[k][c] / cnt[k][c]
    out['kernels'][k] = e
out['gb_per_step'] = round(gb / 1e9, 2)
fused = [k for k in out['kernels'] if 'k_wino_fused' in k]
if fused:
    lps = sum(out['kernels'][k]['launches_per_step'] for k in fused)
    gbs = sum(out['kernels'][k]['hbm_gb_per_step_corrected'] for k in fused)
    out['dominant'] = {'kernel': 'ffr::k_wino_fused<false|true>', 'launches_per_step': lps, 'hbm_gb_per_step': round(gbs, 3),
                       'hbm_bytes_per_launch': int(gbs * 1e9 / lps)}
    mb = sum(out['kernels'][k].get('SQ_VALU_MFMA_BUSY_CYCLES_per_launch', 0) * out['kernels'][k]['launches_per_step'] for k in fused)
    ga = sum(out['kernels'][k].get('GRBM_GUI_ACTIVE_per_launch', 0) * out['kernels'][k]['launches_per_step'] for k in fused)
    if mb and ga:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs of the chip, GRBM_GUI_ACTIVE the 8 XCDs
        out['dominant']['mfma_busy_frac_of_kernel_time'] = round((mb / 1024.0) / (ga / 8.0), 4)
print(json.dumps(out, indent=1))
json.dump(out, open('$OUT/summary.json', 'w'), indent=1)
PY
