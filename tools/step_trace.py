#!/usr/bin/env python3
"""Per-launch durations of one forward from a rocprofv3 kernel-trace CSV (last complete step)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last k_stem marks the start of the last forward
idx = [i for i, r in enumerate(rows) if 'k_stem' in r['Kernel_Name']]
start = idx[-1]
tot = {}
for r in rows[start:]:
    name = r['Kernel_Name'].split('(')[0].replace('void ffr::', '').replace('ffr::', '')
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[name] = tot.get(name, 0) + us
    if len(sys.argv) > 2:
        print('%-34s grid %-8s %9.1f us' % (name[:34], r.get('Grid_Size', r.get('Grid_Size_X', '?')), us))
print({k: round(v / 1e3, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
print('sum ms', round(sum(tot.values()) / 1e3, 3), ' wall ms', (int(rows[-1]['End_Timestamp']) - int(rows[start]['Start_Timestamp'])) / 1e6)
