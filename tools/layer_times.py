#!/usr/bin/env python3
"""Per-launch durations of one forward from a rocprofv3 --kernel-trace CSV (conv kernels only).
usage: layer_times.py <kernel_trace.csv> [substring ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
keys = sys.argv[2:] or ['wino', 'igemm', 'gemm_stream']
idx = [i for i, r in enumerate(rows) if 'k_stem' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
tot = {}
for r in rows[s:e]:
    n = r['Kernel_Name'].split('(')[0]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
    tot[n] = tot.get(n, 0) + d
    if any(k in n for k in keys):
        print('%-34s blocks %7d  %8.1f us' % (n[-34:], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), d))
print('--- per step, us')
for n, d in sorted(tot.items(), key=lambda x: -x[1]):
    print('%10.1f  %s' % (d, n[-70:]))
print('%10.1f  total kernel time; wall %.1f' % (sum(tot.values()), (int(rows[e]['Start_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1000))
