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
wall = (int(rows[e]['Start_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1000
print('%10.1f  total kernel time; wall %.1f' % (sum(tot.values()), wall))
# time in which NO kernel runs (launch gaps; kernels of the two streams may overlap: union of the busy intervals)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows[s:e])
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for a, b in iv[1:]:
    if a > cur_e:
        busy += cur_e - cur_s
        gaps.append((a - cur_e) / 1000)
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
busy += cur_e - cur_s
gaps.sort()
print('%10.1f  us with at least one kernel running; %.1f us idle in %d gaps (median %.2f, p90 %.2f, max %.1f us)'
      % (busy / 1000, wall - busy / 1000, len(gaps), gaps[len(gaps) // 2] if gaps else 0, gaps[int(len(gaps) * 0.9)] if gaps else 0,
         gaps[-1] if gaps else 0))
