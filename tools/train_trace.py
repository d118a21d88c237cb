"""Kernel sequence of the LAST training iteration in a rocprofv3 --kernel-trace directory, from the first
k_copy_slice after the encoder (start of the RecNet forward) on: name, duration, running total."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# last k_stem = start of the last iteration
last = max(i for i, r in enumerate(rows) if 'k_stem' in r[2])
it = rows[last:]
start = next(i for i, r in enumerate(it) if 'k_copy_slice' in r[2])
agg = {}
tot = 0.0
for s, e, n in it[start:]:
    short = n.split('(')[0].replace('void ', '').replace('ffr::', '')[:60]
    agg.setdefault(short, [0, 0.0])
    agg[short][0] += 1
    agg[short][1] += (e - s) / 1e3
    tot += (e - s) / 1e3
print('RecNet fwd + losses + bwd + adam: %.1f us of kernels, wall %.1f us' % (tot, (it[-1][1] - it[start][0]) / 1e3))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%-62s x%-4d %9.1f us' % (k, c, t))
