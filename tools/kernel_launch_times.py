#!/usr/bin/env python3
"""python tools/kernel_launch_times.py <p_kernel_trace.csv> <substring> [last N]: duration (us) and grid of the last N launches whose
kernel name contains <substring>, in launch order (rocprofv3 --kernel-trace --output-format csv)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for r in rows[-n:]:
    name = r['Kernel_Name'].replace('ffr::', '').replace('void ', '')[:48]
    grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
    wg = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 256)) or 256)
    print('%-48s %9.1f us  %6d blocks' % (name, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, grid // max(1, wg)))
