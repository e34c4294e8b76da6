#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 kernel trace:
    python tools/timeline.py gpurun_out/<tag>/stats [n_rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))[-n:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-44s start %8.1f  dur %7.1f  gap %6.1f us  grid %8s" % (r['Kernel_Name'][:44], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Grid_Size_X']))
    prev_end = max(prev_end, e)
