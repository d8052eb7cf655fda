#!/usr/bin/env python
"""Kernels between the LAST occurrence of a start kernel and the next occurrence of a stop kernel in a rocprofv3 kernel trace:
    python tools/trace_window.py <dir> <start-substring> <stop-substring>"""
import csv, glob, os, sys
d, a, b = sys.argv[1:4]
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ia = max(i for i, r in enumerate(rows) if a in r['Kernel_Name'] and any(b in q['Kernel_Name'] for q in rows[i:]))
# first start kernel of that group (consecutive window: walk back while the previous start is less than 5 ms away)
ib = next(i for i in range(ia, len(rows)) if b in rows[i]['Kernel_Name'])
t0 = int(rows[ia]['Start_Timestamp'])
prev = t0
for r in rows[ia:ib + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-78s %9.2f us  gap %8.2f  t %9.2f' % (r['Kernel_Name'][:78], (e - s) / 1e3, (s - prev) / 1e3, (s - t0) / 1e3))
    prev = e
