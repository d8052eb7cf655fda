#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/limiter_pass.sh, plus the kernel's average duration from the same
pass's kernel trace -> profiles/<tag>_limiters_<workload>.csv (one row per kernel, one column per counter)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
names = []
for p in ('a', 'b'):
    d = os.path.join(src, 'lim_%s_%s' % (wl, p))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            per[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] not in names:
                names.append(r['Counter_Name'])
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
out = os.path.join(ROOT, 'profiles', '%s_limiters_%s.csv' % (tag, wl))
with open(out, 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'launches', 'avg_us_under_pmc'] + names)
    for k in sorted(per, key=lambda k: -sum(dur.get(k, [0]))):
        n = max(len(v) for v in per[k].values())
        row = [k.replace('void admp::', '')[:90], n, round(sum(dur[k]) / max(len(dur[k]), 1), 1)]
        row += [int(sum(per[k][c]) / len(per[k][c])) if per[k].get(c) else '' for c in names]
        w.writerow(row)
print(open(out).read()[:6000])
