#!/usr/bin/env python
"""Steady-state kernel timeline of one bench step from a `rocprofv3 --kernel-trace --output-format csv` run:
prints, for the last complete step, every dispatch with its duration and the idle gap before it.
    python tools/timeline.py <dir with *_kernel_trace.csv> [anchor-kernel-substring] [steps]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_prepare_sites'
    files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
    if len(idx) < 3:
        print('anchor not found often enough'); return
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    a, b = idx[-2 - nsteps], idx[-2]
    prev_end = int(rows[a - 1]['End_Timestamp'])
    busy = 0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%-70s %8.2f us   gap %7.2f us' % (r['Kernel_Name'][:70], (e - s) / 1e3, (s - prev_end) / 1e3))
        busy += e - s
        prev_end = e
    total = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
    print('step period %.2f us, kernels busy %.2f us, idle %.2f us, %d dispatches' % (total / 1e3, busy / 1e3, (total - busy) / 1e3, b - a))


if __name__ == '__main__':
    main()
