#!/usr/bin/env python3
"""A/B of the moving S1 loop (tools/s1_trace.py) with and without environment settings, runs interleaved:
    python tools/ab.py "ADMP_X=1 ADMP_Y=0" [pairs=5] [steps=1000] [workload=S1]
prints every run and the median / minimum of each side (run-to-run noise on a box is ~2 %)."""
import os, re, statistics, subprocess, sys
env_b = dict(kv.split('=', 1) for kv in sys.argv[1].split()) if len(sys.argv) > 1 and sys.argv[1] else {}
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = sys.argv[3] if len(sys.argv) > 3 else '1000'
wl = sys.argv[4] if len(sys.argv) > 4 else 'S1'
here = os.path.dirname(os.path.abspath(__file__))
res = {'A': [], 'B': []}
for k in range(pairs):
    for side in ('B', 'A') if k % 2 else ('A', 'B'):
        env = dict(os.environ)
        if side == 'B':
            env.update(env_b)
        out = subprocess.run([sys.executable, os.path.join(here, 's1_trace.py'), steps, wl], env=env, capture_output=True, text=True)
        m = re.search(r'([0-9.]+) ms/step', out.stdout)
        if not m:
            print(out.stdout[-500:], out.stderr[-500:]); sys.exit(1)
        res[side].append(float(m.group(1)))
for side, label in (('A', 'default'), ('B', sys.argv[1] if len(sys.argv) > 1 else '')):
    v = res[side]
    print('%s: %s  median %.4f  min %.4f   [%s]' % (side, ' '.join('%.4f' % x for x in v), statistics.median(v), min(v), label))
