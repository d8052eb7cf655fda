#!/usr/bin/env python3
"""Per-kernel ms of a step (HIP events around every launch), static and moving geometry:
    python tools/kernels.py [S1|S2|S3] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'S3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = bench.make_workload(name)
f, a = bench.make_force(w)
fr = bench.ThermalFrames(w, torch.device('cuda', 0))
for label, frames in (('static', None), ('moving', fr)):
    dt, _, cyc = bench.run_timed(f, a, steps, 3, frames, only=False)
    kb = bench.kernel_breakdown(f, a, frames, steps + 3, steps)[0]
    print('%s %s: %.3f ms/step  %s' % (name, label, dt / steps * 1e3, cyc))
    print('   ' + '  '.join('%s %.4f' % (k, v) for k, v in kb.items()))
