#!/usr/bin/env python3
"""Plain step timing (no per-kernel events): usage python tools/step_time.py [S1|S2|S3] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'S3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = bench.make_workload(name)
f, a = bench.make_force(w)
U = None
for _ in range(3):
    bench.step(f, a, U); U = f.U_ind
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    bench.step(f, a, U); U = f.U_ind
torch.cuda.synchronize()
print('%s: %.3f ms/step' % (name, (time.perf_counter() - t0) / steps * 1e3))
