#!/usr/bin/env python3
"""Per-step wall times of the moving S1 loop from the first call on (does the step time settle, and when?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
w = bench.make_workload('S1')
f, a = bench.make_force(w)
frames = bench.ThermalFrames(w, torch.device('cuda'))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seq = [frames.step_frame(k) for k in range(n)]
U = None
marks = []
torch.cuda.synchronize()
for k in range(n):
    t0 = time.perf_counter()
    bench.step(f, a, U, seq[k]); U = f.U_ind
    marks.append((time.perf_counter() - t0) * 1e3)
m = np.asarray(marks)
for lo in range(0, n, 20):
    seg = m[lo:lo + 20]
    print('steps %3d-%3d  mean %.4f  median %.4f  min %.4f  max %.4f' % (lo, lo + len(seg) - 1, seg.mean(), np.median(seg), seg.min(), seg.max()))
print('scf stats', f.scf_stats())
