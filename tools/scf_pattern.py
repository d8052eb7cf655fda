#!/usr/bin/env python
"""Sequence of Jacobi updates per step on the moving-atom trajectory of bench.py (is a failed first check predictable?).
    python tools/scf_pattern.py [S1|S2|S3] [steps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import time
import torch
import bench as B

name = sys.argv[1] if len(sys.argv) > 1 else 'S1'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
w = B.make_workload(name)
f, a = B.make_force(w)
frames = B.ThermalFrames(w, torch.device('cuda', 0))
U = None
seq, ms = [], []
fr = [frames.step_frame(k) for k in range(steps)]
torch.cuda.synchronize()
for k in range(steps):
    t0 = time.perf_counter()
    B.step(f, a, U, fr[k])
    U = f.U_ind
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) * 1e3)
    seq.append(int(f.n_cycle))
print('ms per step (synchronised):', ' '.join('%.3f' % m for m in ms[-12:]))
print(name, 'updates per step:', ''.join(str(min(s, 9)) for s in seq), 'mean %.2f' % (sum(seq[5:]) / float(len(seq) - 5)))
