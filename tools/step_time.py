#!/usr/bin/env python3
"""Plain step timing (no per-kernel events): usage python tools/step_time.py [S1|S2|S3] [steps] [skin]
skin: on the Verlet list of rc + 1 A built on the GPU, cutoff honoured (what an MD loop runs on)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'S3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = bench.make_workload(name)
f, a = bench.make_force(w)
U = None
kw = {}
if len(sys.argv) > 3 and sys.argv[3] == 'skin':
    dt = torch.float32 if w['prec'] == 'single' else torch.float64
    f.update_neighbors(torch.as_tensor(w['pos'], dtype=dt, device='cuda'), w['box'], rc=bench.RC + bench.SKIN)
    kw = {'pairs': None}
    name += ' (skin list, %d pairs)' % f.n_pairs
for _ in range(3):
    bench.step(f, a, U, **kw); U = f.U_ind
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    bench.step(f, a, U, **kw); U = f.U_ind
torch.cuda.synchronize()
print('%s: %.3f ms/step' % (name, (time.perf_counter() - t0) / steps * 1e3))
