#!/usr/bin/env python3
"""ms of admp_set_pairs from an EXPLICIT (Np, 2) device pair list -- the reference's calling convention (a `pairs` array
per rebuild, admp/pme.py:671-683) -- with the per-kernel breakdown:   python tools/set_pairs_time.py [S1|S2|S3] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'S3'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = bench.make_workload(name)
f, a = bench.make_force(w)
pairs = a['pairs']
other = pairs.clone()
f.set_pairs(pairs)
best = 1e9
for r in range(reps):
    p = other if r % 2 == 0 else pairs          # a different tensor object every time: no cache hit
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f.set_pairs(p)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print('%s: admp_set_pairs of %d pairs (%d table entries): %.3f ms (best of %d)' % (name, pairs.shape[0], 2 * f.n_pairs, best * 1e3, reps))
