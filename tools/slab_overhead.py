#!/usr/bin/env python3
"""Host overhead of the staged (slab) driver: SlabPme with ONE rank against the fused single-GPU entry point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from admp_amd.parallel import ThreadComm
name = sys.argv[1] if len(sys.argv) > 1 else 'S2'
w = bench.make_workload(name)
for label, comm in (('fused', None), ('staged-1rank', ThreadComm(ThreadComm.World(1), 0))):
    f, a = bench.make_force(w, comm)
    U = None
    for _ in range(3):
        bench.step(f, a, U); U = f.U_ind
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        bench.step(f, a, U); U = f.U_ind
    torch.cuda.synchronize()
    print('%s %s: %.3f ms/step' % (name, label, (time.perf_counter() - t0) / 20 * 1e3))
