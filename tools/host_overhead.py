#!/usr/bin/env python
"""cProfile of the Python side of one S1 get_forces step (where the host time of a step goes)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402

w = bench.make_workload('S1')
f, a = bench.make_force(w)
U = None
for _ in range(50):
    bench.step(f, a, U)
    U = f.U_ind
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    bench.step(f, a, U)
    U = f.U_ind
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(22)
