#!/usr/bin/env python
"""Floor of the S1 step without the Python wrapper: the C entry point called in a tight loop with prebuilt arguments
(what a C/C++ host would pay), next to the regular get_forces loop."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from admp_amd import _lib

w = bench.make_workload('S1')
f, a = bench.make_force(w)
U = None
for _ in range(30):
    bench.step(f, a, U); U = f.U_ind
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500):
    bench.step(f, a, U); U = f.U_ind
torch.cuda.synchronize()
print('python get_forces loop: %.4f ms/step' % ((time.perf_counter() - t0) / 500 * 1e3))
L, h = f._L, f._h
na = f.n_atoms
pos, Q, pol, th = a['positions'], a['Q_local'], a['pol'], a['tholes']
Ut = torch.as_tensor(U).clone()
grad = torch.empty((na, 3), dtype=pos.dtype, device=pos.device)
box = _lib.darr(f._host64(a['box'], 9)); mS = _lib.darr(f._host64(a['mScales'])); pS = _lib.darr(f._host64(a['pScales']))
E = (ctypes.c_double * 4)(); ncyc, conv = ctypes.c_int(0), ctypes.c_int(1)
P = lambda t: ctypes.c_void_p(t.data_ptr())     # noqa: E731
args = (h, P(pos), box, P(Q), P(pol), P(th), 5, mS, pS, pS, P(Ut), 30, 10.0, E, P(grad), None, ctypes.byref(ncyc), ctypes.byref(conv), 1)
for _ in range(30):
    L.admp_pme_energy_grad(*args)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500):
    L.admp_pme_energy_grad(*args)
torch.cuda.synchronize()
print('C entry point loop:      %.4f ms/step (cycles %d)' % ((time.perf_counter() - t0) / 500 * 1e3, ncyc.value))
