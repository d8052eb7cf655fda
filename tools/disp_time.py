#!/usr/bin/env python
"""Time the dispersion-PME and Tang-Toennies calculators (ms per get_forces, per-kernel breakdown).
usage: python tools/disp_time.py [S1|S2|S3] [skin]       (skin: on the Verlet list of rc + 1 A, cutoff honoured, as in an MD loop)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402
from admp_amd import settings                                                 # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'S1'
w = bench.make_workload(name)
settings.PRECISION = w['prec']
from admp_amd.disp_pme import ADMPDispPmeForce                                # noqa: E402
from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad   # noqa: E402

dt = torch.float32 if w['prec'] == 'single' else torch.float64
dev = 'cuda'
pos = torch.as_tensor(w['pos'], dtype=dt, device=dev)
par = w['par']
cl = torch.as_tensor(par['c_list'], dtype=dt, device=dev)
disp = ADMPDispPmeForce(w['box'], w['cov'], 4.0, 1e-4, 10)
if w['K'] is not None:
    for k in ('K1', 'K2', 'K3'):
        disp.update_env(k, w['K'])
pairs = w['pairs']
mS = par['mScales']
tt_obj = generate_pairwise_interaction(TT_damping_qq_c6_kernel, w['cov'], static_args={})
if len(sys.argv) > 2 and sys.argv[2] == 'skin':
    f, a = bench.make_force(w)
    f.update_neighbors(pos, w['box'], rc=bench.RC + bench.SKIN)
    for obj in (disp, tt_obj):
        obj.share_neighbors(f)
    pairs = None
    name += ' (skin list, %d pairs)' % f.n_pairs


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_disp = timeit(lambda: disp.get_forces(pos, w['box'], pairs, cl, mS))
disp.profile(True)
disp.profile_reset()
for _ in range(5):
    disp.get_forces(pos, w['box'], pairs, cl, mS)
rep = disp.profile_report()
disp.profile(False)
print('%s dispersion pmax=10: %.3f ms per get_forces; kernels (ms/call): %s' % (
    name, t_disp, {k: round(v[0] / 5, 4) for k, v in sorted(rep.items())}))
tt = value_and_grad(tt_obj)
T = lambda k: torch.as_tensor(par[k], dtype=dt, device=dev)      # noqa: E731
a_, b_, q_, c6 = T('a_list'), T('b_list'), T('q_list'), cl[:, 0].contiguous()
print('%s Tang-Toennies: %.3f ms per call' % (name, timeit(lambda: tt(pos, w['box'], pairs, mS, a_, b_, q_, c6))))
