#!/usr/bin/env python3
"""Per-kernel ms of the dispersion-PME and Tang-Toennies calculators on a skin list (what bench.md_all_terms times):
    python tools/terms_breakdown.py [S1|S2|S3] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from admp_amd.disp_pme import ADMPDispPmeForce
from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
name = sys.argv[1] if len(sys.argv) > 1 else 'S3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = bench.make_workload(name)
dt = torch.float32 if w['prec'] == 'single' else torch.float64
par = w['par']
cl = torch.as_tensor(par['c_list'], dtype=dt, device='cuda')
p = torch.as_tensor(w['pos'], dtype=dt, device='cuda')
disp = ADMPDispPmeForce(w['box'], w['cov'], bench.RC, 1e-4, 10)
if w['K'] is not None:
    for k in ('K1', 'K2', 'K3'):
        disp.update_env(k, w['K'])
tt_obj = generate_pairwise_interaction(TT_damping_qq_c6_kernel, w['cov'], static_args={})
tt = value_and_grad(tt_obj)
a_, b_, q_ = (torch.as_tensor(par[k], dtype=dt, device='cuda') for k in ('a_list', 'b_list', 'q_list'))
c6 = cl[:, 0].contiguous()
mS = par['mScales']
for label, obj, call in (('dispersion', disp, lambda: disp.get_forces(p, w['box'], None, cl, mS)),
                         ('tang-toennies', tt_obj, lambda: tt(p, w['box'], None, mS, a_, b_, q_, c6))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    obj.update_neighbors(p, w['box'], rc=bench.RC + bench.SKIN)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    obj.update_neighbors(p, w['box'], rc=bench.RC + bench.SKIN)
    torch.cuda.synchronize(); t_build = min(t_build, time.perf_counter() - t0)
    for _ in range(2):
        call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    obj.profile(True); obj.profile_reset()
    for _ in range(steps):
        call()
    rep = obj.profile_report(); obj.profile(False)
    print('%s %s: %.3f ms/call, list rebuild %.3f ms' % (name, label, ms, t_build * 1e3))
    print('   ' + '  '.join('%s %.4f' % (k, v[0] / steps) for k, v in sorted(rep.items())))
