"""Per-kernel times of the dispersion and Tang-Toennies calculators inside the all-terms loop of bench.md_all_terms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from admp_amd.disp_pme import ADMPDispPmeForce
from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
name = sys.argv[1] if len(sys.argv) > 1 else 'S3'
w = bench.make_workload(name)
dt = torch.float32 if w['prec'] == 'single' else torch.float64
par = w['par']
f, a = bench.make_force(w)
fr = bench.ThermalFrames(w, torch.device('cuda', 0))
cl = torch.as_tensor(par['c_list'], dtype=dt, device='cuda')
disp = ADMPDispPmeForce(w['box'], w['cov'], bench.RC, 1e-4, 10)
if w['K'] is not None:
    for k in ('K1', 'K2', 'K3'):
        disp.update_env(k, w['K'])
tt_obj = generate_pairwise_interaction(TT_damping_qq_c6_kernel, w['cov'], static_args={})
tt = value_and_grad(tt_obj)
a_, b_, q_ = (torch.as_tensor(par[k], dtype=dt, device='cuda') for k in ('a_list', 'b_list', 'q_list'))
c6 = cl[:, 0].contiguous()
mS = par['mScales']
seq = [fr.step_frame(k) for k in range(12)]
f.update_neighbors(seq[0], w['box'], rc=bench.RC + bench.SKIN)
for o in (disp, tt_obj):
    o.share_neighbors(f)
U = None
def one(k, U):
    _, g = bench.step(f, a, U, seq[k], pairs=None)
    _, g2 = disp.get_forces(seq[k], w['box'], None, cl, mS)
    _, g3 = tt(seq[k], w['box'], None, mS, a_, b_, q_, c6)
    return f.U_ind
for k in range(3):
    U = one(k, U)
for o in (f, disp, tt_obj):
    o.profile(True); o.profile_reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(3, 8):
    U = one(k, U)
torch.cuda.synchronize(); print('all terms (events on): %.3f ms/step' % ((time.perf_counter() - t0) / 5 * 1e3))
for label, o in (('pme', f), ('disp', disp), ('tt', tt_obj)):
    rep = o.profile_report(); o.profile(False)
    print(label, '%.3f' % sum(v[0] / 5 for v in rep.values()), '  '.join('%s %.4f' % (k, v[0] / 5) for k, v in sorted(rep.items())))
