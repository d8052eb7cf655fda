import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
from admp_amd.disp_pme import ADMPDispPmeForce
w = bench.make_workload('S3')
dt = torch.float32
par = w['par']
cl = torch.as_tensor(par['c_list'], dtype=dt, device='cuda')
disp = ADMPDispPmeForce(w['box'], w['cov'], bench.RC, 1e-4, 10)
for k in ('K1','K2','K3'): disp.update_env(k, w['K'])
fr = bench.ThermalFrames(w, torch.device('cuda',0))
seq = [fr.step_frame(k) for k in range(12)]
mS = par['mScales']
disp.update_neighbors(seq[0], w['box'], rc=bench.RC + bench.SKIN)
for mode in ('same', 'moving'):
    for _ in range(2): disp.get_forces(seq[0], w['box'], None, cl, mS)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(10):
        disp.get_forces(seq[k if mode == 'moving' else 0], w['box'], None, cl, mS)
    torch.cuda.synchronize(); print(mode, (time.perf_counter() - t0) / 10 * 1e3, 'ms/call')
disp.profile(True); disp.profile_reset()
for k in range(5): disp.get_forces(seq[k], w['box'], None, cl, mS)
rep = disp.profile_report(); disp.profile(False)
print('  '.join('%s %.4f' % (k, v[0] / 5) for k, v in sorted(rep.items())))
# the same calls on a table borrowed from the PME calculator (site classes compiled in)
f, a = bench.make_force(w)
f.update_neighbors(seq[0], w['box'], rc=bench.RC + bench.SKIN)
U = None
for k in range(3):
    bench.step(f, a, U, seq[k], pairs=None); U = f.U_ind
disp.share_neighbors(f)
for _ in range(2): disp.get_forces(seq[0], w['box'], None, cl, mS)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(10):
    disp.get_forces(seq[k], w['box'], None, cl, mS)
torch.cuda.synchronize(); print('borrowed table', (time.perf_counter() - t0) / 10 * 1e3, 'ms/call')
disp.profile(True); disp.profile_reset()
for k in range(5): disp.get_forces(seq[k], w['box'], None, cl, mS)
rep = disp.profile_report(); disp.profile(False)
print('  '.join('%s %.4f' % (k, v[0] / 5) for k, v in sorted(rep.items())))
