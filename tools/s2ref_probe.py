import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
w2 = bench.make_workload('S2ref')
f2, a2 = bench.make_force(w2)
fr2 = bench.ThermalFrames(w2, torch.device('cuda', 0))
for rep in range(3):
    dtm, _, cycm = bench.run_timed(f2, a2, 10, 3, fr2, only=False)
    dts, _, cycs = bench.run_timed(f2, a2, 10, 3, None, only=False)
    print('moving %.3f %s' % (dtm / 10 * 1e3, cycm)); print('static %.3f %s %s' % (dts / 10 * 1e3, cycs, f2.scf_stats()))
