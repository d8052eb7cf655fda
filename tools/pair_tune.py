#!/usr/bin/env python3
"""Time the pair kernels on a synthetic box for one (LPR, MINW) setting (read from the environment by the library).
usage: ADMP_PAIR_LPR=8 ADMP_PAIR_MINW=1 python tools/pair_tune.py [S2|S3] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

name = sys.argv[1] if len(sys.argv) > 1 else 'S2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = bench.make_workload(name)
f, a = bench.make_force(w)
dt, rep, cyc = bench.run_timed(f, a, steps, 2)
print('LPR=%s MINW=%s %s: step %.3f ms | ' % (os.environ.get('ADMP_PAIR_LPR', '8'), os.environ.get('ADMP_PAIR_MINW', '1'),
                                          name, dt / steps * 1e3) +
      ' '.join('%s=%.4f' % (k, v[0] / max(v[1], 1)) for k, v in sorted(rep.items())))
