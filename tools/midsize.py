#!/usr/bin/env python3
"""Step time and spread time of mid-size boxes (between the scan-spread and brick-spread regimes).
usage: python tools/midsize.py n_waters [single|double]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]); prec = sys.argv[2] if len(sys.argv) > 2 else 'single'
bench.WORKLOADS['X'] = (n, prec, None, 'mid-size box')
w = bench.make_workload('X')
f, a = bench.make_force(w)
dt, rep, cyc = bench.run_timed(f, a, 30, 5)
kb = bench.kernel_breakdown(f, a, None, 0, 10)[0]
print('%d atoms %s K=%d scan_max=%s brick_min=%s: %.4f ms/step, spread %.4f ms, kernels %s' % (
    3 * n, prec, f.K1, os.environ.get('ADMP_SPREAD_SCAN_MAX', '8192'), os.environ.get('ADMP_SPREAD_BRICK_MIN', '20000'),
    dt / 30 * 1e3, kb.get('spread', 0), kb))
