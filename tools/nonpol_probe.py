#!/usr/bin/env python3
"""Probe: non-polarizable get_forces on S3 (pair kernel at 3 waves/SIMD) for comparison with the polarizable one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from admp_amd import settings
from admp_amd.pme import ADMPPmeForce
w = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else 'S3')
settings.PRECISION = w['prec']
f = ADMPPmeForce(w['box'], w['at'], w['ai'], w['cov'], 4.0, 1e-4, 2, lpol=False)
for k in ('K1', 'K2', 'K3'):
    f.update_env(k, w['K'])
dt = torch.float32
p = torch.as_tensor(w['pos'], dtype=dt, device='cuda'); Q = torch.as_tensor(w['par']['Q_local'], dtype=dt, device='cuda')
for _ in range(2):
    f.get_forces(p, w['box'], w['pairs'], Q, w['par']['mScales'])
f.profile(True); f.profile_reset()
for _ in range(5):
    f.get_forces(p, w['box'], w['pairs'], Q, w['par']['mScales'])
print({k: round(v[0] / v[1], 4) for k, v in f.profile_report().items()})
