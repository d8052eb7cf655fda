#!/usr/bin/env python3
"""Golden fixture of BASELINE configs[2] AT SIZE: 98 304-atom seeded water box, K = 128, rc 4 A, from the float64 CPU oracle
(chunked evaluation, oracle/admp_oracle.py CHUNK_*): energy parts + gradient, non-polarizable and polarizable (SCF from zero
with the reference's defaults: POL_CONV 10, 30 cycles).  Written to tests/golden/s2_98304.npz; gradients and dipoles are
stored as float32 (1.2 MB each; 6e-8 relative -- the GPU check's bar is 1e-2, its resolution ~1e-6).

    python tests/golden/make_s2_golden.py            (about an hour on 8 cores, < 20 GB)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from admp_amd import systems as S          # noqa: E402  (host-side input preparation only)
from oracle import admp_oracle as O        # noqa: E402

N_MOL, SEED, RC, K = 32768, 20240, 4.0, 128


def main():
    t0 = time.time()
    pos, box = S.synthetic_water_box(N_MOL, seed=SEED)
    at, ai, cov = S.water_topology(N_MOL)
    pairs = S.build_pairs(pos, box, RC)
    kappa = O.setup_ewald_parameters(RC, 1e-4, box)[0]
    O.CHUNK_ATOMS, O.CHUNK_PAIRS = 8192, 131072
    out = dict(n_mol=N_MOL, seed=SEED, rc=RC, K=K, kappa=kappa, n_pairs=len(pairs),
               pos_checksum=np.array([pos.sum(), (pos ** 2).sum()]))
    for lpol in (False, True):
        par = S.water_parameters(N_MOL, polarizable=lpol)
        sysm = O.PmeSystem(at, ai, cov, kappa, (K, K, K), 2, lpol)
        if lpol:
            hist = []
            r = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                                      par['pScales'])
            out.update(pol_parts=np.array(r['parts']), pol_grad=r['grad'].astype(np.float32),
                       pol_U=r['U_ind'].astype(np.float32), pol_n_cycle=r['n_cycle'], pol_lconverg=r['lconverg'])
        else:
            r = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'])
            out.update(np_parts=np.array(r['parts']), np_grad=r['grad'].astype(np.float32))
        print('lpol=%s E=%.6f parts=%s  %.0f s' % (lpol, r['E'], r['parts'], time.time() - t0), flush=True)
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 's2_98304.npz'), **out)
    print('done in %.0f s' % (time.time() - t0))


if __name__ == '__main__':
    main()
