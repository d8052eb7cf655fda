#!/usr/bin/env python
"""Counterpart of the reference's examples/water_pol_1024/run_admp.py: polarizable MPID water, induced
dipoles by Jacobi SCF, energy and gradient at the converged dipoles (warm-started second call).

    python examples/water_pol_1024/run_admp.py [box.pdb]

Default geometry: the seeded synthetic liquid box of 1024 waters (L = 31.289 A) -- on the reference's
shipped water1024.pdb (random placement, 0.67 A contacts) the Jacobi SCF of the reference itself diverges
(SURVEY.md section 4), so that file is only useful for the non-polarizable example.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from admp_amd import systems as S            # noqa: E402
from admp_amd.pme import ADMPPmeForce        # noqa: E402

if __name__ == '__main__':
    if len(sys.argv) > 1:
        positions, box = S.load_pdb_positions(sys.argv[1])
    else:
        positions, box = S.synthetic_water_box(1024, seed=20240)
    n_mol = len(positions) // 3
    axis_type, axis_indices, covalent_map = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, polarizable=True)
    Q_local, pol, tholes = par['Q_local'], par['pol'], par['tholes']
    mScales, pScales, dScales = par['mScales'], par['pScales'], par['dScales']
    rc, ethresh, lmax = 4, 1e-4, 2

    pairs = S.build_pairs(positions, box, rc)

    pme_force = ADMPPmeForce(box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax, lpol=True)
    pme_force.update_env('kappa', 0.657065221219616)
    E, F = pme_force.get_forces(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales)
    print('# Electrostatic Energy (kJ/mol)')
    E, F = pme_force.get_forces(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales,
                                U_init=pme_force.U_ind)
    print(E)
    print('# SCF cycles (warm start): %d, converged: %s' % (pme_force.n_cycle, pme_force.lconverg))
    U_ind = pme_force.U_ind
    print('# |U_ind| of the first three oxygens (e*A):', [float((U_ind[3 * i] ** 2).sum() ** 0.5) for i in range(3)])
