#!/usr/bin/env python
"""Counterpart of the reference's examples/water_1024/run_admp.py on the MI355X path:
non-polarizable multipolar PME, dispersion PME and Tang-Toennies damping of a 1024-water box.

    python examples/water_1024/run_admp.py [water1024.pdb]

Without a PDB argument the geometry of the reference example is taken from the committed fixture
tests/golden/p1_water1024.npz (the reference tree itself is not shipped).  Data preparation follows
run_admp.py:23-112 of the reference (MPID water parameters, axis types, covalent map, rc = 4 A,
ethresh = 1e-4, kappa override), with `admp_amd.systems` standing in for admp.parser / jax_md.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from admp_amd import systems as S                                                   # noqa: E402
from admp_amd.pme import ADMPPmeForce                                               # noqa: E402
from admp_amd.disp_pme import ADMPDispPmeForce                                      # noqa: E402
from admp_amd.pairwise import (generate_pairwise_interaction, TT_damping_qq_c6_kernel,   # noqa: E402
                               value_and_grad)

if __name__ == '__main__':
    if len(sys.argv) > 1:
        positions, box = S.load_pdb_positions(sys.argv[1])
    else:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'p1_water1024.npz'))
        positions, box = g['positions'], g['box']
    n_mol = len(positions) // 3
    axis_type, axis_indices, covalent_map = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, polarizable=False)
    Q_local, c_list = par['Q_local'], par['c_list']
    mScales = par['mScales']
    rc, ethresh, lmax, pmax = 4, 1e-4, 2, 10

    pairs = S.build_pairs(positions, box, rc)          # jax_md neighbour list in the reference (:109-112)

    # electrostatic
    pme_force = ADMPPmeForce(box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax)
    pme_force.update_env('kappa', 0.657065221219616)
    E, F = pme_force.get_forces(positions, box, pairs, Q_local, mScales)
    print('Electrostatic Energy (kJ/mol)')
    print(E)

    # dispersion
    disp_pme_force = ADMPDispPmeForce(box, covalent_map, rc, ethresh, pmax)
    disp_pme_force.update_env('kappa', 0.657065221219616)
    E, F = disp_pme_force.get_forces(positions, box, pairs, c_list, mScales)
    print('Dispersion Energy (kJ/mol)')
    print(E)

    # short range damping
    TT_damping_qq_c6 = value_and_grad(generate_pairwise_interaction(TT_damping_qq_c6_kernel, covalent_map, static_args={}))
    E, F = TT_damping_qq_c6(positions, box, pairs, mScales, par['a_list'], par['b_list'], par['q_list'], c_list[:, 0])
    print('Tang-Tonnies Damping (kJ/mol)')
    print(E)
