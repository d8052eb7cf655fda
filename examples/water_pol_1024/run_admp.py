#!/usr/bin/env python
"""Counterpart of the reference's examples/water_pol_1024/run_admp.py: polarizable MPID water, induced dipoles by Jacobi
SCF, energy and gradient at the converged dipoles (warm-started second call).  The reference's statements
(run_admp.py:19-141): inputs read from `water1024.pdb` / `mpidwater.xml` next to the script (examples/make_inputs.py writes
them; the geometry is the seeded synthetic liquid box -- on the reference's shipped random placement, 0.67 A contacts, the
Jacobi SCF of the reference itself diverges).

    python examples/make_inputs.py && python examples/water_pol_1024/run_admp.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as jnp                                                              # noqa: E402  (set-up arrays only)
import admp.settings                                                             # noqa: E402,F401
from admp.multipole import convert_cart2harm                                     # noqa: E402
from admp.pme import ADMPPmeForce                                                # noqa: E402
from admp.parser import *                                                        # noqa: E402,F401,F403
from admp_amd.neighbor import NeighborList                                       # noqa: E402

if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    pdbinfo = read_pdb(os.path.join(here, 'water1024.pdb'))
    serials, names, resNames, resSeqs = (pdbinfo[k] for k in ('serials', 'names', 'resNames', 'resSeqs'))
    positions = jnp.asarray(pdbinfo['positions'])
    charges = pdbinfo['charges']
    lx, ly, lz, _, _, _ = pdbinfo['box']
    box = jnp.eye(3) * jnp.array([lx, ly, lz])

    mScales = jnp.array([0.0, 0.0, 0.0, 1.0, 1.0])
    pScales = jnp.array([0.0, 0.0, 0.0, 1.0, 1.0])
    dScales = jnp.array([0.0, 0.0, 0.0, 1.0, 1.0])
    rc = 4        # in Angstrom
    ethresh = 1e-4
    n_atoms = len(serials)

    atomTemplate, residueTemplate = read_xml(os.path.join(here, 'mpidwater.xml'))
    atomDicts, residueDicts = init_residues(serials, names, resNames, resSeqs, positions, charges, atomTemplate,
                                            residueTemplate)
    Q = np.vstack([(atom.c0, atom.dX * 10, atom.dY * 10, atom.dZ * 10, atom.qXX * 300, atom.qYY * 300, atom.qZZ * 300,
                    atom.qXY * 300, atom.qXZ * 300, atom.qYZ * 300) for atom in atomDicts.values()])
    Q_local = convert_cart2harm(Q, 2)
    axis_type = np.array([atom.axisType for atom in atomDicts.values()])
    axis_indices = np.vstack([atom.axis_indices for atom in atomDicts.values()])
    covalent_map = assemble_covalent(residueDicts, n_atoms)

    # induction parameters: nm^3 -> A^3 through float32, like the reference (run_admp.py:58-69)
    pol = np.vstack([(atom.polarizabilityXX, atom.polarizabilityYY, atom.polarizabilityZZ) for atom in atomDicts.values()])
    pol = 1000 * jnp.mean(pol.astype(np.float32), axis=1)
    tholes = np.vstack([atom.thole for atom in atomDicts.values()])
    tholes = jnp.mean(tholes.astype(np.float32), axis=1)

    lmax = 2
    pairs = NeighborList(box, rc).allocate(positions)

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
