#!/usr/bin/env python
"""Counterpart of the reference's examples/water_1024/run_admp.py on the MI355X path: non-polarizable multipolar PME,
dispersion PME and Tang-Toennies damping of a 1024-water box.  Statement for statement the reference's driver
(run_admp.py:20-139) -- PDB and force-field XML read from the working directory, multipoles / axes / covalent map assembled
from them, neighbour list, the three calculators -- with numpy in the place of jax.numpy, admp_amd.neighbor in the place of
jax_md, and `value_and_grad` from admp.pairwise.

    python examples/make_inputs.py && cd examples/water_1024 && python run_admp.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as jnp                                                              # noqa: E402  (set-up arrays only)
import admp.settings                                                             # noqa: E402,F401
from admp.multipole import convert_cart2harm                                     # noqa: E402
from admp.pme import ADMPPmeForce                                                # noqa: E402
from admp.disp_pme import ADMPDispPmeForce                                       # noqa: E402
from admp.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad      # noqa: E402
from admp.parser import *                                                        # noqa: E402,F401,F403
from admp_amd.neighbor import NeighborList                                       # noqa: E402  (jax_md.partition in the reference)

if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    pdb = os.path.join(here, 'water1024.pdb')
    xml = os.path.join(here, 'mpidwater.xml')
    pdbinfo = read_pdb(pdb)
    serials = pdbinfo['serials']
    names = pdbinfo['names']
    resNames = pdbinfo['resNames']
    resSeqs = pdbinfo['resSeqs']
    positions = pdbinfo['positions']
    box = pdbinfo['box']              # a, b, c, alpha, beta, gamma
    charges = pdbinfo['charges']
    positions = jnp.asarray(positions)
    lx, ly, lz, _, _, _ = box
    box = jnp.eye(3) * jnp.array([lx, ly, lz])

    mScales = jnp.array([0.0, 0.0, 0.0, 1.0, 1.0])
    pScales = jnp.array([0.0, 0.0, 0.0, 1.0, 1.0])
    dScales = jnp.array([0.0, 0.0, 0.0, 1.0, 1.0])

    rc = 4        # in Angstrom
    ethresh = 1e-4

    n_atoms = len(serials)

    atomTemplate, residueTemplate = read_xml(xml)
    atomDicts, residueDicts = init_residues(serials, names, resNames, resSeqs, positions, charges, atomTemplate,
                                            residueTemplate)

    # e, e nm -> e A, e nm^2 -> e A^2 (the reference's factors, run_admp.py:49-51)
    Q = np.vstack([(atom.c0, atom.dX * 10, atom.dY * 10, atom.dZ * 10, atom.qXX * 300, atom.qYY * 300, atom.qZZ * 300,
                    atom.qXY * 300, atom.qXZ * 300, atom.qYZ * 300) for atom in atomDicts.values()])
    Q_local = convert_cart2harm(Q, 2)
    axis_type = np.array([atom.axisType for atom in atomDicts.values()])
    axis_indices = np.vstack([atom.axis_indices for atom in atomDicts.values()])
    covalent_map = assemble_covalent(residueDicts, n_atoms)

    lmax = 2
    pmax = 10

    # dispersion coefficients and Tang-Toennies parameters per atom (run_admp.py:66-97)
    c_list = np.zeros((3, n_atoms))
    a_list = np.zeros(n_atoms)
    q_list = np.zeros(n_atoms)
    b_list = np.zeros(n_atoms)
    nmol = int(n_atoms / 3)
    for i in range(nmol):
        a, b, c = i * 3, i * 3 + 1, i * 3 + 2
        c_list[0][[a, b, c]] = 37.19677405, 7.6111103, 7.6111103
        c_list[1][[a, b, c]] = 85.26810658, 11.90220148, 11.90220148
        c_list[2][[a, b, c]] = 134.44874488, 15.05074749, 15.05074749
        q_list[[a, b, c]] = -0.741706, 0.370853, 0.370853
        b_list[[a, b, c]] = 2.00095977, 1.999519942, 1.999519942      # Bohr^-1
        a_list[[a, b, c]] = 458.3777, 0.0317, 0.0317                  # Hartree
    c_list = c_list.T

    # neighbour list (jax_md.partition.neighbor_list(...).allocate(positions) in the reference)
    neighbor_list_fn = NeighborList(box, rc)
    nbr = neighbor_list_fn.allocate(positions)
    pairs = nbr

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
    E, F = TT_damping_qq_c6(positions, box, pairs, mScales, a_list, b_list, q_list, c_list[:, 0])
    print('Tang-Tonnies Damping (kJ/mol)')
    print(E)
