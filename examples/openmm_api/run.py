#!/usr/bin/env python
"""Counterpart of the reference's examples/openmm_api/run.py: the force field of an XML file as generators, the two
potentials in the potential_fn(positions, box, pairs, params) convention and their gradients with respect to the force-field
parameters.  Same statements, with this package's front-end (admp.api: no OpenMM) and neighbour list in place of
openmm.app / jax_md, and `param_gradient` in place of `jax.grad(..., argnums=3)`.

    python examples/make_inputs.py && cd examples/openmm_api && python run.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from admp.api import Hamiltonian, Topology, param_gradient        # noqa: E402
from admp_amd.neighbor import NeighborList                        # noqa: E402

if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    H = Hamiltonian(os.path.join(here, 'forcefield.xml'))
    pdb = Topology.from_pdb(os.path.join(here, 'water1024.pdb'))
    rc = 4.0
    # generator stores all force field parameters
    generator = H.getGenerators()
    disp_generator = generator[0]
    pme_generator = generator[1]
    potentials = H.createPotential(pdb, nonbondedCutoff=rc)
    # pot_fn is the actual energy calculator
    pot_disp = potentials[0]
    pot_pme = potentials[1]

    # construct inputs
    positions = pdb.positions
    box = pdb.box
    # neighbor list
    pairs = NeighborList(box, rc).allocate(positions)

    print(pot_disp(positions, box, pairs, disp_generator.params))
    param_grad = param_gradient(pot_disp, positions, box, pairs, generator[0].params)
    print(param_grad['mScales'])
    print({k: np.asarray(param_grad[k]) for k in ('A', 'B', 'Q', 'C6', 'C8', 'C10')})

    print(pot_pme(positions, box, pairs, pme_generator.params))
    param_grad = param_gradient(pot_pme, positions, box, pairs, generator[1].params)
    print(param_grad['mScales'])
