#!/usr/bin/env python
"""Counterpart of the reference's examples/openmm_api/run.py:40-46: the two potentials in the
potential_fn(positions, box, pairs, params) convention and their gradient with respect to the force-field parameters
(the reference prints param_grad['mScales']).  The OpenMM XML front-end is not rebuilt: the parameters come from
admp_amd.systems (the mpidwater / ADMPDispForce water values of the reference's examples).

    python examples/param_grad/run.py [--waters 1024]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from admp_amd import systems as S                                                             # noqa: E402
from admp_amd.api import pme_potential, disp_potential, param_gradient                       # noqa: E402
from admp_amd.disp_pme import ADMPDispPmeForce                                               # noqa: E402
from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel         # noqa: E402
from admp_amd.pme import ADMPPmeForce                                                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--waters', type=int, default=1024)
    n = ap.parse_args().waters
    pos, box = S.synthetic_water_box(n, seed=20240)
    at, ai, cov = S.water_topology(n)
    par = S.water_parameters(n, polarizable=True)
    pairs = S.build_pairs(pos, box, 4.0)
    rc = 4.0
    pme = ADMPPmeForce(box, at, ai, cov, rc, 1e-4, 2, lpol=True)
    disp = ADMPDispPmeForce(box, cov, rc, 1e-4, 10)
    tt = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})

    # dispersion + Tang-Toennies potential: per-atom-type tables in the reference's XML units (admp/api.py:185-193)
    types = np.tile([0, 1, 1], n)                              # O, H, H
    first = [0, 1]                                             # an atom of each type
    c = par['c_list']
    disp_params = {'mScales': par['mScales'],
                   'A': par['a_list'][first] * 2625.5, 'B': par['b_list'][first] / 0.0529177249, 'Q': par['q_list'][first],
                   'C6': c[first, 0] ** 2 / 1e6, 'C8': c[first, 1] ** 2 / 1e8, 'C10': c[first, 2] ** 2 / 1e10}
    pot_disp = disp_potential(disp, tt, types)
    print(pot_disp(pos, box, pairs, disp_params))
    print(param_gradient(pot_disp, pos, box, pairs, disp_params)['mScales'])

    pme_params = {'mScales': par['mScales'], 'pScales': par['pScales'], 'dScales': par['dScales'],
                  'Q_local': par['Q_local'], 'U_ind': None}
    pot_pme = pme_potential(pme, par['pol'], par['tholes'])
    print(pot_pme(pos, box, pairs, pme_params))
    g = param_gradient(pot_pme, pos, box, pairs, pme_params)
    print(g['mScales'])
    print('dE/dQ_local', np.asarray(g['Q_local']).shape, 'dE/dpol(O)', float(np.asarray(g['pol'])[0]),
          'dE/dthole(O)', float(np.asarray(g['tholes'])[0]))


if __name__ == '__main__':
    main()
