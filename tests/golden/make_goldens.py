"""Generate the committed golden fixtures of the PME path with the float64 oracle.

Run in the build container:  python tests/golden/make_goldens.py
  p1_water1024.npz   shipped example geometry of the reference (examples/water_1024/water1024.pdb: a DATA file,
                     3072 atoms, 50 A cube; positions are stored in the fixture because the reference tree does
                     not travel), rc 4, kappa 0.657065221219616, K 154: non-polarizable electrostatics,
                     dispersion PME (pmax 10) and Tang-Toennies energies + gradients.
  s1_water_pol.npz   seeded synthetic liquid box, 1024 waters, L 31.289: polarizable run (SCF from zero),
                     U_ind, n_cycle, energy parts, gradient, dE/dQ_local.
  toy_water2.npz     the 2-water toy of examples/water_pol_1024/water2.pdb, polarizable, rc 8.
The reference cannot be executed here (needs jax), so these are outputs of the restatement in oracle/,
whose pinning status is stated in oracle/__init__.py.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from admp_amd import systems as S          # noqa: E402
from oracle import admp_oracle as O        # noqa: E402

REF = '/root/reference'
KAPPA = 0.657065221219616                  # examples/water_1024/run_admp.py:116


def p1():
    pos, box = S.load_pdb_positions(os.path.join(REF, 'examples', 'water_1024', 'water1024.pdb'))
    nm = len(pos) // 3
    at, ai, cov = S.water_topology(nm)
    par = S.water_parameters(nm, polarizable=False)
    pairs = S.build_pairs(pos, box, 4.0)
    _, K1, K2, K3 = O.setup_ewald_parameters(4.0, 1e-4, box)
    sysm = O.PmeSystem(at, ai, cov, KAPPA, (K1, K2, K3), 2, False)
    es = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], want_dQ=True)
    dp = O.disp_energy_and_grad(pos, box, pairs, par['c_list'], par['mScales'], cov, KAPPA, (K1, K2, K3), 10)
    tt = O.tt_energy_and_grad(pos, box, pairs, par['mScales'], cov, par['a_list'], par['b_list'], par['q_list'],
                              par['c_list'][:, 0])
    np.savez_compressed(os.path.join(HERE, 'p1_water1024.npz'), positions=pos, box=box, pairs=pairs, rc=4.0,
                        kappa=KAPPA, K=np.array([K1, K2, K3]),
                        es_parts=np.array(es['parts']), es_grad=es['grad'], es_dQ=es['dQ_local'],
                        disp_parts=np.array(dp['parts']), disp_grad=dp['grad'], tt_E=tt['E'], tt_grad=tt['grad'])
    print('p1', es['E'], dp['E'], tt['E'])


def s1():
    nm = 1024
    pos, box = S.synthetic_water_box(nm, seed=20240)
    at, ai, cov = S.water_topology(nm)
    par = S.water_parameters(nm, polarizable=True)
    pairs = S.build_pairs(pos, box, 4.0)
    kappa, K1, K2, K3 = O.setup_ewald_parameters(4.0, 1e-4, box)
    sysm = O.PmeSystem(at, ai, cov, kappa, (K1, K2, K3), 2, True)
    r = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                              par['pScales'], want_dQ=True)
    np.savez_compressed(os.path.join(HERE, 's1_water_pol.npz'), n_mol=nm, seed=20240, box=box, n_pairs=len(pairs),
                        kappa=kappa, K=np.array([K1, K2, K3]), parts=np.array(r['parts']), grad=r['grad'],
                        dQ=r['dQ_local'], U_ind=r['U_ind'], n_cycle=r['n_cycle'], lconverg=r['lconverg'],
                        pos_checksum=np.array([pos.sum(), (pos ** 2).sum()]))
    print('s1', r['E'], r['parts'], r['n_cycle'])


def toy():
    pos, box = S.load_pdb_positions(os.path.join(REF, 'examples', 'water_pol_1024', 'water2.pdb'))
    at, ai, cov = S.water_topology(2)
    par = S.water_parameters(2, polarizable=True)
    pairs = S.build_pairs(pos, box, 8.0)
    kappa, K1, K2, K3 = O.setup_ewald_parameters(8.0, 1e-4, box)
    sysm = O.PmeSystem(at, ai, cov, kappa, (K1, K2, K3), 2, True)
    r = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                              par['pScales'])
    np.savez_compressed(os.path.join(HERE, 'toy_water2.npz'), positions=pos, box=box, pairs=pairs, rc=8.0, kappa=kappa,
                        K=np.array([K1, K2, K3]), parts=np.array(r['parts']), grad=r['grad'], U_ind=r['U_ind'],
                        n_cycle=r['n_cycle'])
    print('toy', r['E'], r['parts'], r['n_cycle'], r['U_ind'][0], r['U_ind'][3])
    # The one reference-held number of the hot path that belongs to a shipped geometry: examples/water_pol_1024/dipole_2,
    # the induced dipoles of this 2-water toy from the MPID OpenMM plugin (nm e; what the commented loop of run_admp.py:142-145
    # compared U_ind with).  Stored as data next to the positions it belongs to.
    import json
    mpid = np.loadtxt(os.path.join(REF, 'examples', 'water_pol_1024', 'dipole_2'))
    with open(os.path.join(HERE, 'ref_water2_mpid_dipoles.json'), 'w') as fh:
        json.dump({'source': 'examples/water_pol_1024/dipole_2 + water2.pdb of the reference (data files)',
                   'unit': 'nm e (x10 -> e A)', 'positions_A': pos.tolist(), 'box_A': box.tolist(),
                   'induced_dipoles_nm_e': mpid.tolist()}, fh, indent=1)


if __name__ == '__main__':
    t = time.time()
    toy()
    p1()
    s1()
    print('done in %.0f s' % (time.time() - t))
