#!/usr/bin/env python
"""Velocity-Verlet NVE loop on the full MPID-style water potential: multipolar (optionally polarizable) PME +
dispersion PME + Tang-Toennies damping from libadmp_hip, harmonic bonds / angles (examples' mpidwater.xml:16-21)
and the integrator as HIP kernels too (admp_amd/md.py: one kernel for the bonded terms, one per half step; round 3 spent
~60 torch launches per step on them).  The reference has no integrator; this driver turns the hot
path into a real MD loop: it reports the energy drift (a direct check that the hand-coded adjoints are the gradient of the
energies) and the achieved ns/day including neighbour rebuilds.  Per step the host waits only where the calculators
themselves do (each returns its energy as a number, like the reference's get_forces); the bonded and kinetic energies
stay on the device until a line is logged.

    python examples/md/nve_water.py [--waters 1024] [--steps 200] [--dt 0.5] [--pol] [--single] [--mesh K] [--log 10] [--prune M] [--predict]

--mesh K: K1 = K2 = K3 = K instead of the reference's rule (admp/pme.py:146-172), e.g. 128 for the 98 304-atom box of
BASELINE configs[2] (the rule gives 305 = 5 * 61 there: every convolution then runs on the two-level DFT kernels, 0.7 ms
each -- with a tight SCF that, not the host, is most of a step).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from admp_amd import settings, systems as S                                                   # noqa: E402

KB = 0.0083144626          # kJ/mol/K
MASS = (15.999, 1.008, 1.008)
K_BOND, R0 = 376560.0 / 100.0, 0.9572          # kJ/mol/A^2 (xml: per nm^2)
K_ANG, TH0 = 460.24, 1.82421813418


def bonded(pos, n_mol):
    """torch restatement of the bonded terms (the checker of tests/test_gpu_examples.py; the loop uses admp_amd.md)"""
    m = pos.reshape(n_mol, 3, 3)
    a, b = m[:, 1] - m[:, 0], m[:, 2] - m[:, 0]
    ra, rb = a.norm(dim=1), b.norm(dim=1)
    th = torch.acos(torch.clamp((a * b).sum(1) / (ra * rb), -1.0, 1.0))
    return 0.5 * K_BOND * ((ra - R0) ** 2 + (rb - R0) ** 2).sum() + 0.5 * K_ANG * ((th - TH0) ** 2).sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--waters', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--dt', type=float, default=0.5, help='fs')
    ap.add_argument('--pol', action='store_true')
    ap.add_argument('--single', action='store_true')
    ap.add_argument('--rebuild', type=int, default=10)
    ap.add_argument('--prune', type=int, default=0, help='steps between prunings of an inner list (admp_prune_pairs; 0: walk the whole skin '
                    'list).  Pays from ~200k atoms on; the drift of an NVE run grows with it: the multipolar kernels evaluate every listed '
                    'pair, and the set of pairs beyond rc changes at every prune')
    ap.add_argument('--minimize', type=int, default=200)
    ap.add_argument('--temp', type=float, default=300.0)
    ap.add_argument('--mesh', type=int, default=0, help='PME mesh size per dimension (0: the reference rule)')
    ap.add_argument('--log', type=int, default=10, help='steps between energy records (each costs two host reads)')
    ap.add_argument('--thresh', type=float, default=1e-2, help='SCF threshold of --pol (the reference default 10 does not conserve energy)')
    ap.add_argument('--predict', type=int, default=0, choices=(0, 1, 2, 3), help='start the SCF from a polynomial predictor over the last k + 1 '
                    'converged dipoles (Kolafa\'s always-stable predictor coefficients: k = 1 is 2 U(n-1) - U(n-2)) instead of U(n-1): a few '
                    'tensor operations per step; at a tight threshold it saves one to two field evaluations per step')
    opt = ap.parse_args()
    settings.PRECISION = 'single' if opt.single else 'double'
    if opt.pol:
        settings.POL_CONV = opt.thresh      # a tight SCF: the reference's default (10) is too loose for energy conservation
        settings.MAX_N_POL = 60
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    from admp_amd.md import HarmonicBonded, VelocityVerlet

    n_mol = opt.waters
    pos0, box = S.synthetic_water_box(n_mol, seed=20240)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, polarizable=opt.pol)
    dt = torch.float32 if opt.single else torch.float64
    dev = 'cuda'
    rc, skin = 4.0, 1.0
    pme = ADMPPmeForce(box, at, ai, cov, rc, 1e-4, 2, lpol=opt.pol)
    disp = ADMPDispPmeForce(box, cov, rc, 1e-4, 10)
    if opt.mesh:
        for obj in (pme, disp):
            for k in ('K1', 'K2', 'K3'):
                obj.update_env(k, opt.mesh)
    tt_obj = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
    tt = value_and_grad(tt_obj)
    o = 3 * np.arange(n_mol)
    bonds = np.stack([np.concatenate([o, o]), np.concatenate([o + 1, o + 2])], axis=1)
    angles = np.stack([o + 1, o, o + 2], axis=1)
    bond = HarmonicBonded(3 * n_mol, bonds, np.tile([K_BOND, R0], (2 * n_mol, 1)), angles, np.tile([K_ANG, TH0], (n_mol, 1)))

    class Lists:
        """Verlet lists with a skin, rebuilt every --rebuild steps (the pair kernels have no cutoff test of their own,
        like the reference).  The PME calculator compiles the neighbour table on the GPU straight from the positions
        (`update_neighbors`, search fused with the table build); the other two borrow it (`share_neighbors` -- the
        reference's drivers hand one `pairs` array to every force object) and all are then called with pairs=None."""
        def allocate(self, p):
            pme.update_neighbors(p, box, rc=rc + skin)
            for obj in (disp, tt_obj):
                obj.share_neighbors(pme)
            return None

        def prune(self, p):
            """inner list (admp_prune_pairs): the entries of the skin list within rc + the share of the skin the atoms can use up
            until the next prune; the multipolar kernels evaluate every listed pair, so a shorter list is less work"""
            pme.prune_neighbors(p, box, rc + skin * opt.prune / float(opt.rebuild))
    nbl = Lists()
    pos = torch.as_tensor(pos0, dtype=dt, device=dev)
    mass = torch.as_tensor(np.tile(MASS, n_mol), dtype=dt, device=dev)[:, None]
    T = lambda k: torch.as_tensor(par[k], dtype=dt, device=dev)      # noqa: E731
    Q, pol, thole, cl = T('Q_local'), T('pol'), T('tholes'), T('c_list')
    a_, b_, q_, c6 = T('a_list'), T('b_list'), T('q_list'), cl[:, 0].contiguous()
    mS, pS, dS = par['mScales'], par['pScales'], par['dScales']
    state = {'U': None}

    def forces(p, pairs):
        """(potential energy of the three calculators -- numbers they return anyway --, +dE/dr of everything); the bonded
        energy of this evaluation is in bond.energy_words (read by epot_now() when a line is logged)"""
        if opt.pol:
            U0 = state['U']
            hist = state.setdefault('hist', [])
            if opt.predict and len(hist) == opt.predict + 1:
                coef = {1: (2.0, -1.0), 2: (2.5, -2.0, 0.5), 3: (2.8, -2.8, 1.2, -0.2)}[opt.predict]
                U0 = coef[0] * hist[-1]
                for c, Uh in zip(coef[1:], reversed(hist[:-1])):
                    U0 = U0 + c * Uh
            e1, g = pme.get_forces(p, box, pairs, Q, pol, thole, mS, pS, dS, U_init=U0)
            state['U'] = pme.U_ind
            if opt.predict:
                hist.append(state['U'])
                del hist[:-(opt.predict + 1)]
            state['cyc'] = state.get('cyc', 0) + pme.n_cycle + 1
            state['n'] = state.get('n', 0) + 1
        else:
            e1, g = pme.get_forces(p, box, pairs, Q, mS)
        e2, g2 = disp.get_forces(p, box, pairs, cl, mS)
        e3, g3 = tt(p, box, pairs, mS, a_, b_, q_, c6)
        g.add_(g2).add_(g3)
        bond.reset_energy()
        bond.add_forces(p, box, g)
        return e1 + e2 + e3, g

    def epot_now(e123):
        return float(e123) + bond.energy()

    g = torch.Generator(device=dev).manual_seed(1)
    # units: A, fs, amu, kJ/mol.  1 kJ/mol/amu = (1e-2 A/fs)^2;  1 (kJ/mol/A)/amu = 1e-4 A/fs^2
    vel = torch.randn(pos.shape, generator=g, device=dev, dtype=dt) * torch.sqrt(KB * opt.temp / mass) * 1e-2
    acc_unit = 1e-4
    h = opt.dt
    vv = VelocityVerlet(pme, np.tile(MASS, n_mol), h)
    pairs = nbl.allocate(pos)
    e123, grad = forces(pos, pairs)
    for it in range(opt.minimize):          # the synthetic box is not equilibrated: capped steepest descent first
        pos = pos - grad * (0.02 / max(float(grad.norm(dim=1).max()), 1e-12))
        if (it + 1) % opt.rebuild == 0:
            pairs = nbl.allocate(pos)
        e123, grad = forces(pos, pairs)
        if it % 50 == 0 or it == opt.minimize - 1:
            print('minimize %4d  Epot %14.4f' % (it, epot_now(e123)))
    pairs = nbl.allocate(pos)
    e123, grad = forces(pos, pairs)
    pos, vel = pos.contiguous(), vel.contiguous()
    log = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(opt.steps):
        vv.kick_drift(pos, vel, grad)                                  # v(t + h/2), r(t + h): in place, one kernel
        if (step + 1) % opt.rebuild == 0:
            pairs = nbl.allocate(pos)
        if opt.prune and (step + 1) % opt.prune == 0:
            nbl.prune(pos)
        e123, grad = forces(pos, pairs)
        rec = step % opt.log == 0 or step == opt.steps - 1
        vv.kick(pos, vel, grad, want_ekin=rec)                         # v(t + h)
        if rec:
            epot, ekin = epot_now(e123), vv.kinetic_energy()
            log.append((step, epot, ekin, epot + ekin))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    e0 = log[0][3]
    for (s, ep, ek, et) in log:
        print('step %5d  Epot %14.4f  Ekin %12.4f  Etot %14.4f  drift %+.3e' % (s, ep, ek, et, (et - e0) / abs(e0)))
    ns_day = opt.steps * h * 1e-6 / wall * 86400.0
    if opt.pol:
        print('# mean SCF cycles per evaluation (thresh %g): %.1f' % (settings.POL_CONV, state['cyc'] / state['n']))
        # which form the library chose for every polarizable call (residual history, engine.hip pme()) and what wrong guesses cost
        print('# SCF forms over the run (real dynamics): %s' % pme.scf_stats())
    print('# %d waters, %s, %s, dt %.2f fs: %.3f ms/step, %.2f ns/day (all terms, list rebuilt every %d steps); '
          'relative energy drift %.2e, T_final %.1f K' % (n_mol, 'polarizable' if opt.pol else 'fixed multipoles',
                                                         settings.PRECISION, h, wall / opt.steps * 1e3, ns_day, opt.rebuild,
                                                         (log[-1][3] - e0) / abs(e0), 2 * log[-1][2] / (3 * 3 * n_mol * KB)))


if __name__ == '__main__':
    main()
