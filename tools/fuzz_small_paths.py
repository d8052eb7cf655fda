#!/usr/bin/env python3
"""Fuzz of the small-system paths of round 4 (spread inside the forward plane transform, closing work in the gather, last
chained residual in the closing gather, field kernels riding in the x pass, one stream) against the same library with all of
them switched off: random water boxes of several sizes on direct-DFT meshes, a few warm-started steps each (so that the
speculative / chained / plain SCF forms all occur), energies / gradient / dipoles / cycle counts compared.
    python tools/fuzz_small_paths.py            # runs both legs in child processes and compares"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OFF = dict(ADMP_FUSE_SPREAD_MAX='0', ADMP_FUSE_FIN_MAX='0', ADMP_FIELD_RIDER='0', ADMP_CHAIN_LAST_FIELD='1', ADMP_OVERLAP_MIN='0')
CASES = [(125, (97, 97, 97), 11), (343, (61, 97, 53), 12), (700, (97, 67, 101), 13), (1500, (31, 97, 97), 14), (2600, (97, 97, 97), 15),
         (1024, (113, 59, 71), 16)]


def leg(path):
    sys.path.insert(0, ROOT)
    import torch
    from admp_amd import settings, systems as S
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    settings.REFERENCE_KPOINT_ORDER = False
    out = {}
    for n_mol, K, seed in CASES:
        pos, box = S.synthetic_water_box(n_mol, seed=seed)
        at, ai, cov = S.water_topology(n_mol)
        par = S.water_parameters(n_mol, polarizable=True)
        pairs = S.build_pairs(pos, box, 4.0)
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        f.K1, f.K2, f.K3 = K
        f.refresh_calculators()
        rng = np.random.default_rng(seed)
        U = None
        noise = rng.normal(size=pos.shape)
        # growing displacements first (first checks fail: plain, then chained calls), then the same geometry again and again
        # (first checks pass: speculative calls)
        for step, scale in enumerate((0.0, 0.01, 0.02, 0.03, 0.04, 0.04, 0.04, 0.04, 0.045, 0.05)):
            p = pos + scale * noise
            E, G = f.get_forces(p, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'],
                                par['dScales'], U_init=U)
            U = f.U_ind
            key = '%d_%d' % (n_mol, step)
            out[key + '_E'] = np.asarray(f.energy_parts)
            out[key + '_G'] = np.asarray(G)
            out[key + '_U'] = np.asarray(U)
            out[key + '_c'] = np.asarray([f.n_cycle, int(f.lconverg)])
        out['%d_forms' % n_mol] = np.asarray([f.scf_stats()[k] for k in ('plain', 'speculative', 'chained')])
    np.savez(path, **out)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--leg':
        leg(sys.argv[2])
        sys.exit(0)
    res = {}
    for name, env in (('on', {}), ('off', OFF)):
        path = '/tmp/fuzz_%s.npz' % name
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--leg', path], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-3000:])
            sys.exit(1)
        res[name] = dict(np.load(path))
    worst = 0.0
    for k, a in res['on'].items():
        b = res['off'][k]
        if k.endswith('_forms'):
            print(k, 'forms on', a, 'off', b)
            continue
        if k.endswith('_c'):
            assert (a == b).all(), (k, a, b)
            continue
        d = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        worst = max(worst, d)
        assert d < 1e-9, (k, d)
    print('fuzz ok: %d arrays, worst relative difference %.2e' % (len(res['on']), worst))
