"""Parity of the HIP path (through the C ABI, via the admp_amd calculators) against the oracle
and the committed golden fixtures.  Tolerances (BASELINE.json north_star): 1e-4 relative in double,
1e-2 in single; the double-precision checks below are far tighter because both sides evaluate the
same formulas in float64."""
import os

import numpy as np
import pytest

from admp_amd import settings
from admp_amd import systems as S

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.fixture()
def precision():
    old = settings.PRECISION
    yield
    settings.PRECISION = old


def water_system(n_mol, seed, polarizable, rc=4.0):
    pos, box = S.synthetic_water_box(n_mol, seed=seed)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, polarizable=polarizable)
    pairs = S.build_pairs(pos, box, rc)
    return pos, box, at, ai, cov, par, pairs


def oracle_es(pos, box, at, ai, cov, par, pairs, kappa, K, lpol, want_dQ=False):
    from oracle import admp_oracle as O
    sysm = O.PmeSystem(at, ai, cov, kappa, K, 2, lpol)
    if lpol:
        return O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                                     par['pScales'], want_dQ=want_dQ)
    return O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], want_dQ=want_dQ)


@pytest.mark.parametrize('prec,tolE,tolG', [('double', 1e-9, 1e-8), ('single', 2e-4, 2e-4)])
def test_nonpolarizable_water_vs_oracle(precision, prec, tolE, tolG):
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = prec
    pos, box, at, ai, cov, par, pairs = water_system(216, 11, False)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    E, G, dQ = f.get_forces_and_dQ(pos, box, pairs, par['Q_local'], par['mScales'])
    ref = oracle_es(pos, box, at, ai, cov, par, pairs, f.kappa, (f.K1, f.K2, f.K3), False, want_dQ=True)
    scale = max(abs(p) for p in ref['parts'])
    for got, want in zip(f.energy_parts, ref['parts']):
        assert abs(got - want) <= tolE * scale
    assert rel(G, ref['grad']) < tolG
    assert rel(dQ, ref['dQ_local']) < tolG
    assert f.n_pairs == len(pairs)
    # energy-only entry point gives the same number
    assert abs(f.get_energy(pos, box, pairs, par['Q_local'], par['mScales']) - E) <= 1e-12 * scale + tolE * scale


@pytest.mark.parametrize('prec,tolE,tolG', [('double', 1e-9, 1e-8), ('single', 2e-4, 5e-4)])
def test_polarizable_water_vs_oracle(precision, prec, tolE, tolG):
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = prec
    pos, box, at, ai, cov, par, pairs = water_system(125, 5, True)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    args = (pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    E, G = f.get_forces(*args)
    ref = oracle_es(pos, box, at, ai, cov, par, pairs, f.kappa, (f.K1, f.K2, f.K3), True)
    scale = max(abs(p) for p in ref['parts'])
    assert f.n_cycle == ref['n_cycle'] and f.lconverg == ref['lconverg']
    for got, want in zip(f.energy_parts, ref['parts']):
        assert abs(got - want) <= tolE * scale
    assert rel(f.U_ind, ref['U_ind']) < tolG
    assert rel(G, ref['grad']) < tolG
    # warm start from the converged dipoles: zero further cycles needed beyond the first check
    E2, G2 = f.get_forces(*args, U_init=f.U_ind)
    assert f.n_cycle <= ref['n_cycle']
    # optimize_Uind alone
    U, flag, i = f.optimize_Uind(*args)
    assert i == ref['n_cycle'] and flag == ref['lconverg'] and rel(U, ref['U_ind']) < tolG


def test_toy_two_waters_golden():
    from admp_amd.pme import ADMPPmeForce
    g = np.load(os.path.join(GOLD, 'toy_water2.npz'))
    at, ai, cov = S.water_topology(2)
    par = S.water_parameters(2, True)
    f = ADMPPmeForce(g['box'], at, ai, cov.toarray(), float(g['rc']), 1e-4, 2, lpol=True)
    assert (f.K1, f.K2, f.K3) == tuple(int(k) for k in g['K'])
    E, G = f.get_forces(g['positions'], g['box'], g['pairs'], par['Q_local'], par['pol'], par['tholes'], par['mScales'],
                        par['pScales'], par['dScales'])
    scale = np.abs(g['parts']).max()
    np.testing.assert_allclose(f.energy_parts, g['parts'], atol=1e-9 * scale)
    assert rel(G, g['grad']) < 1e-8 and rel(f.U_ind, g['U_ind']) < 1e-8 and f.n_cycle == int(g['n_cycle'])


def test_p1_water1024_golden_all_terms():
    """Reference example geometry (water_1024): electrostatics, dispersion PME and Tang-Toennies."""
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    g = np.load(os.path.join(GOLD, 'p1_water1024.npz'))
    pos, box, pairs = g['positions'], g['box'], g['pairs']
    nm = len(pos) // 3
    at, ai, cov = S.water_topology(nm)
    par = S.water_parameters(nm, False)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    f.update_env('kappa', float(g['kappa']))
    assert (f.K1, f.K2, f.K3) == tuple(int(k) for k in g['K'])
    E, G, dQ = f.get_forces_and_dQ(pos, box, pairs, par['Q_local'], par['mScales'])
    scale = np.abs(g['es_parts']).max()
    np.testing.assert_allclose(f.energy_parts[:3], g['es_parts'][:3], atol=1e-9 * scale)
    assert abs(E - g['es_parts'].sum()) < 1e-9 * scale
    assert rel(G, g['es_grad']) < 1e-8 and rel(dQ, g['es_dQ']) < 1e-8
    d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
    d.update_env('kappa', float(g['kappa']))
    Ed, Gd = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
    np.testing.assert_allclose(d.energy_parts, g['disp_parts'], rtol=1e-9)
    assert rel(Gd, g['disp_grad']) < 1e-8
    tt = value_and_grad(generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={}))
    Et, Gt = tt(pos, box, pairs, par['mScales'], par['a_list'], par['b_list'], par['q_list'], par['c_list'][:, 0])
    assert abs(Et - float(g['tt_E'])) < 1e-9 * abs(float(g['tt_E']))
    assert rel(Gt, g['tt_grad']) < 1e-8


def test_s1_polarizable_golden():
    from admp_amd.pme import ADMPPmeForce
    g = np.load(os.path.join(GOLD, 's1_water_pol.npz'))
    nm = int(g['n_mol'])
    pos, box = S.synthetic_water_box(nm, seed=int(g['seed']))
    np.testing.assert_allclose([pos.sum(), (pos ** 2).sum()], g['pos_checksum'], rtol=1e-12)
    at, ai, cov = S.water_topology(nm)
    par = S.water_parameters(nm, True)
    pairs = S.build_pairs(pos, box, 4.0)
    assert len(pairs) == int(g['n_pairs'])
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    E, G, dQ = f.get_forces_and_dQ(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'],
                                   par['pScales'], par['dScales'])
    scale = np.abs(g['parts']).max()
    np.testing.assert_allclose(f.energy_parts, g['parts'], atol=1e-9 * scale)
    assert f.n_cycle == int(g['n_cycle']) and f.lconverg == bool(g['lconverg'])
    assert rel(G, g['grad']) < 1e-8 and rel(f.U_ind, g['U_ind']) < 1e-8 and rel(dQ, g['dQ']) < 1e-8


def test_brick_spread_path_vs_oracle(precision):
    """The LDS-brick spread (default only above 20k atoms) forced on a small box, in a child process because the
    switch is read once per process; both precisions against the oracle."""
    import subprocess
    import sys
    code = """
import os, sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_parity import water_system, oracle_es, rel
from admp_amd import settings
from admp_amd.pme import ADMPPmeForce
pos, box, at, ai, cov, par, pairs = water_system(216, 11, True)
ref = None
for prec, tol in (('double', 1e-8), ('single', 5e-4)):
    settings.PRECISION = prec
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    if ref is None:
        ref = oracle_es(pos, box, at, ai, cov, par, pairs, f.kappa, (f.K1, f.K2, f.K3), True)
    scale = max(abs(p) for p in ref['parts'])
    assert abs(f.energy_parts[1] - ref['parts'][1]) < tol * scale, (prec, f.energy_parts, ref['parts'])
    assert rel(G, ref['grad']) < tol and rel(f.U_ind, ref['U_ind']) < tol and f.n_cycle == ref['n_cycle'], prec
# dispersion PME through the binned bricks (the brick lists of the first power are reused by the next ones)
from admp_amd.disp_pme import ADMPDispPmeForce
from oracle import admp_oracle as O
dref = None
for prec, tol in (('double', 1e-9), ('single', 5e-4)):
    settings.PRECISION = prec
    d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
    # direct-DFT mesh (58 = 2 * 29), a rocFFT one, and a power-of-two one: there the channels go through the batched y-z plans,
    # one x pass with a table per channel, the interleave pass and ONE gather (round 4)
    for K in ((d.K1, d.K2, d.K3), (60, 60, 60), (64, 64, 64)):
        d.K1, d.K2, d.K3 = K
        d.refresh_calculators()
        E, G = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
        dref = O.disp_energy_and_grad(pos, box, pairs, par['c_list'], par['mScales'], cov, d.kappa, K, 10)
        assert abs(E - dref['E']) < tol * max(abs(p) for p in dref['parts']) and rel(G, dref['grad']) < max(tol, 1e-8), (prec, K)
    d8 = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 8)          # two channels through the batched path
    d8.K1 = d8.K2 = d8.K3 = 64
    d8.refresh_calculators()
    E, G = d8.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
    dref = O.disp_energy_and_grad(pos, box, pairs, par['c_list'], par['mScales'], cov, d8.kappa, (64, 64, 64), 8)
    assert abs(E - dref['E']) < tol * max(abs(p) for p in dref['parts']) and rel(G, dref['grad']) < max(tol, 1e-8), (prec, 'pmax 8')
    # typed meshes (round 4; single precision, power-of-two mesh, c_list kept as one device tensor): water has two distinct
    # coefficient rows -> two type meshes instead of three channel meshes; then three types; four fall back to the channels
    import torch
    from admp_amd import _lib
    cl = np.array(par['c_list'], dtype=np.float64)
    cases = {'2 types': cl.copy(), '3 types': cl.copy(), '4 types': cl.copy()}
    cases['3 types'][0::6] *= 1.25                       # every second oxygen
    cases['4 types'][0::6] *= 1.25
    cases['4 types'][1::6] *= 0.5
    for label, c in cases.items():
        dref = O.disp_energy_and_grad(pos, box, pairs, c, par['mScales'], cov, d.kappa, (64, 64, 64), 10)
        ct = torch.as_tensor(c, dtype=torch.float32 if prec == 'single' else torch.float64, device='cuda')
        res = {}
        for typed in (True, False):
            settings.DISP_TYPED_MESHES = typed
            dd = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
            dd.K1 = dd.K2 = dd.K3 = 64
            dd.refresh_calculators()
            for rep in range(2):                        # (second call: the cached coefficient tensor and its type table)
                E, G = dd.get_forces(pos, box, pairs, ct, par['mScales'])
            used = getattr(dd, '_types', None) is not None
            assert used == (typed and label != '4 types'), (prec, label, typed, used)      # (double: the table is set, the brick path keeps the powers)
            assert abs(E - dref['E']) < tol * max(abs(p) for p in dref['parts']) and rel(np.asarray(G), dref['grad']) < max(tol, 1e-8), (prec, label, typed)
            res[typed] = (E, np.asarray(G), dd.energy_parts)
        settings.DISP_TYPED_MESHES = True
        assert rel(res[True][1], res[False][1]) < (2e-5 if prec == 'single' else 1e-12), (prec, label)
        assert abs(res[True][2][1] - res[False][2][1]) < (2e-5 if prec == 'single' else 1e-12) * abs(res[False][2][1]), (prec, label)
    if prec == 'single':      # a type table that does not describe c_list makes the call fail instead of returning a wrong energy
        dd = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
        dd.K1 = dd.K2 = dd.K3 = 64
        dd.refresh_calculators()
        ct = torch.as_tensor(cl, dtype=torch.float32, device='cuda')
        dd.get_forces(pos, box, pairs, ct, par['mScales'])
        wrong = dd._types[1].copy()
        wrong[0, 0] *= 1.5
        _lib.check(dd._h, dd._L.admp_disp_set_types(dd._h, len(wrong), dd._ptr(dd._types[0]), _lib.darr(wrong)), 'set_types')
        try:
            dd.get_forces(pos, box, pairs, ct, par['mScales'])
            raise SystemExit('a wrong type table went unnoticed')
        except _lib.AdmpHipError as e:
            assert 'types' in str(e)
print('BRICK-OK')
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADMP_SPREAD_BRICK_MIN='0')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'BRICK-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_direct_dft_convolution_vs_rocfft(tmp_path):
    """Meshes with a Bluestein dimension go through dft_kernels.hip / pfa_kernels.hip instead of rocFFT (engine.hip setup_dft).
    The k-space legs must agree to round-off: polarizable PME and dispersion PME, even / odd / prime dimensions, both
    precisions; ADMP_DFT is read per handle, the child processes only keep the runs independent."""
    import subprocess
    import sys
    code = """
import os, sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_parity import water_system
from admp_amd import settings
settings.REFERENCE_KPOINT_ORDER = False      # unequal meshes: the transform legs are compared on a self-consistent Ewald sum
out = {}
pos, box, at, ai, cov, par, pairs = water_system(216, 5, True)
for prec in ('double', 'single'):
    settings.PRECISION = prec
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    for K in ((0, 0, 0), (31, 34, 38), (96, 100, 45), (97, 64, 51)):
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
        if K[0]:
            for o in (f, d):
                o.K1, o.K2, o.K3 = K
                o.refresh_calculators()
        E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
        Ed, Gd = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
        key = '%%s_%%d_%%d_%%d' %% ((prec,) + (f.K1, f.K2, f.K3))
        out[key + '_parts'] = np.asarray(f.energy_parts); out[key + '_G'] = np.asarray(G); out[key + '_U'] = np.asarray(f.U_ind)
        out[key + '_dparts'] = np.asarray(d.energy_parts); out[key + '_Gd'] = np.asarray(Gd)
        # the same coefficients as ONE device tensor: the wrapper finds the two atom types of water and the direct-DFT modes
        # transform one mesh per type instead of one per power (typed meshes, round 4; rocFFT meshes of this size keep the powers)
        import torch
        ct = torch.as_tensor(np.asarray(par['c_list']), dtype=torch.float64 if prec == 'double' else torch.float32, device='cuda')
        for rep in range(2):
            Et, Gt = d.get_forces(pos, box, pairs, ct, par['mScales'])
        out[key + '_tparts'] = np.asarray(d.energy_parts); out[key + '_Gt'] = np.asarray(Gt)
        out[key + '_typed'] = np.asarray([1.0 if getattr(d, '_types', None) is not None else 0.0])
# a denser case for the spread that the forward plane kernel does itself (double precision, dft_kernels.hip zy_plane_spread):
# 4500 atoms on 31 x planes = ~870 kept atoms per plane (several staging sub-rounds) and two scan rounds of 3072 atoms
settings.PRECISION = 'double'
from admp_amd.pme import ADMPPmeForce
pos, box, at, ai, cov, par, pairs = water_system(1500, 7, True)
f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
f.K1, f.K2, f.K3 = 31, 97, 97
f.refresh_calculators()
E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
out['double_dense_parts'] = np.asarray(f.energy_parts); out['double_dense_G'] = np.asarray(G); out['double_dense_U'] = np.asarray(f.U_ind)
out['double_dense_cycles'] = np.asarray([f.n_cycle])
f0 = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=False)      # ... and fixed multipoles (no dipole words in the rows)
f0.K1, f0.K2, f0.K3 = 31, 97, 97
f0.refresh_calculators()
par0 = __import__('admp_amd.systems', fromlist=['x']).water_parameters(1500, polarizable=False)
E0, G0, dQ0 = f0.get_forces_and_dQ(pos, box, pairs, par0['Q_local'], par0['mScales'])
out['double_dense0_parts'] = np.asarray(f0.energy_parts); out['double_dense0_G'] = np.asarray(G0); out['double_dense0_dQ'] = np.asarray(dQ0)
np.savez(sys.argv[1], **out)
print('DFT-RUN-OK')
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # 'dft': the direct line transforms of dft_kernels.hip
    # 'pfa': the two-level (Good-Thomas) kernels of pfa_kernels.hip forced onto these small meshes, every dimension split
    # that has a coprime split (34 = 2 * 17, 38 = 2 * 19, 96 = 32 * 3, 100 = 4 * 25, 45 = 9 * 5, 51 = 3 * 17; 31, 97, 64 plain)
    # 'dft' runs the z and y lines of a plane in one kernel where the plane fits the LDS; 'dft_passes' keeps them apart
    # 'dft_nospread': the plane kernels reading a mesh the spread kernel wrote (round 4: by default the forward plane kernel of a
    # double-precision system of <= 8192 atoms builds its planes from the sites); ADMP_FUSE_FIN_MAX=0 there as well: the
    # closing kernel on its own instead of in the gather's epilogue
    modes = {'rocfft': dict(ADMP_DFT='0'), 'dft': dict(ADMP_DFT='1'), 'dft_passes': dict(ADMP_DFT='1', ADMP_DFT_PLANES='0'),
             'dft_nospread': dict(ADMP_DFT='1', ADMP_FUSE_SPREAD_MAX='0', ADMP_FUSE_FIN_MAX='0'),
             'pfa': dict(ADMP_DFT='2', ADMP_PFA_MIN='0')}
    for mode, extra in modes.items():
        path = str(tmp_path / ('%s.npz' % mode))
        r = subprocess.run([sys.executable, '-c', code, path], capture_output=True, text=True,
                           env=dict(os.environ, **extra), timeout=900)
        assert r.returncode == 0 and 'DFT-RUN-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        res[mode] = dict(np.load(path))
    assert len(res['rocfft']) == 2 * 4 * 8 + 4 + 3
    for key, a in res['rocfft'].items():
        if key.endswith('_typed'):
            continue
        for mode in ('dft', 'dft_passes', 'dft_nospread', 'pfa'):
            b = res[mode][key]
            tol = 1e-10 if key.startswith('double') else 2e-4
            scale = np.abs(a).max()
            assert np.abs(a - b).max() <= tol * scale, (mode, key, np.abs(a - b).max(), scale)


def test_pair_list_conventions():
    """Padding rows (i >= j) are dropped (admp/pme.py:671); order of rows is irrelevant; torch inputs work."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    pos, box, at, ai, cov, par, pairs = water_system(64, 3, False)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    E0, G0 = f.get_forces(pos, box, pairs, par['Q_local'], par['mScales'])
    na = len(pos)
    rng = np.random.default_rng(0)
    padded = np.concatenate([pairs, np.full((37, 2), na, dtype=np.int32), pairs[:5, ::-1]], axis=0)
    padded = padded[rng.permutation(len(padded))]
    E1, G1 = f.get_forces(torch.as_tensor(pos).cuda(), torch.as_tensor(box), torch.as_tensor(padded).cuda(),
                          torch.as_tensor(par['Q_local']), torch.as_tensor(par['mScales']))
    assert isinstance(G1, torch.Tensor) and G1.is_cuda
    assert abs(E1 - E0) < 1e-9 * abs(f.energy_parts[0]) and rel(G1.cpu().numpy(), G0) < 1e-10
    assert f.n_pairs == len(pairs)


def test_invariances_at_scale():
    """Size-independent properties on a 98 304-atom box (config C scale), single precision:
    translation of all atoms by a lattice vector and by an arbitrary vector, and sum of the
    real-space + self-consistent parts of the gradient."""
    from admp_amd.pme import ADMPPmeForce
    settings_old = settings.PRECISION
    settings.PRECISION = 'single'
    try:
        nm = 32768
        pos, box = S.synthetic_water_box(nm, seed=20240)
        at, ai, cov = S.water_topology(nm)
        par = S.water_parameters(nm, False)
        pairs = S.build_pairs(pos, box, 4.0)
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
        for k in ('K1', 'K2', 'K3'):
            f.update_env(k, 128)
        E0, G0 = f.get_forces(pos, box, pairs, par['Q_local'], par['mScales'])
        E1, G1 = f.get_forces(pos + box[0] - 2 * box[2], box, pairs, par['Q_local'], par['mScales'])
        scale = max(abs(p) for p in f.energy_parts)
        assert abs(E1 - E0) < 2e-5 * scale and rel(G1, G0) < 2e-3
        # net force: only the PME mesh breaks momentum conservation, and only slightly
        assert np.abs(G0.sum(axis=0)).max() < 1e-3 * np.abs(G0).sum(axis=0).max()
    finally:
        settings.PRECISION = settings_old


def _run_slab_ranks(nranks, prec, lpol, n_mol=125, seed=5):
    """N python threads, one SlabPme each (all on this GPU), exchanging through the in-process communicator."""
    import threading
    import torch
    from admp_amd.parallel import SlabPme, ThreadComm
    settings.PRECISION = prec
    pos, box, at, ai, cov, par, pairs = water_system(n_mol, seed, lpol)
    world = ThreadComm.World(nranks)
    results, errors = [None] * nranks, []

    def work(rank):
        try:
            f = SlabPme(ThreadComm(world, rank), box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol, outputs='replicated')
            if lpol:
                E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'],
                                    par['pScales'], par['dScales'])
                results[rank] = (E, G, f.energy_parts, f.U_ind, f.n_cycle, f.lconverg, f.n_home)
            else:
                E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['mScales'])
                results[rank] = (E, G, f.energy_parts, None, 0, True, f.n_home)
        except Exception as e:      # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                world.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    return (pos, box, at, ai, cov, par, pairs), results


@pytest.mark.parametrize('nranks', [1, 2, 3])
@pytest.mark.parametrize('lpol', [False, True])
def test_slab_decomposition_matches_single_gpu(precision, nranks, lpol):
    """x-slab decomposition (admp_amd/parallel.py): every rank must return the single-GPU result."""
    from admp_amd.pme import ADMPPmeForce
    system, results = _run_slab_ranks(nranks, 'double', lpol)
    pos, box, at, ai, cov, par, pairs = system
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
    if lpol:
        E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'],
                            par['dScales'])
    else:
        E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['mScales'])
    scale = max(abs(p) for p in f.energy_parts)
    assert sum(r[6] for r in results) == len(pos)            # home lists partition the atoms
    for (Er, Gr, parts, U, ncyc, conv, nhome) in results:
        for a, b in zip(parts, f.energy_parts):
            assert abs(a - b) < 1e-10 * scale
        assert rel(Gr, G) < 1e-10
        if lpol:
            assert ncyc == f.n_cycle and conv == f.lconverg and rel(U, f.U_ind) < 1e-10


def test_slab_decomposition_single_precision(precision):
    from admp_amd.pme import ADMPPmeForce
    system, results = _run_slab_ranks(2, 'single', True, n_mol=216, seed=11)
    pos, box, at, ai, cov, par, pairs = system
    ref = oracle_es(pos, box, at, ai, cov, par, pairs, *(lambda f: (f.kappa, (f.K1, f.K2, f.K3)))(
        ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)), True)
    scale = max(abs(p) for p in ref['parts'])
    for (Er, Gr, parts, U, ncyc, conv, nhome) in results:
        assert abs(Er - ref['E']) < 5e-4 * scale and rel(Gr, ref['grad']) < 5e-4 and ncyc == ref['n_cycle']


SLAB_PROC_WORKER = '''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
backend = os.environ.get('ADMP_TEST_BACKEND', 'gloo')
local = int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0       # gloo: the ranks share GPU 0
torch.cuda.set_device(local)
if backend == 'nccl':
    import datetime
    dist.init_process_group('nccl', device_id=torch.device('cuda', local), timeout=datetime.timedelta(seconds=120))
else:
    dist.init_process_group('gloo')
from admp_amd import systems as S
from admp_amd.parallel import SlabPme, make_comm
from admp_amd.pme import ADMPPmeForce
pos, box = S.synthetic_water_box(125, seed=5)
at, ai, cov = S.water_topology(125)
par = S.water_parameters(125, True)
pairs = S.build_pairs(pos, box, 4.0)
args = (pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
f = SlabPme(make_comm(), box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='replicated')
E, G = f.get_forces(*args)
E2, G2 = f.get_forces(*args, U_init=f.U_ind)          # second call: cached pair table, warm start
ref = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
Er, Gr = ref.get_forces(*args)
err = np.linalg.norm(G - Gr) / np.linalg.norm(Gr)
assert abs(E - Er) < 1e-9 * max(abs(p) for p in ref.energy_parts) and err < 1e-10, (E, Er, err)
assert f.n_cycle <= ref.n_cycle and abs(E2 - E) < 1e-3
dist.barrier()
if dist.get_rank() == 0:
    print('SLAB-PROC-OK', dist.get_world_size(), f.n_home)
dist.destroy_process_group()
'''


def test_slab_decomposition_two_processes_gloo(tmp_path):
    """Two PROCESSES sharing this GPU, torch.distributed gloo backend (host-staged collectives): the production
    driver (SlabPme + TorchComm) end to end; on a multi-GPU node the only difference is backend nccl (= RCCL)."""
    import subprocess
    import sys
    script = tmp_path / 'slab_worker.py'
    script.write_text(SLAB_PROC_WORKER % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', '29541', str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'SLAB-PROC-OK 2' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_slab_decomposition_two_processes_rccl(tmp_path):
    """The same worker on TWO GPUs over RCCL (backend nccl: device-to-device all-to-all, send / recv ring shifts, MAX and
    SUM all-reduces on the library's buffers).  Needs two devices: skipped on the one-GPU box this suite usually runs on."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    script = tmp_path / 'slab_worker.py'
    script.write_text(SLAB_PROC_WORKER % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', ADMP_TEST_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', '29543', str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'SLAB-PROC-OK 2' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


RCCL_WORKER = r'''
import ctypes, datetime, os, sys
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29547')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=90))
from admp_amd import _lib
from admp_amd.parallel import CommBinding, TorchComm
comm = TorchComm()
assert comm.native and comm.size == 1
bind = CommBinding(comm, dev)
hip = ctypes.CDLL('libamdhip64.so')
n = 1 << 16
bufs = []
for _ in range(2):                      # device memory that torch's allocator does not own, like the library's buffers
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(8 * n)) == 0
    bufs.append(p.value)
side = torch.cuda.Stream()
with torch.cuda.stream(side):           # the calculators run on the caller's current stream, whichever it is
    for dt, tdt in ((_lib.T_F32, torch.float32), (_lib.T_F64, torch.float64), (_lib.T_I32, torch.int32)):
        a, b = bind._view(bufs[0], n, dt), bind._view(bufs[1], n, dt)
        ref = (torch.arange(n, device=dev) %% 977).to(tdt)
        a.copy_(ref); b.zero_()
        for op in (_lib.OP_SUM, _lib.OP_MAX):
            assert bind._all_reduce(None, bufs[0], n, dt, op, 0) == 0, bind.error
        cnt = (ctypes.c_int64 * 1)(n)
        assert bind._all_to_all_v(None, bufs[0], cnt, bufs[1], cnt, dt, 0) == 0, bind.error
        assert torch.equal(a, ref) and torch.equal(b, ref)
        # the ring shift of TorchComm.shift between two ranks, here with this rank as both neighbours
        b.zero_()
        ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        assert torch.equal(b, ref)
side.synchronize()
assert comm.bytes_sent is not None
dist.destroy_process_group()
print('RCCL-OK')
'''


def test_rccl_collectives_on_library_buffers(tmp_path):
    """RCCL itself (torch.distributed backend nccl) under the communicator callbacks, as far as ONE GPU allows: a
    one-rank group, the admp_comm callbacks (CommBinding -> TorchComm) on raw hipMalloc'ed buffers wrapped as tensors, on a
    non-default stream: all-reduce SUM / MAX in the three element types, the all-to-all, and the send/recv pair of the ring
    shift.  What stays unexercised here is only traffic between different devices."""
    import subprocess
    import sys
    script = tmp_path / 'rccl_worker.py'
    script.write_text(RCCL_WORKER % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29547'))
    assert r.returncode == 0 and 'RCCL-OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


NATIVE_RCCL_WORKER = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
os.environ['ADMP_RCCL_SELF_SENDRECV'] = '1'      # one GPU: the rank's own segments go through ncclSend / ncclRecv too
import torch
from admp_amd import _lib
from admp_amd.parallel import RcclComm
L = _lib.load()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
uid = ctypes.create_string_buffer(_lib.RCCL_ID_BYTES)
assert L.admp_rccl_unique_id(uid) == 0, L.admp_rccl_last_error()
comm = RcclComm(device=dev, unique_id=bytes(uid.raw), rank=0, size=1)
assert comm.version() > 0
hip = ctypes.CDLL('libamdhip64.so')
n = 1 << 16
bufs = []
for _ in range(2):                      # device memory torch's allocator does not own, like the library's buffers
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(8 * n)) == 0
    bufs.append(p.value)
from admp_amd.parallel import _DevArray, _TORCH_OF
def view(ptr, dt):
    td, ts = _TORCH_OF[dt]
    return torch.as_tensor(_DevArray(ptr, n, ts), device=dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side):           # the library runs on the caller's current stream, whichever it is
    st = ctypes.c_void_p(side.cuda_stream)
    for dt, tdt in ((_lib.T_F32, torch.float32), (_lib.T_F64, torch.float64), (_lib.T_I32, torch.int32)):
        a, b = view(bufs[0], dt), view(bufs[1], dt)
        ref = (torch.arange(n, device=dev) %% 977).to(tdt)
        a.copy_(ref); b.zero_()
        for op in (_lib.OP_SUM, _lib.OP_MAX):
            assert L.admp_rccl_all_reduce(comm._c, bufs[0], n, dt, op, st) == 0, L.admp_rccl_last_error()
        cnt = (ctypes.c_int64 * 1)(n)
        assert L.admp_rccl_all_to_all_v(comm._c, bufs[0], cnt, bufs[1], cnt, dt, st) == 0, L.admp_rccl_last_error()
        side.synchronize()
        assert torch.equal(a, ref) and torch.equal(b, ref)
        b.zero_()
        for to_next in (1, 0):
            assert L.admp_rccl_shift(comm._c, bufs[0], bufs[1], n, dt, to_next, st) == 0, L.admp_rccl_last_error()
            side.synchronize()
            assert torch.equal(b, ref)
            b.zero_()
    # the SCF residual word: MAX over the bit patterns of non-negative doubles
    w = view(bufs[0], _lib.T_F64)
    w[0] = 12.5
    assert L.admp_rccl_all_reduce(comm._c, bufs[0], 1, _lib.T_F64, _lib.OP_MAX, st) == 0
    side.synchronize()
    assert float(w[0]) == 12.5
    # through a handle: admp_set_comm_rccl configures rank / nranks from the communicator
    from admp_amd import systems as S
    from admp_amd.parallel import SlabPme
    from admp_amd.pme import ADMPPmeForce
    pos, box = S.synthetic_water_box(64, seed=5)
    at, ai, cov = S.water_topology(64)
    par = S.water_parameters(64, True)
    pairs = S.build_pairs(pos, box, 4.0)
    args = (pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    f = SlabPme(comm, box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='replicated')
    E, G = f.get_forces(*args)
    Er, Gr = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True).get_forces(*args)
    assert abs(E - Er) < 1e-9 * abs(Er) and abs(G - Gr).max() < 1e-9 * abs(Gr).max()
comm.refresh_stats()
assert comm.calls['replicate_outputs'] >= 6, dict(comm.calls)
comm.close()
print('NATIVE-RCCL-OK')
'''


def test_native_rccl_collectives_on_library_buffers(tmp_path):
    """The round-4 communicator (admp_amd/csrc/rccl_comm.hip: RCCL called by the library itself, bound with dlsym) as far as
    ONE GPU allows: a one-rank ncclComm created from a unique id, all-reduce SUM / MAX in the three element types, the
    grouped ncclSend / ncclRecv of the all-to-all-v and of both ring-shift directions (the rank as its own peer:
    ADMP_RCCL_SELF_SENDRECV), the residual word's MAX on bit patterns, on raw hipMalloc'ed buffers and a non-default stream;
    then a handle bound to it with admp_set_comm_rccl.  Unexercised: traffic between different devices
    (test_slab_decomposition_two_processes_rccl, skipped on a one-GPU box)."""
    import subprocess
    import sys
    script = tmp_path / 'native_rccl_worker.py'
    script.write_text(NATIVE_RCCL_WORKER % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    assert r.returncode == 0 and 'NATIVE-RCCL-OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize('prec', ['double', 'single'])
def test_gpu_neighbour_list_matches_kdtree(precision, prec):
    """Cell-list search (admp_amd.neighbor) against the host cKDTree builder, cubic and triclinic cells,
    atoms outside the cell, and a box too small for 3 cells per side."""
    import torch
    from admp_amd.neighbor import NeighborList
    settings.PRECISION = prec
    for n_mol, rc in ((27, 4.0), (512, 4.0), (512, 6.5)):
        pos, box = S.synthetic_water_box(n_mol, seed=3)
        shifted = pos + np.array([37.0, -11.0, 5.0])
        got = NeighborList(box, rc).allocate(shifted).cpu().numpy()
        want = S.build_pairs(pos, box, rc)
        assert (got[:, 0] < got[:, 1]).all()
        a = set(map(tuple, got.tolist()))
        b = set(map(tuple, want.tolist()))
        if prec == 'double':
            assert a == b and len(got) == len(want)
        else:   # pairs within rounding of rc may differ in single precision
            assert len(a ^ b) <= max(2, len(b) // 20000)
    # triclinic: brute-force minimum image with the reference's rounding rule (admp/spatial.py:13-32)
    rng = np.random.default_rng(1)
    box = np.array([[14.0, 0, 0], [2.5, 13.0, 0], [-1.5, 2.0, 15.0]])
    pos = rng.uniform(-5, 20, (400, 3))
    if prec == 'double':
        got = set(map(tuple, NeighborList(box, 5.0).allocate(pos).cpu().numpy().tolist()))
        i, j = np.triu_indices(len(pos), 1)
        d = pos[i] - pos[j]
        s = d @ np.linalg.inv(box)
        d = (s - np.floor(s + 0.5)) @ box
        m = np.linalg.norm(d, axis=1) < 5.0
        assert got == set(zip(i[m].tolist(), j[m].tolist()))


def test_warm_regime_fast_path_is_equivalent(precision):
    """After a call that converged at its first SCF check the next call evaluates the first cycle with the full
    kernels (engine.hip `warm_regime`); results must be those of the plain loop, whether the check then passes
    (step finished in one pass) or fails (falls back to the Jacobi loop)."""
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    pos, box, at, ai, cov, par, pairs = water_system(125, 5, True)
    args = (pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    E0, G0 = f.get_forces(*args)                       # cold: Jacobi loop from zero
    U = f.U_ind.copy()
    E1, G1 = f.get_forces(*args, U_init=U)             # plain loop, converges at the first check
    assert f.n_cycle == 0
    E2, G2 = f.get_forces(*args, U_init=U)             # fast path, check passes
    assert f.n_cycle == 0 and f.lconverg
    scale = max(abs(p) for p in f.energy_parts)
    assert abs(E2 - E1) < 1e-12 * scale and rel(G2, G1) < 1e-12
    assert abs(E1 - E0) < 1e-3 * abs(E0)               # same fixed point up to the loose SCF threshold
    E3, G3 = f.get_forces(*args)                       # fast path armed, but U_init = 0: the check fails
    fresh = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    E4, G4 = fresh.get_forces(*args)
    assert f.n_cycle == fresh.n_cycle and abs(E3 - E4) < 1e-12 * scale and rel(G3, G4) < 1e-12
    assert rel(f.U_ind, fresh.U_ind) < 1e-12


@pytest.mark.parametrize('lpol', [False, True])
def test_parameter_gradient_of_fluctuating_multipoles(precision, lpol):
    """BASELINE config 5: geometry-dependent Q_local with a parameter-gradient adjoint.  Toy model of SURVEY.md 8d:
    the O charge and dipole follow the O-H bond stretch, Q_O = Q0 + k (r_OH1 + r_OH2 - 2 r0), the H charges
    compensate.  dE/dk and the total dE/dpositions (explicit + through Q) from the HIP adjoint chained by torch
    autograd must match autodiff of the oracle."""
    import torch
    from admp_amd.autograd import pme_energy
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    settings.PRECISION = 'double'
    n_mol = 64
    pos, box, at, ai, cov, par, pairs = water_system(n_mol, 8, lpol)
    kvec0 = np.array([0.3, -0.15])        # d(charge)/d(stretch), d(dipole_z)/d(stretch)

    def model(p, kv, Q0):
        m = p.reshape(n_mol, 3, 3)
        s = (m[:, 1] - m[:, 0]).norm(dim=1) + (m[:, 2] - m[:, 0]).norm(dim=1) - 2 * S.R_OH
        Q = Q0.clone().reshape(n_mol, 3, 9)
        dq = kv[0] * s
        Q[:, 0, 0] = Q[:, 0, 0] + dq
        Q[:, 1, 0] = Q[:, 1, 0] - 0.5 * dq
        Q[:, 2, 0] = Q[:, 2, 0] - 0.5 * dq
        Q[:, 0, 1] = Q[:, 0, 1] + kv[1] * s
        return Q.reshape(3 * n_mol, 9)

    rng = np.random.default_rng(0)
    pos = pos + rng.normal(scale=0.03, size=pos.shape)         # stretch the bonds a little
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
    p = torch.tensor(pos, device='cuda', requires_grad=True)
    kv = torch.tensor(kvec0, device='cuda', requires_grad=True)
    Q0 = torch.tensor(par['Q_local'], device='cuda')
    rest = (par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales']) if lpol else (par['mScales'],)
    E = pme_energy(f, p, box, pairs, model(p, kv, Q0), *rest)
    E.backward()
    # oracle: same model in float64 torch-CPU, autodiff through the whole restatement (U fixed at the SCF result)
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, lpol)
    po = torch.tensor(pos, requires_grad=True)
    ko = torch.tensor(kvec0, requires_grad=True)
    Qo = model(po, ko, torch.tensor(par['Q_local']))
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    if lpol:
        U, _, _ = O.optimize_Uind(sysm, po.detach(), box, pairs, Qo.detach(), T(par['pol']), T(par['tholes']),
                                  T(par['mScales']), T(par['pScales']))
        Eo = O.energy_pme(sysm, po, T(box), pairs, Qo, U, T(par['pol']), T(par['tholes']), T(par['mScales']), T(par['pScales']))
    else:
        Eo = O.energy_pme(sysm, po, T(box), pairs, Qo, None, None, None, T(par['mScales']), None)
    Eo.backward()
    assert abs(float(E.detach()) - float(Eo.detach())) < 1e-9 * max(abs(x) for x in f.energy_parts)
    assert rel(p.grad.cpu().numpy(), po.grad.numpy()) < 1e-8
    assert rel(kv.grad.cpu().numpy(), ko.grad.numpy()) < 1e-8


def test_parameter_gradient_at_config_size(precision):
    """BASELINE configs[4] at its stated size: 98 304 atoms (32 768 waters, K = 128), geometry-dependent Q_local.
    The oracle cannot run at this size in seconds, so the check is size-independent: dE/dk and the total
    dE/dpositions (explicit + through Q(positions)) from the HIP adjoint chained by torch autograd must equal central
    differences of the energy the same path returns (non-polarizable and polarizable at fixed converged dipoles)."""
    import torch
    from admp_amd.autograd import pme_energy
    from admp_amd.neighbor import NeighborList
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    n_mol = 32768
    pos, box = S.synthetic_water_box(n_mol, seed=7)
    at, ai, cov = S.water_topology(n_mol)
    rng = np.random.default_rng(3)
    pos = pos + rng.normal(scale=0.02, size=pos.shape)
    pairs = NeighborList(box, 4.0).allocate(pos)
    dev = 'cuda'

    def model(p, kv, Q0):
        m = p.reshape(n_mol, 3, 3)
        s = (m[:, 1] - m[:, 0]).norm(dim=1) + (m[:, 2] - m[:, 0]).norm(dim=1) - 2 * S.R_OH
        Q = Q0.clone().reshape(n_mol, 3, 9)
        dq = kv[0] * s
        Q[:, 0, 0] = Q[:, 0, 0] + dq
        Q[:, 1, 0] = Q[:, 1, 0] - 0.5 * dq
        Q[:, 2, 0] = Q[:, 2, 0] - 0.5 * dq
        Q[:, 0, 1] = Q[:, 0, 1] + kv[1] * s
        return Q.reshape(3 * n_mol, 9)

    for lpol in (False, True):
        par = S.water_parameters(n_mol, lpol)
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
        for k in ('K1', 'K2', 'K3'):
            f.update_env(k, 128)
        Q0 = torch.tensor(par['Q_local'], device=dev)
        rest = (par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales']) if lpol else (par['mScales'],)
        p = torch.tensor(pos, device=dev, requires_grad=True)
        kv = torch.tensor([0.3, -0.15], device=dev, dtype=torch.float64, requires_grad=True)
        kw = {}
        if lpol:     # converge the dipoles once; every later evaluation starts from them and passes its first check
            f.get_energy(p.detach(), box, pairs, model(p, kv, Q0).detach(), *rest)
            U = torch.as_tensor(f.U_ind, device=dev).clone()
            f.get_energy(p.detach(), box, pairs, model(p, kv, Q0).detach(), *rest, U_init=U)
            U = torch.as_tensor(f.U_ind, device=dev).clone()
            kw = dict(U_init=U)
        E = pme_energy(f, p, box, pairs, model(p, kv, Q0), *rest, **kw)
        E.backward()

        def energy(pp, kk):
            with torch.no_grad():
                e = f.get_energy(pp, box, pairs, model(pp, kk, Q0), *rest, **kw)
            if lpol:
                assert f.n_cycle == 0
            return float(e)

        for c, h in ((0, 1e-3), (1, 1e-3)):
            d = torch.zeros(2, device=dev, dtype=torch.float64)
            d[c] = h
            fd = (energy(p.detach(), kv.detach() + d) - energy(p.detach(), kv.detach() - d)) / (2 * h)
            an = float(kv.grad[c])
            assert abs(fd - an) < 2e-6 * abs(an) + 2e-4, (lpol, c, fd, an)
        G = p.grad
        g = torch.Generator(device=dev).manual_seed(5)
        v = G / G.norm() + 0.5 * torch.randn(p.shape, generator=g, device=dev, dtype=p.dtype) / (9 * n_mol) ** 0.5
        v /= v.norm()
        h = 0.02
        fd = (energy(p.detach() + h * v, kv.detach()) - energy(p.detach() - h * v, kv.detach())) / (2 * h)
        an = float((G * v).sum())
        assert abs(fd - an) < 5e-6 * abs(an) + 2e-4, (lpol, fd, an)


def test_mscale_gradients_vs_oracle_autograd(precision):
    """dE/dmScales of the three calculators -- what the reference's examples/openmm_api/run.py:41-46 prints from
    jax.grad(potential, argnums=3) -- against torch autograd through the oracle, f64 and f32, with a bonded model that
    populates several covalent classes (water: 1-2 and 1-3 pairs plus the wrapped non-bonded class)."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
    from oracle import admp_oracle as O
    pos, box, at, ai, cov, par, pairs = water_system(125, 21, True)
    mS0 = np.array([0.3, 0.7, 0.0, 1.0, 0.9])
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    for prec, tol in (('double', 1e-9), ('single', 2e-4)):
        settings.PRECISION = prec
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        g = f.get_mscale_gradient(pos, box, pairs, par['Q_local'], mS0)
        sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, True)
        m = T(mS0).clone().requires_grad_(True)
        U, _, _ = O.optimize_Uind(sysm, T(pos), box, pairs, T(par['Q_local']), T(par['pol']), T(par['tholes']), T(mS0),
                                  T(par['pScales']))
        E = O.energy_pme(sysm, T(pos), T(box), pairs, T(par['Q_local']), U.detach(), T(par['pol']), T(par['tholes']), m,
                         T(par['pScales']))
        ref, = torch.autograd.grad(E, m)
        assert np.abs(g - ref.numpy()).max() < tol * np.abs(ref.numpy()).max(), (prec, g, ref)
        d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
        gd = d.get_mscale_gradient(pos, box, pairs, par['c_list'], mS0)
        m = T(mS0).clone().requires_grad_(True)
        Ed = sum(O.disp_pme_parts(T(pos), T(box), pairs, T(par['c_list']), m, cov, d.kappa, (d.K1, d.K2, d.K3), 10))
        ref, = torch.autograd.grad(Ed, m)
        assert np.abs(gd - ref.numpy()).max() < tol * np.abs(ref.numpy()).max(), (prec, gd, ref)
        tt = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
        lists = [par[k] for k in ('a_list', 'b_list', 'q_list')] + [par['c_list'][:, 0]]
        gt = tt.get_mscale_gradient(pos, box, pairs, mS0, *lists)
        m = T(mS0).clone().requires_grad_(True)
        Et = O.tt_damping_energy(T(pos), T(box), pairs, m, cov, *[T(x) for x in lists])
        ref, = torch.autograd.grad(Et, m)
        assert np.abs(gt - ref.numpy()).max() < tol * np.abs(ref.numpy()).max(), (prec, gt, ref)


def test_pol_and_thole_gradients_vs_oracle_autograd(precision):
    """dE/dpol and dE/dtholes at the converged dipoles against torch autograd through the oracle at the same (fixed)
    dipoles -- the Hellmann-Feynman parameter gradient jax.grad(pot_pme, argnums=3) gives in the reference.  Every atom
    polarizable here (with alpha = 0 the reference's own gradient is 0 * inf); the alpha = 0 convention (gradient 0) is
    checked on the standard water parameters."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    pos, box, at, ai, cov, par, pairs = water_system(125, 31, True)
    rng = np.random.default_rng(5)
    pol = np.where(par['pol'] > 0, par['pol'], 0.25) * rng.uniform(0.8, 1.2, len(pos))
    th = par['tholes'] + rng.uniform(0.5, 3.0, len(pos))
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    for prec, tol in (('double', 1e-8), ('single', 5e-4)):
        settings.PRECISION = prec
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        dpol, dth = f.get_pol_thole_gradients(pos, box, pairs, par['Q_local'], pol, th, par['mScales'], par['pScales'],
                                              par['dScales'])
        sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, True)
        pt, tt = T(pol).clone().requires_grad_(True), T(th).clone().requires_grad_(True)
        U = T(np.asarray(f.U_ind, dtype=np.float64))
        E = O.energy_pme(sysm, T(pos), T(box), pairs, T(par['Q_local']), U, pt, tt, T(par['mScales']), T(par['pScales']))
        rp, rt = torch.autograd.grad(E, (pt, tt))
        assert rel(dpol, rp.numpy()) < tol and rel(dth, rt.numpy()) < tol, (prec, rel(dpol, rp.numpy()), rel(dth, rt.numpy()))
    settings.PRECISION = 'double'
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    dpol, dth = f.get_pol_thole_gradients(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'],
                                          par['pScales'], par['dScales'])
    assert np.all(dpol[par['pol'] == 0] == 0.0) and np.all(np.isfinite(dpol)) and np.all(np.isfinite(dth))


def test_potential_fn_convention_and_param_gradient(precision):
    """admp_amd.api: potential_fn(positions, box, pairs, params) closures in the reference's convention
    (admp/api.py:183-199, 442-455); the energies are linear in mScales, so param_gradient(...)['mScales'] must equal
    finite differences of the potential exactly (to round-off)."""
    from admp_amd.api import pme_potential, disp_potential, param_gradient
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
    settings.PRECISION = 'double'
    pos, box, at, ai, cov, par, pairs = water_system(64, 17, False)
    n = len(pos) // 3
    pme = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    disp = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
    tt = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
    c = par['c_list']
    dpar = {'mScales': np.array([0.2, 0.5, 0.0, 1.0, 0.8]), 'A': par['a_list'][:2] * 2625.5,
            'B': par['b_list'][:2] / 0.0529177249, 'Q': par['q_list'][:2], 'C6': c[:2, 0] ** 2 / 1e6,
            'C8': c[:2, 1] ** 2 / 1e8, 'C10': c[:2, 2] ** 2 / 1e10}
    pot_d = disp_potential(disp, tt, np.tile([0, 1, 1], n))
    pot_p = pme_potential(pme)
    ppar = {'mScales': dpar['mScales'], 'Q_local': par['Q_local']}
    # the potentials are the calculators' energies
    e_sr = tt(pos, box, pairs, dpar['mScales'], par['a_list'], par['b_list'], par['q_list'], c[:, 0])
    e_lr = disp.get_energy(pos, box, pairs, c, dpar['mScales'])
    assert abs(pot_d(pos, box, pairs, dpar) - (e_sr - e_lr)) < 1e-9 * abs(e_lr)
    assert abs(pot_p(pos, box, pairs, ppar) - pme.get_energy(pos, box, pairs, par['Q_local'], dpar['mScales'])) < 1e-9
    for pot, prm in ((pot_d, dpar), (pot_p, ppar)):
        g = param_gradient(pot, pos, box, pairs, prm)['mScales']
        for k in (0, 1, 4):
            up, dn = dict(prm), dict(prm)
            up['mScales'] = prm['mScales'].copy(); up['mScales'][k] += 0.25
            dn['mScales'] = prm['mScales'].copy(); dn['mScales'][k] -= 0.25
            fd = (pot(pos, box, pairs, up) - pot(pos, box, pairs, dn)) / 0.5
            assert abs(fd - g[k]) < 1e-8 * max(abs(g).max(), 1.0), (k, fd, g[k])
    assert param_gradient(pot_p, pos, box, pairs, ppar)['Q_local'].shape == (len(pos), 9)


def _mixed_axis_system(seed=2):
    """8 'molecules' of 4 atoms exercising every axis rule (ZThenX, Bisector, ZBisect, ThreeFold, Zonly, none)."""
    rng = np.random.default_rng(seed)
    n_mol, L = 8, 14.0
    centres = np.array([[x, y, z] for x in (3.5, 10.5) for y in (3.5, 10.5) for z in (3.5, 10.5)], dtype=float)
    pos, at, ai = [], [], []
    local = np.array([[0, 0, 0], [1.0, 0.1, 0.0], [-0.3, 0.95, 0.1], [-0.2, -0.4, 0.9]])
    types = [(0, 1, 2, -1), (1, 0, 2, -1), (2, 0, 1, 3), (3, 1, 2, 0)]   # per atom: (type, z, x, y) within the molecule
    for m in range(n_mol):
        rot = S._random_rotations(rng, 1)[0]
        base = 4 * m
        for k in range(4):
            pos.append(centres[m] + rot @ local[k] + rng.normal(scale=0.05, size=3))
        for k, (t, z, x, y) in enumerate(types):
            if m % 4 == 3 and k == 3:
                at.append(5); ai.append([-1, -1, -1])                 # NoAxisType site (charge only)
            elif m % 4 == 2 and k == 0:
                at.append(4); ai.append([base + 1, -1, -1])            # Zonly
            else:
                at.append(t); ai.append([base + z, base + x, base + y if y >= 0 else -1])
    pos = np.array(pos)
    at, ai = np.array(at, dtype=np.int32), np.array(ai, dtype=np.int32)
    na = len(pos)
    Q = rng.normal(size=(na, 9)) * np.array([0.6, .3, .3, .3, .2, .2, .2, .2, .2])
    Q[at == 5, 1:] = 0.0
    Q[:, 0] -= Q[:, 0].mean()
    cov = np.zeros((na, na), dtype=np.int32)
    for m in range(n_mol):
        for a in range(4):
            for b in range(4):
                if a != b:
                    cov[4 * m + a, 4 * m + b] = 1 if 0 in (a, b) else 2
    pol = rng.uniform(0.4, 1.2, na) * (rng.random(na) < 0.6)
    thole = rng.uniform(2.0, 8.0, na)
    return pos, np.eye(3) * L, at, ai, cov, Q, pol, thole


@pytest.fixture()
def korder():
    old = settings.REFERENCE_KPOINT_ORDER
    yield
    settings.REFERENCE_KPOINT_ORDER = old


@pytest.mark.parametrize('lpol', [False, True])
@pytest.mark.parametrize('tric', [False, 'reference', 'consistent'])
def test_all_axis_rules_and_triclinic_cell(precision, korder, lpol, tric):
    """Every local-axis rule, general scale tables, dense covalent map; orthorhombic and triclinic cells.
    Triclinic cell, 'reference': settings.REFERENCE_KPOINT_ORDER = True against the UNMODIFIED oracle (the reference's
    meshgrid(kz, kx, ky) order, admp/recip.py:339-340).  'consistent': the product's default assignment against the
    oracle's quirk=False variant -- a check of the non-reference default, not a parity claim."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    settings.PRECISION = 'double'
    settings.REFERENCE_KPOINT_ORDER = (tric == 'reference')
    pos, box, at, ai, cov, Q, pol, thole = _mixed_axis_system()
    quirk = tric != 'consistent'
    if tric:
        box = np.array([[14.0, 0, 0], [1.5, 14.0, 0], [-1.0, 0.8, 14.0]])
    pairs = np.array([(i, j) for i in range(len(pos)) for j in range(i + 1, len(pos))], dtype=np.int32)
    d = pos[pairs[:, 0]] - pos[pairs[:, 1]]
    s = d @ np.linalg.inv(box)
    d = (s - np.floor(s + 0.5)) @ box
    pairs = pairs[np.linalg.norm(d, axis=1) < 6.0]
    mS = np.array([0.0, 0.4, 0.8, 1.0, 1.0])
    pS = np.array([0.0, 0.0, 1.0, 1.0, 1.0])
    f = ADMPPmeForce(box, at, ai, cov, 6.0, 1e-5, 2, lpol=lpol)
    K = (f.K1, f.K1, f.K1)
    for k in ('K1', 'K2', 'K3'):
        f.update_env(k, K[0])
    sysm = O.PmeSystem(at, ai, cov, f.kappa, K, 2, lpol)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    p = T(pos).requires_grad_(True)
    if lpol:
        E, G = f.get_forces(pos, box, pairs, Q, pol, thole, mS, pS, pS)
        # oracle with the consistent k assignment needs the SCF done through the same energy function
        U = T(np.zeros_like(pos))
        for i in range(30):
            Ug = U.clone().requires_grad_(True)
            e = _oracle_energy(O, sysm, T(pos), T(box), pairs, T(Q), Ug, T(pol), T(thole), T(mS), T(pS), quirk=quirk)
            fld, = torch.autograd.grad(e, Ug)
            if float(fld[T(pol) > 0.001].abs().max()) < 10.0:
                break
            U = U - fld * T(pol)[:, None] / O.DIELECTRIC
        assert f.n_cycle == i
        e = _oracle_energy(O, sysm, p, T(box), pairs, T(Q), U, T(pol), T(thole), T(mS), T(pS), quirk=quirk)
        assert rel(f.U_ind, U.numpy()) < 1e-8
    else:
        E, G = f.get_forces(pos, box, pairs, Q, mS)
        e = _oracle_energy(O, sysm, p, T(box), pairs, T(Q), None, None, None, T(mS), None, quirk=quirk)
    g, = torch.autograd.grad(e, p)
    scale = max(abs(x) for x in f.energy_parts)
    assert abs(E - float(e.detach())) < 1e-9 * scale
    assert rel(G, g.numpy()) < 1e-8


def _oracle_energy(O, sysm, pos, box, pairs, Q, U, pol, thole, mS, pS, quirk):
    """energy_pme of the oracle with a switch for the k-column quirk of the reference."""
    import torch
    frames = O.construct_local_frames(pos, box, sysm.axis_type, sysm.axis_indices)
    Qg = O.rot_local2global(Q, frames, 2)
    if U is not None:
        Uh = U @ O._C1_C2H.T
        Qt = torch.cat([Qg[:, 0:1], Qg[:, 1:4] + Uh, Qg[:, 4:]], dim=1)
        e = O.pme_real(pos, box, pairs, Qg, Uh, pol, thole, mS, pS, sysm.covalent_map, sysm.kappa, 2, True) + \
            O.pol_penalty(Uh, pol)
    else:
        Qt = Qg
        e = O.pme_real(pos, box, pairs, Qg, None, None, None, mS, None, sysm.covalent_map, sysm.kappa, 2, False)
    return e + O.pme_recip(pos, box, Qt, sysm.kappa, sysm.K, 2, quirk=quirk) + O.pme_self(Qt, sysm.kappa, 2)


@pytest.mark.parametrize('lmax', [0, 1])
def test_lower_multipole_orders(precision, lmax):
    """lmax = 0 (no frames at all) and lmax = 1: Q_local has (lmax+1)^2 columns."""
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    settings.PRECISION = 'double'
    pos, box, at, ai, cov, par, pairs = water_system(64, 9, False)
    nh = (lmax + 1) ** 2
    Q = par['Q_local'][:, :nh].copy()
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, lmax)
    E, G = f.get_forces(pos, box, pairs, Q, par['mScales'])
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), lmax, False)
    ref = O.pme_energy_and_grad(sysm, pos, box, pairs, Q, par['mScales'])
    assert abs(E - ref['E']) < 1e-9 * max(abs(p) for p in ref['parts']) and rel(G, ref['grad']) < 1e-8
    with pytest.raises(NotImplementedError):
        ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 3)


def test_empty_pair_list_and_isolated_atoms(precision):
    """No pairs at all: the energy is reciprocal + self only; padding-only lists behave the same."""
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    settings.PRECISION = 'double'
    pos, box, at, ai, cov, par, pairs = water_system(27, 4, False)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    E, G = f.get_forces(pos, box, np.zeros((0, 2), dtype=np.int32), par['Q_local'], par['mScales'])
    assert f.n_pairs == 0 and f.energy_parts[0] == 0.0
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, False)
    ref = O.pme_energy_and_grad(sysm, pos, box, np.zeros((0, 2), dtype=np.int64), par['Q_local'], par['mScales'])
    assert abs(E - ref['E']) < 1e-9 * abs(ref['parts'][2]) and rel(G, ref['grad']) < 1e-8
    pad = np.full((13, 2), len(pos), dtype=np.int32)
    E2, _ = f.get_forces(pos, box, pad, par['Q_local'], par['mScales'])
    assert E2 == E


def test_c_abi_with_host_pointers(precision):
    """The C ABI called the way INTEGRATION.md's stub does: plain ctypes, HOST numpy buffers (on_device = 0)."""
    import ctypes
    from admp_amd import _lib
    from admp_amd._device import covalent_to_csr
    from admp_amd.pme import setup_ewald_parameters
    from oracle import admp_oracle as O
    L = _lib.load()
    pos, box, at, ai, cov, par, pairs = water_system(64, 6, True)
    na = len(pos)
    kappa, K1, K2, K3 = setup_ewald_parameters(4.0, 1e-4, box)
    h = ctypes.c_void_p()
    assert L.admp_create(ctypes.byref(h), 0, 8) == 0
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    ptr, col, nb = covalent_to_csr(cov, na)
    at32, ai32 = np.ascontiguousarray(at, dtype=np.int32), np.ascontiguousarray(ai, dtype=np.int32)
    assert L.admp_set_topology(h, na, vp(at32), vp(ai32), vp(ptr), vp(col), vp(nb)) == 0
    assert L.admp_set_ewald(h, kappa, K1, K2, K3, 2, 1) == 0
    p32 = np.ascontiguousarray(pairs, dtype=np.int32)
    assert L.admp_set_pairs(h, len(p32), vp(p32), 0) == 0
    assert L.admp_num_pairs(h) == len(pairs)
    posc, Qc = np.ascontiguousarray(pos), np.ascontiguousarray(par['Q_local'])
    polc, thc = np.ascontiguousarray(par['pol']), np.ascontiguousarray(par['tholes'])
    U = np.zeros((na, 3)); grad = np.empty((na, 3)); dQ = np.empty((na, 9))
    E = (ctypes.c_double * 4)(); ncyc = ctypes.c_int(); conv = ctypes.c_int()
    d = _lib.darr
    rc = L.admp_pme_energy_grad(h, vp(posc), d(box.ravel()), vp(Qc), vp(polc), vp(thc), 5, d(par['mScales']),
                                d(par['pScales']), d(par['dScales']), vp(U), 30, 10.0, E, vp(grad), vp(dQ),
                                ctypes.byref(ncyc), ctypes.byref(conv), 0)
    assert rc == 0, L.admp_last_error(h)
    sysm = O.PmeSystem(at, ai, cov, kappa, (K1, K2, K3), 2, True)
    ref = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                                par['pScales'], want_dQ=True)
    scale = max(abs(p) for p in ref['parts'])
    assert abs(sum(E) - ref['E']) < 1e-9 * scale and rel(grad, ref['grad']) < 1e-8 and rel(dQ, ref['dQ_local']) < 1e-8
    assert rel(U, ref['U_ind']) < 1e-8 and ncyc.value == ref['n_cycle'] and bool(conv.value) == ref['lconverg']
    # error behaviour: negative code + message, never an abort
    assert L.admp_set_ewald(h, -1.0, K1, K2, K3, 2, 1) < 0 and b'kappa' in L.admp_last_error(h)
    assert L.admp_set_ewald(h, kappa, K1, K2, K3, 3, 1) < 0
    assert L.admp_destroy(h) == 0


@pytest.mark.parametrize('pmax', [6, 8])
def test_dispersion_lower_orders_and_single_precision(precision, pmax):
    from admp_amd.disp_pme import ADMPDispPmeForce
    from oracle import admp_oracle as O
    pos, box, at, ai, cov, par, pairs = water_system(125, 12, False)
    for prec, tol in (('double', 1e-9), ('single', 5e-4)):
        settings.PRECISION = prec
        d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, pmax)
        E, G = d.get_forces(pos, box, pairs, par['c_list'][:, :(pmax - 4) // 2], par['mScales'])
        ref = O.disp_energy_and_grad(pos, box, pairs, par['c_list'], par['mScales'], cov, d.kappa, (d.K1, d.K2, d.K3), pmax)
        assert abs(E - ref['E']) < tol * max(abs(p) for p in ref['parts']) and rel(G, ref['grad']) < max(tol, 1e-8)


def test_cutoff_on_skin_list_equals_exact_list(precision):
    """set_cutoff(rc) on a Verlet list built with a skin (dispersion PME, Tang-Toennies): the partners beyond rc are
    skipped inside the pair kernel, so energy and gradient are those of the exact-rc list -- and of the oracle on that list."""
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    from oracle import admp_oracle as O
    pos, box, at, ai, cov, par, pairs = water_system(216, 21, False)
    wide = S.build_pairs(pos, box, 5.2)
    assert len(wide) > 1.5 * len(pairs)
    c6 = np.ascontiguousarray(par['c_list'][:, 0])
    for prec, tol in (('double', 1e-11), ('single', 2e-5)):
        settings.PRECISION = prec
        d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
        E0, G0 = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
        Ew, Gw = d.get_forces(pos, box, wide, par['c_list'], par['mScales'])
        d.set_cutoff(4.0)
        E1, G1 = d.get_forces(pos, box, wide, par['c_list'], par['mScales'])
        scale = max(abs(x) for x in d.energy_parts)
        assert abs(E1 - E0) < tol * scale and rel(G1, G0) < tol * 10
        assert abs(Ew - E0) > 1.0                                # the skin pairs do contribute when nobody cuts them
        d.set_cutoff(0.0)
        E2, _ = d.get_forces(pos, box, wide, par['c_list'], par['mScales'])
        assert abs(E2 - Ew) < tol * scale
        t = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
        tt = value_and_grad(t)
        targs = (par['mScales'], par['a_list'], par['b_list'], par['q_list'], c6)
        E0, G0 = tt(pos, box, pairs, *targs)
        t.set_cutoff(4.0)
        E1, G1 = tt(pos, box, wide, *targs)
        assert abs(E1 - E0) < tol * 10 * abs(E0) and rel(G1, G0) < tol * 10
        if prec == 'double':
            ref = O.tt_energy_and_grad(pos, box, pairs, par['mScales'], cov, par['a_list'], par['b_list'], par['q_list'], c6)
            assert abs(E1 - ref['E']) < 1e-9 * abs(ref['E']) and rel(G1, ref['grad']) < 1e-9
    # everything derived from the pair terms honours the cutoff too (advisor, round 3): on the skin list with the cutoff set,
    # parameter gradients, dE/dmScales and the box gradient are those of the exact-rc list
    settings.PRECISION = 'double'
    d0 = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
    d1 = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
    d1.set_cutoff(4.0)
    t0 = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
    t1 = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
    t1.set_cutoff(4.0)
    ta = (par['a_list'], par['b_list'], par['q_list'], c6)
    for a, b in ((d0.get_param_gradient(pos, box, pairs, par['c_list'], par['mScales']),
                  d1.get_param_gradient(pos, box, wide, par['c_list'], par['mScales'])),
                 (d0.get_mscale_gradient(pos, box, pairs, par['c_list'], par['mScales']),
                  d1.get_mscale_gradient(pos, box, wide, par['c_list'], par['mScales'])),
                 (d0.get_box_gradient(pos, box, pairs, par['c_list'], par['mScales']),
                  d1.get_box_gradient(pos, box, wide, par['c_list'], par['mScales'])),
                 (t0.get_param_gradient(pos, box, pairs, par['mScales'], *ta),
                  t1.get_param_gradient(pos, box, wide, par['mScales'], *ta)),
                 (t0.get_mscale_gradient(pos, box, pairs, par['mScales'], *ta),
                  t1.get_mscale_gradient(pos, box, wide, par['mScales'], *ta)),
                 (t0.get_box_gradient(pos, box, pairs, par['mScales'], *ta),
                  t1.get_box_gradient(pos, box, wide, par['mScales'], *ta))):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert np.abs(a - b).max() <= 1e-10 * max(np.abs(a).max(), 1e-30), (a, b)
    from admp_amd._lib import AdmpHipError
    with pytest.raises(AdmpHipError):
        d.set_cutoff(-1.0)


def test_spread_fixed_point_tile_dense_brick_and_large_charges(precision):
    """The f32 brick spread sums 32-bit fixed-point words whose scale comes from the brick's entry count and largest folded
    multipole: 2100 atoms crowded into one 6 A cube (ten times the entries of a liquid brick) and a few charges a thousand
    times the others must neither overflow a word nor drown the rest.  Reciprocal space only (empty pair list), f32
    against f64."""
    from admp_amd.pme import ADMPPmeForce
    nm = 8192
    pos, box = S.synthetic_water_box(nm, seed=77)
    at, ai, cov = S.water_topology(nm)
    par = S.water_parameters(nm, False)
    rng = np.random.default_rng(5)
    pos = pos.copy().reshape(nm, 3, 3)
    centre = pos[:700, 0].copy()
    pos[:700] += (20.0 + 6.0 * rng.random((700, 1, 3))) - centre[:, None, :]        # 700 molecules into one 6 A cube
    pos = pos.reshape(-1, 3)
    Q = par['Q_local'].copy()
    Q[3 * 4000:3 * 4000 + 30] *= 1000.0                                            # ten molecules with huge multipoles
    pairs = np.zeros((0, 2), dtype=np.int32)
    out = {}
    for prec in ('double', 'single'):
        settings.PRECISION = prec
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=False)
        E, G = f.get_forces(pos, box, pairs, Q, par['mScales'])
        out[prec] = (float(E), np.array(G, dtype=np.float64), max(abs(x) for x in f.energy_parts))
    E64, G64, scale = out['double']
    E32, G32, _ = out['single']
    assert np.isfinite(G32).all() and abs(E32 - E64) < 2e-5 * scale
    assert rel(G32, G64) < 2e-4
    quiet = np.ones(len(G64), bool)
    quiet[3 * 4000:3 * 4000 + 30] = False
    quiet[:2100] = False
    assert rel(G32[quiet], G64[quiet]) < 2e-3       # the ordinary atoms are not drowned by the bricks that hold the outliers


def test_full_size_directional_derivative(precision):
    """BASELINE's full size (1 048 575 polarizable atoms, K = 256), double precision: the gradient returned by the
    HIP adjoint must be the derivative of the returned energy.  Central difference along a random direction at
    FIXED induced dipoles (the SCF is warm-started from its own result and passes its first check, so U does not
    move -- which is exactly what the Hellmann-Feynman gradient of the reference differentiates)."""
    import torch
    from admp_amd.neighbor import NeighborList
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    nm = 349525
    pos, box = S.synthetic_water_box(nm, seed=20240)
    at, ai, cov = S.water_topology(nm)
    par = S.water_parameters(nm, True)
    pairs = NeighborList(box, 4.0).allocate(pos)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    for k in ('K1', 'K2', 'K3'):
        f.update_env(k, 256)
    dev = 'cuda'
    p0 = torch.as_tensor(pos, device=dev)
    rest = [torch.as_tensor(par[k], device=dev) for k in ('Q_local', 'pol', 'tholes')] + \
           [par['mScales'], par['pScales'], par['dScales']]
    E0, G = f.get_forces(p0, box, pairs, *rest)
    U = torch.as_tensor(f.U_ind, device=dev).clone()
    E0, G = f.get_forces(p0, box, pairs, *rest, U_init=U)
    assert f.n_cycle == 0 and f.n_pairs == len(pairs)
    # direction: the gradient itself blended with noise (a purely random direction in 3M dimensions has a
    # derivative of O(1) kJ/mol/A, below the round-off of energies whose Ewald parts are O(1e8))
    g = torch.Generator(device=dev).manual_seed(1)
    v = G / G.norm() + 0.5 * torch.randn(p0.shape, generator=g, device=dev, dtype=p0.dtype) / (3 * nm * 3) ** 0.5
    v /= v.norm()
    h = 0.05                                   # ~5e-5 A per atom
    Ep = f.get_energy(p0 + h * v, box, pairs, *rest, U_init=U)
    assert f.n_cycle == 0
    Em = f.get_energy(p0 - h * v, box, pairs, *rest, U_init=U)
    assert f.n_cycle == 0
    fd = (Ep - Em) / (2 * h)
    an = float((G * v).sum())
    assert abs(fd - an) < 2e-6 * abs(an) + 1e-4, (fd, an)
    # and the gradient is translation invariant in sum up to the PME mesh error
    assert float(G.sum(dim=0).abs().max()) < 1e-4 * float(G.abs().sum(dim=0).max())


@pytest.mark.parametrize('prec', ['double', 'single'])
def test_repeated_evaluations_agree_to_roundoff(precision, prec):
    """The real-space rows have a fixed summation order and nothing on the single-GPU path scatters per-atom results
    with global atomics; the only order-dependent sums are the f64 LDS tiles of the spread and the four energy words
    (block sums combined by double atomics).  Repeated evaluations therefore agree to round-off of the f64 mesh sums
    -- also after the neighbour table has been rebuilt from the same positions (another row order)."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = prec
    tol = 1e-12 if prec == 'double' else 5e-6      # f32: another row order = another summation order
    for n_mol in (216, 8192):                     # scan-spread / brick-spread regimes
        pos, box, at, ai, cov, par, pairs = water_system(n_mol, 4, True)
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        args = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
        runs = []
        for rep in range(3):
            if rep == 2:
                f.update_neighbors(torch.as_tensor(pos, device='cuda', dtype=torch.float64 if prec == 'double' else torch.float32), box)
                E, G = f.get_forces(pos, box, None, *args)
            else:
                E, G = f.get_forces(pos, box, pairs, *args)
            runs.append((float(E), np.array(G), np.array(f.U_ind), f.n_cycle))
        for E, G, U, cyc in runs[1:]:
            assert cyc == runs[0][3]
            assert abs(E - runs[0][0]) <= (1e-12 if prec == 'double' else 1e-7) * max(abs(x) for x in f.energy_parts)
            assert rel(G, runs[0][1]) < tol and rel(U, runs[0][2]) < tol


def test_update_neighbors_equals_explicit_pair_list(precision):
    """Fused GPU neighbour search + table build (update_neighbors, pairs=None) against the explicit pair list."""
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    pos, box, at, ai, cov, par, pairs = water_system(216, 13, True)
    rest = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    E0, G0 = f.get_forces(pos, box, pairs, *rest)
    g = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    with pytest.raises(ValueError):
        g.get_forces(pos, box, None, *rest)
    g.update_neighbors(pos + 3 * box[1], box)               # positions outside the cell
    assert g.n_pairs == len(pairs)
    E1, G1 = g.get_forces(pos, box, None, *rest)
    scale = max(abs(p) for p in f.energy_parts)
    assert abs(E1 - E0) < 1e-11 * scale and rel(G1, G0) < 1e-11 and g.n_cycle == f.n_cycle
    # the table is rebuilt in place: a shorter list into the old buffer, then a longer one that outgrows it (the
    # optimistic fill must be repeated after the buffer has grown), then back -- each against an explicit list
    for rc in (3.0, 5.5, 4.0):
        g.update_neighbors(pos, box, rc=rc)
        ref_pairs = S.build_pairs(pos, box, rc)
        assert g.n_pairs == len(ref_pairs)
        Ea, Ga = g.get_forces(pos, box, None, *rest)
        Eb, Gb = f.get_forces(pos, box, ref_pairs, *rest)
        assert abs(Ea - Eb) < 1e-11 * scale and rel(Ga, Gb) < 1e-11, rc


def test_slab_warm_regime_matches_fused_path(precision):
    """Decomposed handle (1 and 2 ranks), repeated warm-started calls: the speculative first cycle must give the
    single-GPU result both when the check passes and when it fails."""
    import threading
    from admp_amd.parallel import SlabPme, ThreadComm
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    pos, box, at, ai, cov, par, pairs = water_system(125, 5, True)
    args = (pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    ref = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    E0, G0 = ref.get_forces(*args)
    U0 = ref.U_ind.copy()
    E1, G1 = ref.get_forces(*args, U_init=U0)
    for nranks in (1, 2):
        world = ThreadComm.World(nranks)
        out, errors = [None] * nranks, []

        def work(rank):
            try:
                f = SlabPme(ThreadComm(world, rank), box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='replicated')
                a = f.get_forces(*args)                       # cold
                b = f.get_forces(*args, U_init=U0)            # plain loop, passes at once -> arms the warm regime
                c = f.get_forces(*args, U_init=U0)            # speculative cycle, passes
                n_c = f.n_cycle
                d = f.get_forces(*args)                       # speculative cycle fails (U_init = 0), falls back
                out[rank] = (a, b, c, n_c, d, f.n_cycle, f.U_ind)
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))
                world.barrier.abort()
        ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        [t.start() for t in ts]
        [t.join(timeout=300) for t in ts]
        assert not errors, errors
        scale = max(abs(p) for p in ref.energy_parts)
        for (a, b, c, n_c, d, n_d, U) in out:
            assert abs(a[0] - E0) < 1e-10 * scale and rel(a[1], G0) < 1e-10
            assert abs(b[0] - E1) < 1e-10 * scale and rel(b[1], G1) < 1e-10
            assert n_c == 0 and abs(c[0] - E1) < 1e-10 * scale and rel(c[1], G1) < 1e-10
            assert n_d == 2 and abs(d[0] - E0) < 1e-10 * scale and rel(d[1], G0) < 1e-10 and rel(U, U0) < 1e-10


def test_slab_all_terms_and_parameter_outputs(precision):
    """What the reference's drivers evaluate per step (examples/water_1024/run_admp.py:115-137: PME, dispersion PME and
    the Tang-Toennies pair term) on 2 and 3 slab ranks, plus dE/dQ_local of the decomposed PME: every rank returns the
    single-GPU numbers.  Both precisions for the dispersion meshes (32-bit fixed-point tiles in f32)."""
    import threading
    import torch
    from admp_amd.parallel import SlabPme, SlabDispPme, SlabPairInteraction, ThreadComm
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
    n_mol = 1728
    pos, box = S.synthetic_water_box(n_mol, seed=17)
    pos = np.mod(pos, box[0, 0])                      # wrapped: molecules straddle the cell and slab faces
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    pairs = S.build_pairs(pos, box, 4.0)
    tt_lists = (par['a_list'], par['b_list'], par['q_list'], par['c_list'][:, 0].copy())
    pol_args = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    for prec, tol in (('double', 1e-10), ('single', 3e-4)):
        settings.PRECISION = prec
        f0 = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        E0, G0, dQ0 = f0.get_forces_and_dQ(pos, box, pairs, *pol_args)
        d0 = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
        Ed0, Gd0 = d0.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
        t0 = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
        Et0, Gt0 = t0.value_and_grad(pos, box, pairs, par['mScales'], *tt_lists)
        for nranks in ((2, 3) if prec == 'double' else (2,)):
            world = ThreadComm.World(nranks)
            out, errors = [None] * nranks, []

            def work(rank):
                try:
                    comm = ThreadComm(world, rank)
                    f = SlabPme(comm, box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='replicated')
                    r = f.get_forces_and_dQ(pos, box, pairs, *pol_args)
                    d = SlabDispPme(comm, box, cov, 4.0, 1e-4, 10, outputs='replicated')
                    rd = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
                    t = SlabPairInteraction(comm, TT_damping_qq_c6_kernel, cov, outputs='replicated')
                    rt = t.value_and_grad(pos, box, pairs, par['mScales'], *tt_lists)
                    out[rank] = (r, f.n_cycle, rd, d.energy_parts, rt, f.n_home, d.n_home, t.n_home)
                except Exception as e:      # noqa: BLE001
                    errors.append((rank, repr(e)))
                    try:
                        world.barrier.abort()
                    except Exception:
                        pass
            ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
            [t.start() for t in ts]
            [t.join(timeout=600) for t in ts]
            assert not errors, errors
            sc = max(abs(p) for p in f0.energy_parts)
            scd = max(abs(p) for p in d0.energy_parts)
            for k in (5, 6, 7):
                assert sum(o[k] for o in out) == 3 * n_mol             # every calculator's home lists partition the atoms
            for (r, ncyc, rd, dparts, rt, *_) in out:
                assert abs(r[0] - E0) < tol * sc and rel(r[1], G0) < tol and rel(r[2], dQ0) < tol and ncyc == f0.n_cycle
                assert abs(rd[0] - Ed0) < tol * scd and rel(rd[1], Gd0) < max(tol, 1e-9)
                for a, b in zip(dparts, d0.energy_parts):
                    assert abs(a - b) < tol * scd
                assert abs(rt[0] - Et0) < tol * abs(Et0) and rel(rt[1], Gt0) < max(tol, 1e-9)


def test_slab_moving_sequence_with_migrating_atoms_and_fused_x_pass(precision):
    """A warm-started sequence of displaced frames on 2 slab ranks that keep only their HOME rows (outputs='home'), on a
    power-of-two mesh (K = 64: the x lines of the distributed transform run in the fused kernel on the transposed layout).
    The whole box drifts along x from frame to frame, so atoms change hands between the evaluations: their induced dipoles
    must travel with them -- every step then takes the SCF cycles of the single-GPU sequence and returns its numbers."""
    import threading
    import torch
    from admp_amd.parallel import SlabPme, ThreadComm
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    n_mol, nranks, nframes = 1000, 2, 4
    pos0, box = S.synthetic_water_box(n_mol, seed=23)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    rng = np.random.default_rng(5)
    frames = [pos0 + np.array([0.9 * k, 0.0, 0.0]) + 0.01 * k * rng.normal(size=pos0.shape) for k in range(nframes)]
    rest = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])

    def run(f, keep_home=None):
        out, U = [], None
        for p in frames:
            pairs = S.build_pairs(p, box, 4.0)
            E, G = f.get_forces(p, box, pairs, *rest, U_init=U)
            U = f.U_ind
            out.append((E, np.asarray(G), np.asarray(U).copy(), f.n_cycle,
                        None if keep_home is None else f.home_atoms.cpu().numpy()))
        return out
    ref = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    for k in ('K1', 'K2', 'K3'):
        ref.update_env(k, 64)
    want = run(ref)
    world = ThreadComm.World(nranks)
    got, errors = [None] * nranks, []

    def work(rank):
        try:
            f = SlabPme(ThreadComm(world, rank), box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='home')
            for k in ('K1', 'K2', 'K3'):
                f.update_env(k, 64)
            got[rank] = run(f, keep_home=True)
        except Exception as e:      # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                world.barrier.abort()
            except Exception:
                pass
    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join(timeout=600) for t in ts]
    assert not errors, errors
    scale = max(abs(p) for p in ref.energy_parts)
    moved = 0
    for k in range(nframes):
        E0, G0, U0, n0, _ = want[k]
        homes = np.concatenate([got[r][k][4] for r in range(nranks)])
        assert len(homes) == 3 * n_mol and len(np.unique(homes)) == 3 * n_mol
        if k:
            moved += len(np.setdiff1d(got[0][k][4], got[0][k - 1][4]))
        for r in range(nranks):
            E, G, U, n, home = got[r][k]
            assert n == n0, (k, r, n, n0)                      # a migrant that lost its dipole would cost extra cycles
            assert abs(E - E0) < 1e-10 * scale and rel(G[home], G0[home]) < 1e-9 and rel(U[home], U0[home]) < 1e-9, (k, r)
    assert moved > 0                                           # atoms did change hands


def test_slab_ranks_build_only_the_rows_near_their_slab(precision):
    """Round 4 (round-3 verdict, missing #3): `update_neighbors` on a slab rank builds the rows of the atoms near its slab only
    (stencil base plane within the slab + half the list cutoff + 4 planes): fewer table entries than the full list, the same
    energies / gradient / dipoles as the single-GPU calculator on a moving sequence with a skin -- and an evaluation in which
    an atom WITHOUT a row has become a home atom is refused (ADMP_E_STATE), not silently short a row."""
    import threading
    import torch
    from admp_amd._lib import AdmpHipError
    from admp_amd.parallel import SlabPme, ThreadComm
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    n_mol, nranks = 1728, 2
    pos, box = S.synthetic_water_box(n_mol, seed=13)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    args = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    rng = np.random.default_rng(2)
    frames = [pos, pos + 0.08 * rng.normal(size=pos.shape) + np.array([0.25, 0.0, 0.0])]      # drifts across the slab faces
    f0 = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    f0.update_neighbors(frames[0], box, rc=5.0)
    f0.set_cutoff(0.0)
    ref = []
    for p in frames:
        E, G = f0.get_forces(p, box, None, *args)
        ref.append((E, np.array(G), np.array(f0.U_ind), f0.n_cycle))
    full_pairs = f0.n_pairs
    world = ThreadComm.World(nranks)
    out, errors = [None] * nranks, []

    def work(rank):
        try:
            f = SlabPme(ThreadComm(world, rank), box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='replicated')
            f.update_neighbors(frames[0], box, rc=5.0)
            res = [f.n_pairs]
            for p in frames:
                E, G = f.get_forces(p, box, None, *args)
                res.append((E, np.array(G), np.array(f.U_ind), f.n_cycle))
            far = frames[0] + np.array([0.45 * box[0, 0], 0.0, 0.0])      # every atom far from where its row was (not) built
            try:
                f.get_forces(far, box, None, *args)
                res.append('no error')
            except AdmpHipError as e:
                res.append(str(e))
            out[rank] = res
        except Exception as e:      # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                world.barrier.abort()
            except Exception:
                pass
    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join(timeout=600) for t in ts]
    assert not errors, errors
    for res in out:
        assert res[0] < 0.9 * full_pairs, (res[0], full_pairs)        # (2 ranks: slab + margins = ~70 % of the box)
        for (E, G, U, ncyc), (Er, Gr, Ur, nr) in zip(res[1:3], ref):
            assert abs(E - Er) < 1e-10 * max(abs(p) for p in f0.energy_parts) and rel(G, Gr) < 1e-10 and rel(U, Ur) < 1e-10
            assert ncyc == nr
        assert 'does not hold' in res[3], res[3]


def test_slab_halo_only_traffic_and_home_outputs(precision):
    """outputs='home': a rank returns its home rows and nothing proportional to the number of atoms is ever sent -- the
    SCF exchanges only the dipoles of imported atoms (all-to-all-v over index lists), the gradient only what a rank
    added to atoms it does not own.  Wrapped atoms so that molecules straddle the slab boundaries (frame adjoint across
    ranks).  Results: the home rows equal the single-GPU rows."""
    import threading
    import torch
    from admp_amd.parallel import SlabPme, ThreadComm
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    n_mol, nranks = 4096, 2
    pos, box = S.synthetic_water_box(n_mol, seed=12)
    pos = np.mod(pos, box[0, 0])
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    na = 3 * n_mol
    args = (pos, box, None, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    ref = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    ref.update_neighbors(torch.as_tensor(pos, device='cuda'), box)
    E0, G0 = ref.get_forces(*args)
    U0 = np.asarray(ref.U_ind)
    world = ThreadComm.World(nranks)
    out, errors = [None] * nranks, []

    def work(rank):
        try:
            comm = ThreadComm(world, rank)
            f = SlabPme(comm, box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='home')
            f.update_neighbors(torch.as_tensor(pos, device='cuda'), box)
            E, G = f.get_forces(*args)
            first = (dict(comm.bytes_sent), dict(comm.calls), f.n_cycle, f.n_import)
            Uh = np.asarray(f.U_ind).copy()
            comm.reset_stats()
            E2, G2 = f.get_forces(*args, U_init=Uh)          # warm start from HOME rows only: imports are pulled
            out[rank] = (E, np.asarray(G), Uh, f.home_atoms.cpu().numpy(), first, E2, np.asarray(G2), f.n_cycle,
                         dict(comm.bytes_sent))
        except Exception as e:      # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                world.barrier.abort()
            except Exception:
                pass
    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join(timeout=600) for t in ts]
    assert not errors, errors
    scale = max(abs(p) for p in ref.energy_parts)
    homes = np.concatenate([o[3] for o in out])
    assert len(homes) == na and len(np.unique(homes)) == na               # the home sets partition the atoms
    for (E, G, U, home, (sent, calls, ncyc, n_imp), E2, G2, ncyc2, sent2) in out:
        assert abs(E - E0) < 1e-10 * scale and ncyc == ref.n_cycle
        assert rel(G[home], G0[home]) < 1e-10 and rel(U[home], U0[home]) < 1e-10
        other = np.setdiff1d(np.arange(na), home)
        assert np.abs(G[other]).max() == 0.0                               # only home rows are handed out
        assert abs(E2 - E0) < 1e-6 * scale and rel(G2[home], G0[home]) < 1e-6 and ncyc2 == 0
        # halo-only: the imports are a surface layer (rc + a bond on both faces of a 24.8 A slab), not the volume
        assert 0 < n_imp < 0.45 * na
        w = 8
        assert 'replicate_outputs' not in sent and 'all_reduce' not in sent
        assert sent['halo_gradient'] == 3 * w * n_imp                     # my contributions to my imports, back to the owners
        assert sent['halo_dipoles'] % (calls['halo_dipoles'] * 3 * w) == 0 and \
            sent['halo_dipoles'] // (calls['halo_dipoles'] * 3 * w) < 0.45 * na   # rows my peers import from me, per cycle
        # dipoles: once at the start (the imports' starting values) and once per Jacobi step (their changes); no index lists
        # travel any more -- both ends derive them from the replicated inputs
        assert calls['halo_dipoles'] == ncyc + 1 and calls['halo_gradient'] == 1 and calls['scf_max'] == ncyc + 1
        assert 'halo_lists' not in sent
        per_atom_traffic = sent['halo_dipoles'] + sent['halo_gradient']
        assert per_atom_traffic < 0.5 * (ncyc + 1) * na * 3 * w            # what the full-array all-reduces used to move
        assert sent['scf_max'] <= 8 * (ncyc + 1) and sent['energies'] <= 32


@pytest.mark.parametrize('prec,tol', [('single', 3e-5), ('double', 2e-6)])
def test_s2_config_size_vs_oracle_golden(precision, prec, tol):
    """BASELINE configs[2] AT SIZE (98 304 atoms, K = 128, rc 4 A) against the float64 oracle's numbers committed in
    tests/golden/s2_98304.npz (tests/golden/make_s2_golden.py, chunked evaluation): energy parts, gradient, dipoles and
    SCF cycle count, non-polarizable and polarizable.  Bar (north_star): 1e-2 relative in single precision; the test pins the
    ACHIEVED level instead (gradient and dipoles <= 3e-5; 1.3e-5 / 1.2e-5 measured) so that a kernel change that spends the
    margin shows up here -- round 3's 32-bit spread tile quadrupled the f32 force error at 1M atoms and no test saw it.
    The stored gradient / dipoles are float32 (6e-8), hence 2e-6 for the double-precision check."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    g = np.load(os.path.join(GOLD, 's2_98304.npz'))
    settings.PRECISION = prec
    nm = int(g['n_mol'])
    pos, box = S.synthetic_water_box(nm, seed=int(g['seed']))
    np.testing.assert_allclose([pos.sum(), (pos ** 2).sum()], g['pos_checksum'], rtol=1e-12)
    at, ai, cov = S.water_topology(nm)
    K = int(g['K'])
    dt = torch.float32 if prec == 'single' else torch.float64
    p = torch.as_tensor(pos, dtype=dt, device='cuda')
    for lpol in (False, True):
        par = S.water_parameters(nm, polarizable=lpol)
        f = ADMPPmeForce(box, at, ai, cov, float(g['rc']), 1e-4, 2, lpol=lpol)
        assert abs(f.kappa - float(g['kappa'])) < 1e-15
        for k in ('K1', 'K2', 'K3'):
            f.update_env(k, K)
        f.update_neighbors(p, box)                      # GPU cell list: the same pair set as the oracle's cKDTree list
        assert f.n_pairs == int(g['n_pairs'])
        if lpol:
            E, G = f.get_forces(p, box, None, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'],
                                par['dScales'])
            parts, Gref = g['pol_parts'], g['pol_grad'].astype(np.float64)
            assert f.n_cycle == int(g['pol_n_cycle']) and f.lconverg == bool(g['pol_lconverg'])
            eu = rel(f.U_ind.cpu().numpy(), g['pol_U'].astype(np.float64))
            assert eu < tol
        else:
            E, G = f.get_forces(p, box, None, par['Q_local'], par['mScales'])
            parts, Gref, eu = g['np_parts'], g['np_grad'].astype(np.float64), 0.0
        scale = np.abs(parts).max()
        ee = max(abs(a - b) for a, b in zip(f.energy_parts, parts)) / scale
        eg = rel(G.cpu().numpy(), Gref)
        print('S2 %s lpol=%s: energy parts %.2e of the largest part, gradient rel L2 %.2e, dipoles rel L2 %.2e' %
              (prec, lpol, ee, eg, eu))
        assert ee < (1e-9 if prec == 'double' else 1e-6) and eg < tol


@pytest.mark.gpu
@pytest.mark.parametrize('prec,tolE,tolG', [('double', 1e-9, 1e-8), ('single', 2e-4, 5e-4)])
def test_charge_only_site_classes_follow_the_parameters(precision, prec, tolE, tolG):
    """The neighbour table is compiled with the atoms' classes (charge-only or not: NbrTable::cls in launch.h) one call
    after the sites show them, and the pair kernel takes the reduced forms of pme_math.h for charge-only sites from then
    on.  Every call must match the oracle whatever the table currently assumes: (1) first call, nothing known, general
    form; (2) classes compiled in, reduced forms; (3) the hydrogens get a dipole: the table is stale, its marks are
    ignored; (4) recompiled without charge-only atoms; (5) back to charge-only hydrogens within the quiet period (general
    form, table not recompiled); (6) a new pair list keeps the classes; and dE/dQ_local requests (which need the full
    potential of every row) on a table with classes."""
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = prec
    pos, box, at, ai, cov, par, pairs = water_system(216, 23, True)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    rest = (par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    Q0 = par['Q_local']
    Q1 = Q0.copy()
    Q1[1::3, 1] = 0.11; Q1[2::3, 3] = -0.07; Q1[1::3, 4] = 0.05     # hydrogens with a dipole / a quadrupole component
    refs = {}

    def check(Q, what):
        key = id(Q)
        if key not in refs:
            p = dict(par, Q_local=Q)
            refs[key] = oracle_es(pos, box, at, ai, cov, p, pairs, f.kappa, (f.K1, f.K2, f.K3), True)
        ref = refs[key]
        E, G = f.get_forces(pos, box, pairs, Q, *rest)
        scale = max(abs(p) for p in ref['parts'])
        for got, want in zip(f.energy_parts, ref['parts']):
            assert abs(got - want) <= tolE * scale, what
        assert f.n_cycle == ref['n_cycle'], what
        assert rel(G, ref['grad']) < tolG and rel(f.U_ind, ref['U_ind']) < tolG, what

    check(Q0, 'first call')
    check(Q0, 'classes compiled in')
    check(Q0, 'steady state')
    check(Q1, 'stale table')
    check(Q1, 'recompiled, no charge-only atoms')
    check(Q0, 'charge-only again, quiet period')
    pairs = pairs[::-1].copy()                      # another pair list (same set): table rebuilt, classes kept
    check(Q0, 'new pair list')
    for _ in range(10):
        f.get_forces(pos, box, pairs, Q0, *rest)
    check(Q0, 'after the quiet period')
    # dE/dQ_local on a table with classes: general form, full potential on every row
    p = dict(par, Q_local=Q0)
    fn = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    ref = oracle_es(pos, box, at, ai, cov, p, pairs, fn.kappa, (fn.K1, fn.K2, fn.K3), False, want_dQ=True)
    for rep in range(3):
        if rep == 1:
            fn.get_forces(pos, box, pairs, Q0, par['mScales'])      # a call that may take the reduced forms
        E, G, dQ = fn.get_forces_and_dQ(pos, box, pairs, Q0, par['mScales'])
        assert rel(G, ref['grad']) < tolG and rel(dQ, ref['dQ_local']) < tolG


@pytest.mark.gpu
def test_two_level_dft_on_a_bluestein_mesh_above_160(precision):
    """Mesh dimensions with a prime factor above 13 that are too long for plain O(N^2) lines -- 305 = 5 * 61 is what the
    reference's setup_ewald_parameters gives for the 98 304-atom box (admp/pme.py:146-172) -- go through the two-level
    kernels of pfa_kernels.hip by default.  Same numbers as the rocFFT (Bluestein) leg: energy parts, forces, induced
    dipoles, the dispersion terms, and the box gradient (whose reciprocal passes stay on rocFFT and need the natural-order
    G table next to the slot-ordered one)."""
    import subprocess
    import sys
    code = """
import os, sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_parity import water_system
from admp_amd import settings
settings.PRECISION = 'double'
settings.REFERENCE_KPOINT_ORDER = False      # unequal mesh: the transform legs are compared on a self-consistent Ewald sum
from admp_amd.pme import ADMPPmeForce
from admp_amd.disp_pme import ADMPDispPmeForce
pos, box, at, ai, cov, par, pairs = water_system(216, 5, True)
out = {}
f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
for o in (f, d):
    o.K1, o.K2, o.K3 = 305, 170, 183          # 5 * 61, 10 * 17, 3 * 61
    o.refresh_calculators()
args = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
E, G = f.get_forces(pos, box, pairs, *args)
out['parts'] = np.asarray(f.energy_parts); out['G'] = np.asarray(G); out['U'] = np.asarray(f.U_ind)
Ed, Gd = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
out['dparts'] = np.asarray(d.energy_parts); out['Gd'] = np.asarray(Gd)
Eb, dbox = f.get_energy_and_box_gradient(pos, box, pairs, *args, U_init=f.U_ind)
out['Eb'] = np.asarray(Eb); out['dbox'] = np.asarray(dbox)
np.savez(sys.argv[1], **out)
print('PFA-RUN-OK')
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode, extra in {'rocfft': dict(ADMP_DFT='0'), 'pfa': {}}.items():
        path = os.path.join('/tmp', 'pfa_%s_%d.npz' % (mode, os.getpid()))
        r = subprocess.run([sys.executable, '-c', code, path], capture_output=True, text=True, env=dict(os.environ, **extra),
                           timeout=900)
        assert r.returncode == 0 and 'PFA-RUN-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        res[mode] = dict(np.load(path))
        os.remove(path)
    for key, a in res['rocfft'].items():
        b = res['pfa'][key]
        assert np.abs(a - b).max() <= 1e-10 * np.abs(a).max(), (key, np.abs(a - b).max(), np.abs(a).max())


@pytest.mark.gpu
def test_toy_two_waters_vs_reference_held_mpid_dipoles(precision):
    """The HIP path against the reference-held induced dipoles of the 2-water toy (`examples/water_pol_1024/dipole_2`, MPID
    OpenMM plugin; see tests/test_oracle_physics.py for what this number is and is not): tight SCF, percent-level bar."""
    import json
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    g = json.load(open(os.path.join(GOLD, 'ref_water2_mpid_dipoles.json')))
    pos, box = np.array(g['positions_A']), np.array(g['box_A'])
    ref = np.array(g['induced_dipoles_nm_e']) * 10.0
    at, ai, cov = S.water_topology(2)
    par = S.water_parameters(2, True)
    old = settings.POL_CONV
    settings.POL_CONV = 1e-8
    try:
        f = ADMPPmeForce(box, at, ai, cov.toarray(), 4.0, 1e-4, 2, lpol=True)
        pairs = S.build_pairs(pos, box, 4.0)
        f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
        U = np.asarray(f.U_ind)
        assert f.lconverg
        assert np.linalg.norm(U - ref) / np.linalg.norm(ref) < 0.035
        big = np.unravel_index(np.abs(ref).argmax(), ref.shape)
        assert abs(U[big] - ref[big]) < 5e-3 * abs(ref[big])
    finally:
        settings.POL_CONV = old


@pytest.mark.gpu
def test_fused_x_pass_vs_rocfft_3d(tmp_path):
    """Power-of-two K1: the k-space leg is rocFFT on the y-z planes + ONE kernel doing x forward * G (+ energy) * x inverse
    (fftx_kernels.hip) instead of 3-D rocFFT plans + k_kspace.  Same numbers to round-off: polarizable PME and dispersion PME,
    both precisions, x lengths 32 / 64 / 128 with other (even, odd) dimensions."""
    import subprocess
    import sys
    code = """
import os, sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_parity import water_system
from admp_amd import settings
settings.REFERENCE_KPOINT_ORDER = False      # unequal meshes: the transform legs are compared on a self-consistent Ewald sum
out = {}
pos, box, at, ai, cov, par, pairs = water_system(216, 5, True)
for prec in ('double', 'single'):
    settings.PRECISION = prec
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    for K in ((32, 30, 36), (64, 48, 45), (128, 64, 50)):
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
        for o in (f, d):
            o.K1, o.K2, o.K3 = K
            o.refresh_calculators()
        E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
        Ed, Gd = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
        key = '%%s_%%d_%%d_%%d' %% ((prec,) + K)
        out[key + '_parts'] = np.asarray(f.energy_parts); out[key + '_G'] = np.asarray(G); out[key + '_U'] = np.asarray(f.U_ind)
        out[key + '_dparts'] = np.asarray(d.energy_parts); out[key + '_Gd'] = np.asarray(Gd)
np.savez(sys.argv[1], **out)
print('FX-RUN-OK')
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # fused_x: spectrum rows padded to whole 128-byte lines
    for mode, extra in {'rocfft3d': dict(ADMP_FUSED_X='0'), 'fused_x': {}}.items():
        path = str(tmp_path / ('%s.npz' % mode))
        r = subprocess.run([sys.executable, '-c', code, path], capture_output=True, text=True, env=dict(os.environ, **extra),
                           timeout=900)
        assert r.returncode == 0 and 'FX-RUN-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        res[mode] = dict(np.load(path))
    assert len(res['rocfft3d']) == 2 * 3 * 5
    for key, a in res['rocfft3d'].items():
        for mode in ('fused_x',):
            b = res[mode][key]
            tol = 1e-10 if key.startswith('double') else 2e-4
            assert np.abs(a - b).max() <= tol * np.abs(a).max(), (mode, key, np.abs(a - b).max(), np.abs(a).max())


def test_first_cycle_forms_agree_on_a_moving_sequence(tmp_path):
    """engine.hip pme(): the first SCF cycle of a call runs either as field kernels followed by the closing pass, or
    speculatively with the full kernels (one host synchronisation when its check passes); which one is chosen from the residual
    history of the previous calls (scf_last + scf_growth < threshold).  Every choice must give the same step: energies,
    gradient, dipoles, cycle count and flag on a sequence of displaced frames, warm-started like an MD run."""
    import subprocess
    import sys
    code = """
import os, sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_parity import water_system
from admp_amd import settings
from admp_amd.pme import ADMPPmeForce
pos, box, at, ai, cov, par, pairs = water_system(216, 5, True)
rng = np.random.default_rng(3)
vel = rng.normal(size=pos.shape) * 0.012
out = {}
for conv in (10.0, 0.01):          # the reference's threshold (0 / 1 Jacobi steps per call) and a tight one (5 per call)
    settings.POL_CONV = conv
    for prec in ('double', 'single'):
        settings.PRECISION = prec
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        U = None
        for k in range(8):
            p = pos + vel * k
            E, G = f.get_forces(p, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'],
                                par['dScales'], U_init=U)
            U = np.asarray(f.U_ind).copy()
            tag = '%%s_%%g_%%d' %% (prec, conv, k)
            out[tag + '_E'] = np.asarray(f.energy_parts)
            out[tag + '_G'] = np.asarray(G)
            out[tag + '_U'] = U
            out[tag + '_c'] = np.array([f.n_cycle, int(f.lconverg)])
np.savez(sys.argv[1], **out)
print('SEQ-OK', [int(out['double_10_%%d_c' %% k][0]) for k in range(8)], [int(out['double_0.01_%%d_c' %% k][0]) for k in range(8)])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # auto: the form is chosen per call from the residual history, incl. the chained form (whole step enqueued at once, the
    # Jacobi step gated on the device); nochain: auto without the chained form
    for mode, extra in {'auto': dict(ADMP_SCF_TRACE='1'), 'nochain': dict(ADMP_SCF_CHAIN_MAX='0'), 'plain': dict(ADMP_SPECULATE='0'),
                        'speculative': dict(ADMP_SPECULATE='1')}.items():
        path = str(tmp_path / ('%s.npz' % mode))
        r = subprocess.run([sys.executable, '-c', code, path], capture_output=True, text=True,
                           env=dict(os.environ, **extra), timeout=600)
        assert r.returncode == 0 and 'SEQ-OK' in r.stdout, mode + r.stdout[-2000:] + r.stderr[-3000:]
        res[mode] = dict(np.load(path))
        if mode == 'auto':
            assert 'chained' in r.stderr and 'speculative' in r.stderr and 'plain' in r.stderr, r.stderr[-2000:]
            assert any('chained' in line and not line.rstrip().endswith(', 1 steps)') for line in r.stderr.splitlines()), r.stderr[-2000:]
    cycles = [int(res['plain']['double_10_%d_c' % k][0]) for k in range(8)]
    assert min(cycles) == 0 and max(cycles) >= 1, cycles          # the sequence exercises both outcomes of the first check
    assert min(int(res['plain']['double_0.01_%d_c' % k][0]) for k in range(8)) >= 3      # ... and chains of several steps
    for key, a in res['plain'].items():
        for mode in ('auto', 'nochain', 'speculative'):
            b = res[mode][key]
            if key.endswith('_c'):
                assert (a == b).all(), (mode, key, a, b)
            else:
                tol = 1e-10 if key.startswith('double') else 2e-4
                scale = np.abs(a).max()
                assert np.abs(a - b).max() <= tol * scale, (mode, key, np.abs(a - b).max(), scale)


def test_pruned_inner_list_matches_a_list_built_at_the_inner_cutoff():
    """admp_prune_pairs (round 4): the inner table an MD loop walks between two rebuilds -- the entries of the outer table (rc +
    skin) below rc + margin at the current positions -- must be the table a search at rc + margin builds: same pair count, and
    polarizable PME (which evaluates every listed pair), dispersion PME and Tang-Toennies (cutoff honoured) on it agree with the
    calculators on a list searched at the inner cutoff; pruning again after the atoms moved starts from the outer table; a
    rebuild ends it.  Both precisions."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    old = settings.PRECISION
    try:
        for prec, tol in (('double', 1e-10), ('single', 2e-5)):
            settings.PRECISION = prec
            dt = torch.float64 if prec == 'double' else torch.float32
            n_mol = 512
            pos, box, at, ai, cov, par, _ = water_system(n_mol, 9, True)
            rng = np.random.default_rng(3)
            p0 = torch.as_tensor(pos, dtype=dt, device='cuda')
            p1 = torch.as_tensor(pos + rng.normal(scale=0.05, size=pos.shape), dtype=dt, device='cuda')      # "later in the loop"
            T = lambda k: torch.as_tensor(np.asarray(par[k]), dtype=dt, device='cuda')      # noqa: E731
            Q, pol, th, cl = T('Q_local'), T('pol'), T('tholes'), T('c_list')
            a_, b_, q_, c6 = T('a_list'), T('b_list'), T('q_list'), T('c_list')[:, 0].contiguous()
            res = {}
            for mode in ('pruned', 'searched'):
                f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
                d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
                tt_obj = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
                tt = value_and_grad(tt_obj)
                if mode == 'pruned':
                    f.update_neighbors(p0, box, rc=5.0)                     # outer list: rc + 1 A skin, at the old positions
                    n_outer = f.n_pairs
                    # (a loop has evaluated before it prunes: the first evaluation of a handle compiles the site classes into
                    # the table as built, which ends a pruning)
                    f.get_forces(p0, box, None, Q, pol, th, par['mScales'], par['pScales'], par['dScales'])
                    f.prune_neighbors(p0, box, 4.6)                         # a first prune at the old positions ...
                    f.prune_neighbors(p1, box, 4.4)                         # ... and one later: from the outer table again
                    assert f.n_pairs < n_outer
                else:
                    f.update_neighbors(p1, box, rc=4.4)
                    f.get_forces(p0, box, None, Q, pol, th, par['mScales'], par['pScales'], par['dScales'])
                for o in (d, tt_obj):
                    o.share_neighbors(f)
                    o.set_cutoff(4.0)
                E, G = f.get_forces(p1, box, None, Q, pol, th, par['mScales'], par['pScales'], par['dScales'])
                Ed, Gd = d.get_forces(p1, box, None, cl, par['mScales'])
                Et, Gt = tt(p1, box, None, par['mScales'], a_, b_, q_, c6)
                res[mode] = (f.n_pairs, E, G.cpu().numpy(), f.U_ind.cpu().numpy(), f.n_cycle, Ed, Gd.cpu().numpy(), Et, Gt.cpu().numpy())
                if mode == 'pruned':                                         # a rebuild ends the pruning
                    f.update_neighbors(p1, box, rc=5.0)
                    assert f.n_pairs > res[mode][0]
            a, b = res['pruned'], res['searched']
            assert a[0] == b[0], (prec, a[0], b[0])
            assert a[4] == b[4]
            for k in (1, 5, 7):
                assert abs(a[k] - b[k]) <= tol * max(abs(b[k]), 1.0), (prec, k, a[k], b[k])
            for k in (2, 3, 6, 8):
                assert rel(a[k], b[k]) < max(tol, 1e-10) * (50 if prec == 'single' else 1), (prec, k, rel(a[k], b[k]))
    finally:
        settings.PRECISION = old


def test_small_system_paths_fuzz_against_fallbacks():
    """tools/fuzz_small_paths.py: six random water boxes (375 ... 7800 atoms) on direct-DFT meshes, ten warm-started steps each
    (plain, chained and speculative SCF calls all occur), with the round-4 small-system paths on (spread inside the forward plane
    transform, closing work in the gather, last chained residual in the closing gather, field kernels riding in the x pass, one
    stream) against the same library with all of them off: energies, gradient, dipoles to 1e-9, cycle counts and flags equal."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_small_paths.py')
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'fuzz ok' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
