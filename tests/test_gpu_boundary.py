"""Boundary rows of SURVEY.md section 8 that round 2 closed, checked on the GPU through the C ABI:
the reference's literal k-point order (a18), the bare calculators energy_fn / grad_U_fn / grad_pos_fn and the
construct_local_frames / pme_recip attributes (a1), the `admp` import name (b), the numpy pair-list cache."""
import os

import numpy as np
import pytest

from admp_amd import settings
from admp_amd import systems as S

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.fixture()
def env():
    old = (settings.PRECISION, settings.REFERENCE_KPOINT_ORDER)
    settings.PRECISION = 'double'
    yield
    settings.PRECISION, settings.REFERENCE_KPOINT_ORDER = old


def orthorhombic_water(n_mol=96, seed=3, scale=(1.0, 1.25, 1.6)):
    """liquid box stretched to unequal edges (molecules kept rigid) so that the reference rule gives K1 != K2 != K3"""
    pos, box = S.synthetic_water_box(n_mol, seed=seed)
    mol = pos.reshape(n_mol, 3, 3)
    com = mol.mean(axis=1, keepdims=True)
    sc = np.asarray(scale)
    pos2 = (com * sc + (mol - com)).reshape(-1, 3)
    box2 = box * sc[None, :]
    at, ai, cov = S.water_topology(n_mol)
    return pos2, box2, at, ai, cov


@pytest.mark.parametrize('lpol', [False, True])
def test_reference_kpoint_order_vs_oracle_quirk(env, lpol):
    """a18: through the reference's import name and with the DEFAULT settings the HIP tables reproduce the reference's
    meshgrid(kz, kx, ky) order (admp/recip.py:339-340) on a box where it matters (K1 != K2 != K3); the oracle runs
    UNMODIFIED (quirk=True default).  The drop-in says once that this order is not a consistent Ewald sum here."""
    import warnings
    import admp.settings
    from admp.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    pos, box, at, ai, cov = orthorhombic_water()
    par = S.water_parameters(len(pos) // 3, polarizable=lpol)
    pairs = S.build_pairs(pos, box, 4.0)
    assert admp.settings is settings and settings.REFERENCE_KPOINT_ORDER is True      # the default of a drop-in
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
    assert len({f.K1, f.K2, f.K3}) == 3
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, lpol)
    with pytest.warns(UserWarning, match='literal order'):
        if lpol:
            E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'],
                                par['dScales'])
            ref = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                                        par['pScales'])
            assert f.n_cycle == ref['n_cycle'] and rel(f.U_ind, ref['U_ind']) < 1e-8
        else:
            E, G = f.get_forces(pos, box, pairs, par['Q_local'], par['mScales'])
            ref = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'])
    scale = max(abs(p) for p in ref['parts'])
    for got, want in zip(f.energy_parts, ref['parts']):
        assert abs(got - want) <= 1e-9 * scale
    assert rel(G, ref['grad']) < 1e-8
    # and the consistent assignment (opt-in) is a DIFFERENT number here -- the switch is not a no-op; it does not warn
    settings.REFERENCE_KPOINT_ORDER = False
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        f2 = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
        if lpol:
            f2.get_forces(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'],
                          par['dScales'])
        else:
            f2.get_forces(pos, box, pairs, par['Q_local'], par['mScales'])
    assert abs(f2.energy_parts[1] - ref['parts'][1]) > 1e-6 * scale


def test_reference_kpoint_order_dispersion(env):
    from admp_amd.disp_pme import ADMPDispPmeForce
    from oracle import admp_oracle as O
    pos, box, at, ai, cov = orthorhombic_water()
    par = S.water_parameters(len(pos) // 3)
    pairs = S.build_pairs(pos, box, 4.0)
    settings.REFERENCE_KPOINT_ORDER = True
    d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
    E, G = d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])
    ref = O.disp_energy_and_grad(pos, box, pairs, par['c_list'], par['mScales'], cov, d.kappa, (d.K1, d.K2, d.K3), 10)
    np.testing.assert_allclose(d.energy_parts, ref['parts'], rtol=1e-9)
    assert rel(G, ref['grad']) < 1e-8


def test_bare_calculators_at_given_dipoles(env):
    """energy_fn / grad_U_fn / grad_pos_fn (admp/pme.py:69-78) at dipoles that are NOT the SCF solution, against autograd
    of the oracle's energy_pme with respect to Uind_global and positions."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.pairwise import grad
    from oracle import admp_oracle as O
    n_mol = 64
    pos, box = S.synthetic_water_box(n_mol, seed=21)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    pairs = S.build_pairs(pos, box, 4.0)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    rng = np.random.default_rng(0)
    U = rng.normal(size=(3 * n_mol, 3)) * 0.05 * (par['pol'] > 0)[:, None]
    args = (pos, box, pairs, par['Q_local'], U, par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    E = f.energy_fn(*args)
    gU = f.grad_U_fn(*args)
    gp = f.grad_pos_fn(*args)
    assert grad(f.energy_fn, argnums=4) == f.grad_U_fn and grad(f.energy_fn, argnums=0) == f.grad_pos_fn
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, True)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    p, Ut = T(pos).requires_grad_(True), T(U).requires_grad_(True)
    e = O.energy_pme(sysm, p, T(box), pairs, T(par['Q_local']), Ut, T(par['pol']), T(par['tholes']), T(par['mScales']),
                     T(par['pScales']))
    rp, rU = torch.autograd.grad(e, [p, Ut])
    assert abs(E - float(e.detach())) < 1e-9 * max(abs(x) for x in f.energy_parts)
    assert rel(gp, rp.numpy()) < 1e-8
    site = par['pol'] > 0
    assert rel(gU[site], rU.numpy()[site]) < 1e-8
    # the SCF loop written by hand with the bare calculators (admp/pme.py:130-138) lands on optimize_Uind's result
    Uc = np.zeros_like(U)
    for i in range(30):
        fld = f.grad_U_fn(pos, box, pairs, par['Q_local'], Uc, *args[5:])
        if np.abs(fld[site]).max() < settings.POL_CONV:
            break
        Uc = Uc - fld * par['pol'][:, None] / 1389.35455846
    U2, flag, n = f.optimize_Uind(pos, box, pairs, par['Q_local'], *args[5:])
    assert n == i and rel(Uc, U2) < 1e-10


def test_local_frames_and_pme_recip_attributes(env):
    import torch
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    n_mol = 27
    pos, box = S.synthetic_water_box(n_mol, seed=4)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, False)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    fr = f.construct_local_frames(pos, box)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    ref = O.construct_local_frames(T(pos), T(box), at, ai)
    assert fr.shape == (3 * n_mol, 3, 3) and rel(fr, ref.numpy()) < 1e-12
    Qg = O.rot_local2global(T(par['Q_local']), ref, 2)
    e = f.pme_recip(pos, box, Qg.numpy())
    want = float(O.pme_recip(T(pos), T(box), Qg, f.kappa, (f.K1, f.K2, f.K3), 2))
    assert abs(e - want) < 1e-9 * abs(want)


def test_admp_import_name_and_value_and_grad(env):
    """b: the reference's import lines resolve to the HIP calculators; value_and_grad(get_energy) is get_forces."""
    import admp.settings
    from admp.multipole import convert_cart2harm       # noqa: F401
    from admp.pme import ADMPPmeForce
    from admp.disp_pme import ADMPDispPmeForce
    from admp.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad      # noqa: F401
    import admp_amd.pme
    assert ADMPPmeForce is admp_amd.pme.ADMPPmeForce and admp.settings is settings
    n_mol = 27
    pos, box = S.synthetic_water_box(n_mol, seed=4)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, False)
    pairs = S.build_pairs(pos, box, 4.0)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    E, G = value_and_grad(f.get_energy)(pos, box, pairs, par['Q_local'], par['mScales'])
    E2, G2 = f.get_forces(pos, box, pairs, par['Q_local'], par['mScales'])
    assert abs(E - E2) < 1e-9 * abs(E) and rel(G, G2) < 1e-10      # mesh sums go through atomics: round-off only
    d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
    Ed, Gd = value_and_grad(d.get_energy)(pos, box, pairs, par['c_list'], par['mScales'])
    assert abs(Ed - d.get_forces(pos, box, pairs, par['c_list'], par['mScales'])[0]) < 1e-9 * abs(Ed)


def test_initial_dipoles_are_read_only_and_results_are_new_arrays(env):
    """admp/pme.py:104-109: `U_init` is an input (jnp arrays are immutable) and `pme.U_ind` a new array every call.  The
    wrapper hands the library the caller's array as a read-only source (admp_set_dipole_source) and a fresh output array --
    no device copy of its own; the caller's tensor must come back untouched, also when it is the previous call's U_ind."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    n_mol = 125
    pos, box = S.synthetic_water_box(n_mol, seed=6)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    pairs = S.build_pairs(pos, box, 4.0)
    rest = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    dev = lambda x: torch.as_tensor(x, device='cuda')
    p0 = dev(pos)
    E0, _ = f.get_forces(p0, box, pairs, *rest)
    U0 = f.U_ind
    assert isinstance(U0, torch.Tensor)
    keep = U0.clone()
    p1 = p0 + 0.05 * torch.randn(p0.shape, generator=torch.Generator(device='cuda').manual_seed(3), device='cuda', dtype=p0.dtype)
    E1, _ = f.get_forces(p1, box, pairs, *rest, U_init=U0)        # warm start from the previous result, geometry moved
    U1 = f.U_ind
    assert U1.data_ptr() != U0.data_ptr() and torch.equal(U0, keep)
    assert float((U1 - U0).abs().max()) > 1e-6                     # the dipoles did move
    cold = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    Ec, _ = cold.get_forces(p1, box, pairs, *rest, U_init=keep)    # the same call on another calculator
    assert abs(Ec - E1) < 1e-9 * max(abs(x) for x in cold.energy_parts)
    assert float((cold.U_ind - U1).abs().max()) < 1e-9 * float(U1.abs().max())


def test_numpy_pair_list_refilled_in_place(env):
    """a caller who rewrites the same ndarray must get a fresh neighbour table (the whole array is hashed)."""
    from admp_amd.pme import ADMPPmeForce
    n_mol = 64
    pos, box = S.synthetic_water_box(n_mol, seed=8)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, False)
    full = S.build_pairs(pos, box, 4.0)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    E_full = f.get_energy(pos, box, full, par['Q_local'], par['mScales'])
    buf = full.copy()
    assert abs(f.get_energy(pos, box, buf, par['Q_local'], par['mScales']) - E_full) < 1e-9 * abs(E_full)
    keep = len(full) - 500
    buf[keep:] = 0                                  # rows with i >= j are padding (admp/pme.py:671): 500 pairs dropped
    E_cut = f.get_energy(pos, box, buf, par['Q_local'], par['mScales'])
    assert f.n_pairs == keep and abs(E_cut - E_full) > 1e-6 * abs(E_full)
    buf[5000 if len(full) > 6000 else 10] = buf[5000 if len(full) > 6000 else 10][::-1]   # a single interior row swapped (i > j: dropped)
    f.get_energy(pos, box, buf, par['Q_local'], par['mScales'])
    assert f.n_pairs == keep - 1


def test_literal_kpoint_order_divergence_is_the_references(env):
    """GPU twin of tests/test_oracle_physics.py::test_literal_kpoint_order_makes_the_scf_diverge_on_an_unequal_mesh: on the
    96 x 100 x 45 mesh the default (literal) k-point order makes the reference's own Jacobi SCF diverge; the drop-in must
    return what the reference returns -- all 30 cycles, lconverg False, the same (huge) dipoles -- and say so."""
    import warnings
    from admp.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    n_mol = 216
    pos, box = S.synthetic_water_box(n_mol, seed=5)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    pairs = S.build_pairs(pos, box, 4.0)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    for k, v in zip(('K1', 'K2', 'K3'), (96, 100, 45)):
        f.update_env(k, v)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        U, flag, i = f.optimize_Uind(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'],
                                     par['dScales'])
    assert any('did not converge' in str(x.message) for x in w)
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (96, 100, 45), 2, True)
    Ur, flag_r, i_r = O.optimize_Uind(sysm, pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'],
                                      par['pScales'])
    assert (flag, i) == (flag_r, i_r) == (False, 29)
    assert rel(U, Ur.numpy()) < 1e-8 and np.abs(U).max() > 1e20


@pytest.mark.parametrize('lpol', [False, True])
def test_group_closing_kernel_matches_oracle(env, lpol, monkeypatch):
    """The closing kernels of large molecular systems (used above 8192 atoms) forced at a size the oracle can do: k_finish_rows
    (round 4: one lane per atom, frames exchanged through LDS inside workgroups of whole frame groups) and k_finish_groups
    (round 2: one lane per molecule; ADMP_FINISH_ROWS=0): energies, gradient, dE/dQ_local; and the pull form on the same
    inputs gives the same numbers."""
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    n_mol = 125
    pos, box = S.synthetic_water_box(n_mol, seed=31)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, polarizable=lpol)
    pairs = S.build_pairs(pos, box, 4.0)
    out = {}
    # 'gather': the form small systems take by default since round 4 -- the closing work in the epilogue of the gather
    # (k_gather_staged<.., FIN>, one workgroup per run of whole frame groups); ADMP_FUSE_FIN_MAX=0 gives the separate kernels
    for mode, gmin, rows, fin in (('rows', '0', '1', '0'), ('groups', '0', '0', '0'), ('pull', '100000000', '1', '0'),
                                  ('gather', '100000000', '1', '8192')):
        monkeypatch.setenv('ADMP_FINISH_GROUPS_MIN', gmin)
        monkeypatch.setenv('ADMP_FINISH_ROWS', rows)
        monkeypatch.setenv('ADMP_FUSE_FIN_MAX', fin)
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
        if lpol:
            out[mode] = f.get_forces_and_dQ(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'],
                                            par['pScales'], par['dScales']) + (f.energy_parts,)
        else:
            out[mode] = f.get_forces_and_dQ(pos, box, pairs, par['Q_local'], par['mScales']) + (f.energy_parts,)
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, lpol)
    if lpol:
        ref = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                                    par['pScales'], want_dQ=True)
    else:
        ref = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], want_dQ=True)
    scale = max(abs(p) for p in ref['parts'])
    for mode in ('rows', 'groups', 'pull', 'gather'):
        E, G, dQ, parts = out[mode]
        for got, want in zip(parts, ref['parts']):
            assert abs(got - want) <= 1e-9 * scale, mode
        assert rel(G, ref['grad']) < 1e-8 and rel(dQ, ref['dQ_local']) < 1e-8, mode
    assert rel(out['groups'][1], out['pull'][1]) < 1e-12 and rel(out['rows'][1], out['pull'][1]) < 1e-12
    assert rel(out['gather'][1], out['pull'][1]) < 1e-12 and rel(out['gather'][2], out['pull'][2]) < 1e-12


def test_pscale_gradient_vs_oracle(env):
    """dE/dpScales: entries with pscale = 0 against torch autograd through the oracle (finite there); entries with
    pscale = 1, where autodiff through the Fermi switch gives NaN (in the reference as well: exp overflow), against
    central differences of the oracle energy -- the switch is exactly flat on both sides.  dE/ddScales = 0."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.api import pme_potential, param_gradient
    from oracle import admp_oracle as O
    n_mol = 64
    pos, box = S.synthetic_water_box(n_mol, seed=41)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, True)
    pairs = S.build_pairs(pos, box, 4.0)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
    args = (pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    g = f.get_pscale_gradient(*args)
    U = np.asarray(f.U_ind)
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, True)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731

    def energy(pS):
        return O.energy_pme(sysm, T(pos), T(box), pairs, T(par['Q_local']), T(U), T(par['pol']), T(par['tholes']),
                            T(par['mScales']), pS)
    pS = T(par['pScales']).clone().requires_grad_(True)
    auto, = torch.autograd.grad(energy(pS), pS)
    auto = auto.numpy()
    assert np.isnan(auto[4]) and np.all(np.isfinite(auto[:2]))            # the reference's own behaviour
    scale = np.abs(g).max()
    assert scale > 1.0
    for k in (0, 1):                                                       # 1-2 and 1-3 pairs: pscale 0
        assert abs(g[k] - auto[k]) < 1e-8 * scale, (k, g[k], auto[k])
    h = 1e-4
    for k in (3, 4):                                                       # pscale 1 (k = 4: every non-bonded pair)
        pp, pm = par['pScales'].copy(), par['pScales'].copy()
        pp[k] += h
        pm[k] -= h
        fd = float(energy(T(pp)) - energy(T(pm))) / (2 * h)
        assert abs(g[k] - fd) < 1e-6 * scale, (k, g[k], fd)
    assert g[2] == 0.0                                                     # no 1-4 pairs in water
    out = param_gradient(pme_potential(f, par['pol'], par['tholes']), pos, box, pairs,
                         dict(Q_local=par['Q_local'], mScales=par['mScales'], pScales=par['pScales'], dScales=par['dScales']))
    assert np.allclose(out['pScales'], g, rtol=1e-9, atol=1e-9 * scale) and not out['dScales'].any()


@pytest.mark.parametrize('prec,tol', [('double', 1e-11), ('single', 2e-5)])
def test_traced_pair_kernels_on_gpu(env, prec, tol):
    """a22: generate_pairwise_interaction with ARBITRARY Python kernels (traced by admp_amd/xp.py, compiled by hiprtc):
    (1) the reference's TT kernel restated as a Python function gives the numbers of the hand-written HIP kernel;
    (2) a kernel with branches (switched, screened Lennard-Jones) against a torch restatement of the reference's driver
        (pairs i<j, mScales[nbonds-1] with wrap, minimum image, sum of kernel values) and its autograd gradient."""
    import torch
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    from tests.test_host_logic import _tt_kernel, _switched_lj
    from oracle import admp_oracle as O
    settings.PRECISION = prec
    n_mol = 216
    pos, box = S.synthetic_water_box(n_mol, seed=23)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol)
    pairs = S.build_pairs(pos, box, 4.5)
    mS = np.array([0.0, 0.3, 0.7, 1.0, 1.0])
    abqc = (par['a_list'], par['b_list'], par['q_list'], par['c_list'][:, 0])
    named = value_and_grad(generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={}))
    traced = generate_pairwise_interaction(_tt_kernel, cov, static_args={})
    E0, G0 = named(pos, box, pairs, mS, *abqc)
    E1, G1 = value_and_grad(traced)(pos, box, pairs, mS, *abqc)
    assert abs(E1 - E0) < tol * abs(E0) and rel(G1, G0) < 10 * tol
    assert abs(traced(pos, box, pairs, mS, *abqc) - E1) < 1e-12 * abs(E1)           # energy-only call
    # (2) a kernel the library has never seen, two parameter lists
    rng = np.random.default_rng(2)
    sig = np.tile(np.array([3.1, 1.8, 1.8]), n_mol) * rng.uniform(0.95, 1.05, 3 * n_mol)
    eps = np.tile(np.array([0.65, 0.05, 0.05]), n_mol)
    lj = generate_pairwise_interaction(_switched_lj, cov, static_args={})
    E2, G2 = value_and_grad(lj)(pos, box, pairs, mS, sig, eps)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    p = T(pos).requires_grad_(True)
    pi, pj, drv, m = O._pair_distances(p, T(box), pairs, T(mS), cov)
    dr = torch.linalg.norm(drv, dim=1)
    s, e = 0.5 * (T(sig)[pi] + T(sig)[pj]), torch.sqrt(T(eps)[pi] * T(eps)[pj])
    x6 = (s / dr) ** 6
    t = dr - 3.0
    sw = torch.where(dr < 3.0, torch.ones_like(dr), torch.where(dr > 4.0, torch.zeros_like(dr), 1.0 - t * t * (3.0 - 2.0 * t)))
    ref = torch.sum(m * 4.0 * e * (x6 * x6 - x6) * sw * torch.erfc(0.3 * dr) * dr ** -0.25)
    g, = torch.autograd.grad(ref, p)
    assert abs(E2 - float(ref)) < max(tol, 1e-12) * abs(float(ref)) * 10 and rel(G2, g.numpy()) < 10 * tol
    assert 'admp_pair_custom' in lj.source


@pytest.mark.gpu
def test_calculators_share_one_neighbour_table():
    """admp_share_neighbors: dispersion PME and a pair potential walk the PME calculator's compiled table (the reference's
    drivers hand ONE `pairs` array to every force object, examples/water_pol_1024/run_admp.py:117-136).  Results equal the
    ones with a table of their own; the loan follows the lender's list updates, ends with a list of the borrower's own, and
    a destroyed lender is an error, not a dangling pointer."""
    import gc
    from admp_amd import settings
    from admp_amd import systems as S
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    old = settings.PRECISION
    settings.PRECISION = 'double'
    try:
        n_mol = 216
        pos, box = S.synthetic_water_box(n_mol, seed=3)
        at, ai, cov = S.water_topology(n_mol)
        par = S.water_parameters(n_mol, polarizable=True)
        pairs = S.build_pairs(pos, box, 4.0)
        pairs5 = S.build_pairs(pos, box, 5.0)
        f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True)
        pme_args = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
        mS = par['mScales']
        c6 = np.ascontiguousarray(par['c_list'][:, 0])

        def fresh():
            d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
            t = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
            return d, t

        def both(d, t, p):
            Ed, Gd = d.get_forces(pos, box, p, par['c_list'], mS)
            Et, Gt = value_and_grad(t)(pos, box, p, mS, par['a_list'], par['b_list'], par['q_list'], c6)
            return float(Ed), np.asarray(Gd), float(Et), np.asarray(Gt)

        d0, t0 = fresh()
        ref4 = both(d0, t0, pairs)
        ref5 = both(d0, t0, pairs5)
        d, t = fresh()
        f.get_forces(pos, box, pairs, *pme_args)
        f.get_forces(pos, box, pairs, *pme_args)            # second call: the lender's table now carries site classes
        d.share_neighbors(f)
        t.share_neighbors(f)
        assert d.n_pairs == f.n_pairs == len(pairs)
        for got, want in zip(both(d, t, None), ref4):
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
        Ef, Gf = f.get_forces(pos, box, pairs5, *pme_args)  # the lender takes another list: the borrowers follow
        for got, want in zip(both(d, t, None), ref5):
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
        f.update_neighbors(pos, box, rc=4.0)                # ... also one built on the GPU
        for got, want in zip(both(d, t, None), ref4):
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
        for got, want in zip(both(d, t, pairs5), ref5):     # a list of their own ends the loan
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
        E2, G2 = f.get_forces(pos, box, None, *pme_args)    # the lender is unaffected
        d.share_neighbors(f)
        d._lender = None                                    # (the wrapper keeps the lender alive; not so a C caller)
        del f
        gc.collect()
        with pytest.raises(Exception):
            d.get_forces(pos, box, None, par['c_list'], mS)
    finally:
        settings.PRECISION = old


@pytest.mark.gpu
def test_pair_parameter_cache_follows_in_place_writes():
    """The packed per-atom parameter rows of a pair interaction are cached while the lists are the same, unwritten torch
    tensors: an in-place write (version counter) or a new tensor must be seen."""
    import torch
    from admp_amd import settings
    from admp_amd import systems as S
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
    old = settings.PRECISION
    settings.PRECISION = 'double'
    try:
        n_mol = 64
        pos, box = S.synthetic_water_box(n_mol, seed=9)
        at, ai, cov = S.water_topology(n_mol)
        par = S.water_parameters(n_mol, polarizable=False)
        pairs = S.build_pairs(pos, box, 4.0)
        t = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
        dev = 'cuda'
        a_, b_, q_, c_ = (torch.as_tensor(par[k] if k != 'c' else par['c_list'][:, 0].copy(), dtype=torch.float64, device=dev)
                          for k in ('a_list', 'b_list', 'q_list', 'c'))
        E0 = float(t(pos, box, pairs, par['mScales'], a_, b_, q_, c_))
        assert abs(float(t(pos, box, pairs, par['mScales'], a_, b_, q_, c_)) - E0) <= 1e-12 * abs(E0)    # cached rows (atomic sums: order-dependent last bits)
        a_.mul_(1.1)                                                                     # in-place write
        E1 = float(t(pos, box, pairs, par['mScales'], a_, b_, q_, c_))
        E1_fresh = float(t(pos, box, pairs, par['mScales'], a_.clone(), b_, q_, c_))     # a new tensor
        assert abs(E1 - E0) > 1e-6 * abs(E0) and abs(E1 - E1_fresh) <= 1e-12 * abs(E1)
    finally:
        settings.PRECISION = old


def test_force_field_front_end_and_dispersion_parameter_gradients(env, tmp_path):
    """f3: Hamiltonian(xml).createPotential(topology) -> [pot_disp, pot_pme] (admp/api.py:469-488) on a water box written to
    files, and `param_gradient(pot_disp, ...)` -- the counterpart of jax.grad(pot_disp, argnums=3) of the reference's
    examples/openmm_api/run.py:41-43 -- for mScales AND the per-type tables A, B, Q, C6, C8, C10, against torch autograd
    through the oracle with the same unit conversions (admp/api.py:185-193)."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
    import make_inputs
    from admp.api import Hamiltonian, Topology, param_gradient
    from oracle import admp_oracle as O
    n_mol = 64
    pos0, box = S.synthetic_water_box(n_mol, seed=9)
    make_inputs.write_pdb(str(tmp_path / 'w.pdb'), pos0, box)
    make_inputs.write_forcefield_xml(str(tmp_path / 'ff.xml'))
    H = Hamiltonian(str(tmp_path / 'ff.xml'))
    top = Topology.from_pdb(str(tmp_path / 'w.pdb'))
    disp_g, pme_g = H.getGenerators()
    pot_disp, pot_pme = H.createPotential(top, nonbondedCutoff=4.0)
    pos, box = top.positions, top.box                       # (the PDB keeps three decimals)
    pairs = S.build_pairs(pos, box, 4.0)
    E = pot_disp(pos, box, pairs, disp_g.params)
    g = param_gradient(pot_disp, pos, box, pairs, disp_g.params)
    # oracle: the same potential as a torch function of the per-type tables
    idx = torch.as_tensor(disp_g.map_atomtype)
    tabs = {k: torch.tensor(np.asarray(disp_g.params[k], dtype=np.float64), requires_grad=True) for k in
            ('A', 'B', 'Q', 'C6', 'C8', 'C10', 'mScales')}
    a = tabs['A'][idx] / 2625.5
    b = tabs['B'][idx] * 0.0529177249
    q = tabs['Q'][idx]
    c = torch.stack([torch.sqrt(tabs['C6'][idx] * 1e6), torch.sqrt(tabs['C8'][idx] * 1e8), torch.sqrt(tabs['C10'][idx] * 1e10)], dim=1)
    d = disp_g.disp_force
    cov = disp_g.covalent_map
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))          # noqa: E731
    e_sr = O.tt_damping_energy(T(pos), T(box), pairs, tabs['mScales'], cov, a, b, q, c[:, 0])
    e_lr = sum(O.disp_pme_parts(T(pos), T(box), pairs, c, tabs['mScales'], cov, d.kappa, (d.K1, d.K2, d.K3), 10))
    e_ref = e_sr - e_lr
    keys = ('A', 'B', 'Q', 'C6', 'C8', 'C10', 'mScales')
    grads = torch.autograd.grad(e_ref, [tabs[k] for k in keys])
    assert abs(E - float(e_ref)) < 1e-9 * max(abs(float(e_sr)), abs(float(e_lr)))
    for k, want in zip(keys, grads):
        want = want.numpy()
        got = np.asarray(g[k], dtype=np.float64)
        assert got.shape == want.shape, k
        assert np.abs(got - want).max() <= 1e-8 * max(np.abs(want).max(), 1e-30), (k, got, want)
    # the PME potential of the front-end is the calculator's energy with the generator's per-atom parameters
    par = S.water_parameters(n_mol, True)
    Ep = pot_pme(pos, box, pairs, pme_g.params)
    at, ai, cov2 = S.water_topology(n_mol)
    assert (pme_g.axis_types == at).all() and (pme_g.axis_indices[:, :2] == ai[:, :2]).all()
    np.testing.assert_allclose(pme_g.params['Q_local'], par['Q_local'], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(pme_g.params['pol'], par['pol'], rtol=1e-12)
    from admp_amd.pme import ADMPPmeForce
    f = ADMPPmeForce(box, at, ai, cov2, 4.0, 1e-5, 2, lpol=True)
    Ed = f.get_energy(pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
    assert abs(Ep - Ed) < 1e-9 * max(abs(p) for p in f.energy_parts)
    gp = param_gradient(pot_pme, pos, box, pairs, pme_g.params)
    assert set(gp) >= {'mScales', 'pScales', 'dScales', 'Q_local', 'pol', 'tholes'}


def test_front_end_mscale_gradient_vs_reference_held_value(env, tmp_path):
    """GPU twin of tests/test_oracle_physics.py::test_oracle_vs_reference_held_mscale_gradient: the reference's example
    examples/openmm_api/run.py:40-43 (Hamiltonian -> createPotential -> pot_disp -> grad(..., argnums=3)['mScales']) run
    through this package's front-end on the shipped water1024 geometry; the 1-2 component against the value the reference
    holds in examples/openmm_api/ref_out (committed as data, tests/golden/ref_openmm_api_mscale_grad.json) at 2 %, and
    against the oracle's autograd at 1e-8.  The other components of ref_out belong to a different geometry."""
    import json
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
    import make_inputs
    from admp.api import Hamiltonian, Topology, param_gradient
    from oracle import admp_oracle as O
    here = os.path.dirname(os.path.abspath(__file__))
    ref = json.load(open(os.path.join(here, 'golden', 'ref_openmm_api_mscale_grad.json')))
    d = np.load(os.path.join(here, 'golden', 'p1_water1024.npz'))
    make_inputs.write_pdb(str(tmp_path / 'water1024.pdb'), d['positions'], d['box'])
    make_inputs.write_forcefield_xml(str(tmp_path / 'forcefield.xml'))
    H = Hamiltonian(str(tmp_path / 'forcefield.xml'))
    top = Topology.from_pdb(str(tmp_path / 'water1024.pdb'))
    disp_g = H.getGenerators()[0]
    pot_disp = H.createPotential(top, nonbondedCutoff=4.0)[0]
    pos, box = top.positions, top.box
    pairs = S.build_pairs(pos, box, 4.0)
    g = np.asarray(param_gradient(pot_disp, pos, box, pairs, disp_g.params)['mScales'], dtype=np.float64)
    want = ref['dE_dmScales'][0]
    assert abs(g[0] - want) <= 0.02 * abs(want), (g, want)
    assert g[2] == 0.0 and g[3] == 0.0
    par = S.water_parameters(len(pos) // 3, True)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))          # noqa: E731
    mS = torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0], dtype=torch.float64, requires_grad=True)
    c = T(par['c_list'])
    e = O.tt_damping_energy(T(pos), T(box), pairs, mS, disp_g.covalent_map, T(par['a_list']), T(par['b_list']), T(par['q_list']),
                            c[:, 0]) - sum(O.disp_pme_parts(T(pos), T(box), pairs, c, mS, disp_g.covalent_map,
                                                            disp_g.disp_force.kappa, (24, 24, 24), 10))
    go, = torch.autograd.grad(e, [mS])
    assert np.abs(g - go.numpy()).max() <= 1e-8 * np.abs(go.numpy()).max(), (g, go)


def test_per_atom_parameter_gradients_of_dispersion_and_tang_toennies(env):
    """admp_disp_param_grad / admp_tt_param_grad (per-atom lists) against oracle autograd, both precisions, pmax 6 / 10."""
    import torch
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
    from oracle import admp_oracle as O
    n_mol = 64
    pos, box = S.synthetic_water_box(n_mol, seed=4)
    _, _, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol)
    pairs = S.build_pairs(pos, box, 4.0)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))          # noqa: E731
    mS = np.array([0.3, 0.0, 0.0, 1.0, 1.0])                                # a bonded class with a non-zero scale
    lists = [torch.tensor(par[k], requires_grad=True) for k in ('a_list', 'b_list', 'q_list')]
    c6 = torch.tensor(par['c_list'][:, 0].copy(), requires_grad=True)
    e = O.tt_damping_energy(T(pos), T(box), pairs, T(mS), cov, lists[0], lists[1], lists[2], c6)
    want_tt = [x.numpy() for x in torch.autograd.grad(e, lists + [c6])]
    for prec, tol in (('double', 1e-9), ('single', 2e-4)):
        settings.PRECISION = prec
        tt = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
        got = tt.get_param_gradient(pos, box, pairs, mS, par['a_list'], par['b_list'], par['q_list'], par['c_list'][:, 0].copy())
        for gk, wk in zip(got, want_tt):
            assert rel(gk, wk) < tol
        for pmax in (6, 10):
            nc = (pmax - 4) // 2
            d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, pmax)
            cl = torch.tensor(par['c_list'][:, :nc].copy(), requires_grad=True)
            c3 = torch.cat([cl, torch.zeros((3 * n_mol, 3 - nc), dtype=torch.float64)], dim=1)
            ed = sum(O.disp_pme_parts(T(pos), T(box), pairs, c3, T(mS), cov, d.kappa, (d.K1, d.K2, d.K3), pmax))
            want = torch.autograd.grad(ed, cl)[0].numpy()
            got = d.get_param_gradient(pos, box, pairs, par['c_list'][:, :nc].copy(), mS)
            assert np.asarray(got).shape == want.shape and rel(got, want) < tol, (prec, pmax)
