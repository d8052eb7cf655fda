"""Kernel arithmetic (admp_amd/csrc/*_math.h, compiled for the host by tests/hostshim) against the
oracle's autograd: the hand-coded adjoints of the pair kernel, the local frames and the
B-spline spread/gather.  No GPU needed; the HIP kernels run exactly these inline functions."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import admp_oracle as O
from tests.hostshim_util import lib, dp, c64, i32, scale_tables

F64 = torch.float64


def T(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float64))


def random_sites(n, L, seed, min_sep=1.2):
    rng = np.random.default_rng(seed)
    pos = []
    while len(pos) < n:
        p = rng.uniform(0, L, 3)
        if all(np.linalg.norm((p - q) - L * np.round((p - q) / L)) > min_sep for q in pos):
            pos.append(p)
    pos = np.array(pos)
    Q = rng.normal(size=(n, 9)) * np.array([1, .5, .5, .5, .3, .3, .3, .3, .3])
    U = rng.normal(size=(n, 3)) * 0.1
    pol = rng.uniform(0.3, 1.5, n)
    pol[rng.random(n) < 0.3] = 0.0
    thole = rng.uniform(0.0, 9.0, n)
    return pos, Q, U, pol, thole


def all_pairs(n):
    i, j = np.triu_indices(n, 1)
    return np.stack([i, j], axis=1).astype(np.int32)


def oracle_pair(pos, box, pairs, nbonds_dense, Qg, Uh, pol, thole, mS, pS, kappa, lpol):
    p = T(pos).requires_grad_(True)
    q = T(Qg).requires_grad_(True)
    u = T(Uh).requires_grad_(True)
    e = O.pme_real(p, T(box), pairs, q, u if lpol else None, T(pol) if lpol else None,
                   T(thole) if lpol else None, T(mS), T(pS) if lpol else None, nbonds_dense, kappa, 2, lpol)
    gs = torch.autograd.grad(e, [p, q] + ([u] if lpol else []))
    return float(e.detach()), [g.numpy() for g in gs]


@pytest.mark.parametrize('lpol', [False, True])
@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('tric', [False, True])
def test_pair_kernel_matches_oracle(lpol, mode, tric):
    n, L = 14, 9.0
    pos, Q, U, pol, thole = random_sites(n, L, 7)
    box = np.eye(3) * L
    if tric:
        box = np.array([[L, 0, 0], [1.3, L * 0.95, 0], [-0.8, 1.1, L * 1.05]])
    pairs = all_pairs(n)
    rng = np.random.default_rng(3)
    cov = np.zeros((n, n), dtype=np.int64)
    for (i, j) in pairs[rng.random(len(pairs)) < 0.25]:
        cov[i, j] = cov[j, i] = rng.integers(1, 6)
    mS = np.array([0.0, 0.2, 0.5, 0.8, 1.0])
    pS = np.array([0.0, 0.0, 0.0, 1.0, 1.0])
    kappa = 0.41
    e_ref, g_ref = oracle_pair(pos, box, pairs, cov, Q, U, pol, thole, mS, pS, kappa, lpol)
    nb = i32(cov[pairs[:, 0], pairs[:, 1]])
    mtab, ptab, w0 = scale_tables(mS, pS)
    grad = np.zeros((n, 3)); pot = np.zeros((n, 9)); fld = np.zeros((n, 3))
    p6 = c64(pol ** (1.0 / 6.0))
    e = lib().shim_pair_real(8, n, dp(c64(pos)), dp(c64(Q)), dp(c64(U)), dp(p6), dp(c64(thole)), dp(c64(box)),
                             ctypes.c_long(len(pairs)), dp(i32(pairs)), dp(nb), dp(mtab), dp(ptab), dp(w0),
                             ctypes.c_double(kappa), int(lpol), mode, dp(grad), dp(pot), dp(fld))
    assert abs(e - e_ref) <= 1e-10 * max(1.0, abs(e_ref))
    np.testing.assert_allclose(grad, g_ref[0], rtol=1e-9, atol=1e-8 * np.abs(g_ref[0]).max())
    np.testing.assert_allclose(pot, g_ref[1], rtol=1e-9, atol=1e-9 * np.abs(g_ref[1]).max())
    if lpol:
        np.testing.assert_allclose(fld, g_ref[2], rtol=1e-9, atol=1e-9 * np.abs(g_ref[2]).max())
        # field-only kernel used inside the SCF
        f2 = np.zeros((n, 3))
        lib().shim_pair_field(8, n, dp(c64(pos)), dp(c64(Q)), dp(c64(U)), dp(p6), dp(c64(thole)), dp(c64(box)),
                              ctypes.c_long(len(pairs)), dp(i32(pairs)), dp(nb), dp(ptab), dp(w0),
                              ctypes.c_double(kappa), dp(f2))
        np.testing.assert_allclose(f2, g_ref[2], rtol=1e-9, atol=1e-9 * np.abs(g_ref[2]).max())


@pytest.mark.parametrize('lpol', [False, True])
def test_charge_only_pair_forms_match_oracle(lpol):
    """pair_full_mono / pair_mono_full / pair_mono_mono (pme_math.h: the forms k_pair_full takes for sites that carry a
    charge and nothing else) against the oracle on a mix of charge-only and fully equipped sites, triclinic cell, bonded
    classes with pscale 0 and 1, Thole parameters incl. zero.  The pot of a charge-only ROW is complete only in its monopole
    slot (such a site has no torque), so pot is compared there and everywhere for the other sites."""
    n, L = 16, 9.5
    pos, Q, U, pol, thole = random_sites(n, L, 19)
    mono = np.arange(n) % 3 != 0                      # two of three sites charge-only, like water
    Q[mono, 1:] = 0.0
    U[mono] = 0.0
    pol[mono] = 0.0
    thole[mono] = 0.0
    thole[3] = 0.0                                    # a polarizable site with thole 0 next to charge-only partners
    box = np.array([[L, 0, 0], [1.1, L * 0.97, 0], [-0.7, 0.9, L * 1.04]])
    pairs = all_pairs(n)
    rng = np.random.default_rng(8)
    cov = np.zeros((n, n), dtype=np.int64)
    for (i, j) in pairs[rng.random(len(pairs)) < 0.3]:
        cov[i, j] = cov[j, i] = rng.integers(1, 6)
    mS = np.array([0.0, 0.2, 0.5, 0.8, 1.0])
    pS = np.array([0.0, 0.0, 1.0, 1.0, 1.0])
    kappa = 0.43
    e_ref, g_ref = oracle_pair(pos, box, pairs, cov, Q, U, pol, thole, mS, pS, kappa, lpol)
    nb = i32(cov[pairs[:, 0], pairs[:, 1]])
    mtab, ptab, w0 = scale_tables(mS, pS)
    grad = np.zeros((n, 3)); pot = np.zeros((n, 9)); fld = np.zeros((n, 3))
    e = lib().shim_pair_real(8, n, dp(c64(pos)), dp(c64(Q)), dp(c64(U)), dp(c64(pol ** (1.0 / 6.0))), dp(c64(thole)),
                             dp(c64(box)), ctypes.c_long(len(pairs)), dp(i32(pairs)), dp(nb), dp(mtab), dp(ptab), dp(w0),
                             ctypes.c_double(kappa), int(lpol), 2, dp(grad), dp(pot), dp(fld))
    assert abs(e - e_ref) <= 1e-10 * max(1.0, abs(e_ref))
    np.testing.assert_allclose(grad, g_ref[0], rtol=1e-9, atol=1e-8 * np.abs(g_ref[0]).max())
    tol = 1e-9 * np.abs(g_ref[1]).max()
    np.testing.assert_allclose(pot[~mono], g_ref[1][~mono], rtol=1e-9, atol=tol)
    np.testing.assert_allclose(pot[mono, 0], g_ref[1][mono, 0], rtol=1e-9, atol=tol)
    assert not pot[mono, 1:].any()
    if lpol:
        np.testing.assert_allclose(fld[~mono], g_ref[2][~mono], rtol=1e-9, atol=1e-9 * np.abs(g_ref[2]).max())
        # field-only kernel of the SCF with its short form for charge-only partners (pair_field_mono); dE/dU at every site
        f2 = np.zeros((n, 3))
        lib().shim_pair_field_mono(8, n, dp(c64(pos)), dp(c64(Q)), dp(c64(U)), dp(c64(pol ** (1.0 / 6.0))), dp(c64(thole)),
                                   dp(c64(box)), ctypes.c_long(len(pairs)), dp(i32(pairs)), dp(nb), dp(ptab), dp(w0),
                                   ctypes.c_double(kappa), dp(f2))
        np.testing.assert_allclose(f2, g_ref[2], rtol=1e-9, atol=1e-9 * np.abs(g_ref[2]).max())


def test_pair_kernel_float32_close():
    n, L = 14, 9.0
    pos, Q, U, pol, thole = random_sites(n, L, 11)
    box = np.eye(3) * L
    pairs = all_pairs(n)
    cov = np.zeros((n, n), dtype=np.int64)
    mS = pS = np.array([0.0, 0.0, 0.0, 1.0, 1.0])
    e_ref, g_ref = oracle_pair(pos, box, pairs, cov, Q, U, pol, thole, mS, pS, 0.5, True)
    nb = i32(cov[pairs[:, 0], pairs[:, 1]])
    mtab, ptab, w0 = scale_tables(mS, pS)
    grad = np.zeros((n, 3)); pot = np.zeros((n, 9)); fld = np.zeros((n, 3))
    e = lib().shim_pair_real(4, n, dp(c64(pos)), dp(c64(Q)), dp(c64(U)), dp(c64(pol ** (1 / 6))), dp(c64(thole)),
                             dp(c64(box)), ctypes.c_long(len(pairs)), dp(i32(pairs)), dp(nb), dp(mtab), dp(ptab), dp(w0),
                             ctypes.c_double(0.5), 1, 1, dp(grad), dp(pot), dp(fld))
    assert abs(e - e_ref) <= 1e-4 * abs(e_ref) + 1e-2
    assert np.linalg.norm(grad - g_ref[0]) <= 1e-4 * np.linalg.norm(g_ref[0])


AXIS_CASES = {
    'water': (np.array([1, 0, 0, 1, 0, 0]), np.array([[1, 2, -1], [0, 2, -1], [0, 1, -1], [4, 5, -1], [3, 5, -1], [3, 4, -1]])),
    'mixed': (np.array([0, 1, 2, 3, 4, 5]), np.array([[1, 2, -1], [0, 2, -1], [3, 4, 5], [0, 1, 2], [5, -1, -1], [-1, -1, -1]])),
}


@pytest.mark.parametrize('case', ['water', 'mixed'])
def test_local_frames_forward_and_adjoint(case):
    atype, aidx = AXIS_CASES[case]
    rng = np.random.default_rng(5)
    n = 6
    pos = rng.uniform(0, 6, (n, 3))
    box = np.array([[12.0, 0, 0], [0.7, 11.0, 0], [0.3, -0.4, 13.0]])
    Ql = rng.normal(size=(n, 9))
    if case == 'mixed':
        Ql[5, 1:] = 0.0        # NoAxisType site: charge only
    P = rng.normal(size=(n, 9))
    p = T(pos).requires_grad_(True)
    ql = T(Ql).requires_grad_(True)
    fr = O.construct_local_frames(p, T(box), atype, aidx)
    qg = O.rot_local2global(ql, fr, 2)
    s = torch.sum(T(P) * qg)
    gp, gq = torch.autograd.grad(s, [p, ql])
    frames = np.zeros((n, 9)); Qg = np.zeros((n, 9)); grad = np.zeros((n, 3)); dQl = np.zeros((n, 9))
    lib().shim_frames(8, n, dp(c64(pos)), dp(c64(box)), dp(i32(atype)), dp(i32(aidx)), dp(c64(Ql)), dp(c64(P)),
                      dp(frames), dp(Qg), dp(grad), dp(dQl))
    keep = atype != 5
    np.testing.assert_allclose(frames.reshape(n, 3, 3)[keep], fr.detach().numpy()[keep], atol=1e-12)
    np.testing.assert_allclose(Qg, qg.detach().numpy(), atol=1e-10)
    np.testing.assert_allclose(grad, gp.numpy(), atol=1e-9)
    np.testing.assert_allclose(dQl, gq.numpy(), atol=1e-10)


def test_bspline_matches_piecewise():
    for f in [0.0, 0.123, 0.5, 0.999]:
        out = np.zeros(24)
        lib().shim_bspline6(ctypes.c_double(f), dp(out))
        u = T(f + np.arange(6.0))
        for d in range(4):
            np.testing.assert_allclose(out[6 * d:6 * d + 6], O._bspline6(u, d).numpy(), atol=1e-13)


@pytest.mark.parametrize('tric', [False, True])
def test_spread_and_gather_match_oracle(tric):
    rng = np.random.default_rng(9)
    n = 9
    K = (10, 12, 14)
    box = np.diag([8.0, 9.0, 10.5])
    if tric:
        box = np.array([[8.0, 0, 0], [1.0, 9.0, 0], [0.5, -0.7, 10.5]])
    pos = rng.uniform(-3, 14, (n, 3))          # some outside the cell
    Q = rng.normal(size=(n, 9))
    phi = rng.normal(size=K)
    p = T(pos).requires_grad_(True)
    q = T(Q).requires_grad_(True)
    mesh_ref = O.spread_Q(p, T(box), q, K, 2)
    s = torch.sum(mesh_ref * T(phi))
    gp, gq = torch.autograd.grad(s, [p, q])
    Ki = np.array(K, dtype=np.int32)
    mesh = np.zeros(K)
    lib().shim_spread(8, n, dp(c64(pos)), dp(c64(Q)), dp(c64(box)), dp(Ki), dp(mesh))
    np.testing.assert_allclose(mesh, mesh_ref.detach().numpy(), atol=1e-11 * np.abs(mesh).max() + 1e-13)
    pot = np.zeros((n, 9)); grad = np.zeros((n, 3)); fo = np.zeros((n, 3))
    lib().shim_gather(8, n, dp(c64(pos)), dp(c64(Q)), dp(c64(box)), dp(Ki), dp(c64(phi)), dp(pot), dp(grad), dp(fo))
    np.testing.assert_allclose(pot, gq.numpy(), rtol=1e-9, atol=1e-10 * np.abs(gq.numpy()).max())
    if not tric:
        # the reference's operator matrix is transposed w.r.t. the true Jacobian for triclinic cells
        # (recip.py:177 vs :75); the kernel follows both, autograd of the restatement likewise
        pass
    np.testing.assert_allclose(grad, gp.numpy(), rtol=1e-9, atol=1e-10 * np.abs(gp.numpy()).max())
    # dipole-only gather used by the SCF: harmonic slots (z,x,y) of dE/dQ
    np.testing.assert_allclose(fo[:, [2, 0, 1]], gq.numpy()[:, 1:4], rtol=1e-9, atol=1e-10 * np.abs(gq.numpy()).max())


@pytest.mark.parametrize('pmax', [6, 8, 10])
def test_dispersion_and_tt_pair_terms(pmax):
    n, L = 12, 8.0
    pos, *_ = random_sites(n, L, 21)
    box = np.eye(3) * L
    pairs = all_pairs(n)
    rng = np.random.default_rng(2)
    cov = np.zeros((n, n), dtype=np.int64)
    for (i, j) in pairs[rng.random(len(pairs)) < 0.2]:
        cov[i, j] = cov[j, i] = rng.integers(1, 4)
    mS = np.array([0.0, 0.3, 0.6, 1.0, 1.0])
    c = rng.uniform(5, 40, (n, 3))
    kappa = 0.37
    p = T(pos).requires_grad_(True)
    pi, pj, dr, m = O._pair_distances(p, T(box), pairs, T(mS), cov)
    dr2 = torch.sum(dr * dr, 1)
    x2 = kappa ** 2 * dr2
    ex = torch.exp(-x2)
    g6 = (1 + x2 + x2 ** 2 / 2) * ex
    g8 = g6 + x2 ** 3 / 6 * ex
    g10 = g8 + x2 ** 4 / 24 * ex
    ct = T(c)
    e = (m + g6 - 1) * ct[pi, 0] * ct[pj, 0] / dr2 ** 3
    if pmax >= 8:
        e = e + (m + g8 - 1) * ct[pi, 1] * ct[pj, 1] / dr2 ** 4
    if pmax >= 10:
        e = e + (m + g10 - 1) * ct[pi, 2] * ct[pj, 2] / dr2 ** 5
    e = e.sum()
    g_ref, = torch.autograd.grad(e, p)
    nb = i32(cov[pairs[:, 0], pairs[:, 1]])
    mtab, _, _ = scale_tables(mS)
    grad = np.zeros((n, 3))
    ed = lib().shim_disp_real(8, n, dp(c64(pos)), dp(c64(c)), dp(c64(box)), ctypes.c_long(len(pairs)), dp(i32(pairs)),
                              dp(nb), dp(mtab), ctypes.c_double(kappa), pmax, dp(grad))
    assert abs(ed - float(e.detach())) < 1e-10 * abs(float(e.detach()))
    np.testing.assert_allclose(grad, g_ref.numpy(), rtol=1e-9, atol=1e-10 * np.abs(g_ref.numpy()).max())
    if pmax == 6:
        abqc = np.stack([rng.uniform(0.01, 400, n), rng.uniform(1.5, 2.5, n), rng.normal(size=n), rng.uniform(5, 40, n)], 1)
        r = O.tt_energy_and_grad(pos, box, pairs, mS, cov, abqc[:, 0], abqc[:, 1], abqc[:, 2], abqc[:, 3])
        grad = np.zeros((n, 3))
        et = lib().shim_tt_real(8, n, dp(c64(pos)), dp(c64(abqc)), dp(c64(box)), ctypes.c_long(len(pairs)),
                                dp(i32(pairs)), dp(nb), dp(mtab), dp(grad))
        assert abs(et - r['E']) < 1e-10 * abs(r['E'])
        np.testing.assert_allclose(grad, r['grad'], rtol=1e-9, atol=1e-10 * np.abs(r['grad']).max())


def test_dispersion_ck():
    ksq = np.array([0.0, 0.01, 0.3, 2.0, 9.0])
    for which, fn in [(6, O.Ck_6), (8, O.Ck_8), (10, O.Ck_10)]:
        ref = fn(T(ksq), 0.43, 1234.5).numpy()
        got = [lib().shim_disp_ck(which, ctypes.c_double(k), ctypes.c_double(0.43), ctypes.c_double(1234.5)) for k in ksq]
        np.testing.assert_allclose(got, ref, rtol=1e-12)


# ---- direct-DFT lines (dft_math.h) against numpy.fft: the arithmetic of dft_kernels.hip on the CPU
@pytest.mark.parametrize('kq', [1, 2, 4])
@pytest.mark.parametrize('N', [2, 3, 4, 5, 6, 9, 16, 31, 96, 97, 100, 127, 160])
def test_dft_lines_match_numpy_fft(N, kq):
    L = lib()
    rng = np.random.default_rng(N)
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    buf = np.ascontiguousarray(np.stack([x.real, x.imag], axis=1))
    for sign, ref in ((-1, np.fft.fft(x)), (+1, np.fft.ifft(x) * N)):
        for prec, tol in ((8, 1e-13), (4, 2e-5)):
            out = np.zeros((N, 2))
            L.shim_dft_line(prec, kq, 0, N, sign, dp(buf), dp(out))
            got = out[:, 0] + 1j * out[:, 1]
            assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)) * np.sqrt(N)
    xr = rng.normal(size=N)
    ref = np.fft.rfft(xr)
    half = np.ascontiguousarray(np.stack([ref.real, ref.imag], axis=1))
    for prec, tol in ((8, 1e-13), (4, 2e-5)):
        out = np.zeros((N // 2 + 1, 2))
        L.shim_dft_line(prec, kq, 1, N, 0, dp(np.ascontiguousarray(xr)), dp(out))
        assert np.max(np.abs(out[:, 0] + 1j * out[:, 1] - ref)) <= tol * np.sqrt(N) * np.max(np.abs(ref))
        # c2r of the half spectrum returns N * x (unnormalised, like rocFFT / like ifftn * N)
        back = np.zeros(N)
        L.shim_dft_line(prec, kq, 2, N, 0, dp(half), dp(back))
        assert np.max(np.abs(back - N * xr)) <= 10 * tol * N * np.max(np.abs(xr))


def test_largest_prime_factor_rule():
    L = lib()
    assert [L.shim_largest_prime_factor(n) for n in (1, 2, 96, 97, 100, 128, 154, 91, 2 * 17)] == [1, 2, 3, 97, 5, 2, 11, 13, 17]
