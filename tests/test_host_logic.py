"""CPU-side checks: the C-ABI library loads and exports every symbol include/admp_hip.h declares
(no compute calls without a GPU), and the host logic around it (covalent map -> CSR, Ewald
parameters, pair-list builder, scale-table wrap, synthetic box generator)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    from admp_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, 'include', 'admp_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(admp_[a-z_0-9]+)\s*\(', header))
    assert len(declared) >= 15
    from admp_amd import _lib
    assert declared == set(_lib.PROTOTYPES), 'ctypes prototypes and header disagree'
    lib = ctypes.CDLL(built_lib)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    L = _lib.load()
    assert b'gfx950' in L.admp_version()


def test_no_gpu_fails_loudly(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from admp_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.admp_create(ctypes.byref(h), 0, 8) != 0          # ADMP_E_NOGPU, no fallback
    from admp_amd.pme import ADMPPmeForce
    from admp_amd import systems as S
    at, ai, cov = S.water_topology(2)
    with pytest.raises(RuntimeError):
        ADMPPmeForce(np.eye(3) * 20.0, at, ai, cov, 4.0, 1e-4, 2)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'admp_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f
                assert 'hostshim_util' not in text, f


def test_ewald_parameters_match_reference_rule():
    from admp_amd.pme import setup_ewald_parameters
    kappa, K1, K2, K3 = setup_ewald_parameters(4, 1e-4, np.eye(3) * 50.0)
    assert abs(kappa - 0.7296057664681077) < 1e-15 and (K1, K2, K3) == (154, 154, 154)
    kappa, K1, K2, K3 = setup_ewald_parameters(4.0, 1e-4, np.diag([31.289, 40.0, 25.0]))
    assert (K1, K2, K3) == (97, 123, 77)


def test_covalent_map_to_csr_dense_and_sparse():
    from admp_amd._device import covalent_to_csr
    from admp_amd import systems as S
    _, _, cov = S.water_topology(3)
    for form in (cov, cov.toarray()):
        ptr, col, val = covalent_to_csr(form, 9)
        assert ptr.tolist() == [0, 2, 4, 6, 8, 10, 12, 14, 16, 18]
        assert col[:6].tolist() == [1, 2, 0, 2, 0, 1] and val[:6].tolist() == [1, 1, 1, 2, 1, 2]
    ptr, col, val = covalent_to_csr(None, 4)
    assert ptr.tolist() == [0] * 5 and len(col) == 0
    with pytest.raises(ValueError):
        covalent_to_csr(np.full((2, 2), 99), 2)


def test_scale_table_wraps_like_python_indexing():
    from tests.hostshim_util import scale_tables
    mS = np.array([0.1, 0.2, 0.3, 0.4, 0.5])
    mtab, ptab, w0 = scale_tables(mS, np.array([0.0, 0.0, 0.0, 1.0, 1.0]))
    assert mtab[0] == mS[-1]            # nbonds = 0 -> mScales[-1] (admp/pme.py:682-683)
    assert mtab[1:6].tolist() == mS.tolist()
    assert w0[1] == pytest.approx(1.0) and w0[4] == 0.0 and w0[0] == 0.0


def test_pair_builder_matches_brute_force():
    from admp_amd import systems as S
    pos, box = S.synthetic_water_box(27, seed=1)
    pairs = S.build_pairs(pos + 100.0, box, 4.0)          # positions outside the cell are wrapped
    L = box[0, 0]
    i, j = np.triu_indices(len(pos), 1)
    d = pos[i] - pos[j]
    d -= L * np.round(d / L)
    r = np.linalg.norm(d, axis=1)
    want = set(zip(i[r < 4.0].tolist(), j[r < 4.0].tolist()))
    assert set(map(tuple, pairs.tolist())) == want and (pairs[:, 0] < pairs[:, 1]).all()


def test_synthetic_box_is_seeded_and_physical():
    from admp_amd import systems as S
    a, box = S.synthetic_water_box(216, seed=20240)
    b, _ = S.synthetic_water_box(216, seed=20240)
    assert np.array_equal(a, b)
    assert abs(box[0, 0] - (216 / S.WATER_DENSITY) ** (1 / 3)) < 1e-12
    mol = a.reshape(-1, 3, 3)
    np.testing.assert_allclose(np.linalg.norm(mol[:, 1] - mol[:, 0], axis=1), S.R_OH, atol=1e-12)
    pairs = S.build_pairs(a, box, 3.0)
    inter = pairs[:, 0] // 3 != pairs[:, 1] // 3
    d = a[pairs[inter, 0]] - a[pairs[inter, 1]]
    d -= box[0, 0] * np.round(d / box[0, 0])
    assert np.linalg.norm(d, axis=1).min() >= 1.5 - 1e-9


def test_bench_pair_kernel_bytes_formula():
    import bench
    total, per_pair = bench.pair_kernel_bytes(13_826_823, 1_048_575, 4, True)
    assert abs(per_pair - (8 + 68 + 60 + 128 / (13_826_823 / 1_048_575))) < 1e-9     # 145.6 B/pair
    assert abs(per_pair - 145.7) < 0.1


def test_fft_friendly_mesh_is_opt_in():
    """settings.FFT_FRIENDLY_MESH rounds the reference's K up to a 7-smooth size; off by default (reference mesh)."""
    from admp_amd import settings
    from admp_amd.pme import setup_ewald_parameters, next_smooth
    box = np.eye(3) * 31.289
    assert setup_ewald_parameters(4.0, 1e-4, box)[1:] == (97, 97, 97)
    assert [next_smooth(n) for n in (1, 97, 98, 154, 305, 128)] == [1, 98, 98, 160, 315, 128]
    settings.FFT_FRIENDLY_MESH = True
    try:
        assert setup_ewald_parameters(4.0, 1e-4, box)[1:] == (98, 98, 98)
    finally:
        settings.FFT_FRIENDLY_MESH = False


def test_admp_alias_package_resolves_to_admp_amd():
    """`from admp.pme import ADMPPmeForce` (the reference's import lines) works and is the same module object."""
    import admp.settings
    import admp_amd.settings
    from admp.pme import ADMPPmeForce, setup_ewald_parameters      # noqa: F401
    from admp.disp_pme import ADMPDispPmeForce                      # noqa: F401
    from admp.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel      # noqa: F401
    from admp.multipole import convert_cart2harm                    # noqa: F401
    import admp_amd.pme
    assert admp.settings is admp_amd.settings and admp.pme is admp_amd.pme
    assert hasattr(admp.settings, 'REFERENCE_KPOINT_ORDER') and admp.settings.REFERENCE_KPOINT_ORDER is True


def _tt_kernel(dr, m, ai, aj, bi, bj, qi, qj, ci, cj):
    """the reference's TT_damping_qq_c6_kernel (admp/pairwise.py:96-113), its arithmetic restated against admp_amd.xp"""
    from admp_amd import xp as jnp
    a = jnp.sqrt(ai * aj)
    b = jnp.sqrt(bi * bj)
    c = ci * cj
    q = qi * qj
    br = b * (dr * 1.889726878)
    ebr = jnp.exp(-br)
    poly = 1 + br + br ** 2 / 2 + br ** 3 / 6 + br ** 4 / 24 + br ** 5 / 120 + br ** 6 / 720
    return (2625.5 * a * ebr + (-2625.5) * ebr * (1 + br) * q / br + ebr * poly * c / dr ** 6) * m


def _switched_lj(dr, m, si, sj, ei, ej):
    """a kernel with a branch: Lennard-Jones, smoothly switched off between 3 and 4 A, erfc-screened"""
    from admp_amd import xp
    s = 0.5 * (si + sj)
    e = xp.sqrt(ei * ej)
    x6 = (s / dr) ** 6
    lj = 4.0 * e * (x6 * x6 - x6)
    t = (dr - 3.0) / 1.0
    sw = xp.where(dr < 3.0, 1.0, xp.where(dr > 4.0, 0.0, 1.0 - t * t * (3.0 - 2.0 * t)))
    return m * lj * sw * xp.erfc(0.3 * dr) * xp.power(dr, -0.25)


@pytest.mark.parametrize('kernel,npar', [(_tt_kernel, 4), (_switched_lj, 2)])
def test_traced_pair_kernel_value_and_derivative(tmp_path, kernel, npar):
    """admp_amd/xp.py: a Python pair kernel is traced once into C++/HIP source.  The traced expression and its forward-mode
    derivative (compiled here as HOST code with g++: no GPU needed) equal the Python function and its central difference."""
    import subprocess
    from admp_amd import xp
    src = xp.generate_host_source(kernel, npar)
    cpp, so = tmp_path / 'k.cpp', tmp_path / 'k.so'
    cpp.write_text(src)
    r = subprocess.run(['g++', '-O2', '-shared', '-fPIC', '-o', str(so), str(cpp)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    L = ctypes.CDLL(str(so))
    rng = np.random.default_rng(5)
    out = (ctypes.c_double * 2)()
    for dr in (0.9, 2.2, 2.99, 3.4, 3.999, 5.5):
        pi, pj = rng.uniform(0.5, 3.0, npar), rng.uniform(0.5, 3.0, npar)
        args = [v for pair in zip(pi, pj) for v in pair]
        L.eval(ctypes.c_double(dr), ctypes.c_double(0.7), (ctypes.c_double * npar)(*pi), (ctypes.c_double * npar)(*pj), out)
        want = kernel(dr, 0.7, *args)                      # the same function on plain floats
        h = 1e-6
        fd = (kernel(dr + h, 0.7, *args) - kernel(dr - h, 0.7, *args)) / (2 * h)
        assert abs(out[0] - want) <= 1e-13 * max(1.0, abs(want))
        assert abs(out[1] - fd) <= 1e-6 * max(1.0, abs(fd))
    dev = xp.generate_source(kernel, npar)                 # the device flavour: same body inside the pair kernel
    assert 'extern "C" __global__' in dev and 'admp_pair_custom' in dev
    with pytest.raises(TypeError):                         # Python control flow on traced values is refused, not mis-traced
        xp.trace_pair_kernel(lambda dr, m: dr if dr > 1 else m, 0)


def test_parser_reproduces_example_inputs(tmp_path):
    """admp.parser (this package's reader of the drivers' PDB / XML inputs) on the files examples/make_inputs.py writes:
    multipoles, axis types / anchors and covalent map equal the hard-wired water preparation of admp_amd/systems.py, i.e.
    what the reference's tests/test_sptial.py:74-84 lists for O, H1, H2 waters."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'make_inputs.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    from admp.parser import read_pdb, read_xml, init_residues, assemble_covalent
    from admp.multipole import convert_cart2harm
    from admp_amd import systems as S
    d = os.path.join(ROOT, 'examples', 'water_pol_1024')
    info = read_pdb(os.path.join(d, 'water1024.pdb'))
    at, rt = read_xml(os.path.join(d, 'mpidwater.xml'))
    atoms, residues = init_residues(info['serials'], info['names'], info['resNames'], info['resSeqs'], info['positions'],
                                    info['charges'], at, rt)
    n = len(info['serials'])
    assert n == 3072 and info['box'][:3] == [31.289, 31.289, 31.289]
    Q = np.vstack([(a.c0, a.dX * 10, a.dY * 10, a.dZ * 10, a.qXX * 300, a.qYY * 300, a.qZZ * 300, a.qXY * 300, a.qXZ * 300,
                    a.qYZ * 300) for a in atoms.values()])
    t, idx, cov = S.water_topology(n // 3)
    par = S.water_parameters(n // 3, True)
    assert np.abs(convert_cart2harm(Q, 2) - par['Q_local']).max() < 1e-15
    assert (np.array([a.axisType for a in atoms.values()]) == t).all()
    assert (np.vstack([a.axis_indices for a in atoms.values()]) == idx).all()
    assert (assemble_covalent(residues, n) == cov.toarray()).all()
    pol = np.vstack([(a.polarizabilityXX, a.polarizabilityYY, a.polarizabilityZZ) for a in atoms.values()]).astype(np.float32)
    assert np.abs(1000 * np.mean(pol, axis=1) - par['pol']).max() < 1e-12


def test_force_field_front_end_without_gpu(tmp_path):
    """admp.api (reference admp/api.py:120-488): the XML force field is read into generators holding the parameter tables, the
    PDB into a topology; atom types, bonds, covalent map and local-frame rules of water come out as the reference's
    front-end derives them (tests/test_sptial.py:74-84 axis atoms).  Host side only -- no calculator is built."""
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    import make_inputs
    from admp_amd import systems as S
    from admp_amd import api, parser
    import admp.api
    assert admp.api is api
    n_mol = 8
    pos, box = S.synthetic_water_box(n_mol, seed=3)
    make_inputs.write_pdb(str(tmp_path / 'w.pdb'), pos, box)
    make_inputs.write_forcefield_xml(str(tmp_path / 'ff.xml'))
    H = api.Hamiltonian(str(tmp_path / 'ff.xml'))
    disp_g, pme_g = H.getGenerators()
    assert isinstance(disp_g, api.ADMPDispGenerator) and isinstance(pme_g, api.ADMPPmeGenerator)
    assert set(disp_g.params) == {'A', 'B', 'Q', 'C6', 'C8', 'C10', 'mScales'} and disp_g.pmax == 10
    np.testing.assert_allclose(np.sqrt(disp_g.params['C6'] * 1e6), S.C6, rtol=1e-12)        # admp/api.py:190
    np.testing.assert_allclose(disp_g.params['A'] / 2625.5, S.TT_A, rtol=1e-12)
    assert pme_g.lpol and pme_g.lmax == 2 and list(pme_g.params['mScales']) == [0, 0, 0, 1, 1]
    top = api.Topology.from_pdb(str(tmp_path / 'w.pdb'))
    assert top.n_atoms == 3 * n_mol and len(top.residues) == n_mol
    np.testing.assert_allclose(top.box, box, atol=1e-3)
    typed = api._Typed(H._templates, top)
    assert typed.types[:3] == ['380', '381', '381'] and sorted(typed.bonds[:2]) == [(1, 0), (2, 0)]
    cov = api.build_covalent_map(top.n_atoms, typed.bonds, 6).toarray()
    _, _, cov_ref = S.water_topology(n_mol)
    assert (cov == cov_ref.toarray()).all()
    # local-frame rule and anchors of the first molecule (O: bisector of the hydrogens; H: z = O, x = the other H)
    rules = []
    for t, kz, kx in (('380', '-381', '-381'), ('381', '380', '381')):
        d = {'type': t, 'kz': kz, 'kx': kx, 'ky': ''}
        parser._axis_rule(d)
        rules.append(d['axisType'])
    assert rules == [parser.Bisector, parser.ZThenX]
    assert api._cutoff_angstrom(4.0) == 4.0

    class Q:                                   # an OpenMM-style quantity
        _value, unit = 0.4, 'nanometer'
    assert abs(api._cutoff_angstrom(Q()) - 4.0) < 1e-12
