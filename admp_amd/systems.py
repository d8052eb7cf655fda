"""Input preparation for the water workloads of the PME path.

Everything the reference's example drivers build before they call
``ADMPPmeForce.get_forces`` (reference ``examples/water_1024/run_admp.py:23-112``,
``examples/water_pol_1024/run_admp.py:19-132``): the MPID water parameters,
local-axis topology, covalent map, a pair list -- plus the seeded synthetic
liquid boxes the benchmarks run on (sets S1/S2/S3).

Host-side numpy only; nothing here is on the timed path.
"""
import math

import numpy as np

# --- MPID water model (values of the reference's examples/water_pol_1024/mpidwater.xml:27-40;
#     unit conversions of examples/water_pol_1024/run_admp.py:46-50,63-64: nm->A x10, nm^2->A^2 x300 (sic),
#     nm^3->A^3 x1000) -------------------------------------------------------------------------------
_O_CART = [-1.0614, 0.0, 0.0, -0.023671684 * 10,
           0.000150963 * 300, 0.00008707 * 300, -0.000238034 * 300, 0.0, 0.0, 0.0]
_H_CART = [0.5307] + [0.0] * 9
#: polarizability goes through float32 in the reference driver (run_admp.py:63-64)
POL_O = float(np.float32(1000) * np.mean(np.array([0.00088] * 3, dtype=np.float32)))
THOLE_O = 8.0
R_OH = 0.9572
ANG_HOH = math.radians(104.52)
WATER_DENSITY = 0.03343          # molecules / A^3

# dispersion / Tang-Toennies per-atom lists of the drivers (run_admp.py:66-97)
C6 = (37.19677405, 7.6111103)
C8 = (85.26810658, 11.90220148)
C10 = (134.44874488, 15.05074749)
TT_Q = (-0.741706, 0.370853)
TT_B = (2.00095977, 1.999519942)
TT_A = (458.3777, 0.0317)

_RT3 = 1.73205080757             # the reference's truncated sqrt(3) (admp/multipole.py:14)


def convert_cart2harm(theta, lmax=2):
    """Cartesian [q, dx,dy,dz, xx,yy,zz,xy,xz,yz] -> real spherical harmonics
    [00, 10,11c,11s, 20,21c,21s,22c,22s] (reference admp/multipole.py:36-77)."""
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    out = [theta[:, 0:1]]
    if lmax >= 1:
        out.append(theta[:, [3, 1, 2]])
    if lmax >= 2:
        xx, yy, zz, xy, xz, yz = (theta[:, 4 + k] for k in range(6))
        s = 1.0 / _RT3
        out.append(np.stack([zz, 2 * s * xz, 2 * s * yz, s * (xx - yy), 2 * s * xy], axis=1))
    return np.concatenate(out, axis=1)


def water_topology(n_mol):
    """axis types / axis atoms / covalent map for O,H1,H2-ordered waters.

    O is a Bisector site (z = H1, x = H2), each H is ZThenX (z = O, x = other H)
    (reference tests/test_sptial.py:74-84).  The covalent map is returned as a
    scipy CSR matrix (O-H = 1, H-H = 2); ``.toarray()`` gives the reference's
    dense Na x Na form for small systems.
    """
    import scipy.sparse as sp
    na = 3 * n_mol
    o = 3 * np.arange(n_mol)
    axis_type = np.tile(np.array([1, 0, 0], dtype=np.int32), n_mol)
    axis_indices = np.empty((na, 3), dtype=np.int32)
    axis_indices[o] = np.stack([o + 1, o + 2, -np.ones_like(o)], axis=1)
    axis_indices[o + 1] = np.stack([o, o + 2, -np.ones_like(o)], axis=1)
    axis_indices[o + 2] = np.stack([o, o + 1, -np.ones_like(o)], axis=1)
    rows = np.concatenate([o, o, o + 1, o + 1, o + 2, o + 2])
    cols = np.concatenate([o + 1, o + 2, o, o + 2, o, o + 1])
    vals = np.concatenate([np.ones(2 * n_mol), np.ones(n_mol), 2 * np.ones(n_mol),
                           np.ones(n_mol), 2 * np.ones(n_mol)]).astype(np.int32)
    cov = sp.csr_matrix((vals, (rows, cols)), shape=(na, na), dtype=np.int32)
    return axis_type, axis_indices, cov


def water_parameters(n_mol, polarizable=True):
    """Per-atom parameter arrays in the units the PME path takes."""
    na = 3 * n_mol
    q_cart = np.tile(np.array([_O_CART, _H_CART, _H_CART]), (n_mol, 1))
    Q_local = convert_cart2harm(q_cart, 2)
    pol = np.tile(np.array([POL_O if polarizable else 0.0, 0.0, 0.0]), n_mol)
    tholes = np.tile(np.array([THOLE_O, 0.0, 0.0]), n_mol)

    def per_atom(pair):
        return np.tile(np.array([pair[0], pair[1], pair[1]]), n_mol)
    c_list = np.stack([per_atom(C6), per_atom(C8), per_atom(C10)], axis=1)      # (Na, 3)
    return dict(Q_local=Q_local, pol=pol, tholes=tholes, c_list=c_list,
                a_list=per_atom(TT_A), b_list=per_atom(TT_B), q_list=per_atom(TT_Q),
                mScales=np.array([0.0, 0.0, 0.0, 1.0, 1.0]),
                pScales=np.array([0.0, 0.0, 0.0, 1.0, 1.0]),
                dScales=np.array([0.0, 0.0, 0.0, 1.0, 1.0]), n_atoms=na)


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)


def synthetic_water_box(n_mol, seed=20240, density=WATER_DENSITY, jitter=0.3,
                        min_oo=2.6, min_any=1.5, max_sweeps=200):
    """Seeded liquid-density cubic water box (SURVEY.md 8d "performance set").

    Rigid molecules on a jittered simple-cubic lattice with uniformly random
    orientations; molecules with an O-O contact < ``min_oo`` or any
    intermolecular atom contact < ``min_any`` are redrawn until none is left.
    Returns (positions (3*n_mol,3) ordered O,H1,H2, box (3,3)).
    """
    rng = np.random.default_rng(seed)
    L = (n_mol / density) ** (1.0 / 3.0)
    n = int(math.ceil(n_mol ** (1.0 / 3.0) - 1e-9))
    a = L / n
    sites = rng.permutation(n ** 3)[:n_mol] if n ** 3 > n_mol else np.arange(n_mol)
    sites = np.sort(sites)
    ijk = np.stack(np.unravel_index(sites, (n, n, n)), axis=1)
    site_to_mol = -np.ones(n ** 3, dtype=np.int64)
    site_to_mol[sites] = np.arange(n_mol)
    half = 0.5 * ANG_HOH
    body = np.array([[0.0, 0.0, 0.0],
                     [R_OH * math.sin(half), 0.0, R_OH * math.cos(half)],
                     [-R_OH * math.sin(half), 0.0, R_OH * math.cos(half)]])

    def draw(idx):
        centre = (ijk[idx] + 0.5) * a + rng.uniform(-jitter, jitter, size=(len(idx), 3))
        rot = _random_rotations(rng, len(idx))
        return centre[:, None, :] + np.einsum('nij,kj->nki', rot, body)

    mol = draw(np.arange(n_mol))                      # (n_mol, 3 atoms, 3)
    offs = np.array([(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)
                     if (dx, dy, dz) != (0, 0, 0)])
    active = np.arange(n_mol)                         # molecules whose contacts must be (re)checked
    for _ in range(max_sweeps):
        bad = np.zeros(n_mol, dtype=bool)
        prio = rng.random(n_mol)                      # of a clashing pair, the lower priority is redrawn
        for off in offs:
            nb_ijk = (ijk[active] + off) % n
            nb = site_to_mol[np.ravel_multi_index(nb_ijk.T, (n, n, n))]
            ok = (nb >= 0) & (nb != active)
            sel = np.nonzero(ok)[0]
            d = mol[active[sel]][:, :, None, :] - mol[nb[sel]][:, None, :, :]
            d -= L * np.round(d / L)
            r2 = np.einsum('mabk,mabk->mab', d, d)    # (m, 3, 3) squared distances
            clash = (r2[:, 0, 0] < min_oo ** 2) | (r2.reshape(len(sel), 9).min(axis=1) < min_any ** 2)
            ca, cb = active[sel[clash]], nb[sel[clash]]
            bad[np.where(prio[ca] < prio[cb], ca, cb)] = True
        active = np.nonzero(bad)[0]
        if len(active) == 0:
            break
        mol[active] = draw(active)
    else:
        raise RuntimeError('synthetic_water_box: could not remove close contacts')
    positions = mol.reshape(3 * n_mol, 3)
    return positions, np.eye(3) * L


def tile_box(positions, box, reps):
    """Periodic replication (molecule order kept within each image)."""
    box = np.asarray(box, dtype=np.float64)
    out = []
    for ix in range(reps[0]):
        for iy in range(reps[1]):
            for iz in range(reps[2]):
                out.append(positions + ix * box[0] + iy * box[1] + iz * box[2])
    return np.concatenate(out, axis=0), box * np.array(reps, dtype=np.float64)[:, None]


def build_pairs(positions, box, rc):
    """Half pair list (i < j, minimum-image r < rc) for an orthorhombic box.

    Stand-in for ``jax_md.partition.neighbor_list(..., format=OrderedSparse)``
    of the reference drivers (examples/water_1024/run_admp.py:109-112).
    """
    from scipy.spatial import cKDTree
    box = np.asarray(box, dtype=np.float64)
    if np.abs(box - np.diag(np.diag(box))).max() > 0:
        raise NotImplementedError('build_pairs: orthorhombic boxes only')
    L = np.diag(box)
    wrapped = np.mod(positions, L)
    wrapped = np.where(wrapped >= L, 0.0, wrapped)
    tree = cKDTree(wrapped, boxsize=L)
    pairs = tree.query_pairs(rc, output_type='ndarray')
    pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
    return pairs.astype(np.int32)


def load_pdb_positions(path):
    """Minimal CRYST1/ATOM/HETATM reader (columns as in the PDB v3.3 spec)."""
    pos, cell = [], None
    with open(path) as fh:
        for line in fh:
            if line.startswith('CRYST1'):
                cell = [float(line[6:15]), float(line[15:24]), float(line[24:33])]
            elif line.startswith(('ATOM', 'HETATM')):
                pos.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
            elif line.startswith('END'):
                break
    return np.array(pos), np.diag(cell)
