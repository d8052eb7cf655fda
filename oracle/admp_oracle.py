"""Float64 CPU restatement of ADMP's multipolar / polarizable PME path.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py`` for who may import this and
for the pinning status ("parity unpinned" for the hot path itself).

Written from the formulas of the reference (file:line citations are into the
reference tree) with ``torch`` (CPU, float64); ``torch.autograd`` plays the
role ``jax.value_and_grad`` plays there (reference ``admp/pme.py:108``): the
reference *defines* its forces as the reverse-mode derivative of the energy,
so the derivative of a faithful energy restatement is the force oracle.

Conventions (reference ``admp/pme.py:176-218``): positions (Na,3) in Angstrom,
box (3,3) lattice vectors in rows, multipoles as real spherical harmonics
[00, 10, 11c, 11s, 20, 21c, 21s, 22c, 22s], energies in kJ/mol.
"""
import math

import numpy as np
import torch

DIELECTRIC = 1389.35455846          # admp/pme.py:16
DEFAULT_THOLE_WIDTH = 0.3           # admp/pme.py:17
POL_CONV = 10.0                     # admp/settings.py:29
MAX_N_POL = 30                      # admp/settings.py:30
RT3 = 1.73205080757                 # admp/multipole.py:14 (truncated literal, kept on purpose)
SQRT_PI = 1.7724538509055159        # admp/recip.py:19

F64 = torch.float64

# Memory control for large systems (the reference materialises (Na, 216, 9) and (Np, ~100) intermediates, 16 GB at 1M atoms):
# when set, spread_Q works through the atoms and pme_real through the pairs in chunks of this many rows, and
# torch.utils.checkpoint recomputes a chunk's intermediates in the backward pass.  Same arithmetic, bounded memory.
CHUNK_ATOMS = None
CHUNK_PAIRS = None


def _t(x):
    if isinstance(x, torch.Tensor):
        return x.to(F64)
    return torch.as_tensor(np.asarray(x, dtype=np.float64))


# --------------------------------------------------------------------------------------
# geometry helpers  (admp/spatial.py)
# --------------------------------------------------------------------------------------
def pbc_shift(dr, box, box_inv):
    """Minimum image by fractional rounding (admp/spatial.py:13-32)."""
    ds = dr @ box_inv
    ds = ds - torch.floor(ds + 0.5)
    return ds @ box


def _normalize(v):
    return v / torch.linalg.norm(v, dim=-1, keepdim=True)


def construct_local_frames(positions, box, axis_types, axis_indices):
    """Per-atom local frames, rows = (x, y, z) (admp/spatial.py:76-142).

    axis types: 0 ZThenX, 1 Bisector, 2 ZBisect, 3 ThreeFold, 4 Zonly, 5 none
    (admp/spatial.py:58-64).  axis_indices columns are the (z, x, y) atoms.
    """
    positions = _t(positions)
    box = _t(box)
    axis_types = np.asarray(axis_types)
    axis_indices = np.asarray(axis_indices)
    box_inv = torch.linalg.inv(box)
    n = positions.shape[0]
    z_atoms = torch.as_tensor(axis_indices[:, 0].astype(np.int64))
    x_atoms = torch.as_tensor(axis_indices[:, 1].astype(np.int64))
    y_atoms = torch.as_tensor(axis_indices[:, 2].astype(np.int64))
    zonly = torch.as_tensor(axis_types == 4)
    bis = torch.as_tensor(axis_types == 1)
    zbis = torch.as_tensor(axis_types == 2)
    three = torch.as_tensor(axis_types == 3)

    noaxis = torch.as_tensor(axis_types == 5)
    uses_y = zbis | three
    self_idx = torch.arange(n)
    ex = torch.tensor([1.0, 0.0, 0.0], dtype=F64).expand(n, 3)

    def unit_to(atoms, used, fallback):
        # where an axis atom is not used by the site's rule the reference still indexes it
        # (index -1 wraps) but discards the value; here the unused slots get a constant unit
        # vector so that no 0/0 enters the autograd graph.
        safe = torch.where(used, atoms, (self_idx + 1) % n)
        v = _normalize(pbc_shift(positions[safe] - positions, box, box_inv))
        return torch.where(used[:, None], v, fallback)

    ez = torch.tensor([0.0, 0.0, 1.0], dtype=F64).expand(n, 3)
    vec_z = unit_to(z_atoms, ~noaxis, ez)
    # Z-only: x axis is e_x unless z is (nearly) along x, then e_y (spatial.py:103-105)
    xz = torch.round(torch.abs(vec_z[:, 0])).detach()
    vx_zonly = torch.stack([1.0 - xz, xz, torch.zeros_like(xz)], dim=1)
    vx_other = unit_to(x_atoms, ~(zonly | noaxis), ex)
    vec_x = torch.where(zonly[:, None], vx_zonly, vx_other)
    vec_y_raw = unit_to(y_atoms, uses_y, ex)
    # Bisector (spatial.py:112-114)
    vec_z = torch.where(bis[:, None], _normalize(vec_z + vec_x), vec_z)
    # Z-bisect (spatial.py:116-121)
    vec_x = torch.where(zbis[:, None], _normalize(vec_x + vec_y_raw), vec_x)
    # ThreeFold (spatial.py:123-134)
    vec_z = torch.where(three[:, None], _normalize(vec_z + vec_x + vec_y_raw), vec_z)
    # Gram-Schmidt, then y = z cross x (spatial.py:137-142)
    proj = torch.sum(vec_x * vec_z, dim=1, keepdim=True)
    vec_x = _normalize(vec_x - vec_z * proj)
    vec_y = torch.linalg.cross(vec_z, vec_x)
    return torch.stack([vec_x, vec_y, vec_z], dim=1)


def build_quasi_internal(r1, r2, dr, norm_dr):
    """Pair frame with z along dr (admp/spatial.py:149-178).

    The helper axis is e_x unless the RAW coordinates share y and z
    (spatial.py:172), then e_y.
    """
    vz = dr / norm_dr[:, None]
    use_ex = (r1[:, 1] != r2[:, 1]) | (r1[:, 2] != r2[:, 2])
    ex = torch.tensor([1.0, 0.0, 0.0], dtype=F64)
    ey = torch.tensor([0.0, 1.0, 0.0], dtype=F64)
    vx = torch.where(use_ex[:, None], vz + ex, vz + ey)
    vx = vx - vz * torch.sum(vz * vx, dim=1, keepdim=True)
    vx = vx / torch.linalg.norm(vx, dim=1, keepdim=True)
    vy = torch.linalg.cross(vz, vx)
    return torch.stack([vx, vy, vz], dim=1)


# --------------------------------------------------------------------------------------
# multipole algebra  (admp/multipole.py)
# --------------------------------------------------------------------------------------
_C1_H2C = torch.tensor([[0.0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=F64)     # multipole.py:17-19
_C1_C2H = _C1_H2C.T.contiguous()                                            # multipole.py:20
_IR3 = 1.0 / RT3
_C2_C2H = torch.tensor([[0, 0, 1, 0, 0, 0],
                        [0, 0, 0, 0, 2 * _IR3, 0],
                        [0, 0, 0, 0, 0, 2 * _IR3],
                        [_IR3, -_IR3, 0, 0, 0, 0],
                        [0, 0, 0, 2 * _IR3, 0, 0]], dtype=F64)              # multipole.py:22-26


def convert_cart2harm(theta, lmax=2):
    """[q, dx,dy,dz, xx,yy,zz,xy,xz,yz] -> harmonics (admp/multipole.py:36-77)."""
    theta = _t(theta)
    out = [theta[:, 0:1]]
    if lmax >= 1:
        out.append(theta[:, 1:4] @ _C1_C2H.T)
    if lmax >= 2:
        out.append(theta[:, 4:10] @ _C2_C2H.T)
    return torch.cat(out, dim=1)


def _quad_rot_matrix(R):
    """5x5 l=2 rotation built from frame entries (admp/multipole.py:124-171)."""
    xx, xy, xz = R[:, 0, 0], R[:, 0, 1], R[:, 0, 2]
    yx, yy, yz = R[:, 1, 0], R[:, 1, 1], R[:, 1, 2]
    zx, zy, zz = R[:, 2, 0], R[:, 2, 1], R[:, 2, 2]
    rt3 = RT3
    c = {}
    c[0, 0] = (3 * zz ** 2 - 1) / 2
    c[0, 1] = rt3 * zx * zz
    c[0, 2] = rt3 * zy * zz
    c[0, 3] = (rt3 * (-2 * zy ** 2 - zz ** 2 + 1)) / 2
    c[0, 4] = rt3 * zx * zy
    c[1, 0] = rt3 * xz * zz
    c[1, 1] = 2 * xx * zz - yy
    c[1, 2] = yx + 2 * xy * zz
    c[1, 3] = -2 * xy * zy - xz * zz
    c[1, 4] = xx * zy + zx * xy
    c[2, 0] = rt3 * yz * zz
    c[2, 1] = 2 * yx * zz + xy
    c[2, 2] = -xx + 2 * yy * zz
    c[2, 3] = -2 * yy * zy - yz * zz
    c[2, 4] = yx * zy + zx * yy
    c[3, 0] = rt3 * (-2 * yz ** 2 - zz ** 2 + 1) / 2
    c[3, 1] = -2 * yx * yz - zx * zz
    c[3, 2] = -2 * yy * yz - zy * zz
    c[3, 3] = (4 * yy ** 2 + 2 * zy ** 2 + 2 * yz ** 2 + zz ** 2 - 3) / 2
    c[3, 4] = -2 * yx * yy - zx * zy
    c[4, 0] = rt3 * xz * yz
    c[4, 1] = xx * yz + yx * xz
    c[4, 2] = xy * yz + yy * xz
    c[4, 3] = -2 * xy * yy - xz * yz
    c[4, 4] = xx * yy + yx * xy
    # reference stores C2_gl[a][b] = c[b, a] and then swaps axes (multipole.py:163-171):
    # the matrix applied is M[j, k] = c[j, k]
    rows = [torch.stack([c[j, k] for k in range(5)], dim=1) for j in range(5)]
    return torch.stack(rows, dim=1)


def rot_global2local(Q, R, lmax=2):
    """Rotate harmonics into the frame R (rows x,y,z) (admp/multipole.py:92-179)."""
    out = [Q[:, 0:1]]
    zxy = [2, 0, 1]
    if lmax >= 1:
        R1 = R[:, zxy][:, :, zxy]
        out.append(torch.einsum('nij,nj->ni', R1, Q[:, 1:4]))
    if lmax >= 2:
        out.append(torch.einsum('njk,nk->nj', _quad_rot_matrix(R), Q[:, 4:9]))
    return torch.cat(out, dim=1)


def rot_local2global(Q, R, lmax=2):
    """admp/multipole.py:183-201: global2local with the transposed frame."""
    return rot_global2local(Q, R.transpose(-2, -1), lmax)


def rot_ind_global2local(U, R):
    """Harmonic-ordered dipole rotation (admp/multipole.py:80-89)."""
    zxy = [2, 0, 1]
    R1 = R[:, zxy][:, :, zxy]
    return torch.einsum('nij,nj->ni', R1, U)


# --------------------------------------------------------------------------------------
# real space  (admp/pme.py:258-735)
# --------------------------------------------------------------------------------------
def setup_ewald_parameters(rc, ethresh, box):
    """admp/pme.py:146-172 (uses only the box diagonal)."""
    box = np.asarray(box, dtype=np.float64)
    kappa = math.sqrt(-math.log(2 * ethresh)) / rc
    K = [int(math.ceil(2 * kappa * box[d, d] / 3 / ethresh ** 0.2)) for d in range(3)]
    return kappa, K[0], K[1], K[2]


def _ewald_bvec(dr, kappa):
    """rInvVec, x^n, X and bVec[1..4] (admp/pme.py:283-300)."""
    rinv = 1.0 / dr
    R = [DIELECTRIC * rinv ** i for i in range(9)]
    x = [(kappa * dr) ** i for i in range(10)]
    X = 2 * torch.exp(-x[2]) / math.sqrt(math.pi)
    b = [None] * 6
    b[1] = -torch.erf(x[1])
    tmp = x[1]
    dfac, cnt = 1, 1
    for i in range(2, 6):
        b[i] = b[i - 1] + tmp * X / dfac
        cnt += 2
        dfac *= cnt
        tmp = tmp * 2 * x[2]
    return R, x, X, b


def calc_e_perm(dr, m, kappa, lmax=2):
    """Ten permanent-permanent radial coefficients (admp/pme.py:258-334)."""
    R, x, X, b = _ewald_bvec(dr, kappa)
    z = torch.zeros_like(dr)
    cc = R[1] * (m + b[2] - x[1] * X)
    cd = dd0 = dd1 = cq = dq0 = dq1 = qq0 = qq1 = qq2 = z
    if lmax >= 1:
        cd = R[2] * (m + b[2])
        dd0 = -2 / 3 * R[3] * (3 * (m + b[3]) + x[3] * X)
        dd1 = R[3] * (m + b[3] - (2 / 3) * x[3] * X)
    if lmax >= 2:
        cq = (m + b[3]) * R[3]
        dq0 = R[4] * (3 * (m + b[3]) + (4 / 3) * x[5] * X)
        dq1 = -math.sqrt(3) * R[4] * (m + b[3])
        qq0 = R[5] * (6 * (m + b[4]) + (4 / 45) * (-3 + 10 * x[2]) * x[5] * X)
        qq1 = -(4 / 15) * R[5] * (15 * (m + b[4]) + x[5] * X)
        qq2 = R[5] * (m + b[4] - (4 / 15) * x[5] * X)
    return cc, cd, dd0, dd1, cq, dq0, dq1, qq0, qq1, qq2


def _trim0(x, thresh=1e-8):        # admp/pme.py:351-362
    return torch.where(x < thresh, torch.full_like(x, thresh), x)


def _trim_inf(x, thresh=1e8):      # admp/pme.py:365-376
    return torch.where(x < thresh, x, torch.full_like(x, thresh))


def calc_e_ind(dr, thole1, thole2, dmp, p, kappa, lmax=2):
    """Seven induced-dipole radial coefficients (admp/pme.py:379-475).

    ``dscales`` is accepted by the reference but never used (uscales = 1,
    pme.py:472), so it is not an argument here.
    """
    # Fermi switch between DEFAULT_THOLE_WIDTH (excluded pairs) and thole_i+thole_j (pme.py:337-348, 411)
    u_sw = (p - 1e-3) / 1e-5
    w0 = 1.0 / (torch.exp(u_sw) + 1.0)
    a = w0 * DEFAULT_THOLE_WIDTH + (1.0 - w0) * (thole1 + thole2)
    dmp = _trim0(dmp)
    u = _trim_inf(dr / dmp)
    au = a * u
    expau = torch.where(au < 50, torch.exp(-torch.clamp(au, max=50.0)), torch.zeros_like(au))
    au2 = _trim_inf(au * au)
    au3 = _trim_inf(au2 * au)
    au4 = _trim_inf(au3 * au)
    th_c = 1.0 - expau * (1.0 + au + 0.5 * au2)
    th_d0 = 1.0 - expau * (1.0 + au + 0.5 * au2 + au3 / 4.0)
    th_d1 = 1.0 - expau * (1.0 + au + 0.5 * au2)
    th_q0 = 1.0 - expau * (1.0 + au + 0.5 * au2 + au3 / 6.0 + au4 / 18.0)
    th_q1 = 1.0 - expau * (1.0 + au + 0.5 * au2 + au3 / 6.0)
    R, x, X, b = _ewald_bvec(dr, kappa)
    z = torch.zeros_like(dr)
    cud = 2.0 * R[2] * (p * th_c + b[2])
    dud0 = dud1 = udq0 = udq1 = z
    if lmax >= 1:
        dud0 = -2.0 * 2.0 / 3.0 * R[3] * (3.0 * (p * th_d0 + b[3]) + x[3] * X)
        dud1 = 2.0 * R[3] * (p * th_d1 + b[3] - 2.0 / 3.0 * x[3] * X)
    if lmax >= 2:
        udq0 = 2.0 * R[4] * (3.0 * (p * th_q0 + b[3]) + 4 / 3 * x[5] * X)
        udq1 = -2.0 * math.sqrt(3) * R[4] * (p * th_q1 + b[3])
    udud0 = -2.0 / 3.0 * R[3] * (3.0 * (th_d0 + b[3]) + x[3] * X)
    udud1 = R[3] * (th_d1 + b[3] - 2.0 / 3.0 * x[3] * X)
    return cud, dud0, dud1, udq0, udq1, udud0, udud1


def pme_real_pair_energies(norm_dr, QI, QJ, UI, UJ, th1, th2, dmp, m, p, kappa, lmax, lpol):
    """Per-pair quasi-internal-frame energy (admp/pme.py:479-624), lmax = 2 layout."""
    cc, cd, dd0, dd1, cq, dq0, dq1, qq0, qq1, qq2 = calc_e_perm(norm_dr, m, kappa, lmax)
    n = norm_dr.shape[0]
    Vij = [torch.zeros(n, dtype=F64) for _ in range(9)]
    Vji = [torch.zeros(n, dtype=F64) for _ in range(9)]
    I = [QI[:, k] for k in range(QI.shape[1])] + [torch.zeros(n, dtype=F64)] * (9 - QI.shape[1])
    J = [QJ[:, k] for k in range(QJ.shape[1])] + [torch.zeros(n, dtype=F64)] * (9 - QJ.shape[1])
    Vij[0] = cc * I[0]
    Vji[0] = cc * J[0]
    if lpol:
        cud, dud0, dud1, udq0, udq1, udud0, udud1 = calc_e_ind(norm_dr, th1, th2, dmp, p, kappa, lmax)
        Vij[0] = Vij[0] - cud * UI[:, 0]
        Vji[0] = Vji[0] + cud * UJ[:, 0]
    if lmax >= 1:
        Vij[0] = Vij[0] - cd * I[1]
        Vji[1] = -cd * J[0]
        Vij[1] = cd * I[0]
        Vji[0] = Vji[0] + cd * J[1]
        Vij[1] = Vij[1] + dd0 * I[1]
        Vji[1] = Vji[1] + dd0 * J[1]
        Vij[2] = dd1 * I[2]
        Vji[2] = dd1 * J[2]
        Vij[3] = dd1 * I[3]
        Vji[3] = dd1 * J[3]
        if lpol:
            Vij[1] = Vij[1] + dud0 * UI[:, 0]
            Vji[1] = Vji[1] + dud0 * UJ[:, 0]
            Vij[2] = Vij[2] + dud1 * UI[:, 1]
            Vji[2] = Vji[2] + dud1 * UJ[:, 1]
            Vij[3] = Vij[3] + dud1 * UI[:, 2]
            Vji[3] = Vji[3] + dud1 * UJ[:, 2]
    if lmax >= 2:
        Vij[0] = Vij[0] + cq * I[4]
        Vji[4] = cq * J[0]
        Vij[4] = cq * I[0]
        Vji[0] = Vji[0] + cq * J[4]
        Vij[1] = Vij[1] + dq0 * I[4]
        Vji[4] = Vji[4] + dq0 * J[1]
        Vij[4] = Vij[4] - dq0 * I[1]
        Vji[1] = Vji[1] - dq0 * J[4]
        Vij[2] = Vij[2] + dq1 * I[5]
        Vji[5] = dq1 * J[2]
        Vij[3] = Vij[3] + dq1 * I[6]
        Vji[6] = dq1 * J[3]
        Vij[5] = -(dq1 * I[2])
        Vji[2] = Vji[2] - dq1 * J[5]
        Vij[6] = -(dq1 * I[3])
        Vji[3] = Vji[3] - dq1 * J[6]
        Vij[4] = Vij[4] + qq0 * I[4]
        Vji[4] = Vji[4] + qq0 * J[4]
        Vij[5] = Vij[5] + qq1 * I[5]
        Vji[5] = Vji[5] + qq1 * J[5]
        Vij[6] = Vij[6] + qq1 * I[6]
        Vji[6] = Vji[6] + qq1 * J[6]
        Vij[7] = qq2 * I[7]
        Vji[7] = qq2 * J[7]
        Vij[8] = qq2 * I[8]
        Vji[8] = qq2 * J[8]
        if lpol:
            Vji[4] = Vji[4] + udq0 * UJ[:, 0]
            Vij[4] = Vij[4] - udq0 * UI[:, 0]
            Vji[5] = Vji[5] + udq1 * UJ[:, 1]
            Vji[6] = Vji[6] + udq1 * UJ[:, 2]
            Vij[5] = Vij[5] - udq1 * UI[:, 1]
            Vij[6] = Vij[6] - udq1 * UI[:, 2]
    e = 0.5 * (sum(J[k] * Vij[k] for k in range(9)) + sum(I[k] * Vji[k] for k in range(9)))
    if lpol:
        Vijdd = [udud0 * UI[:, 0], udud1 * UI[:, 1], udud1 * UI[:, 2]]
        Vjidd = [udud0 * UJ[:, 0], udud1 * UJ[:, 1], udud1 * UJ[:, 2]]
        e = e + 0.5 * (sum(UJ[:, k] * Vijdd[k] for k in range(3)) + sum(UI[:, k] * Vjidd[k] for k in range(3)))
    return e


def pair_nbonds(covalent_map, pairs):
    """covalent_map[i, j] for every pair; dense ndarray or scipy sparse accepted."""
    i, j = pairs[:, 0], pairs[:, 1]
    if len(i) == 0:
        return np.zeros(0, dtype=np.int64)
    if hasattr(covalent_map, 'tocsr'):
        return np.asarray(covalent_map.tocsr()[i, j]).ravel().astype(np.int64)
    return np.asarray(covalent_map)[i, j].astype(np.int64)


def filter_pairs(pairs):
    """Keep rows with i < j (admp/pme.py:671)."""
    pairs = np.asarray(pairs)
    return pairs[pairs[:, 0] < pairs[:, 1]].astype(np.int64)


def _scale_lookup(scales, nbonds):
    """scales[nbonds-1] with python/jax negative-index wrap (admp/pme.py:681-683)."""
    idx = torch.as_tensor((nbonds - 1) % len(scales))
    return scales[idx]


def pme_real(positions, box, pairs, Q_global, U_harm, pol, tholes, mScales, pScales,
             covalent_map, kappa, lmax, lpol):
    """admp/pme.py:628-729; chunked over the pairs if CHUNK_PAIRS is set."""
    pairs = filter_pairs(pairs)
    if CHUNK_PAIRS is not None and len(pairs) > CHUNK_PAIRS:
        from torch.utils.checkpoint import checkpoint
        tot = torch.zeros((), dtype=F64)
        dummy = torch.zeros(1, dtype=F64)
        Uh = U_harm if U_harm is not None else dummy
        for p0 in range(0, len(pairs), CHUNK_PAIRS):
            chunk = pairs[p0:p0 + CHUNK_PAIRS]
            tot = tot + checkpoint(
                lambda pos, b, Qg, Uu, chunk=chunk: _pme_real_rows(pos, b, chunk, Qg, Uu if U_harm is not None else None, pol, tholes,
                                                      mScales, pScales, covalent_map, kappa, lmax, lpol),
                positions, box, Q_global, Uh, use_reentrant=False)
        return tot
    return _pme_real_rows(positions, box, pairs, Q_global, U_harm, pol, tholes, mScales, pScales, covalent_map, kappa,
                          lmax, lpol)


def _pme_real_rows(positions, box, pairs, Q_global, U_harm, pol, tholes, mScales, pScales,
                   covalent_map, kappa, lmax, lpol):
    nb = pair_nbonds(covalent_map, pairs)
    pi = torch.as_tensor(pairs[:, 0])
    pj = torch.as_tensor(pairs[:, 1])
    box_inv = torch.linalg.inv(box)
    r1, r2 = positions[pi], positions[pj]
    m = _scale_lookup(mScales, nb)
    dr = pbc_shift(r1 - r2, box, box_inv)
    norm = torch.linalg.norm(dr, dim=-1)
    Ri = build_quasi_internal(r1, r2, dr, norm)
    QI = rot_global2local(Q_global[pi], Ri, lmax)
    QJ = rot_global2local(Q_global[pj], Ri, lmax)
    if lpol:
        p = _scale_lookup(pScales, nb)
        dmp = (pol[pi] * pol[pj]) ** (1.0 / 6.0)         # pme.py:732-735
        UI = rot_ind_global2local(U_harm[pi], Ri)
        UJ = rot_ind_global2local(U_harm[pj], Ri)
        e = pme_real_pair_energies(norm, QI, QJ, UI, UJ, tholes[pi], tholes[pj], dmp, m, p, kappa, lmax, True)
    else:
        e = pme_real_pair_energies(norm, QI, QJ, None, None, None, None, None, m, None, kappa, lmax, False)
    return torch.sum(e)


def pme_self(Q, kappa, lmax=2):
    """admp/pme.py:738-757."""
    nh = (lmax + 1) ** 2
    l_list = np.array([0] + [1] * 3 + [2] * 5)[:nh]
    l_fac2 = np.array([1] + [3] * 3 + [15] * 5)[:nh]
    factor = torch.as_tensor(kappa / np.sqrt(np.pi) * (2 * kappa ** 2) ** l_list / l_fac2)
    return -torch.sum(factor[None, :] * Q ** 2) * DIELECTRIC


def pol_penalty(U, pol):
    """admp/pme.py:760-774."""
    pol_pi = _trim0(pol)
    return torch.sum(0.5 / pol_pi[:, None] * U ** 2) * DIELECTRIC


# --------------------------------------------------------------------------------------
# reciprocal space  (admp/recip.py)
# --------------------------------------------------------------------------------------
_BINOM6 = [1.0, -6.0, 15.0, -20.0, 15.0, -6.0, 1.0]


def _bspline6(u, deriv=0):
    """Order-6 cardinal B-spline and derivatives on [0,6) by truncated powers.

    Algebraically identical to the piecewise polynomials of
    admp/recip.py:80-137 (first four pieces are literally this sum); zero outside.
    """
    p = 5 - deriv
    fac = {0: 120.0, 1: 24.0, 2: 6.0, 3: 2.0}[deriv]
    out = torch.zeros_like(u)
    for k in range(7):
        out = out + _BINOM6[k] * torch.clamp(u - k, min=0.0) ** p
    out = out / fac
    inside = (u >= 0) & (u < 6)
    return torch.where(inside, out, torch.zeros_like(out))


def _shifts():
    """admp/recip.py:27-29: meshgrid(...).T.reshape -> rows ordered (sx slowest?) -- order is
    irrelevant for the sum; we use C order over (sx, sy, sz)."""
    r = np.arange(-3, 3)
    g = np.stack(np.meshgrid(r, r, r, indexing='ij'), axis=-1).reshape(-1, 3)
    return torch.as_tensor(g.astype(np.float64)), torch.as_tensor(g.astype(np.int64))


def kpts_integer(K):
    """FFT-ordered integer frequencies [0..(K-1)//2, -(K//2)..-1] (admp/recip.py:332-341)."""
    lo = (-(K - 1)) // 2
    hi = (K + 1) // 2
    return np.roll(np.arange(lo, hi), lo)


def spread_Q(positions, box, Q, K, lmax):
    """Mesh of B-spline-spread multipoles (admp/recip.py:215-329, 368-392); chunked over the atoms if CHUNK_ATOMS is set."""
    na = positions.shape[0]
    if CHUNK_ATOMS is None or na <= CHUNK_ATOMS:
        return _spread_rows(positions, box, Q, K, lmax)
    from torch.utils.checkpoint import checkpoint
    mesh = None
    for a0 in range(0, na, CHUNK_ATOMS):
        sl = slice(a0, min(na, a0 + CHUNK_ATOMS))
        part = checkpoint(lambda p, b, q: _spread_rows(p, b, q, K, lmax), positions[sl], box, Q[sl], use_reentrant=False)
        mesh = part if mesh is None else mesh + part
    return mesh


def _spread_rows(positions, box, Q, K, lmax):
    N = torch.as_tensor(np.asarray(K, dtype=np.float64))
    box_inv = torch.linalg.inv(box)
    Nstar = (N.reshape(1, 3) * box_inv).T                      # recip.py:52
    Rm = torch.einsum('ij,kj->ki', Nstar, positions)           # recip.py:75
    m_u0 = torch.ceil(Rm.detach())                             # integer part: no gradient (recip.py:76)
    u0 = (m_u0 - Rm) + 3.0                                     # recip.py:77
    sh_f, sh_i = _shifts()
    na = positions.shape[0]
    u = (u0[:, None, :] + sh_f[None]).reshape(na * 216, 3)     # recip.py:239
    M = _bspline6(u, 0)
    theta = M[:, 0] * M[:, 1] * M[:, 2]
    harm = [theta]
    if lmax >= 1:
        Mp = _bspline6(u, 1)
        div = torch.stack([Mp[:, 0] * M[:, 1] * M[:, 2],
                           Mp[:, 1] * M[:, 2] * M[:, 0],
                           Mp[:, 2] * M[:, 0] * M[:, 1]], dim=1)
        tp = torch.einsum('ij,kj->ki', -Nstar, div)            # recip.py:177
        harm += [tp[:, 2], tp[:, 0], tp[:, 1]]                 # recip.py:249-255
    if lmax >= 2:
        Mpp = _bspline6(u, 2)
        d00 = Mpp[:, 0] * M[:, 1] * M[:, 2]
        d11 = Mpp[:, 1] * M[:, 0] * M[:, 2]
        d22 = Mpp[:, 2] * M[:, 0] * M[:, 1]
        d01 = Mp[:, 0] * Mp[:, 1] * M[:, 2]
        d02 = Mp[:, 0] * Mp[:, 2] * M[:, 1]
        d12 = Mp[:, 1] * Mp[:, 2] * M[:, 0]
        dd = torch.stack([torch.stack([d00, d01, d02], dim=1),
                          torch.stack([d01, d11, d12], dim=1),
                          torch.stack([d02, d12, d22], dim=1)], dim=1)      # (n,3,3), symmetric
        t2 = torch.einsum('im,jn,kmn->kij', -Nstar, -Nstar, dd)             # recip.py:212
        rt3 = math.sqrt(3.0)
        tr = t2[:, 0, 0] + t2[:, 1, 1] + t2[:, 2, 2]
        harm += [(3 * t2[:, 2, 2] - tr) / 2, rt3 * t2[:, 0, 2], rt3 * t2[:, 1, 2],
                 rt3 / 2 * (t2[:, 0, 0] - t2[:, 1, 1]), rt3 * t2[:, 0, 1]]  # recip.py:264-271
    harm = torch.stack(harm, dim=1).reshape(na, 216, -1)
    Qd = Q[:, 0:1]
    if lmax >= 1:
        Qd = torch.cat([Qd, Q[:, 1:4]], dim=1)
    if lmax >= 2:
        Qd = torch.cat([Qd, Q[:, 4:9] / 3], dim=1)             # recip.py:305
    Qm = torch.sum(Qd[:, None, :] * harm, dim=2)               # (na, 216)
    Ki = torch.as_tensor(np.asarray(K, dtype=np.int64))
    idx = torch.remainder(m_u0.to(torch.int64)[:, None, :] + sh_i[None], Ki[None, None, :])
    flat = (idx[..., 0] * K[1] + idx[..., 1]) * K[2] + idx[..., 2]
    mesh = torch.zeros(K[0] * K[1] * K[2], dtype=F64)
    mesh = mesh.index_add(0, flat.reshape(-1), Qm.reshape(-1))
    return mesh.reshape(K[0], K[1], K[2])


def Ck_1(ksq, kappa, V):       # recip.py:434-435
    return 2 * math.pi / V / ksq * torch.exp(-ksq / 4 / kappa ** 2)


def _sqrt0(x2):
    """sqrt whose derivative at 0 is 0 instead of inf.  Values are those of the reference's jnp.sqrt (recip.py:439,447,456);
    only the BOX gradient of the k = 0 term differs: there the literal sqrt makes autodiff return inf * 0 = NaN for every
    element of dE/dbox (in the reference as well), although that term does not depend on the k vector at all."""
    return torch.sqrt(torch.clamp(x2, min=1e-300))


def Ck_6(ksq, kappa, V):       # recip.py:437-443
    x2 = ksq / 4 / kappa ** 2
    x = _sqrt0(x2)
    f = (1 - 2 * x2) * torch.exp(-x2) + 2 * x2 * x * SQRT_PI * torch.erfc(x)
    return SQRT_PI * math.pi / 2 / V * kappa ** 3 * f / 3


def Ck_8(ksq, kappa, V):       # recip.py:445-452
    x2 = ksq / 4 / kappa ** 2
    x = _sqrt0(x2)
    x4 = x2 * x2
    f = (3 - 2 * x2 + 4 * x4) * torch.exp(-x2) - 4 * x4 * x * SQRT_PI * torch.erfc(x)
    return SQRT_PI * math.pi / 2 / V * kappa ** 5 * f / 45


def Ck_10(ksq, kappa, V):      # recip.py:454-462
    x2 = ksq / 4 / kappa ** 2
    x = _sqrt0(x2)
    x4 = x2 * x2
    x6 = x4 * x2
    f = (15 - 6 * x2 + 4 * x4 - 8 * x6) * torch.exp(-x2) + 8 * x6 * x * SQRT_PI * torch.erfc(x)
    return SQRT_PI * math.pi / 2 / V * kappa ** 7 * f / 1260


def kspace_tables(box, K, quirk=True):
    """Integer k per flattened mesh point, k^2 and theta_k (admp/recip.py:332-365, 400-408).

    quirk=True reproduces the reference's column order: meshgrid(kz, kx, ky) with 'xy'
    indexing gives an array of shape (K1, K3, K2) whose flattened rows are
    (kz[a1], kx[a0], ky[a2]) (recip.py:339-340) -- i.e. the frequencies of mesh axes
    (1, 0, 2) land in k-columns (0, 1, 2).  Harmless iff K1 = K2 = K3 and |a| = |b|
    orthorhombic.  quirk=False is the physically consistent assignment.
    """
    K1, K2, K3 = K
    kx, ky, kz = kpts_integer(K1), kpts_integer(K2), kpts_integer(K3)
    if quirk:
        a, b, c = np.meshgrid(kz, kx, ky)          # default 'xy' indexing
        kint = np.stack([a.ravel(), b.ravel(), c.ravel()], axis=1)
    else:
        a, b, c = np.meshgrid(kx, ky, kz, indexing='ij')
        kint = np.stack([a.ravel(), b.ravel(), c.ravel()], axis=1)
    kint_t = torch.as_tensor(kint.astype(np.float64))
    box_inv = torch.linalg.inv(box)
    kpts = 2 * math.pi * kint_t @ box_inv
    ksq = torch.sum(kpts ** 2, dim=1)
    N = np.array([K1, K2, K3], dtype=np.float64).reshape(1, 1, 3)
    m = np.linspace(-2, 2, 5).reshape(5, 1, 1)
    M6 = np.array([1.0, 26.0, 66.0, 26.0, 1.0]).reshape(5, 1, 1) / 120.0   # bspline(m+3)
    theta_k = np.prod(np.sum(M6 * np.cos(2 * np.pi * m * kint[None] / N), axis=0), axis=1)
    return ksq, torch.as_tensor(theta_k)


def pme_recip(positions, box, Q, kappa, K, lmax, Ck_fn=Ck_1, gamma=False, quirk=True):
    """admp/recip.py:31-426."""
    mesh = spread_Q(positions, box, Q, K, lmax)
    ksq, theta_k = kspace_tables(box, K, quirk)
    V = torch.linalg.det(box)
    S = torch.fft.fftn(mesh).reshape(-1)
    if not gamma:
        Ck = Ck_fn(ksq[1:], kappa, V)
        E = Ck * torch.abs(S[1:] / theta_k[1:]) ** 2
        return torch.sum(E) * DIELECTRIC
    Ck = Ck_fn(ksq, kappa, V)
    return torch.sum(Ck * torch.abs(S / theta_k) ** 2)


# --------------------------------------------------------------------------------------
# top level  (admp/pme.py:30-254)
# --------------------------------------------------------------------------------------
class PmeSystem:
    """Static ("environment") part of an ADMPPmeForce (admp/pme.py:37-55)."""

    def __init__(self, axis_type, axis_indices, covalent_map, kappa, K, lmax, lpol):
        self.axis_type = np.asarray(axis_type)
        self.axis_indices = np.asarray(axis_indices)
        self.covalent_map = covalent_map
        self.kappa = float(kappa)
        self.K = tuple(int(k) for k in K)
        self.lmax = int(lmax)
        self.lpol = bool(lpol)


def energy_pme_parts(sysm, positions, box, pairs, Q_local, Uind_global, pol, tholes,
                     mScales, pScales):
    """(E_real, E_recip, E_self, E_penalty) (admp/pme.py:176-254)."""
    lmax = sysm.lmax
    if lmax > 0:
        frames = construct_local_frames(positions, box, sysm.axis_type, sysm.axis_indices)
        Q_global = rot_local2global(Q_local, frames, lmax)
    else:
        Q_global = Q_local
    zero = torch.zeros((), dtype=F64)
    if sysm.lpol:
        U_h = Uind_global @ _C1_C2H.T                         # pme.py:235
        Q_tot = torch.cat([Q_global[:, 0:1], Q_global[:, 1:4] + U_h, Q_global[:, 4:]], dim=1)
        e_real = pme_real(positions, box, pairs, Q_global, U_h, pol, tholes, mScales, pScales,
                          sysm.covalent_map, sysm.kappa, lmax, True)
        e_pen = pol_penalty(U_h, pol)
    else:
        Q_tot = Q_global
        e_real = pme_real(positions, box, pairs, Q_global, None, None, None, mScales, None,
                          sysm.covalent_map, sysm.kappa, lmax, False)
        e_pen = zero
    e_recip = pme_recip(positions, box, Q_tot, sysm.kappa, sysm.K, lmax)
    e_self = pme_self(Q_tot, sysm.kappa, lmax)
    return e_real, e_recip, e_self, e_pen


def energy_pme(sysm, positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales):
    return sum(energy_pme_parts(sysm, positions, box, pairs, Q_local, Uind_global, pol, tholes,
                                mScales, pScales))


def optimize_Uind(sysm, positions, box, pairs, Q_local, pol, tholes, mScales, pScales,
                  U_init=None, maxiter=MAX_N_POL, thresh=POL_CONV, history=None):
    """Jacobi SCF of the induced dipoles (admp/pme.py:111-143).

    Returns (U, converged_flag, i) with the reference's conventions: ``i`` is the
    loop index at exit and the flag is ``i != maxiter-1`` (pme.py:139-143).
    """
    positions, box, Q_local = _t(positions).detach(), _t(box).detach(), _t(Q_local).detach()
    pol, tholes = _t(pol), _t(tholes)
    mScales, pScales = _t(mScales), _t(pScales)
    U = torch.zeros((positions.shape[0], 3), dtype=F64) if U_init is None else _t(U_init).clone()
    site = pol > 0.001
    i = 0
    for i in range(maxiter):
        Ug = U.clone().requires_grad_(True)
        e = energy_pme(sysm, positions, box, pairs, Q_local, Ug, pol, tholes, mScales, pScales)
        field, = torch.autograd.grad(e, Ug)
        fmax = float(torch.max(torch.abs(field[site]))) if bool(site.any()) else 0.0
        if history is not None:
            history.append(fmax)
        if fmax < thresh:
            break
        U = U - field * pol[:, None] / DIELECTRIC
    flag = (i != maxiter - 1)
    return U.detach(), flag, i


def pme_energy_and_grad(sysm, positions, box, pairs, Q_local, mScales, pol=None, tholes=None,
                        pScales=None, U_init=None, want_dQ=False, want_dbox=False):
    """``get_forces`` of the reference: (E, dE/dpositions) (+ parts, U, flags).

    Polarizable: SCF first, then the gradient at FIXED U (admp/pme.py:81-85).
    want_dbox: also dE/dbox at fixed Cartesian positions -- what ``value_and_grad(get_energy, argnums=1)`` returns in
    the reference (admp/pme.py:108 with argnums; README.md:7 "force and virial").
    """
    positions_t = _t(positions).clone().requires_grad_(True)
    box_t = _t(box).clone().requires_grad_(want_dbox)
    Q_t = _t(Q_local).clone().requires_grad_(want_dQ)
    mS = _t(mScales)
    out = {}
    if sysm.lpol:
        pol_t, th_t, pS = _t(pol), _t(tholes), _t(pScales)
        U, flag, ncyc = optimize_Uind(sysm, positions, box, pairs, Q_local, pol_t, th_t, mS, pS, U_init=U_init)
        out.update(U_ind=U.numpy().copy(), lconverg=flag, n_cycle=ncyc)
        parts = energy_pme_parts(sysm, positions_t, box_t, pairs, Q_t, U, pol_t, th_t, mS, pS)
    else:
        parts = energy_pme_parts(sysm, positions_t, box_t, pairs, Q_t, None, None, None, mS, None)
    e = sum(parts)
    inputs = [positions_t] + ([Q_t] if want_dQ else []) + ([box_t] if want_dbox else [])
    grads = torch.autograd.grad(e, inputs)
    out.update(E=float(e.detach()), parts=[float(p.detach()) for p in parts], grad=grads[0].numpy().copy())
    if want_dQ:
        out['dQ_local'] = grads[1].numpy().copy()
    if want_dbox:
        out['dbox'] = grads[-1].numpy().copy()
    return out


# --------------------------------------------------------------------------------------
# dispersion PME  (admp/disp_pme.py) and Tang-Toennies damping (admp/pairwise.py)
# --------------------------------------------------------------------------------------
def _pair_distances(positions, box, pairs, mScales, covalent_map):
    pairs = filter_pairs(pairs)
    nb = pair_nbonds(covalent_map, pairs)
    pi = torch.as_tensor(pairs[:, 0])
    pj = torch.as_tensor(pairs[:, 1])
    box_inv = torch.linalg.inv(box)
    dr = pbc_shift(positions[pi] - positions[pj], box, box_inv)
    return pi, pj, dr, _scale_lookup(mScales, nb)


def disp_pme_parts(positions, box, pairs, c_list, mScales, covalent_map, kappa, K, pmax):
    """(E_real, E_recip, E_self) of admp/disp_pme.py:80-279."""
    pi, pj, dr, m = _pair_distances(positions, box, pairs, mScales, covalent_map)
    dr2 = torch.sum(dr * dr, dim=1)
    x2 = kappa * kappa * dr2
    x4 = x2 * x2
    ex = torch.exp(-x2)
    g6 = (1 + x2 + 0.5 * x4) * ex
    dr6 = dr2 ** 3
    ci, cj = c_list[pi], c_list[pj]
    e_real = (m + g6 - 1) * ci[:, 0] * cj[:, 0] / dr6
    e_recip = pme_recip(positions, box, c_list[:, 0:1], kappa, K, 0, Ck_6, gamma=True)
    e_self = -kappa ** 6 / 12 * torch.sum(c_list[:, 0] ** 2)
    if pmax >= 8:
        g8 = g6 + x4 * x2 / 6 * ex
        e_real = e_real + (m + g8 - 1) * ci[:, 1] * cj[:, 1] / (dr6 * dr2)
        e_recip = e_recip + pme_recip(positions, box, c_list[:, 1:2], kappa, K, 0, Ck_8, gamma=True)
        e_self = e_self - kappa ** 8 / 48 * torch.sum(c_list[:, 1] ** 2)
    if pmax >= 10:
        g10 = g8 + x4 * x4 / 24 * ex
        e_real = e_real + (m + g10 - 1) * ci[:, 2] * cj[:, 2] / (dr6 * dr2 * dr2)
        e_recip = e_recip + pme_recip(positions, box, c_list[:, 2:3], kappa, K, 0, Ck_10, gamma=True)
        e_self = e_self - kappa ** 10 / 240 * torch.sum(c_list[:, 2] ** 2)
    return torch.sum(e_real), e_recip, e_self


def disp_energy_and_grad(positions, box, pairs, c_list, mScales, covalent_map, kappa, K, pmax, want_dbox=False):
    p = _t(positions).clone().requires_grad_(True)
    b = _t(box).clone().requires_grad_(want_dbox)
    parts = disp_pme_parts(p, b, pairs, _t(c_list), _t(mScales), covalent_map, kappa, K, pmax)
    e = sum(parts)
    gs = torch.autograd.grad(e, [p, b] if want_dbox else [p])
    out = dict(E=float(e.detach()), parts=[float(x.detach()) for x in parts], grad=gs[0].numpy().copy())
    if want_dbox:
        out['dbox'] = gs[1].numpy().copy()
    return out


def tt_damping_energy(positions, box, pairs, mScales, covalent_map, a_list, b_list, q_list, c_list):
    """generate_pairwise_interaction + TT_damping_qq_c6_kernel (admp/pairwise.py:45-113)."""
    pi, pj, drv, m = _pair_distances(positions, box, pairs, mScales, covalent_map)
    dr = torch.linalg.norm(drv, dim=1)
    a = torch.sqrt(a_list[pi] * a_list[pj])
    b = torch.sqrt(b_list[pi] * b_list[pj])
    c = c_list[pi] * c_list[pj]
    q = q_list[pi] * q_list[pj]
    br = b * dr * 1.889726878
    ebr = torch.exp(-br)
    poly = 1 + br + br ** 2 / 2 + br ** 3 / 6 + br ** 4 / 24 + br ** 5 / 120 + br ** 6 / 720
    f = 2625.5 * a * ebr + (-2625.5) * ebr * (1 + br) * q / br + ebr * poly * c / dr ** 6
    return torch.sum(f * m)


def tt_energy_and_grad(positions, box, pairs, mScales, covalent_map, a_list, b_list, q_list, c_list, want_dbox=False):
    p = _t(positions).clone().requires_grad_(True)
    b = _t(box).clone().requires_grad_(want_dbox)
    e = tt_damping_energy(p, b, pairs, _t(mScales), covalent_map, _t(a_list), _t(b_list),
                          _t(q_list), _t(c_list))
    gs = torch.autograd.grad(e, [p, b] if want_dbox else [p])
    out = dict(E=float(e.detach()), grad=gs[0].numpy().copy())
    if want_dbox:
        out['dbox'] = gs[1].numpy().copy()
    return out
