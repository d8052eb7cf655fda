"""ADMPPmeForce -- drop-in counterpart of the reference's admp/pme.py:30-143, running on MI355X.

Same constructor, attributes and callables:

    pme_force = ADMPPmeForce(box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax, lpol=False)
    pme_force.update_env('kappa', 0.657065221219616)
    E, G = pme_force.get_forces(positions, box, pairs, Q_local, mScales)                       # non-polarizable
    E, G = pme_force.get_forces(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales,
                                U_init=pme_force.U_ind)                                         # polarizable

`G` is +dE/dpositions, as in the reference (jax.value_and_grad, admp/pme.py:108).  The work is done by
libadmp_hip (include/admp_hip.h); there is no autodiff: the gradient is the hand-coded adjoint of the kernels.
Differences a caller can see: arrays may be numpy or torch (results come back in the container type of
`positions`); covalent_map may also be a scipy sparse matrix (mandatory beyond ~50k atoms); the x/y k-column
quirk of the reference (admp/recip.py:339-340) is not emulated, see DESIGN.md.
"""
import ctypes
import math
import warnings

import numpy as np
import torch

from . import _lib, settings
from ._device import HipForceBase

DIELECTRIC = 1389.35455846          # admp/pme.py:16
DEFAULT_THOLE_WIDTH = 0.3           # admp/pme.py:17


def setup_ewald_parameters(rc, ethresh, box):
    """kappa, K1, K2, K3 exactly as admp/pme.py:146-172 (OpenMM's rule; only the box diagonal is used)."""
    box = np.asarray(box.detach().cpu() if isinstance(box, torch.Tensor) else box, dtype=np.float64)
    kappa = math.sqrt(-math.log(2 * ethresh)) / rc
    K1 = math.ceil(2 * kappa * box[0, 0] / 3 / ethresh ** 0.2)
    K2 = math.ceil(2 * kappa * box[1, 1] / 3 / ethresh ** 0.2)
    K3 = math.ceil(2 * kappa * box[2, 2] / 3 / ethresh ** 0.2)
    if settings.FFT_FRIENDLY_MESH:
        K1, K2, K3 = (next_smooth(int(k)) for k in (K1, K2, K3))
    return kappa, int(K1), int(K2), int(K3)


def next_smooth(n):
    """smallest m >= n whose prime factors are all <= 7 (settings.FFT_FRIENDLY_MESH)"""
    m = max(int(n), 1)
    while True:
        k = m
        for p in (2, 3, 5, 7):
            while k % p == 0:
                k //= p
        if k == 1:
            return m
        m += 1


class ADMPPmeForce(HipForceBase):
    """Multipolar (optionally polarizable) PME; see the module docstring."""

    def __init__(self, box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax, lpol=False, device=None):
        self.axis_type = axis_type
        self.axis_indices = axis_indices
        self.rc = rc
        self.ethresh = ethresh
        self.lmax = int(lmax)
        kappa, K1, K2, K3 = setup_ewald_parameters(rc, ethresh, box)
        self.kappa = kappa
        self.K1, self.K2, self.K3 = K1, K2, K3
        self.pme_order = 6
        self.covalent_map = covalent_map
        self.lpol = bool(lpol)
        n_atoms = int(covalent_map.shape[0])
        if self.lmax > 2:
            raise NotImplementedError('l > 2 (beyond quadrupole) not supported')      # admp/recip.py:275
        if self.lmax == 0:
            axis_type = axis_indices = None        # no frames needed (admp/pme.py:101-104)
        super().__init__(n_atoms, covalent_map, axis_type, axis_indices, device)
        self.U_ind = None
        self.lconverg = None
        self.n_cycle = None
        self.energy_parts = None
        self._out_E = (ctypes.c_double * 4)()
        self._out_ncyc, self._out_conv = ctypes.c_int(0), ctypes.c_int(1)
        self._out_ncyc_ref, self._out_conv_ref = ctypes.byref(self._out_ncyc), ctypes.byref(self._out_conv)
        self.refresh_calculators()

    # ---- environment (admp/pme.py:89-109) ------------------------------------------------------------------
    def update_env(self, attr, val):
        """Update an environment parameter (kappa, K1, K2, K3, lmax, ...) and refresh the calculators."""
        setattr(self, attr, val)
        self.refresh_calculators()

    def refresh_calculators(self):
        _lib.check(self._h, self._L.admp_set_ewald(self._h, float(self.kappa), int(self.K1), int(self.K2), int(self.K3),
                                                   int(self.lmax), 1 if self.lpol else 0), 'admp_set_ewald')
        self._ref_korder = bool(settings.REFERENCE_KPOINT_ORDER)
        _lib.check(self._h, self._L.admp_set_option(self._h, _lib.OPT_REFERENCE_KPOINTS, int(self._ref_korder)),
                   'admp_set_option')
        # the reference's attributes of the same names are closures over jitted JAX functions (admp/pme.py:102-105);
        # here they expose the corresponding device computations
        self.construct_local_frames = self._construct_local_frames if self.lmax > 0 else None
        self.pme_recip = self._pme_recip
        self.get_energy = self.generate_get_energy()
        self.get_forces = self._generate_get_forces()
        self.get_energy._value_and_grad = self.get_forces          # value_and_grad(pme_force.get_energy) -> get_forces
        self.get_energy._value_and_box_grad = self.get_energy_and_box_gradient      # ... argnums=1
        if self.lpol:
            self.U_ind = np.zeros((self.n_atoms, 3))      # reset with the closures, admp/pme.py:79
            def energy_fn(*a, **k):                       # plain function objects, like the reference's closures
                return self._energy_fn(*a, **k)
            energy_fn._grads = {0: self._grad_pos_fn, 4: self._grad_U_fn}
            energy_fn.__doc__ = self._energy_fn.__doc__
            self.energy_fn, self.grad_U_fn, self.grad_pos_fn = energy_fn, self._grad_U_fn, self._grad_pos_fn

    # ---- calculators ------------------------------------------------------------------------------------------
    def _pad_Q(self, Q_local):
        nh = (self.lmax + 1) ** 2
        if isinstance(Q_local, torch.Tensor):
            if Q_local.dtype == self._dtype and Q_local.is_cuda and Q_local.dim() == 2 and Q_local.shape[1] == 9 and \
                    nh == 9 and Q_local.shape[0] == self.n_atoms and Q_local.is_contiguous() and \
                    not Q_local.requires_grad and Q_local.device.index == self._device.index:
                return Q_local             # already the library's layout and precision
            q = Q_local.detach().to(device=self._device, dtype=self._dtype)
        else:
            q = torch.as_tensor(np.asarray(Q_local, dtype=np.float64), dtype=self._dtype).to(self._device)
        if q.dim() != 2 or q.shape[0] != self.n_atoms or q.shape[1] < nh:
            raise ValueError('Q_local must be (Na, >= (lmax+1)^2)')
        if q.shape[1] == 9 and nh == 9 and q.is_contiguous():
            return q                       # already in the library's layout: no copy
        out = torch.zeros((self.n_atoms, 9), dtype=self._dtype, device=self._device)
        out[:, :nh] = q[:, :nh]
        return out

    def _hint_pol_sites(self, pol, pol_t):
        """ADMP_OPT_KEEP_POL_SITES: tell the library when the polarizabilities are the ones of the previous call (same torch
        tensor, not written since / same numpy content), so that it keeps its list of polarizable sites."""
        if isinstance(pol, torch.Tensor):
            key = ('t', id(pol), pol.data_ptr(), pol._version, tuple(pol.shape))
        else:
            key = ('n', np.asarray(pol, dtype=np.float64).tobytes())
        same = key == getattr(self, '_pol_key', None)
        if same != getattr(self, '_pol_hint', None):
            _lib.check(self._h, self._L.admp_set_option(self._h, _lib.OPT_KEEP_POL_SITES, int(same)), 'admp_set_option')
            self._pol_hint = same
        self._pol_key = key
        self._pol_keep = pol          # keeps id() unique while cached

    def _evaluate(self, positions, box, pairs, Q_local, mScales, pol=None, tholes=None, pScales=None, dScales=None,
                  U_init=None, want_grad=True, want_dQ=False, maxiter=None, thresh=None):
        prev = self._enter_stream()
        try:
            return self._evaluate_on_stream(positions, box, pairs, Q_local, mScales, pol, tholes, pScales, dScales,
                                            U_init, want_grad, want_dQ, maxiter, thresh)
        finally:
            self._leave_stream(prev)

    def _evaluate_on_stream(self, positions, box, pairs, Q_local, mScales, pol, tholes, pScales, dScales, U_init,
                            want_grad, want_dQ, maxiter, thresh):
        L, h, na = self._L, self._h, self.n_atoms
        if (self.K1 != self.K2 or self.K2 != self.K3) and self._ref_korder and not getattr(self, '_warned_k', False):
            self._warned_k = True
            warnings.warn('unequal PME mesh dimensions: the k-points are assigned in the reference\'s literal order '
                          '(admp/recip.py:339-340, meshgrid(kz, kx, ky)), which reproduces its numbers but is not a '
                          'self-consistent Ewald sum on this mesh; settings.REFERENCE_KPOINT_ORDER = False selects the '
                          'consistent assignment')
        self.set_pairs(pairs)
        pos = self._real(positions, (na, 3))
        Q = self._pad_Q(Q_local)
        boxa, _ = self._harr('box', box, 9)
        mSa, ns = self._harr('mS', mScales)
        E, ncyc, conv = self._out_E, self._out_ncyc, self._out_conv      # reused ctypes outputs
        ncyc.value, conv.value = 0, 1
        grad = torch.empty((na, 3), dtype=self._dtype, device=self._device) if want_grad else None
        dQ = torch.empty((na, 9), dtype=self._dtype, device=self._device) if want_dQ else None
        pol_t = th_t = U = None
        pS = dS = None
        if self.lpol:
            pol_t = self._real(pol, (na,))
            th_t = self._real(tholes, (na,))
            self._hint_pol_sites(pol, pol_t)
            pS, _ = self._harr('pS', pScales, ns)
            dS = self._harr('dS', dScales, ns)[0] if dScales is not None else pS
        maxiter = settings.MAX_N_POL if maxiter is None else int(maxiter)
        thresh = settings.POL_CONV if thresh is None else float(thresh)
        P = self._ptr
        if self.lpol:
            if U_init is None:
                U = torch.zeros((na, 3), dtype=self._dtype, device=self._device)
            else:    # the caller's array is not touched: the library reads it and writes the new dipoles into a fresh one
                U_first = self._real(U_init, (na, 3))
                U = torch.empty((na, 3), dtype=self._dtype, device=self._device)
                # armed LAST, right before the call that consumes it: nothing that can raise stands between the two (a
                # pointer left on the handle would be read by the next evaluation, possibly after U_first was freed)
                _lib.check(h, L.admp_set_dipole_source(h, P(U_first)), 'admp_set_dipole_source')
        rc = L.admp_pme_energy_grad(h, P(pos), boxa, P(Q), P(pol_t), P(th_t), ns, mSa, pS, dS, P(U), maxiter, thresh, E,
                                    P(grad), P(dQ), self._out_ncyc_ref, self._out_conv_ref, 1)
        _lib.check(h, rc, 'admp_pme_energy_grad')
        self.energy_parts = tuple(E)        # (real, recip, self, penalty)
        out = {'E': np.float64(E[0] + E[1] + E[2] + E[3])}
        if self.lpol:
            out['U'] = U
            out['flag'] = bool(conv.value)
            out['i'] = int(ncyc.value)
        if want_grad:
            out['grad'] = grad
        if want_dQ:
            out['dQ'] = dQ[:, :(self.lmax + 1) ** 2]
        return out

    # ---- bare calculators with the dipoles as an explicit input (admp/pme.py:69-78) ---------------------------------
    def _at_U(self, positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales, dScales,
              want_pos=False, want_U=False):
        if not self.lpol:
            raise RuntimeError('energy_fn / grad_U_fn / grad_pos_fn exist only for lpol=True (admp/pme.py:67-78)')
        self._use_current_stream()
        L, h, na = self._L, self._h, self.n_atoms
        self.set_pairs(pairs)
        pos = self._real(positions, (na, 3))
        Q = self._pad_Q(Q_local)
        U = self._real(Uind_global, (na, 3))
        pol_t, th_t = self._real(pol, (na,)), self._real(tholes, (na,))
        boxa, _ = self._harr('box', box, 9)
        mSa, ns = self._harr('mS', mScales)
        pS, _ = self._harr('pS', pScales, ns)
        gpos = torch.empty((na, 3), dtype=self._dtype, device=self._device) if want_pos else None
        gU = torch.empty((na, 3), dtype=self._dtype, device=self._device) if want_U else None
        E = self._out_E
        P = self._ptr
        _lib.check(h, L.admp_pme_energy_fixed_dipoles(h, P(pos), boxa, P(Q), P(pol_t), P(th_t), ns, mSa, pS, P(U), E, P(gpos), P(gU),
                                             None), 'admp_pme_energy_fixed_dipoles')
        self.energy_parts = tuple(E)
        return np.float64(E[0] + E[1] + E[2] + E[3]), gpos, gU

    def _energy_fn(self, positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales, dScales):
        """E at the dipoles given (no SCF), admp/pme.py:69-75."""
        return self._at_U(positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales, dScales)[0]

    def _grad_U_fn(self, positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales, dScales):
        """dE/dUind_global (Na,3) -- the "field" of the SCF loop: grad(energy_fn, argnums=4), admp/pme.py:77."""
        r = self._at_U(positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales, dScales, want_U=True)
        return self._like(r[2], positions)

    def _grad_pos_fn(self, positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales, dScales):
        """dE/dpositions (Na,3) at fixed dipoles: grad(energy_fn, argnums=0), admp/pme.py:78."""
        r = self._at_U(positions, box, pairs, Q_local, Uind_global, pol, tholes, mScales, pScales, dScales, want_pos=True)
        return self._like(r[1], positions)

    def _construct_local_frames(self, positions, box):
        """(Na,3,3) local frames, rows x, y, z (admp/spatial.py:76-142) -- a diagnostic: the hot path builds the frames
        inside its first kernel and never materialises them."""
        self._use_current_stream()
        pos = self._real(positions, (self.n_atoms, 3))
        out = torch.empty((self.n_atoms, 3, 3), dtype=self._dtype, device=self._device)
        boxa, _ = self._harr('box', box, 9)
        _lib.check(self._h, self._L.admp_local_frames(self._h, self._ptr(pos), boxa, self._ptr(out)), 'admp_local_frames')
        return self._like(out, positions)

    def _pme_recip(self, positions, box, Q):
        """Reciprocal-space energy of GLOBAL multipoles Q (Na, (lmax+1)^2) (generate_pme_recip, admp/recip.py:21-431, with
        gamma=False): evaluated by a sibling handle without local frames and with an empty pair list."""
        b = getattr(self, '_bare', None)
        if b is None:
            b = self._bare = HipForceBase(self.n_atoms, None, None, None, self._dev_index)
            b._key = None
        key = (float(self.kappa), int(self.K1), int(self.K2), int(self.K3), int(self.lmax), self._ref_korder)
        if b._key != key:
            _lib.check(b._h, b._L.admp_set_ewald(b._h, key[0], key[1], key[2], key[3], key[4], 0), 'admp_set_ewald')
            _lib.check(b._h, b._L.admp_set_option(b._h, _lib.OPT_REFERENCE_KPOINTS, int(key[5])), 'admp_set_option')
            _lib.check(b._h, b._L.admp_set_pairs(b._h, 0, None, 1), 'admp_set_pairs')
            b._key = key
        b._use_current_stream()
        pos = self._real(positions, (self.n_atoms, 3))
        Qp = self._pad_Q(Q)
        E = (ctypes.c_double * 4)()
        one = (ctypes.c_double * 1)(1.0)
        boxa, _ = self._harr('box', box, 9)
        _lib.check(b._h, b._L.admp_pme_energy_grad(b._h, self._ptr(pos), boxa, self._ptr(Qp), None, None, 1, one, None, None,
                                                   None, 1, 0.0, E, None, None, None, None, 1), 'admp_pme_energy_grad')
        return np.float64(E[1])

    def generate_get_energy(self):
        if not self.lpol:
            def get_energy(positions, box, pairs, Q_local, mScales):
                return self._evaluate(positions, box, pairs, Q_local, mScales, want_grad=False)['E']
            return get_energy

        def get_energy(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales, U_init=None):
            # the reference binds U_init to the zeros created with the closure (admp/pme.py:79-81): a call
            # without U_init always starts the SCF from zero
            r = self._evaluate(positions, box, pairs, Q_local, mScales, pol, tholes, pScales, dScales, U_init,
                               want_grad=False)
            self._store_scf(r, positions)
            return r['E']
        return get_energy

    def _store_scf(self, r, like):
        self.U_ind = self._like(r['U'], like)
        self.lconverg = r['flag']
        self.n_cycle = r['i']
        self._warn_unconverged(r)

    def _warn_unconverged(self, r):
        """The reference returns `lconverg = False` silently (admp/pme.py:139-143); a drop-in user whose SCF diverges -- e.g.
        the literal k-point order on an unequal mesh, where the reference's own iteration has a growing mode -- gets told."""
        if not r['flag']:
            warnings.warn('induced-dipole SCF did not converge within %d cycles (lconverg = False); energies and gradients '
                          'are those of the unconverged dipoles, as in the reference' % (r['i'] + 1))

    def _generate_get_forces(self):
        if not self.lpol:
            def get_forces(positions, box, pairs, Q_local, mScales):
                r = self._evaluate(positions, box, pairs, Q_local, mScales)
                return r['E'], self._like(r['grad'], positions)
            return get_forces

        def get_forces(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales, U_init=None):
            r = self._evaluate(positions, box, pairs, Q_local, mScales, pol, tholes, pScales, dScales, U_init)
            self._store_scf(r, positions)
            return r['E'], self._like(r['grad'], positions)
        return get_forces

    def get_forces_and_dQ(self, positions, box, pairs, Q_local, *rest, **kw):
        """(E, dE/dpositions, dE/dQ_local): the parameter gradient the reference exposes through
        jax.grad(..., argnums=3) (examples/openmm_api/run.py:41-46)."""
        if self.lpol:
            pol, tholes, mScales, pScales, dScales = rest
            r = self._evaluate(positions, box, pairs, Q_local, mScales, pol, tholes, pScales, dScales,
                               kw.get('U_init'), want_dQ=True)
            self._store_scf(r, positions)
        else:
            (mScales,) = rest
            r = self._evaluate(positions, box, pairs, Q_local, mScales, want_dQ=True)
        return r['E'], self._like(r['grad'], positions), self._like(r['dQ'], positions)

    def get_pol_thole_gradients(self, positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales, U_init=None):
        """(dE/dpol, dE/dtholes), each (Na,): the 'pol' and 'tholes' entries of `grad(pot_pme, argnums=3)` in the reference.
        The induced dipoles are converged first (like `get_energy`); at the SCF solution dE/dU = 0, so the parameter
        gradient is the partial derivative at fixed dipoles: the Thole-damping part of every pair (admp_thole_sums) plus
        the polarization penalty D |U|^2 / (2 alpha) (admp/pme.py:760-774).  Sites with pol <= 1e-8 get 0 for dE/dpol."""
        if not self.lpol:
            raise RuntimeError('get_pol_thole_gradients needs lpol=True')
        self.get_energy(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales, U_init=U_init)
        with self._on_stream():
            na = self.n_atoms
            pos = self._real(positions, (na, 3))
            Q = self._pad_Q(Q_local)
            pol_t, th_t = self._real(pol, (na,)), self._real(tholes, (na,))
            U = self._real(self.U_ind, (na, 3))
            mS = self._host64(mScales)
            pS = self._host64(pScales, len(mS))
            sx = torch.zeros(na, dtype=self._dtype, device=self._device)      # (a slab rank fills its home rows only)
            sw = torch.zeros(na, dtype=self._dtype, device=self._device)
            rc = self._L.admp_thole_sums(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(Q),
                                         self._ptr(pol_t), self._ptr(th_t), len(mS), _lib.darr(mS), _lib.darr(pS),
                                         self._ptr(U), self._ptr(sx), self._ptr(sw))
            _lib.check(self._h, rc, 'admp_thole_sums')
            live = pol_t > 1e-8
            safe = torch.where(live, pol_t, torch.ones_like(pol_t))
            dpol = torch.where(live, -sx / (6.0 * safe) - DIELECTRIC * (U * U).sum(dim=1) / (2.0 * safe * safe),
                               torch.zeros_like(pol_t))
        return self._like(dpol, positions), self._like(sw, positions)

    def get_box_gradient(self, positions, box, pairs, Q_local, *rest, **kw):
        """dE/dbox (3,3) at fixed Cartesian positions -- `grad(get_energy, argnums=1)` of the reference, the quantity its
        callers turn into the virial (README.md:7).  Arguments as get_energy.  Polarizable: the dipoles are converged first
        and the derivative is taken at fixed dipoles, like every gradient of the reference (admp/pme.py:81-85)."""
        return self.get_energy_and_box_gradient(positions, box, pairs, Q_local, *rest, **kw)[1]

    def get_energy_and_box_gradient(self, positions, box, pairs, Q_local, *rest, **kw):
        na = self.n_atoms
        if self.lpol:
            pol, tholes, mScales, pScales, dScales = rest
            self.get_energy(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales,
                            U_init=kw.get('U_init'))
        else:
            (mScales,) = rest
            pol = tholes = pScales = None
        self._use_current_stream()
        self.set_pairs(pairs)
        pos = self._real(positions, (na, 3))
        Q = self._pad_Q(Q_local)
        boxa, _ = self._harr('box', box, 9)
        mSa, ns = self._harr('mS', mScales)
        pol_t = th_t = U = pS = None
        if self.lpol:
            pol_t, th_t = self._real(pol, (na,)), self._real(tholes, (na,))
            pS, _ = self._harr('pS', pScales, ns)
            U = self._real(self.U_ind, (na, 3))
        E = self._out_E
        dbox = (ctypes.c_double * 9)()
        P = self._ptr
        _lib.check(self._h, self._L.admp_pme_box_grad(self._h, P(pos), boxa, P(Q), P(pol_t), P(th_t), ns, mSa, pS, P(U), E,
                                                      dbox), 'admp_pme_box_grad')
        self.energy_parts = tuple(E)
        return np.float64(E[0] + E[1] + E[2] + E[3]), np.array(dbox[:], dtype=np.float64).reshape(3, 3)

    def get_pscale_gradient(self, positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales, U_init=None):
        """dE/dpScales (len(pScales),): the 'pScales' entry of `grad(pot_pme, argnums=3)` in the reference.  The dipoles are
        converged first and held fixed (admp/pme.py:81-85).  pscale scales the Thole factor of the permanent-induced
        coefficients (admp/pme.py:455-470); the Fermi switch it also drives (pme.py:411) is flat wherever its derivative is
        finite -- the reference's autodiff returns NaN there for pscale > ~0.008, this returns the analytic limit.
        dE/ddScales is identically zero (the reference ignores dScales: uscales = 1, pme.py:472)."""
        if not self.lpol:
            raise RuntimeError('get_pscale_gradient needs lpol=True')
        self.get_energy(positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales, U_init=U_init)
        self._use_current_stream()
        na = self.n_atoms
        pos = self._real(positions, (na, 3))
        Q = self._pad_Q(Q_local)
        mS = self._host64(mScales)
        pS = self._host64(pScales, len(mS))
        out = (ctypes.c_double * len(mS))()
        pol_t, th_t, U = self._real(pol, (na,)), self._real(tholes, (na,)), self._real(self.U_ind, (na, 3))   # kept alive
        rc = self._L.admp_pscale_grad(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(Q),
                                      self._ptr(pol_t), self._ptr(th_t), len(mS), _lib.darr(mS), _lib.darr(pS), self._ptr(U),
                                      out)
        _lib.check(self._h, rc, 'admp_pscale_grad')
        return np.array(out[:], dtype=np.float64)

    def get_mscale_gradient(self, positions, box, pairs, Q_local, mScales):
        """dE/dmScales, shape (len(mScales),): what `grad(pot_pme, argnums=3)(...)['mScales']` gives in the reference
        (examples/openmm_api/run.py:44-46).  The energy is linear in mScales and its induced part carries pScales, so neither
        the values of mScales nor the induced dipoles enter."""
        return self._mscale_gradient(0, positions, box, pairs, self._pad_Q(Q_local), 9, len(self._host64(mScales)))

    def scf_stats(self, reset=False):
        """How the polarizable calls were enqueued and what wrong guesses cost (admp_scf_stats): dict of counters."""
        out = (ctypes.c_int64 * 8)()
        _lib.check(self._h, self._L.admp_scf_stats(self._h, out, 1 if reset else 0), 'admp_scf_stats')
        keys = ('plain', 'speculative', 'speculative_failed', 'chained', 'chained_too_short', 'chained_too_long',
                'wasted_increments', 'jacobi_steps')
        return dict(zip(keys, (int(v) for v in out)))

    def optimize_Uind(self, positions, box, pairs, Q_local, pol, tholes, mScales, pScales, dScales, U_init=None,
                      maxiter=None, thresh=None):
        """Jacobi SCF of the induced dipoles; returns (U, converged, i) like admp/pme.py:111-143."""
        if not self.lpol:
            raise RuntimeError('optimize_Uind needs lpol=True')
        r = self._evaluate(positions, box, pairs, Q_local, mScales, pol, tholes, pScales, dScales, U_init,
                           want_grad=False, maxiter=maxiter, thresh=thresh)
        self._warn_unconverged(r)
        return self._like(r['U'], positions), r['flag'], r['i']
