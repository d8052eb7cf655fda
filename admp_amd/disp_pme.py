"""ADMPDispPmeForce -- counterpart of the reference's admp/disp_pme.py:20-77 on MI355X.

    disp = ADMPDispPmeForce(box, covalent_map, rc, ethresh, pmax)
    disp.update_env('kappa', 0.657065221219616)
    E, G = disp.get_forces(positions, box, pairs, c_list, mScales)      # c_list (Na, (pmax-4)/2): C6[, C8[, C10]]

Real-space (m + g_p - 1) c_i c_j / r^p pair sum, one scalar PME reciprocal pass per power with the
Ck_6/8/10 kernels (gamma point included), and the self term (admp/disp_pme.py:80-279).
"""
import ctypes

import numpy as np
import torch

from . import _lib, settings
from ._device import HipForceBase
from .pme import setup_ewald_parameters


class ADMPDispPmeForce(HipForceBase):
    def __init__(self, box, covalent_map, rc, ethresh, pmax, device=None):
        self.covalent_map = covalent_map
        self.rc = rc
        self.ethresh = ethresh
        self.pmax = int(pmax)
        if self.pmax not in (6, 8, 10):
            raise ValueError('pmax must be 6, 8 or 10')
        kappa, K1, K2, K3 = setup_ewald_parameters(rc, ethresh, box)
        self.kappa = kappa
        self.K1, self.K2, self.K3 = K1, K2, K3
        self.pme_order = 6
        super().__init__(int(covalent_map.shape[0]), covalent_map, None, None, device)
        self.energy_parts = None
        self.refresh_calculators()

    def update_env(self, attr, val):
        setattr(self, attr, val)
        self.refresh_calculators()

    def refresh_calculators(self):
        _lib.check(self._h, self._L.admp_set_ewald(self._h, float(self.kappa), int(self.K1), int(self.K2), int(self.K3),
                                                   0, 0), 'admp_set_ewald')
        _lib.check(self._h, self._L.admp_set_option(self._h, _lib.OPT_REFERENCE_KPOINTS,
                                                    int(bool(settings.REFERENCE_KPOINT_ORDER))), 'admp_set_option')
        self.get_energy = self.generate_get_energy()
        self.get_forces = self._generate_get_forces()
        self.get_energy._value_and_grad = self.get_forces          # value_and_grad(disp.get_energy) -> get_forces
        self.get_energy._value_and_box_grad = self.get_energy_and_box_gradient

    def _evaluate(self, positions, box, pairs, c_list, mScales, want_grad):
        with self._on_stream():
            return self._evaluate_on_stream(positions, box, pairs, c_list, mScales, want_grad)

    def _evaluate_on_stream(self, positions, box, pairs, c_list, mScales, want_grad):
        na = self.n_atoms
        self.set_pairs(pairs)
        pos = self._real(positions, (na, 3))
        c3 = self._packed_c(c_list)
        mS = self._host64(mScales)
        E = (ctypes.c_double * 3)()
        grad = torch.empty((na, 3), dtype=self._dtype, device=self._device) if want_grad else None
        rc = self._L.admp_disp_energy_grad(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(c3),
                                           self.pmax, len(mS), _lib.darr(mS), E, self._ptr(grad), 1)
        _lib.check(self._h, rc, 'admp_disp_energy_grad')
        self.energy_parts = tuple(E)
        return np.float64(E[0] + E[1] + E[2]), grad

    def _packed_c(self, c_list):
        """(Na, 3) device rows (C6, C8, C10; unused powers zero) in the handle's precision.  Packing is a fill + a copy per call:
        the packed tensor is kept while c_list is the same torch tensor, not written since (as pairwise._packed_params does)."""
        na = self.n_atoms
        nc = (self.pmax - 4) // 2
        key = None
        if isinstance(c_list, torch.Tensor):
            key = (id(c_list), c_list.data_ptr(), c_list._version)
            hit = getattr(self, '_c_cache', None)
            if hit is not None and hit[0] == key:
                self._type_misses = 0
                return hit[1]
            c = c_list.detach().to(device=self._device, dtype=self._dtype)
        else:
            c = torch.as_tensor(np.asarray(c_list, dtype=np.float64), dtype=self._dtype).to(self._device)
        if c.dim() != 2 or c.shape[0] != na or c.shape[1] < nc:
            raise ValueError('c_list must be (Na, >= (pmax-4)/2)')
        c3 = torch.zeros((na, 3), dtype=self._dtype, device=self._device)
        c3[:, :nc] = c[:, :nc]
        self._c_cache = (key, c3, c_list) if key is not None else None      # (the tensor stays alive: its id stays unique)
        self._update_types(c3 if key is not None else None)
        return c3

    def _update_types(self, c3):
        """Atom types for the library's typed meshes (include/admp_hip.h admp_disp_set_types): when the rows of c_list take at
        most 3 distinct values the reciprocal part keeps one mesh per type instead of one per power.  Found once per c_list
        tensor (torch.unique on the device: milliseconds at 1M atoms, so only for a tensor that is kept from call to call --
        a caller who hands in new coefficients every call, e.g. while fitting them, gets the per-power path)."""
        misses = getattr(self, '_type_misses', 0)
        types = None
        if c3 is not None and misses < 3 and settings.DISP_TYPED_MESHES:
            rows, inv = torch.unique(c3, dim=0, return_inverse=True)
            if rows.shape[0] <= 3:
                types = (inv.to(torch.int32).contiguous(), rows.double().cpu().numpy())
        self._type_misses = misses + 1          # (reset by every call that finds its tensor in the cache: _evaluate_on_stream)
        if types is None:
            if getattr(self, '_types', None) is not None:
                _lib.check(self._h, self._L.admp_disp_set_types(self._h, 0, None, None), 'admp_disp_set_types')
            self._types = None
            return
        self._types = types                      # (the library reads the index array of every later call: keep it alive)
        _lib.check(self._h, self._L.admp_disp_set_types(self._h, int(types[1].shape[0]), self._ptr(types[0]),
                                                         _lib.darr(types[1])), 'admp_disp_set_types')

    def get_energy_and_box_gradient(self, positions, box, pairs, c_list, mScales):
        """(E, dE/dbox (3,3)) at fixed Cartesian positions: `value_and_grad(get_energy, argnums=1)` of the reference."""
        with self._on_stream():
            na = self.n_atoms
            self.set_pairs(pairs)
            pos = self._real(positions, (na, 3))
            c3 = self._packed_c(c_list)
            mS = self._host64(mScales)
            E = (ctypes.c_double * 3)()
            dbox = (ctypes.c_double * 9)()
            rc = self._L.admp_disp_box_grad(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(c3),
                                            self.pmax, len(mS), _lib.darr(mS), E, dbox)
            _lib.check(self._h, rc, 'admp_disp_box_grad')
        self.energy_parts = tuple(E)
        return np.float64(E[0] + E[1] + E[2]), np.array(dbox[:], dtype=np.float64).reshape(3, 3)

    def get_box_gradient(self, positions, box, pairs, c_list, mScales):
        return self.get_energy_and_box_gradient(positions, box, pairs, c_list, mScales)[1]

    def get_mscale_gradient(self, positions, box, pairs, c_list, mScales):
        """dE/dmScales (len(mScales),): `grad(pot_disp, argnums=3)(...)['mScales']` of the reference
        (examples/openmm_api/run.py:41-43)."""
        nc = (self.pmax - 4) // 2
        c = np.zeros((self.n_atoms, 3)) if not isinstance(c_list, torch.Tensor) else None
        if c is None:
            c = torch.zeros((self.n_atoms, 3), dtype=self._dtype, device=self._device)
            c[:, :nc] = c_list.detach().to(device=self._device, dtype=self._dtype)[:, :nc]
        else:
            c[:, :nc] = np.asarray(c_list, dtype=np.float64)[:, :nc]
        return self._mscale_gradient(1, positions, box, pairs, c, 3, len(self._host64(mScales)), self.pmax)

    def get_param_gradient(self, positions, box, pairs, c_list, mScales):
        """dE/dc_list (Na, (pmax-4)/2): the per-atom form of the 'C6' / 'C8' / 'C10' entries of `grad(pot_disp, argnums=3)` in
        the reference (examples/openmm_api/run.py:41-43, admp/api.py:183-199) -- real-space pair sums, the mesh potential of
        every channel at the atoms, and the self term (admp_disp_param_grad)."""
        with self._on_stream():
            na = self.n_atoms
            self.set_pairs(pairs)
            pos = self._real(positions, (na, 3))
            c3 = self._packed_c(c_list)
            mS = self._host64(mScales)
            out = torch.empty((na, 3), dtype=self._dtype, device=self._device)
            rc = self._L.admp_disp_param_grad(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(c3), self.pmax,
                                              len(mS), _lib.darr(mS), self._ptr(out))
            _lib.check(self._h, rc, 'admp_disp_param_grad')
        return self._like(out[:, :(self.pmax - 4) // 2], positions)

    def generate_get_energy(self):
        def get_energy(positions, box, pairs, c_list, mScales):
            return self._evaluate(positions, box, pairs, c_list, mScales, False)[0]
        return get_energy

    def _generate_get_forces(self):
        def get_forces(positions, box, pairs, c_list, mScales):
            e, g = self._evaluate(positions, box, pairs, c_list, mScales, True)
            return e, self._like(g, positions)
        return get_forces
