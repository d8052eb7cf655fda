"""MD-driver helpers on the GPU (the reference has no integrator; its drivers stop at get_forces).

`HarmonicBonded` evaluates the bonded terms of the drivers' force field (examples/*/mpidwater.xml:16-21: harmonic O-H bonds
and H-O-H angles) with one HIP kernel over explicit lists (include/admp_hip.h admp_md_bonded); `VelocityVerlet` does the two
half steps with one kernel each (admp_md_kick_drift).  Energies accumulate in device words and are read only when the caller
logs, so an MD step adds no host synchronisation of its own to those of the calculators.  Used by examples/md/nve_water.py.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._device import HipForceBase


class HarmonicBonded(HipForceBase):
    """E = sum_bonds k/2 (r - r0)^2 + sum_angles k/2 (theta - theta0)^2 with minimum-image vectors.
    bonds (nb, 2) int, bond_par (nb, 2) = (k, r0); angles (na, 3) int = (i, centre, k), angle_par (na, 2) = (k, theta0 / rad)."""

    def __init__(self, n_atoms, bonds, bond_par, angles, angle_par, device=None):
        super().__init__(n_atoms, None, None, None, device)

        def ints(x, w):
            a = np.ascontiguousarray(np.asarray(x, dtype=np.int32).reshape(-1, w))
            return torch.as_tensor(a).to(self._device)
        self._bidx, self._aidx = ints(bonds, 2), ints(angles, 3)
        self._bpar = self._real(np.asarray(bond_par, dtype=np.float64).reshape(-1, 2))
        self._apar = self._real(np.asarray(angle_par, dtype=np.float64).reshape(-1, 2))
        if len(self._bidx) != len(self._bpar) or len(self._aidx) != len(self._apar):
            raise ValueError('index and parameter lists differ in length')
        for idx in (self._bidx, self._aidx):
            if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.n_atoms):
                raise ValueError('atom index out of range')
        self.energy_words = torch.zeros(2, dtype=torch.float64, device=self._device)      # (bonds, angles), accumulated

    def add_forces(self, positions, box, grad):
        """grad (Na,3 device tensor of the handle's precision) += dE/dpositions; the energies are added to `energy_words`
        (zero them with reset_energy(); read them with energy())."""
        self._use_current_stream()
        pos = self._real(positions, (self.n_atoms, 3))
        if not (isinstance(grad, torch.Tensor) and grad.is_cuda and grad.dtype == self._dtype and grad.is_contiguous()):
            raise ValueError('grad must be a contiguous device tensor of the handle\'s precision')
        boxa, _ = self._harr('box', box, 9)
        P = self._ptr
        _lib.check(self._h, self._L.admp_md_bonded(self._h, P(pos), boxa, len(self._bidx), P(self._bidx), P(self._bpar),
                                                   len(self._aidx), P(self._aidx), P(self._apar), P(self.energy_words),
                                                   P(grad)), 'admp_md_bonded')
        return grad

    def reset_energy(self):
        self.energy_words.zero_()

    def energy(self):
        """(host read) sum of the accumulated words"""
        return float(self.energy_words.sum())

    def get_forces(self, positions, box):
        """(E, dE/dpositions) of one evaluation -- the calculators' convention (one host read)"""
        self.reset_energy()
        g = torch.zeros((self.n_atoms, 3), dtype=self._dtype, device=self._device)
        self.add_forces(positions, box, g)
        return np.float64(self.energy()), self._like(g, positions)


class VelocityVerlet:
    """r, v in A and A/fs, gradients in kJ/mol/A, masses in amu: v -= (dt/2) 1e-4 grad / m; r += dt v (1 kJ/mol/A/amu = 1e-4
    A/fs^2).  Both arrays are updated in place by one kernel per half step (the handle of any calculator lends its stream)."""
    ACC = 1e-4

    def __init__(self, handle_owner, masses, dt_fs):
        self._o = handle_owner
        self.dt = float(dt_fs)
        self.inv_mass = (1.0 / handle_owner._real(np.asarray(masses, dtype=np.float64).reshape(-1))).contiguous()
        self.ekin_word = torch.zeros(1, dtype=torch.float64, device=handle_owner._device)

    def _call(self, pos, vel, grad, drift, want_ekin):
        o = self._o
        o._use_current_stream()
        if want_ekin:
            self.ekin_word.zero_()
        P = o._ptr
        _lib.check(o._h, o._L.admp_md_kick_drift(o._h, vel.shape[0], P(pos), P(vel), P(grad), P(self.inv_mass),
                                                 0.5 * self.dt * self.ACC, self.dt if drift else 0.0,
                                                 P(self.ekin_word) if want_ekin else None), 'admp_md_kick_drift')

    def kick_drift(self, pos, vel, grad):
        """first half: v(t + dt/2), r(t + dt)"""
        self._call(pos, vel, grad, True, False)

    def kick(self, pos, vel, grad, want_ekin=False):
        """second half: v(t + dt); want_ekin: the kinetic energy (kJ/mol) of the new velocities lands in ekin_word"""
        self._call(pos, vel, grad, False, want_ekin)

    def kinetic_energy(self):
        """(host read) sum m v^2 / 2 of the last kick(want_ekin=True), in kJ/mol"""
        return float(self.ekin_word[0]) / self.ACC
