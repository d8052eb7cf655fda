"""Pairwise short-range interactions -- counterpart of the reference's admp/pairwise.py:45-113.

The reference builds a calculator from an arbitrary JAX kernel:

    pot = generate_pairwise_interaction(TT_damping_qq_c6_kernel, covalent_map, static_args={})
    E, G = value_and_grad(pot)(positions, box, pairs, mScales, a_list, b_list, q_list, c_list)

Two kinds of kernel are accepted here:
  * the one the reference ships, `TT_damping_qq_c6_kernel`: a hand-written HIP kernel of libadmp_hip;
  * ANY Python function `kernel(dr, m, p1i, p1j, p2i, p2j, ...)` written against `admp_amd.xp` (the stand-in for
    jax.numpy: sqrt, exp, erf, ..., where): it is traced once into HIP source, compiled for gfx950 at run time with hiprtc
    and run as a row-per-atom pair kernel; the gradient comes from forward-mode dual numbers in the generated code.
`value_and_grad` below plays the role of jax.value_and_grad for these calculators (argnums 0 = positions, 1 = box).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._device import HipForceBase


class _HipPairKernel:
    def __init__(self, name, n_params):
        self.name = name
        self.n_params = n_params

    def __repr__(self):
        return '<HIP pair kernel %s>' % self.name


#: Tang-Toennies damped exchange / charge-penetration / C6 kernel (admp/pairwise.py:94-113);
#: atomic parameters in the reference's order: a (Hartree), b (Bohr^-1), q (e), c6.
TT_damping_qq_c6_kernel = _HipPairKernel('tt_damping_qq_c6', 4)


class _PairInteraction(HipForceBase):
    def __init__(self, kernel, covalent_map, static_args):
        super().__init__(int(covalent_map.shape[0]), covalent_map, None, None, None)
        self.kernel = kernel
        self.static_args = dict(static_args or {})

    def _evaluate(self, positions, box, pairs, mScales, atomic_params, want_grad):
        with self._on_stream():
            return self._evaluate_on_stream(positions, box, pairs, mScales, atomic_params, want_grad)

    def _evaluate_on_stream(self, positions, box, pairs, mScales, atomic_params, want_grad):
        na = self.n_atoms
        if len(atomic_params) != self.kernel.n_params:
            raise TypeError('%s takes %d atomic parameter lists' % (self.kernel.name, self.kernel.n_params))
        self.set_pairs(pairs)
        pos = self._real(positions, (na, 3))
        par = self._packed_params(atomic_params)
        mS = self._host64(mScales)
        E = (ctypes.c_double * 1)()
        grad = torch.empty((na, 3), dtype=self._dtype, device=self._device) if want_grad else None
        rc = self._L.admp_tt_energy_grad(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(par),
                                         len(mS), _lib.darr(mS), E, self._ptr(grad), 1)
        _lib.check(self._h, rc, 'admp_tt_energy_grad')
        return np.float64(E[0]), grad

    def _packed_params(self, atomic_params):
        """(Na, n_params) device rows of the per-atom parameter lists.  Repacking them is a stack + copy per call; the
        packed tensor is kept while every list is the same torch tensor, not written since (same test as the pair list)."""
        key = None
        if all(isinstance(p, torch.Tensor) for p in atomic_params):
            key = tuple((id(p), p.data_ptr(), p._version) for p in atomic_params)
            c = getattr(self, '_par_cache', None)
            if c is not None and c[0] == key:
                return c[1]
        par = torch.stack([self._real(p, (self.n_atoms,)) for p in atomic_params], dim=1).contiguous()
        self._par_cache = (key, par, tuple(atomic_params)) if key is not None else None    # (the lists stay alive: ids unique)
        return par

    def get_mscale_gradient(self, positions, box, pairs, mScales, *atomic_params):
        """dE/dmScales (len(mScales),) of this pair interaction."""
        if len(atomic_params) != self.kernel.n_params:
            raise TypeError('%s takes %d atomic parameter lists' % (self.kernel.name, self.kernel.n_params))
        par = torch.stack([self._real(p, (self.n_atoms,)) for p in atomic_params], dim=1).contiguous()
        return self._mscale_gradient(2, positions, box, pairs, par, self.kernel.n_params, len(self._host64(mScales)))

    def get_param_gradient(self, positions, box, pairs, mScales, *atomic_params):
        """(dE/dp1, dE/dp2, ...), each (Na,): the derivative of the pair energy with respect to the per-atom parameter lists
        -- what `grad(pot_disp, argnums=3)` chains to the 'A' / 'B' / 'Q' / 'C6' tables in the reference
        (admp/api.py:183-199).  Tang-Toennies: a zero a_i or b_i gets 0 (no finite derivative of sqrt(a_i a_j) there)."""
        if len(atomic_params) != self.kernel.n_params:
            raise TypeError('%s takes %d atomic parameter lists' % (self.kernel.name, self.kernel.n_params))
        with self._on_stream():
            na = self.n_atoms
            self.set_pairs(pairs)
            pos = self._real(positions, (na, 3))
            par = self._packed_params(atomic_params)
            mS = self._host64(mScales)
            out = torch.empty((na, self.kernel.n_params), dtype=self._dtype, device=self._device)
            rc = self._L.admp_tt_param_grad(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(par), len(mS),
                                            _lib.darr(mS), self._ptr(out))
            _lib.check(self._h, rc, 'admp_tt_param_grad')
        return tuple(self._like(out[:, k].contiguous(), positions) for k in range(self.kernel.n_params))

    def get_energy_and_box_gradient(self, positions, box, pairs, mScales, *atomic_params):
        """(E, dE/dbox (3,3)) at fixed Cartesian positions: value_and_grad(pair_int, argnums=1) of the reference."""
        with self._on_stream():
            na = self.n_atoms
            if len(atomic_params) != self.kernel.n_params:
                raise TypeError('%s takes %d atomic parameter lists' % (self.kernel.name, self.kernel.n_params))
            self.set_pairs(pairs)
            pos = self._real(positions, (na, 3))
            par = torch.stack([self._real(p, (na,)) for p in atomic_params], dim=1).contiguous()
            mS = self._host64(mScales)
            E = (ctypes.c_double * 1)()
            dbox = (ctypes.c_double * 9)()
            rc = self._L.admp_tt_box_grad(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(par), len(mS),
                                          _lib.darr(mS), E, dbox)
            _lib.check(self._h, rc, 'admp_tt_box_grad')
        return np.float64(E[0]), np.array(dbox[:], dtype=np.float64).reshape(3, 3)

    def get_box_gradient(self, positions, box, pairs, mScales, *atomic_params):
        return self.get_energy_and_box_gradient(positions, box, pairs, mScales, *atomic_params)[1]

    def __call__(self, positions, box, pairs, mScales, *atomic_params):
        return self._evaluate(positions, box, pairs, mScales, atomic_params, False)[0]

    def value_and_grad(self, positions, box, pairs, mScales, *atomic_params):
        e, g = self._evaluate(positions, box, pairs, mScales, atomic_params, True)
        return e, self._like(g, positions)


class _TracedPairInteraction(_PairInteraction):
    """generate_pairwise_interaction for a user-supplied Python kernel: traced (admp_amd/xp.py) and compiled (hiprtc) at
    the first call, when the number of atomic parameter lists is known."""

    def __init__(self, fn, covalent_map, static_args):
        HipForceBase.__init__(self, int(covalent_map.shape[0]), covalent_map, None, None, None)
        self.kernel = fn
        self.static_args = dict(static_args or {})
        self._programs = {}                      # n_params -> program id
        self.source = None

    def _program(self, n_params):
        if n_params not in self._programs:
            from . import xp
            self.source = xp.generate_source(self.kernel, n_params)
            pid = ctypes.c_int(-1)
            rc = self._L.admp_pair_program_build(self._h, self.source.encode(), n_params, ctypes.byref(pid))
            _lib.check(self._h, rc, 'admp_pair_program_build')
            self._programs[n_params] = pid.value
        return self._programs[n_params]

    def _evaluate_on_stream(self, positions, box, pairs, mScales, atomic_params, want_grad):
        na = self.n_atoms
        pid = self._program(len(atomic_params))
        self.set_pairs(pairs)
        pos = self._real(positions, (na, 3))
        par = (torch.stack([self._real(p, (na,)) for p in atomic_params], dim=1).contiguous() if atomic_params
               else None)
        mS = self._host64(mScales)
        E = (ctypes.c_double * 1)()
        grad = torch.empty((na, 3), dtype=self._dtype, device=self._device) if want_grad else None
        rc = self._L.admp_pair_program_energy_grad(self._h, pid, self._ptr(pos), _lib.darr(self._host64(box, 9)),
                                                   self._ptr(par), len(mS), _lib.darr(mS), E, self._ptr(grad), 1)
        _lib.check(self._h, rc, 'admp_pair_program_energy_grad')
        return np.float64(E[0]), grad

    def get_mscale_gradient(self, *a, **k):
        raise NotImplementedError('dE/dmScales is available for the named kernels only')

    def get_param_gradient(self, *a, **k):
        raise NotImplementedError('parameter gradients are available for the named kernels only')

    def get_energy_and_box_gradient(self, *a, **k):
        raise NotImplementedError('the box gradient is available for the named kernels only')


def generate_pairwise_interaction(pair_int_kernel, covalent_map, static_args):
    """(kernel, covalent_map, static_args) -> pair_int(positions, box, pairs, mScales, *atomic_params)
    (admp/pairwise.py:45-91).  `pair_int_kernel`: TT_damping_qq_c6_kernel, or any Python function
    kernel(dr, m, p1i, p1j, ...) written against admp_amd.xp (see the module docstring)."""
    if isinstance(pair_int_kernel, _HipPairKernel):
        return _PairInteraction(pair_int_kernel, covalent_map, static_args)
    if callable(pair_int_kernel):
        return _TracedPairInteraction(pair_int_kernel, covalent_map, static_args)
    raise TypeError('pair_int_kernel must be a kernel of admp_amd.pairwise or a Python function of (dr, m, *params)')


def value_and_grad(fn, argnums=0):
    """Stand-in for jax.value_and_grad on the calculators of this package: the reference obtains its force routines as
    `value_and_grad(get_energy)` (admp/pme.py:108, admp/disp_pme.py:76, examples/water_1024/run_admp.py:101); here the
    hand-coded adjoint that belongs to a calculator is looked up.  argnums = 0 (positions), 1 (box) or (0, 1)."""
    if isinstance(argnums, (tuple, list)):
        parts = [value_and_grad(fn, a) for a in argnums]

        def both(*a, **k):
            outs = [p(*a, **k) for p in parts]
            return outs[0][0], tuple(o[1] for o in outs)
        return both
    if argnums == 1:
        if isinstance(fn, _PairInteraction):
            return fn.get_energy_and_box_gradient
        vb = getattr(fn, '_value_and_box_grad', None)
        if vb is not None:
            return vb
    elif argnums == 0:
        if isinstance(fn, _PairInteraction):
            return fn.value_and_grad
        vg = getattr(fn, '_value_and_grad', None)
        if vg is not None:
            return vg
    else:
        raise NotImplementedError('value_and_grad: argnums 0 (positions) and 1 (box) are available; parameter gradients: '
                                  'admp_amd.api.param_gradient')
    owner = getattr(fn, '__self__', None)
    raise NotImplementedError('value_and_grad: %r is not a calculator of this package' % (owner or fn))


def grad(fn, argnums=0):
    """Stand-in for jax.grad on the calculators of this package (admp/pme.py:77-78: grad(energy_fn, argnums=4 | 0))."""
    table = getattr(fn, '_grads', None)
    if table is None and getattr(fn, '__func__', None) is not None:
        table = getattr(fn.__func__, '_grads', None)
    if table is not None and argnums in table:
        return table[argnums]
    vg = value_and_grad(fn, argnums)
    return lambda *a, **k: vg(*a, **k)[1]
