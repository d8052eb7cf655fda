"""Device plumbing shared by the force classes: handle lifetime, array staging, pair-list cache.

PyTorch is used only for what it is here for: device memory, streams and (in bench/tests)
torch.distributed.  Every computation is done by libadmp_hip through raw device pointers.
"""
import contextlib
import ctypes

import numpy as np
import torch

from . import _lib, settings


try:
    import xxhash

    def _digest(a):
        return xxhash.xxh3_128_digest(a.tobytes())
except ImportError:      # pragma: no cover
    import hashlib

    def _digest(a):
        return hashlib.blake2b(a.tobytes(), digest_size=16).digest()


def _torch_dtype(nbytes):
    return torch.float32 if nbytes == 4 else torch.float64


def covalent_to_csr(covalent_map, n_atoms):
    """covalent_map (dense ndarray / scipy sparse / None) -> CSR triple of int32 arrays."""
    if covalent_map is None:
        return np.zeros(n_atoms + 1, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)
    if hasattr(covalent_map, 'tocsr'):
        csr = covalent_map.tocsr()
        csr.sort_indices()
        ptr, col, val = csr.indptr, csr.indices, csr.data
    else:
        dense = np.asarray(covalent_map)
        if dense.ndim != 2 or dense.shape[0] != dense.shape[1]:
            raise ValueError('covalent_map must be (Na, Na)')
        rows, col = np.nonzero(dense)
        val = dense[rows, col]
        ptr = np.zeros(dense.shape[0] + 1, dtype=np.int64)
        np.add.at(ptr, rows + 1, 1)
        ptr = np.cumsum(ptr)
    val = np.asarray(val).astype(np.int64)
    if val.size and (val.min() < 0 or val.max() > 15):
        raise ValueError('covalent_map entries must lie in 0..15')
    return (np.ascontiguousarray(ptr, dtype=np.int32), np.ascontiguousarray(col, dtype=np.int32),
            np.ascontiguousarray(val, dtype=np.int32))


# the current stream's handle without building a torch.cuda.Stream object per call (4 us of a 0.2 ms step)
_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)

class HipForceBase:
    """Owns one admp_handle (one GPU, one stream) and the static environment."""

    def __init__(self, n_atoms, covalent_map, axis_type=None, axis_indices=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('admp_amd needs a ROCm GPU (torch.cuda.is_available() is False); '
                               'the PME path has no CPU implementation')
        self._L = _lib.load()
        self._nbytes = settings.precision_bytes()
        self._dtype = _torch_dtype(self._nbytes)
        self._dev_index = torch.cuda.current_device() if device is None else int(device)
        self._device = torch.device('cuda', self._dev_index)
        self._h = ctypes.c_void_p()
        rc = self._L.admp_create(ctypes.byref(self._h), self._dev_index, self._nbytes)
        if rc != 0:
            raise _lib.AdmpHipError('admp_create failed (code %d): %s' % (rc, self._L.admp_last_error(None).decode()))
        self.n_atoms = int(n_atoms)
        ptr, col, val = covalent_to_csr(covalent_map, self.n_atoms)
        if len(ptr) != self.n_atoms + 1:
            raise ValueError('covalent_map size does not match the number of atoms')
        at = None if axis_type is None else np.ascontiguousarray(np.asarray(axis_type), dtype=np.int32)
        ai = None if axis_indices is None else np.ascontiguousarray(np.asarray(axis_indices), dtype=np.int32)
        if at is not None and at.shape != (self.n_atoms,):
            raise ValueError('axis_type must have one entry per atom')
        if ai is not None and ai.shape != (self.n_atoms, 3):
            raise ValueError('axis_indices must be (Na, 3)')

        def p(a):
            return None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(self._h, self._L.admp_set_topology(self._h, self.n_atoms, p(at), p(ai), p(ptr), p(col), p(val)),
                   'admp_set_topology')
        self._pairs_key = None
        self._pairs_keep = None
        self._stream = None
        self._hcache = {}

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                self._L.admp_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---- staging -----------------------------------------------------------------------------------------
    def _use_current_stream(self):
        """Bind the library to the caller's CURRENT torch stream (re-bound only when that changes): the torch ops of a
        call and the HIP kernels of the library share ONE stream, so there is nothing to order across streams -- a
        per-call cross-stream event wait costs ~15 us on the GPU, 8 % of a 3072-atom step.  The legacy default stream
        (handle 0, torch's default) is selected explicitly (admp_use_default_stream): a NULL argument of admp_set_stream
        means "library-owned stream"."""
        cur = _RAW_STREAM(self._device.index) if _RAW_STREAM else torch.cuda.current_stream(self._device).cuda_stream
        if cur != self._stream:
            if cur == 0:
                _lib.check(self._h, self._L.admp_use_default_stream(self._h), 'admp_use_default_stream')
            else:
                _lib.check(self._h, self._L.admp_set_stream(self._h, ctypes.c_void_p(cur)), 'admp_set_stream')
            self._stream = cur

    def _on_stream(self):
        self._use_current_stream()
        return contextlib.nullcontext()

    def _enter_stream(self):
        """per-step form of `with self._on_stream()`; returns the token for _leave_stream (nothing to restore)."""
        self._use_current_stream()
        return None

    @staticmethod
    def _leave_stream(prev):
        pass

    def _real(self, x, shape=None):
        """array-like -> contiguous device tensor of the handle's precision."""
        if isinstance(x, torch.Tensor):
            if x.dtype == self._dtype and x.is_cuda and x.is_contiguous() and not x.requires_grad and \
                    x.device.index == self._device.index:
                t = x                                 # already what the library reads: no new tensor object
            else:
                t = x.detach().to(device=self._device, dtype=self._dtype).contiguous()
        else:
            t = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float64)), dtype=self._dtype).to(self._device)
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError('expected shape %s, got %s' % (tuple(shape), tuple(t.shape)))
        return t

    def _harr(self, slot, x, n=None):
        """host values -> (ctypes double array, count), cached per call-site `slot` while the content is unchanged (the
        scale factors and the box are the same objects step after step; marshalling them anew costs microseconds of a
        0.17 ms step)."""
        if isinstance(x, np.ndarray) and x.dtype == np.float64:
            b = x.tobytes()
        elif isinstance(x, torch.Tensor):
            b = x.detach().cpu().numpy().astype(np.float64).tobytes()
        else:
            b = np.asarray(x, dtype=np.float64).tobytes()
        c = self._hcache.get(slot)
        if c is None or c[0] != b:
            cnt = len(b) // 8
            c = (b, (ctypes.c_double * cnt).from_buffer_copy(b), cnt)
            self._hcache[slot] = c
        if n is not None and c[2] != n:
            raise ValueError('expected %d values, got %d' % (n, c[2]))
        return c[1], c[2]

    @staticmethod
    def _host64(x, n=None):
        a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        if n is not None and a.size != n:
            raise ValueError('expected %d values, got %d' % (n, a.size))
        return a

    @staticmethod
    def _like(result, template):
        """return `result` (device tensor) in the container type of `template`."""
        if isinstance(template, torch.Tensor):
            return result if result.device == template.device else result.to(device=template.device)
        return result.cpu().numpy()

    @staticmethod
    def _ptr(t):
        # (a plain int: ctypes converts it for a c_void_p parameter; building a c_void_p object per pointer costs 0.2 us each)
        return None if t is None else t.data_ptr()

    # ---- pair list -----------------------------------------------------------------------------------------
    def _pairs_fingerprint(self, pairs):
        """Identity of a pair list.  torch: the tensor object + its version counter (bumped by every in-place write).
        numpy has no such counter, so the WHOLE array is hashed (xxh3: ~10 GB/s, 30 us for the 42k pairs of 1024 waters) --
        a caller who refills the same ndarray in place gets a fresh table."""
        if isinstance(pairs, torch.Tensor):
            return ('t', id(pairs), pairs.data_ptr(), tuple(pairs.shape), pairs._version)
        a = np.ascontiguousarray(np.asarray(pairs))
        return ('n', a.shape, a.dtype.str, _digest(a))

    def set_pairs(self, pairs):
        """Compile the (Np, 2) pair list into the device neighbour table (rows with i >= j are dropped,
        reference admp/pme.py:671).  Called automatically by get_energy/get_forces when `pairs` changes."""
        if pairs is None:
            if self._pairs_key is None:
                raise ValueError('pairs=None needs a previous set_pairs / update_neighbors')
            return
        if pairs is self._pairs_keep and isinstance(pairs, torch.Tensor) and self._pairs_key is not None and \
                pairs._version == self._pairs_key[-1]:
            return                                   # same tensor object, not written since: the cached table holds
        key = self._pairs_fingerprint(pairs)
        if key == self._pairs_key:
            return
        with self._on_stream():
            self._set_pairs_now(pairs, key)

    def _set_pairs_now(self, pairs, key):
        if isinstance(pairs, torch.Tensor):
            t = pairs.detach().to(device=self._device, dtype=torch.int32).contiguous()
        else:
            a = np.asarray(pairs)
            if a.ndim != 2 or a.shape[1] != 2:
                raise ValueError('pairs must be (Np, 2)')
            t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self._device)
        if t.dim() != 2 or t.shape[1] != 2:
            raise ValueError('pairs must be (Np, 2)')
        _lib.check(self._h, self._L.admp_set_pairs(self._h, t.shape[0], self._ptr(t), 1), 'admp_set_pairs')
        self._pairs_key = key
        self._pairs_keep = pairs      # keeps id() unique while cached
        self._lender = None           # a list of its own ends a loan (share_neighbors)

    def _mscale_gradient(self, kind, positions, box, pairs, params, n_params, n_scales, pmax=0):
        """dE/dmScales (n_scales,) of this calculator: the gradient jax.grad(potential, argnums=3)(...)['mScales'] of the
        reference's examples/openmm_api/run.py:41-46."""
        with self._on_stream():
            self.set_pairs(pairs)
            pos = self._real(positions, (self.n_atoms, 3))
            par = self._real(params, (self.n_atoms, n_params))
            out = (ctypes.c_double * int(n_scales))()
            rc = self._L.admp_mscale_grad(self._h, int(kind), self._ptr(pos), _lib.darr(self._host64(box, 9)), self._ptr(par),
                                          int(pmax), int(n_scales), out, 1)
            _lib.check(self._h, rc, 'admp_mscale_grad')
        return np.array(out[:], dtype=np.float64)

    def update_neighbors(self, positions, box, rc=None):
        """Neighbour search + table compile in one GPU pass (cell list): afterwards pass `pairs=None` to
        get_energy / get_forces.  The MD-loop counterpart of re-running jax_md's neighbour list in the reference's
        drivers; equivalent to `set_pairs(NeighborList(box, rc).allocate(positions))`."""
        rc = float(self.rc if rc is None else rc)
        with self._on_stream():
            pos = self._real(positions, (self.n_atoms, 3))
            _lib.check(self._h, self._L.admp_set_pairs_from_positions(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), rc),
                       'admp_set_pairs_from_positions')
        self._pairs_key = ('from_positions',)
        self._pairs_keep = None
        self._lender = None

    def prune_neighbors(self, positions, box, rc):
        """Inner list of an MD loop (include/admp_hip.h admp_prune_pairs): until the next `update_neighbors` / `set_pairs` this
        calculator and those that borrow its table walk the entries of the current table whose distance at `positions` is
        below `rc` (device tensor).  rc <= 0: back to the table as built."""
        with self._on_stream():
            pos = self._real(positions, (self.n_atoms, 3))
            _lib.check(self._h, self._L.admp_prune_pairs(self._h, self._ptr(pos), _lib.darr(self._host64(box, 9)), float(rc)),
                       'admp_prune_pairs')

    def share_neighbors(self, lender):
        """Walk `lender`'s neighbour table instead of compiling one of its own (include/admp_hip.h admp_share_neighbors):
        the calculators of one system get the same pair list from the reference's drivers -- here one of them is given
        the list (`set_pairs` / `update_neighbors` / the `pairs` argument) and the others borrow its table, then take
        `pairs=None`.  Same atoms and covalent map required.  `lender=None`, or passing a pair list again, ends the loan."""
        with self._on_stream():
            _lib.check(self._h, self._L.admp_share_neighbors(self._h, lender._h if lender is not None else None),
                       'admp_share_neighbors')
        self._lender = lender                       # keeps the lender's handle alive
        self._pairs_key = ('shared', id(lender)) if lender is not None else None
        self._pairs_keep = None

    def set_cutoff(self, rc):
        """Skip the pairs of the list beyond `rc` (minimum image) in the pair kernels: for Verlet lists built with a skin.
        0 = evaluate every listed pair (default; what the reference does).  Dispersion and pair-potential calculators."""
        _lib.check(self._h, self._L.admp_set_cutoff(self._h, float(rc)), 'admp_set_cutoff')

    def set_side_stream(self, on=True):
        """ADMP_OPT_SIDE_STREAM: False keeps every kernel of a call on the handle's stream (clean per-kernel event times in
        measurements); True (default) lets small systems run their pair kernels next to the mesh chain."""
        _lib.check(self._h, self._L.admp_set_option(self._h, _lib.OPT_SIDE_STREAM, 1 if on else 0), 'admp_set_option')

    @property
    def n_pairs(self):
        return int(self._L.admp_num_pairs(self._h))

    # ---- measurement ---------------------------------------------------------------------------------------
    def profile(self, on=True, only=None):
        """Bracket kernel launches with HIP events; `only` restricts it to one label (e.g. 'pair_full')."""
        _lib.check(self._h, self._L.admp_profile_filter(self._h, only.encode() if only else None), 'admp_profile_filter')
        _lib.check(self._h, self._L.admp_profile_enable(self._h, 1 if on else 0), 'admp_profile_enable')

    def profile_reset(self):
        _lib.check(self._h, self._L.admp_profile_reset(self._h), 'admp_profile_reset')

    def profile_report(self):
        """{kernel label: (total_ms, launches)} measured with HIP events on the handle's stream."""
        n = self._L.admp_profile_count(self._h)
        out = {}
        for k in range(max(n, 0)):
            label = ctypes.c_char_p()
            ms = ctypes.c_double()
            cnt = ctypes.c_int64()
            _lib.check(self._h, self._L.admp_profile_entry(self._h, k, ctypes.byref(label), ctypes.byref(ms),
                                                           ctypes.byref(cnt)), 'admp_profile_entry')
            out[label.value.decode()] = (ms.value, cnt.value)
        return out
