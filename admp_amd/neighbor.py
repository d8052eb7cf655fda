"""GPU neighbour search: positions -> the half pair list (i < j, minimum-image r < rc) the PME path consumes.

Counterpart of what the reference's drivers get from jax_md
(`partition.neighbor_list(displacement_fn, box, rc, 0, format=partition.OrderedSparse)` then
`nbr = neighbor_list_fn.allocate(positions); pairs = nbr.idx.T`, examples/water_1024/run_admp.py:109-112):

    nbl = NeighborList(box, rc)
    pairs = nbl.allocate(positions)          # (Np, 2) int32 torch tensor on the GPU, rows i < j
    pairs = nbl.update(positions)            # rebuild (same call; kept for jax_md-style code)

The search is a cell list in libadmp_hip (cell_kernels.hip); any lattice with rc <= half of each box height.
"""
import ctypes

import numpy as np
import torch

from . import _lib, settings
from ._device import _torch_dtype


class NeighborList:
    def __init__(self, box, rc, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('admp_amd.neighbor needs a ROCm GPU')
        self._L = _lib.load()
        self._nbytes = settings.precision_bytes()
        self._dtype = _torch_dtype(self._nbytes)
        self._dev_index = torch.cuda.current_device() if device is None else int(device)
        self._device = torch.device('cuda', self._dev_index)
        self._h = ctypes.c_void_p()
        rc_ = self._L.admp_create(ctypes.byref(self._h), self._dev_index, self._nbytes)
        if rc_ != 0:
            raise _lib.AdmpHipError('admp_create failed (code %d)' % rc_)
        self.box = np.ascontiguousarray(np.asarray(box.detach().cpu() if isinstance(box, torch.Tensor) else box,
                                                   dtype=np.float64)).reshape(3, 3)
        self.rc = float(rc)
        self._stream = None          # bound to the caller's current torch stream at call time (like the calculators)
        self.idx = None

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                self._L.admp_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def allocate(self, positions, box=None):
        """Build the pair list for `positions` (numpy or torch, (Na, 3)); returns (Np, 2) int32 on the GPU."""
        if box is not None:
            self.box = np.ascontiguousarray(np.asarray(box.detach().cpu() if isinstance(box, torch.Tensor) else box,
                                                       dtype=np.float64)).reshape(3, 3)
        cur = torch.cuda.current_stream(self._device).cuda_stream
        if cur != self._stream:
            if cur == 0:
                _lib.check(self._h, self._L.admp_use_default_stream(self._h), 'admp_use_default_stream')
            else:
                _lib.check(self._h, self._L.admp_set_stream(self._h, ctypes.c_void_p(cur)), 'admp_set_stream')
            self._stream = cur
        if True:
            if isinstance(positions, torch.Tensor):
                pos = positions.detach().to(device=self._device, dtype=self._dtype).contiguous()
            else:
                pos = torch.as_tensor(np.ascontiguousarray(np.asarray(positions, dtype=np.float64)),
                                      dtype=self._dtype).to(self._device)
            if pos.dim() != 2 or pos.shape[1] != 3:
                raise ValueError('positions must be (Na, 3)')
            n = ctypes.c_int64(0)
            _lib.check(self._h, self._L.admp_neighbor_count(self._h, pos.shape[0], ctypes.c_void_p(pos.data_ptr()),
                                                            _lib.darr(self.box.reshape(-1)), self.rc, ctypes.byref(n)),
                       'admp_neighbor_count')
            pairs = torch.empty((int(n.value), 2), dtype=torch.int32, device=self._device)
            _lib.check(self._h, self._L.admp_neighbor_fill(self._h, ctypes.c_void_p(pairs.data_ptr())),
                       'admp_neighbor_fill')
        self.idx = pairs
        return pairs

    update = allocate
