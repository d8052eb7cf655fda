"""ctypes loader for tests/hostshim (host-compiled kernel arithmetic; test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'hostshim', 'shim.cpp')
LIB = os.path.join(HERE, 'hostshim', 'libadmp_hostshim.so')
CSRC = os.path.join(os.path.dirname(HERE), 'admp_amd', 'csrc')

_lib = None


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def lib():
    global _lib
    if _lib is None:
        if _stale():
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-o', LIB, SRC])
        _lib = ctypes.CDLL(LIB)
        _lib.shim_pair_real.restype = ctypes.c_double
        _lib.shim_disp_real.restype = ctypes.c_double
        _lib.shim_tt_real.restype = ctypes.c_double
        _lib.shim_disp_ck.restype = ctypes.c_double
    return _lib


def dp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def c64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def scale_tables(mScales, pScales=None):
    """Per-nbonds lookup (nbonds 0..15) of scales[(nbonds-1) % n] and the Fermi weight of
    admp/pme.py:337-348 at that pscale -- mirrors admp_amd's host-side table building."""
    n = len(mScales)
    idx = (np.arange(16) - 1) % n
    mtab = np.asarray(mScales, dtype=np.float64)[idx]
    if pScales is None:
        return mtab, np.zeros(16), np.zeros(16)
    ptab = np.asarray(pScales, dtype=np.float64)[idx]
    with np.errstate(over='ignore'):
        w0 = 1.0 / (np.exp((ptab - 1e-3) / 1e-5) + 1.0)
    return mtab, ptab, w0
