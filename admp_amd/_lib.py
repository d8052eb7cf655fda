"""ctypes binding of libadmp_hip.so (include/admp_hip.h).

There is deliberately no fallback: if the HIP library is missing or cannot be loaded the
import fails loudly -- the PME path exists only as MI355X kernels.
"""
import ctypes

import numpy as np
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ADMP_HIP_LIB') or os.path.join(_HERE, 'lib', 'libadmp_hip.so')      # (override: A/B builds of tools/ab_build.sh)

_c = ctypes
_vp, _i32, _i64, _dbl = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_double
_dp = _c.POINTER(_c.c_double)
_ip = _c.POINTER(_c.c_int)

# name -> (restype, argtypes): every symbol include/admp_hip.h declares
PROTOTYPES = {
    'admp_version': (_c.c_char_p, []),
    'admp_create': (_i32, [_c.POINTER(_vp), _i32, _i32]),
    'admp_destroy': (_i32, [_vp]),
    'admp_last_error': (_c.c_char_p, [_vp]),
    'admp_set_stream': (_i32, [_vp, _vp]),
    'admp_use_default_stream': (_i32, [_vp]),
    'admp_synchronize': (_i32, [_vp]),
    'admp_set_topology': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    'admp_set_ewald': (_i32, [_vp, _dbl, _i32, _i32, _i32, _i32, _i32]),
    'admp_set_pairs': (_i32, [_vp, _i64, _vp, _i32]),
    'admp_share_neighbors': (_i32, [_vp, _vp]),
    'admp_num_pairs': (_i64, [_vp]),
    'admp_set_cutoff': (_i32, [_vp, _dbl]),
    'admp_set_dipole_source': (_i32, [_vp, _vp]),
    'admp_pme_energy_grad': (_i32, [_vp, _vp, _dp, _vp, _vp, _vp, _i32, _dp, _dp, _dp, _vp, _i32, _dbl, _dp, _vp, _vp,
                                    _ip, _ip, _i32]),
    'admp_pme_energy_fixed_dipoles': (_i32, [_vp, _vp, _dp, _vp, _vp, _vp, _i32, _dp, _dp, _vp, _dp, _vp, _vp, _vp]),
    'admp_pme_box_grad': (_i32, [_vp, _vp, _dp, _vp, _vp, _vp, _i32, _dp, _dp, _vp, _dp, _dp]),
    'admp_disp_box_grad': (_i32, [_vp, _vp, _dp, _vp, _i32, _i32, _dp, _dp, _dp]),
    'admp_tt_box_grad': (_i32, [_vp, _vp, _dp, _vp, _i32, _dp, _dp, _dp]),
    'admp_local_frames': (_i32, [_vp, _vp, _dp, _vp]),
    'admp_set_option': (_i32, [_vp, _i32, _i32]),
    'admp_disp_energy_grad': (_i32, [_vp, _vp, _dp, _vp, _i32, _i32, _dp, _dp, _vp, _i32]),
    'admp_prune_pairs': (_i32, [_vp, _vp, _dp, _dbl]),
    'admp_disp_set_types': (_i32, [_vp, _i32, _vp, _dp]),
    'admp_tt_energy_grad': (_i32, [_vp, _vp, _dp, _vp, _i32, _dp, _dp, _vp, _i32]),
    'admp_pair_program_build': (_i32, [_vp, _c.c_char_p, _i32, _ip]),
    'admp_pair_program_energy_grad': (_i32, [_vp, _i32, _vp, _dp, _vp, _i32, _dp, _dp, _vp, _i32]),
    'admp_disp_param_grad': (_i32, [_vp, _vp, _dp, _vp, _i32, _i32, _dp, _vp]),
    'admp_tt_param_grad': (_i32, [_vp, _vp, _dp, _vp, _i32, _dp, _vp]),
    'admp_thole_sums': (_i32, [_vp, _vp, _dp, _vp, _vp, _vp, _i32, _dp, _dp, _vp, _vp, _vp]),
    'admp_pscale_grad': (_i32, [_vp, _vp, _dp, _vp, _vp, _vp, _i32, _dp, _dp, _vp, _dp]),
    'admp_mscale_grad': (_i32, [_vp, _i32, _vp, _dp, _vp, _i32, _i32, _dp, _i32]),
    'admp_md_bonded': (_i32, [_vp, _vp, _dp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    'admp_md_kick_drift': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _dbl, _dbl, _vp]),
    'admp_neighbor_count': (_i32, [_vp, _i32, _vp, _dp, _dbl, _c.POINTER(_i64)]),
    'admp_neighbor_fill': (_i32, [_vp, _vp]),
    'admp_set_pairs_from_positions': (_i32, [_vp, _vp, _dp, _dbl]),
    'admp_slab_configure': (_i32, [_vp, _i32, _i32]),
    'admp_slab_info': (_i32, [_vp, _c.POINTER(_i64)]),
    'admp_scf_stats': (_i32, [_vp, _c.POINTER(_i64), _i32]),
    'admp_set_comm': (_i32, [_vp, _vp]),
    'admp_slab_home': (_i32, [_vp, _vp, _ip, _ip]),
    'admp_rccl_unique_id': (_i32, [_vp]),
    'admp_rccl_create': (_i32, [_c.POINTER(_vp), _i32, _vp, _i32, _i32]),
    'admp_rccl_destroy': (_i32, [_vp]),
    'admp_rccl_abort': (_i32, [_vp]),
    'admp_rccl_stats': (_i32, [_vp, _c.POINTER(_i64), _c.POINTER(_i64), _i32]),
    'admp_rccl_all_reduce': (_i32, [_vp, _vp, _i64, _i32, _i32, _vp]),
    'admp_rccl_all_to_all_v': (_i32, [_vp, _vp, _c.POINTER(_i64), _vp, _c.POINTER(_i64), _i32, _vp]),
    'admp_rccl_shift': (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    'admp_rccl_version': (_i32, [_ip]),
    'admp_rccl_last_error': (_c.c_char_p, []),
    'admp_set_comm_rccl': (_i32, [_vp, _vp]),
    'admp_profile_enable': (_i32, [_vp, _i32]),
    'admp_profile_filter': (_i32, [_vp, _c.c_char_p]),
    'admp_profile_reset': (_i32, [_vp]),
    'admp_profile_count': (_i32, [_vp]),
    'admp_profile_entry': (_i32, [_vp, _i32, _c.POINTER(_c.c_char_p), _dp, _c.POINTER(_i64)]),
}

# communicator of a slab-decomposed handle (include/admp_hip.h: admp_comm)
T_I32, T_F32, T_F64 = 0, 1, 2
OP_SUM, OP_MAX = 0, 1
TAGS = {1: 'ghost_planes', 2: 'transpose', 3: 'halo_dipoles', 4: 'halo_gradient', 5: 'scf_max', 6: 'energies',
        7: 'replicate_outputs'}
_i64p = _c.POINTER(_i64)
ALL_REDUCE_FN = _c.CFUNCTYPE(_i32, _vp, _vp, _i64, _i32, _i32, _i32)
ALL_TO_ALL_V_FN = _c.CFUNCTYPE(_i32, _vp, _vp, _i64p, _vp, _i64p, _i32, _i32)
SHIFT_FN = _c.CFUNCTYPE(_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32)


class AdmpComm(_c.Structure):
    _fields_ = [('ctx', _vp), ('all_reduce', ALL_REDUCE_FN), ('all_to_all_v', ALL_TO_ALL_V_FN), ('shift', SHIFT_FN)]


OPT_REFERENCE_KPOINTS = 1
OPT_KEEP_POL_SITES = 2
OPT_SIDE_STREAM = 3
RCCL_ID_BYTES, RCCL_NTAGS = 128, 8

_lib = None


class AdmpHipError(RuntimeError):
    pass


def load():
    """Load the shared library and attach prototypes (does not touch the GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'admp_amd: %s not found. Build it with `python -m admp_amd.build` (needs hipcc); '
            'there is no CPU fallback for the PME path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(handle, rc, what):
    if rc != 0:
        msg = load().admp_last_error(handle)
        raise AdmpHipError('%s failed (code %d): %s' % (what, rc, msg.decode() if msg else ''))


def darr(values):
    """ctypes double array from a python/numpy sequence."""
    a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    return (ctypes.c_double * a.size).from_buffer_copy(a.tobytes())
