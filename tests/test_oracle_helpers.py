"""The oracle's geometry/multipole helpers against the reference's own unit-test vectors
(reference tests/test_sptial.py, tests/test_multipole.py; arrays in golden/ref_unit_vectors.json)."""
import json
import os

import numpy as np
import numpy.testing as npt
import pytest
import torch

from oracle import admp_oracle as O

with open(os.path.join(os.path.dirname(__file__), 'golden', 'ref_unit_vectors.json')) as fh:
    VEC = json.load(fh)


def T(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float64))


@pytest.mark.parametrize('case', VEC['test_build_quasi_internal'])
def test_build_quasi_internal(case):
    out = O.build_quasi_internal(T(case['r1']), T(case['r2']), T(case['dr']), T(case['norm_dr']))
    # reference compares at default rtol 1e-7 against float32-formatted numbers; 2e-6 abs covers them
    npt.assert_allclose(out.numpy(), np.array(case['expected']), atol=2e-6)


@pytest.mark.parametrize('case', VEC['test_pbc_shift'])
def test_pbc_shift(case):
    out = O.pbc_shift(T(case['drvecs']), T(case['box']), T(case['box_inv']))
    npt.assert_allclose(out.numpy(), np.array(case['expected']), atol=1e-12)


@pytest.mark.parametrize('case', VEC['test_generate_construct_local_frames'])
def test_local_frames(case):
    out = O.construct_local_frames(case['positions'], case['box'], case['axis_types'], case['axis_indices'])
    # expected arrays are float32-derived (they carry ~1e-6 absolute noise against a float64 evaluation)
    npt.assert_allclose(out.numpy(), np.array(case['expected_local_frames']), atol=2e-6)


@pytest.mark.parametrize('case', VEC['test_convert_cart2harm'])
def test_convert_cart2harm(case):
    out = O.convert_cart2harm(case['theta'], 2)
    npt.assert_allclose(out.numpy(), np.array(case['expected']), rtol=1e-6)
    from admp_amd.systems import convert_cart2harm
    npt.assert_allclose(convert_cart2harm(case['theta'], 2), np.array(case['expected']), rtol=1e-6)


@pytest.mark.parametrize('case', VEC['test_rot_global_local'])
def test_rotations(case):
    Qg, Ql, fr = T(case['Q_global']), T(case['Q_local']), T(case['local_frames'])
    npt.assert_allclose(O.rot_local2global(Ql, fr, 2).numpy(), Qg.numpy(), rtol=1e-6, atol=1e-6)
    npt.assert_allclose(O.rot_global2local(Qg, fr, 2).numpy(), Ql.numpy(), rtol=1e-6, atol=1e-6)
