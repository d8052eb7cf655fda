"""Store the numbers the reference's parameter-gradient example holds as data.

``/root/reference/examples/openmm_api/ref_out`` is the printed output of the reference's ``examples/openmm_api/run.py``
(lines 40-43: ``pot_disp(...)`` = E_sr - E_lr, then ``jax.grad(pot_disp, argnums=3)[...]['mScales']``).  The file is read AS
TEXT -- numbers only, nothing of the reference is imported or executed -- and written to ``ref_openmm_api_mscale_grad.json``.
Run in the build container only (needs /root/reference).

What the numbers can and cannot pin: the run that produced them used a geometry that is NOT the
shipped ``water1024.pdb`` (components [1] and [4] of the gradient are orders of magnitude away from anything the shipped file
gives: that run had overlapping H-H contacts), but its energies are mutually consistent with ``examples/water_1024/ref_out``
(221523.0 - 54660.043 = 166 863 vs 166834.94 here; ethresh 1e-5 vs 1e-4).  Component [0] -- the 1-2 (O-H) scale -- only
sees intramolecular pairs, i.e. the rigid water geometry, which IS the same in both files: the one reference-held number of
the dispersion / Tang-Toennies path that can be compared, at the percent level.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == '__main__':
    text = open(os.path.join(REF, 'examples', 'openmm_api', 'ref_out')).read()
    nums = [float(x) for x in re.findall(r'[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?', text)]
    assert len(nums) == 6, nums
    out = {'source': 'examples/openmm_api/ref_out (printed by examples/openmm_api/run.py:40-43)',
           'E_sr_minus_E_lr': nums[0], 'dE_dmScales': nums[1:6],
           'comparable': {'dE_dmScales[0]': 'intramolecular 1-2 pairs only: independent of the box geometry for rigid waters',
                          'others': 'belong to a geometry that is not the shipped water1024.pdb'}}
    with open(os.path.join(HERE, 'ref_openmm_api_mscale_grad.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    print(out)
