"""`admp` -- import-name alias of the reference package, served by admp_amd (the MI355X-native implementation).

The reference's drivers say `from admp.pme import ADMPPmeForce`, `from admp.disp_pme import ADMPDispPmeForce`,
`from admp.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel`, `import admp.settings`,
`from admp.multipole import convert_cart2harm` (examples/water_1024/run_admp.py:13-17).  With this directory on the
path those lines resolve to the HIP calculators.  Each submodule IS the admp_amd module of the same name (one module
object, so e.g. `admp.settings.PRECISION = 'single'` is seen by the calculators); nothing is re-implemented here.
`admp.parser` is this package's own small reader for the examples' PDB / XML inputs, `admp.api` the force-field front-end
(Hamiltonian / generators / potential_fn) without OpenMM.  Not provided: admp.recip / admp.spatial (internals of the
reference's JAX path that have no callable counterpart here).
"""
import importlib
import sys

for _name in ('settings', 'pme', 'disp_pme', 'pairwise', 'multipole', 'parser', 'api'):
    _mod = importlib.import_module('admp_amd.' + _name)
    sys.modules[__name__ + '.' + _name] = _mod
    globals()[_name] = _mod
del _name, _mod
