"""admp_amd -- MI355X-native multipolar / polarizable PME behind ADMP's calculator API.

    from admp_amd.pme import ADMPPmeForce
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad

The compute path is libadmp_hip.so (hand-written HIP for gfx950 + rocFFT), bound through ctypes;
build it with `python -m admp_amd.build`.  Importing this package does not load the library;
constructing a force object does, and fails loudly without it or without a GPU.
"""
__version__ = '0.1.0'
