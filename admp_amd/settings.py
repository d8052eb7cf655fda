"""Global switches of the PME path, same names as the reference's admp/settings.py:6-30.

PRECISION selects the arithmetic of the HIP kernels ('double' -> f64, 'single' -> f32; energies are
accumulated in f64 either way).  DO_JIT / jit_condition are kept for source compatibility and do
nothing: there is no tracing compiler on this path.
"""
PRECISION = 'double'      # 'single' | 'double'   (admp/settings.py:6)
DO_JIT = True             # no-op here            (admp/settings.py:8)

# DEFAULT THRESHOLDS (admp/settings.py:29-30)
POL_CONV = 10.0           # gradient convergence threshold for the induced dipoles, kJ/mol/(e A)
MAX_N_POL = 30            # maximum number of SCF cycles

# Not in the reference (default off = the reference's mesh exactly): round K1..K3 of setup_ewald_parameters UP to the next
# size with prime factors <= 7.  The reference's rule often lands on sizes rocFFT can only do with Bluestein (305 = 5*61
# for a 99 A box: 2.5 ms per transform pair instead of ~0.3 ms); a finer mesh only lowers the PME error, but the numbers
# then differ from the reference's at the level of ethresh -- hence opt-in.
FFT_FRIENDLY_MESH = False


# The reciprocal-space k-point table.  True (default since round 3: a drop-in returns the reference's numbers): the literal
# order of admp/recip.py:339-340 -- `meshgrid(kz, kx, ky)` puts the frequencies of mesh axes (1, 0, 2) into k-columns
# (0, 1, 2) and, for K2 != K3, scrambles the flattened table against the mesh.  On a cubic box with K1 = K2 = K3 (every
# shipped example and benchmark case) both orders give the same energies and forces; elsewhere the reference's order is not
# a consistent Ewald sum (it pairs a mesh axis with another axis' reciprocal vector) and a warning says so.  False: the
# axis-by-axis (physically consistent) assignment.  Read when a calculator is created / refreshed.
REFERENCE_KPOINT_ORDER = True


def jit_condition(*args, **kwargs):
    def deco(func):
        return func
    return deco


def precision_bytes():
    if PRECISION not in ('single', 'double'):
        raise ValueError("settings.PRECISION must be 'single' or 'double'")
    return 4 if PRECISION == 'single' else 8

# Dispersion PME: when the (C6, C8, C10) rows of the atoms take at most three distinct values (atom types), the reciprocal part
# keeps one mesh per type instead of one per power (include/admp_hip.h admp_disp_set_types).  Same sums regrouped; False keeps
# the reference's per-power meshes (admp/disp_pme.py:80-123).
DISP_TYPED_MESHES = True
