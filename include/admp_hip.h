/* libadmp_hip -- C ABI of the MI355X-native multipolar PME path.
 *
 * The reference (Roy-Kid/ADMP) has no FFI layer: its boundary is the Python callable API of
 * admp/pme.py:30-143 (ADMPPmeForce), admp/disp_pme.py:20-77 (ADMPDispPmeForce) and
 * admp/pairwise.py:45-113 (generate_pairwise_interaction + TT kernel).  This header is the layer
 * BENEATH that API: admp_amd/{pme,disp_pme,pairwise}.py keep the reference's class / method names
 * and bind these entry points through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success or a negative ADMP_E_* code; nothing throws or aborts;
 *     admp_last_error() gives the message of the last failure on that handle.
 *   - `real` arrays are float (precision 4) or double (precision 8), fixed at admp_create.
 *   - array arguments marked [dev|host] are device pointers when `on_device` is non-zero, host
 *     pointers otherwise (the library then stages them); the caller owns every buffer.
 *   - one handle = one GPU + one HIP stream; a handle is not thread-safe, distinct handles are
 *     independent.  Energies are always returned as double, in kJ/mol; lengths in Angstrom.
 */
#ifndef ADMP_HIP_H
#define ADMP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct admp_handle admp_handle;

enum {
  ADMP_OK = 0,
  ADMP_E_ARG = -1,      /* bad argument / call order */
  ADMP_E_HIP = -2,      /* HIP runtime error */
  ADMP_E_FFT = -3,      /* rocFFT error */
  ADMP_E_NOGPU = -4,    /* no usable device */
  ADMP_E_STATE = -5,    /* topology / ewald / pairs not set, or an internal precondition violated */
  ADMP_E_COMM = -6      /* a communicator callback of a slab-decomposed handle failed */
};

/* library version, and the gfx target the kernels were built for ("gfx950") */
const char* admp_version(void);

/* ---- lifetime ------------------------------------------------------------------------------ */
/* replaces: ADMPPmeForce.__init__ (admp/pme.py:37-55) -- device side of the object */
int admp_create(admp_handle** out, int device, int precision /* 4 | 8 */);
int admp_destroy(admp_handle* h);
const char* admp_last_error(const admp_handle* h);
/* run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int admp_set_stream(admp_handle* h, void* hip_stream);
/* run on the legacy default (null) stream -- what torch.cuda.current_stream() is unless the caller switched streams */
int admp_use_default_stream(admp_handle* h);
int admp_synchronize(admp_handle* h);

/* ---- static environment -------------------------------------------------------------------- */
/* replaces: axis_type / axis_indices / covalent_map ctor arguments (admp/pme.py:37-51,
 * generate_construct_local_frames admp/spatial.py:44-74).  covalent_map is passed in CSR form
 * (row i: atoms j with covalent_map[i,j] = nbonds > 0, nbonds <= 7: ADMP_E_ARG beyond -- the reference's
 * parser marks 1..2, its scale lists have 5 entries); all pointers are HOST pointers.
 * axis_type[i] in 0..5 (ZThenX, Bisector, ZBisect, ThreeFold, Zonly, NoAxisType);
 * axis_idx[i*3 + {0,1,2}] = z, x, y atom (-1 = none). */
int admp_set_topology(admp_handle* h, int n_atoms, const int32_t* axis_type, const int32_t* axis_idx,
                      const int32_t* excl_rowptr, const int32_t* excl_col, const int32_t* excl_nbonds);

/* replaces: the kappa / K1..K3 / lmax / lpol environment (admp/pme.py:42-50, update_env :89-94) */
int admp_set_ewald(admp_handle* h, double kappa, int K1, int K2, int K3, int lmax, int lpol);

/* behaviour switches that are not part of the reference's argument lists */
enum {
  ADMP_OPT_REFERENCE_KPOINTS = 1,  /* value 1: build the reciprocal-space tables with the reference's literal k-point order
                                      (meshgrid(kz, kx, ky), admp/recip.py:339-340) instead of the axis-by-axis one; the
                                      two agree iff K1 = K2 = K3 on a cubic box.  Default 0. */
  ADMP_OPT_KEEP_POL_SITES = 2,     /* value 1: the caller vouches that the SET of polarizable sites {i : pol_i > 0} of the
                                      following admp_pme_energy_grad calls is the one of the previous call (the values may
                                      change); the library then keeps its list of those sites instead of rebuilding it in
                                      every call.  Default 0 (rebuild every call).  Speed only, never results. */
  ADMP_OPT_SIDE_STREAM = 3         /* value 0: every kernel of a call runs on the handle's stream.  Default 1: on systems up to
                                      200 000 atoms the real-space pair kernels run on a second stream next to the mesh chain
                                      of the same call.  Speed only; switched off by measurements that bracket single kernels
                                      with events (the event time of a kernel that shares the chip is not its duration). */
};
int admp_set_option(admp_handle* h, int option, int value);

/* replaces: `pairs = pairs[pairs[:,0] < pairs[:,1]]` + the per-pair gathers of pme_real
 * (admp/pme.py:671-683).  pairs is (n_rows, 2) int32 [dev|host]; rows with i >= j (padding) are
 * dropped.  The list is compiled into an i-grouped neighbour table that stays valid until the
 * next call. */
int admp_set_pairs(admp_handle* h, int64_t n_rows, const int32_t* pairs, int on_device);

/* The calculators of one system (PME, dispersion PME, pair potentials) are handed the SAME pair list by the reference's
 * drivers (examples/water_pol_1024/run_admp.py:117-136: one `pairs` array for every force object).  Instead of compiling
 * that list once per handle, `h` walks the neighbour table of `lender` (same atoms, same covalent map, same device):
 * whatever list the lender holds when `h` is next called -- also after the lender's admp_set_pairs /
 * admp_set_pairs_from_positions.  lender = NULL, or a pair list of its own, ends the loan; calling `h` after the lender
 * was destroyed is ADMP_E_ARG. */
int admp_share_neighbors(admp_handle* h, admp_handle* lender);
int64_t admp_num_pairs(const admp_handle* h);   /* pairs kept (i < j) */
/* Verlet lists with a skin (MD drivers rebuild the list every few steps with rc + skin): pairs of the list whose
 * minimum-image distance is >= rc are skipped by the pair kernels, so that the result is the one of the exact-rc list
 * whatever the skin.  rc = 0 (default): every listed pair is evaluated -- what the reference does with whatever list it is
 * handed (admp/pme.py:671-729 has no distance test).  Not a speed option (the kernels wait on the partner fetches, not on
 * the pair arithmetic).  Honoured by the dispersion and Tang-Toennies pair kernels and by everything derived from them
 * (admp_disp_energy_grad, admp_tt_energy_grad, their box gradients, admp_disp_param_grad, admp_tt_param_grad, admp_mscale_grad
 * kinds 1 and 2: the derivatives are those of the energy the calculator returns); the multipolar PME kernels evaluate
 * every listed pair. */
int admp_set_cutoff(admp_handle* h, double rc);
/* MD loops with a Verlet skin (no counterpart in the reference, which evaluates whatever list it is given): until the next
 * list build the calculators of `h` -- and of every handle that borrows its table -- walk an INNER table: the entries of the
 * current one whose minimum-image distance at `positions` (DEVICE pointer, (Na,3) real) is below rc, rows compacted in place
 * order.  A driver that rebuilds the outer list (rc + skin) every n steps prunes it to rc + skin_inner every m < n steps,
 * skin_inner = twice what an atom can move in m steps: the multipolar kernels, which evaluate every listed pair, then do
 * (rc + skin_inner)^3 / (rc + skin)^3 of the work, and so do the cutoff-testing dispersion / pair-potential kernels.  Pruning
 * always starts from the table as built; rc <= 0 goes back to it.  Single-rank handles that own their table.  One host
 * synchronisation. */
int admp_prune_pairs(admp_handle* h, const void* positions, const double* box, double rc);
/* One shot, device pointers only: the NEXT admp_pme_energy_grad reads its initial dipoles from U_init ((Na,3), read-only) and
 * uses U_inout purely as output (it starts as a copy made by the first kernel of the evaluation).  The reference's callers
 * pass `U_init=pme.U_ind` and get a new array back (admp/pme.py:104-109: jnp arrays are immutable): with this entry the
 * wrapper hands over a fresh output array without a device copy of its own.  NULL clears it. */
int admp_set_dipole_source(admp_handle* h, const void* U_init);

/* ---- the hot path -------------------------------------------------------------------------- */
/* replaces: ADMPPmeForce.get_energy / get_forces (admp/pme.py:58-86, 108) including the induced
 * dipole SCF optimize_Uind (admp/pme.py:111-143) when the handle is polarizable.
 *   positions  (Na,3) real [dev|host]         box        9 doubles, lattice vectors in rows (host)
 *   Q_local    (Na,9) real [dev|host]         harmonics [00,10,11c,11s,20,21c,21s,22c,22s], zero-padded above lmax
 *   pol,tholes (Na) real [dev|host]           NULL unless polarizable
 *   mScales/pScales/dScales                   n_scales doubles each (host); index (nbonds-1) wraps like the reference
 *   U_inout    (Na,3) real [dev|host]         in: SCF start (global Cartesian), out: converged dipoles; NULL unless polarizable
 *   max_cycle, thresh                         MAX_N_POL / POL_CONV of admp/settings.py:29-30
 *   E_out[4]                                  real-space, reciprocal, self, polarization penalty
 *   dE_dpos    (Na,3) real [dev|host]         +dE/dpositions (the reference returns the gradient, not the force); may be NULL (energy only)
 *   dE_dQlocal (Na,9) real [dev|host]         optional (NULL to skip)
 *   n_cycle, converged                        loop index at exit and `i != max_cycle-1` (admp/pme.py:139-143) */
int admp_pme_energy_grad(admp_handle* h, const void* positions, const double* box, const void* Q_local,
                         const void* pol, const void* tholes, int n_scales, const double* mScales,
                         const double* pScales, const double* dScales, void* U_inout, int max_cycle, double thresh,
                         double* E_out, void* dE_dpos, void* dE_dQlocal, int* n_cycle, int* converged,
                         int on_device);

/* replaces: ADMPPmeForce.energy_fn / grad_U_fn / grad_pos_fn (admp/pme.py:69-78): the bare polarizable energy with the
 * induced dipoles as an explicit input (no SCF), and its derivatives with respect to positions, dipoles and Q_local.
 * All array arguments are DEVICE pointers; U (Na,3) global Cartesian; dE_dpos / dE_dU / dE_dQlocal may each be NULL.
 * dE_dU is the "field" of optimize_Uind (admp/pme.py:133): real + reciprocal + self + polarization-penalty terms.
 * On a slab-decomposed handle (round 4): U must be valid on the rank's home rows (the rows it reads of other ranks are fetched
 * from their owners into a copy); dE_dpos / dE_dU / dE_dQlocal hold the home rows, E_out the global energies. */
int admp_pme_energy_fixed_dipoles(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                         const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                         double* E_out, void* dE_dpos, void* dE_dU, void* dE_dQlocal);

/* replaces: jax.value_and_grad(get_energy, argnums=1) -- the gradient of the energy with respect to the cell matrix at FIXED
 * Cartesian positions, from which callers of the reference form the virial (README.md:7; admp/pme.py:108 and
 * admp/disp_pme.py:76 with argnums).  dE_dbox is 9 doubles (host), row-major like `box` (lattice vectors in rows).
 * Positions / parameters are DEVICE pointers.  For a polarizable handle U holds the induced dipoles to evaluate at (the
 * converged ones: the reference differentiates energy_fn at stop_gradient(U_ind), admp/pme.py:81-85).
 * E_out as in the corresponding energy_grad call.  All three also run on a slab-decomposed handle (round 4: every rank returns
 * the full gradient; the device sums are added over the ranks). */
int admp_pme_box_grad(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                      const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                      double* E_out, double* dE_dbox);
int admp_disp_box_grad(admp_handle* h, const void* positions, const double* box, const void* c_list, int pmax, int n_scales,
                       const double* mScales, double* E_out, double* dE_dbox);
int admp_tt_box_grad(admp_handle* h, const void* positions, const double* box, const void* abqc, int n_scales,
                     const double* mScales, double* E_out, double* dE_dbox);

/* replaces: the `construct_local_frames(positions, box)` attribute (generate_construct_local_frames,
 * admp/spatial.py:44-142): frames_out (Na,3,3) real, rows = local x, y, z axes in the global frame.  DEVICE pointers.
 * Diagnostic only -- the hot path builds the frames inside its first kernel and never materialises them. */
int admp_local_frames(admp_handle* h, const void* positions, const double* box, void* frames_out);

/* replaces: ADMPDispPmeForce.get_energy / get_forces (admp/disp_pme.py:44-77, 80-279).
 *   c_list (Na,3) real: C6, C8, C10 per atom (columns above pmax ignored); E_out[3] = real, recip, self */
int admp_disp_energy_grad(admp_handle* h, const void* positions, const double* box, const void* c_list, int pmax,
                          int n_scales, const double* mScales, double* E_out, void* dE_dpos, int on_device);

/* Optional hint for admp_disp_energy_grad (no counterpart in the reference, which spreads every channel's coefficients,
 * admp/disp_pme.py:80-123): the rows of c_list take only n_types <= 3 distinct values (atom types: water has two).
 *   type_of_atom  (Na) int32, DEVICE pointer owned by the caller (must stay valid until replaced): row index of every atom
 *   coefficients  (n_types,3) host doubles: the distinct rows (C6, C8, C10)
 * A single-rank handle on a power-of-two mesh in single precision then keeps one mesh per TYPE instead of one per channel
 * (S_p(k) = sum_t c_p,t S_t(k): one spread and one gather per atom, n_types transforms each way, the channels combined per k
 * point) -- the same sums regrouped.  Every call that uses the table checks it against c_list and fails with ADMP_E_ARG on a
 * mismatch.  n_types = 0 (or NULL): forget the table.  Other handles ignore it. */
int admp_disp_set_types(admp_handle* h, int n_types, const void* type_of_atom, const double* coefficients);

/* replaces: generate_pairwise_interaction(TT_damping_qq_c6_kernel, ...) (admp/pairwise.py:45-113).
 *   abqc (Na,4) real: a, b, q, c6 per atom */
int admp_tt_energy_grad(admp_handle* h, const void* positions, const double* box, const void* abqc, int n_scales,
                        const double* mScales, double* E_out, void* dE_dpos, int on_device);

/* replaces: generate_pairwise_interaction(pair_int_kernel, covalent_map, static_args) for an ARBITRARY kernel
 * (admp/pairwise.py:45-91): the Python layer traces `pair_int_kernel(dr, m, p1i, p1j, ...)` once into HIP source
 * (admp_amd/xp.py: one statement per arithmetic operation, d/d(dr) by forward-mode dual numbers) and this call compiles
 * it for gfx950 with hiprtc and loads it.  The source must define
 *   extern "C" __global__ void admp_pair_custom(int na, const int* rowptr, const int* col, const int* order,
 *       const REAL_T* pos, const REAL_T* par, const REAL_T* box18, const REAL_T* mscale16, REAL_T* grad, double* energy)
 * (REAL_T is defined on the command line as the handle's precision).  program_id receives a handle-local id. */
int admp_pair_program_build(admp_handle* h, const char* hip_source, int n_params, int* program_id);
/* energy and +dE/dpositions of a built program on the current pair list; params (Na, n_params) real [dev|host],
 * other arguments as admp_tt_energy_grad; E_out[1] */
int admp_pair_program_energy_grad(admp_handle* h, int program_id, const void* positions, const double* box,
                                  const void* params, int n_scales, const double* mScales, double* E_out, void* dE_dpos,
                                  int on_device);

/* replaces: jax.grad(potential, argnums=3)(...)['mScales'] of the reference's parameter-gradient example
 * (examples/openmm_api/run.py:41-46): dE/dmScales[k], k = 0..n_scales-1, of one calculator on the current pair list.
 *   kind 0  multipolar PME         params = Q_local (Na,9) real   (admp/pme.py:681-683: mscales = mScales[nbonds-1])
 *   kind 1  dispersion PME         params = c_list  (Na,3) real, pmax as in admp_disp_energy_grad
 *   kind 2  Tang-Toennies damping  params = abqc    (Na,4) real
 * The pair energy is linear in the scale factor, so the result does not depend on the current mScales values (nor, for a
 * polarizable handle, on the induced dipoles: the induced terms carry pScales).  positions / params [dev|host]. */
int admp_mscale_grad(admp_handle* h, int kind, const void* positions, const double* box, const void* params, int pmax,
                     int n_scales, double* dE_dmScales, int on_device);

/* replaces: the 'C6' / 'C8' / 'C10' and 'A' / 'B' / 'Q' entries of jax.grad(pot_disp, argnums=3) of the reference's
 * dispersion front-end (examples/openmm_api/run.py:41-43 through admp/api.py:183-199), at the level of the per-ATOM lists the
 * calculators take (the caller chains them to its per-type tables and unit conversions, admp_amd/api.py):
 *   admp_disp_param_grad   dE_dc (Na,3) real = d(E_real + E_recip + E_self)/dc_list of admp_disp_energy_grad
 *   admp_tt_param_grad     dE_dabqc (Na,4) real = dE/d(a, b, q, c6) of admp_tt_energy_grad; an atom whose a or b is zero gets 0
 *                          for that entry (the geometric mean sqrt(a_i a_j) has no finite derivative there: NaN in autodiff)
 * All array arguments are DEVICE pointers.  On a slab-decomposed handle (round 4) the rows of the rank's home atoms are filled
 * (per-atom outputs, like the gradient); the class sums of admp_mscale_grad / admp_pscale_grad are added over the ranks. */
int admp_disp_param_grad(admp_handle* h, const void* positions, const double* box, const void* c_list, int pmax, int n_scales,
                         const double* mScales, void* dE_dc);
int admp_tt_param_grad(admp_handle* h, const void* positions, const double* box, const void* abqc, int n_scales,
                       const double* mScales, void* dE_dabqc);

/* replaces: the 'pol' / 'tholes' entries of jax.grad(pot_pme, argnums=3) (parameter gradients of the polarizable model).
 * The polarizabilities and Thole parameters enter the pair energy only through the Thole argument
 * au = a_w r / (alpha_i alpha_j)^(1/6) (admp/pme.py:408-414).  This call returns, for the induced dipoles U given
 * (normally the converged ones), the per-atom sums
 *   sumX[i]  = sum_j dE_ij / d ln(au_ij)            -> dE/dalpha_i = -sumX[i] / (6 alpha_i) - D |U_i|^2 / (2 alpha_i^2)
 *   sumXw[i] = sum_j dE_ij / d ln(au_ij) (1 - w0_ij) / a_w,ij   -> dE/dthole_i = sumXw[i]
 * (the closing formulas are applied by the caller, admp_amd/pme.py).  All array arguments are DEVICE pointers. */
int admp_thole_sums(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                    const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                    void* sumX, void* sumXw);

/* replaces: the 'pScales' entry of jax.grad(pot_pme, argnums=3): dE/dpScales[k], k = 0..n_scales-1, at the induced dipoles U
 * given (DEVICE pointers).  pscale multiplies the Thole factor of the permanent-induced coefficients (admp/pme.py:455-470);
 * its second role, the Fermi switch of the Thole width (pme.py:411), has a derivative below 1e-38 wherever it is finite
 * and is NaN in the reference's autodiff for pscale > ~0.008 (exp overflow): the analytic limit 0 is used.  The 'dScales'
 * entry is identically zero (the reference ignores dScales, uscales = 1, pme.py:472). */
int admp_pscale_grad(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                     const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                     double* dE_dpScales);

/* ---- neighbour search ("next" row of SURVEY.md 8f) --------------------------------------------------------
 * replaces: jax_md.partition.neighbor_list(displacement_fn, box, rc, 0, format=OrderedSparse).allocate(positions)
 * of the reference's drivers (examples/water_1024/run_admp.py:109-112): the producer of `pairs`.
 * Cell list on the GPU, any lattice with rc <= half of every box height.  Two phases so the caller can size
 * the output: count -> allocate (n_pairs, 2) int32 on the device -> fill (rows i < j, grouped by i, j ascending
 * within a cell sweep).  positions is a DEVICE pointer to (n_atoms, 3) reals and must stay valid until fill. */
int admp_neighbor_count(admp_handle* h, int n_atoms, const void* positions, const double* box, double rc,
                        int64_t* n_pairs);
int admp_neighbor_fill(admp_handle* h, int32_t* pairs_out);
/* search + admp_set_pairs fused: builds the handle's neighbour table (all pairs with minimum-image r < rc) straight
 * from DEVICE positions (n_atoms of admp_set_topology), without materialising the pair array. */
int admp_set_pairs_from_positions(admp_handle* h, const void* positions, const double* box, double rc);

/* ---- MD-driver helpers ("next" row f2 of SURVEY.md 8f; the reference has no integrator) -------------------
 * replaces: nothing in the reference -- its drivers stop at get_forces.  What an MD loop around the hot path needs besides
 * it, as device kernels so that a step does not pay dozens of framework launches: the harmonic bonded terms of the drivers'
 * force field (examples/water_1024/mpidwater.xml:16-21: HarmonicBondForce E = k/2 (r - r0)^2, HarmonicAngleForce E = k/2
 * (theta - theta0)^2) over explicit lists, and the half steps of velocity Verlet.  All pointers are DEVICE pointers
 * (box: host); nothing is read back -- energies accumulate into device words the caller reads when it logs.
 *   admp_md_bonded      bond_idx (n_bonds,2) int32, bond_par (n_bonds,2) real = (k, r0); angle_idx (n_angles,3) int32 = (i,
 *                       centre, k), angle_par (n_angles,2) real = (k, theta0 in rad); minimum-image vectors;
 *                       E_dev[0] += bond energy, E_dev[1] += angle energy (doubles); grad_inout (Na,3) real += dE/dr
 *   admp_md_kick_drift  v -= half_dt_acc * grad / m (grad = +dE/dr, inv_mass (Na) real); then, if dt != 0, r += dt * v;
 *                       ekin_dev (optional) += sum m v^2 / 2 after the kick, in the caller's units */
int admp_md_bonded(admp_handle* h, const void* positions, const double* box, int n_bonds, const int32_t* bond_idx,
                   const void* bond_par, int n_angles, const int32_t* angle_idx, const void* angle_par, double* E_dev,
                   void* grad_inout);
int admp_md_kick_drift(admp_handle* h, int n_atoms, void* positions, void* velocities, const void* grad, const void* inv_mass,
                       double half_dt_acc, double dt, double* ekin_dev);

/* ---- multi-GPU: x-slab decomposition ---------------------------------------------------------------------
 * (no counterpart in the reference, which is single-device)  One process per GPU, SPMD: every rank makes
 * the same admp_pme_energy_grad / admp_disp_energy_grad / admp_tt_energy_grad call with the same full input arrays.  Rank s
 * owns the mesh planes [X0,X1) along x and works on its "home" atoms (lowest stencil plane inside the slab); the library
 * runs the whole evaluation -- the same kernels and the same SCF forms as on one GPU -- and calls back into the caller's
 * communicator where ranks exchange data: ghost planes after the spread / before the gather (shift), the two transposes
 * of the distributed 3-D transform (all_to_all_v), the dipoles / gradient contributions of the halo atoms (all_to_all_v
 * over index lists both ends derive by themselves), the SCF residual (all_reduce MAX of one word) and the energies
 * (all_reduce SUM of four words).  The host side binds the callbacks to RCCL (torch.distributed backend "nccl" in
 * admp_amd/parallel.py); nothing proportional to the number of atoms is ever exchanged by the library itself.
 *
 * Callback contract: buffers are DEVICE pointers owned by the library; counts are in elements of `dtype`; a callback is
 * entered on the thread that made the library call, with the handle's stream being the caller's current stream, and must
 * enqueue its communication ordered after the work already on that stream and before whatever is enqueued on it later
 * (torch.distributed collectives on the current stream do exactly that).  Return 0, or non-zero to abort the call with
 * ADMP_E_COMM.  `tag` says what travels (accounting only). */
enum { ADMP_T_I32 = 0, ADMP_T_F32 = 1, ADMP_T_F64 = 2 };
enum { ADMP_OP_SUM = 0, ADMP_OP_MAX = 1 };
enum { ADMP_TAG_GHOST = 1, ADMP_TAG_TRANSPOSE = 2, ADMP_TAG_HALO_DIPOLES = 3, ADMP_TAG_HALO_GRADIENT = 4, ADMP_TAG_SCF_MAX = 5,
       ADMP_TAG_ENERGIES = 6 };
typedef struct admp_comm {
  void* ctx;
  /* in place over `count` elements of every rank */
  int (*all_reduce)(void* ctx, void* buf, int64_t count, int dtype, int op, int tag);
  /* segment t of `send` (send_counts[t] elements, segments back to back in rank order) goes to rank t; segment s of `recv`
   * (recv_counts[s] elements) comes from rank s */
  int (*all_to_all_v)(void* ctx, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                      int dtype, int tag);
  /* ring shift: send `count` elements to rank+1 (to_next != 0) or rank-1, receive as many from the opposite neighbour */
  int (*shift)(void* ctx, const void* send, void* recv, int64_t count, int dtype, int to_next, int tag);
} admp_comm;
/* rank / nranks of this handle (nranks = 1: not decomposed; at most 28 ranks).  Call before the first evaluation. */
int admp_slab_configure(admp_handle* h, int rank, int nranks);
/* the communicator of a decomposed handle (copied; ctx must outlive the handle).  NULL detaches. */
int admp_set_comm(admp_handle* h, const admp_comm* comm);
/* out11 = {X0, X1, Y0, Y1, local planes (X1-X0+ghost), ghost, K1, K2, K3/2+1, rank, nranks} */
int admp_slab_info(admp_handle* h, int64_t* out11);
/* ---- native RCCL communicator (round 4) ---------------------------------------------------------------------
 * The collectives of a decomposed handle issued by the library itself, on the handle's stream: ncclAllReduce, and
 * ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd for the all-to-all-v and the ring shifts -- no callback, no host
 * language in the step.  RCCL is bound at run time (dlopen of librccl.so.1; inside a PyTorch process that is the copy
 * PyTorch already holds), so single-GPU users never load it.
 *   admp_rccl_unique_id   rank 0 creates the 128-byte id (ncclGetUniqueId); the caller distributes it to the other ranks by
 *                         whatever means it has (torch.distributed broadcast in admp_amd/parallel.py, MPI, a file ...)
 *   admp_rccl_create      every rank: ncclCommInitRank on `device` (blocks until all ranks have joined)
 *   admp_set_comm_rccl    binds the communicator to a handle: sets rank / nranks (as admp_slab_configure) and replaces the
 *                         callbacks of admp_set_comm.  One communicator serves any number of handles of the process (PME,
 *                         dispersion PME, pair potentials); it must outlive them.  NULL detaches.
 *   admp_rccl_stats       bytes sent to other ranks / calls per ADMP_TAG_* (arrays of ADMP_RCCL_NTAGS), optionally cleared
 *   admp_rccl_abort       ncclCommAbort: releases peers blocked in a collective of this communicator (error paths)
 *   admp_rccl_all_reduce  in-place SUM / MAX over `count` elements of a caller-owned DEVICE buffer on `hip_stream` (what a
 *                         driver that wants the reference's replicated outputs does once per output array; tag 7)
 *   admp_rccl_all_to_all_v, admp_rccl_shift   the other two collectives of the decomposed evaluation on caller-owned DEVICE
 *                         buffers, with the semantics of the admp_comm callbacks above (for drivers that move their own
 *                         per-atom data between slab ranks, and for tests) */
typedef struct admp_rccl admp_rccl;
enum { ADMP_RCCL_ID_BYTES = 128, ADMP_RCCL_NTAGS = 8 };
int admp_rccl_unique_id(void* out128);
int admp_rccl_create(admp_rccl** out, int device, const void* id128, int rank, int nranks);
int admp_rccl_destroy(admp_rccl* c);
int admp_rccl_abort(admp_rccl* c);
int admp_rccl_stats(admp_rccl* c, int64_t* bytes_out, int64_t* calls_out, int reset);
int admp_rccl_all_reduce(admp_rccl* c, void* buf, int64_t count, int dtype /* ADMP_T_* */, int op /* ADMP_OP_* */, void* hip_stream);
int admp_rccl_all_to_all_v(admp_rccl* c, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                           int dtype, void* hip_stream);
int admp_rccl_shift(admp_rccl* c, const void* send, void* recv, int64_t count, int dtype, int to_next, void* hip_stream);
int admp_rccl_version(int* version);
const char* admp_rccl_last_error(void);
int admp_set_comm_rccl(admp_handle* h, admp_rccl* c);
/* Outputs of a decomposed evaluation: dE_dpos / U_inout / dE_dQlocal hold this rank's HOME rows (the other rows are
 * unspecified); E_out, n_cycle and converged are the global values on every rank.  home_out (device, room for n_atoms
 * int32) receives the home atoms of the last evaluation in ascending order, *n_home their number; n_import (optional) the
 * number of atoms the rank read without owning them. */
int admp_slab_home(admp_handle* h, int32_t* home_out, int* n_home, int* n_import);

/* How the polarizable evaluations of this handle were enqueued (the residual history of the previous calls decides between
 * the plain loop, a speculative first cycle and a chain of device-gated Jacobi steps: engine.hip pme()) and what the guesses
 * cost.  out8 = {plain calls, speculative calls, speculative calls whose first check failed, chained calls, chained calls
 * that needed more steps than enqueued, chained calls that enqueued more than needed, field increments that ran for nothing
 * in those, Jacobi steps in total}; reset != 0 clears the counters.  The results never depend on the form. */
int admp_scf_stats(admp_handle* h, int64_t* out8, int reset);

/* ---- measurement ---------------------------------------------------------------------------- */
/* When enabled every kernel launch is bracketed by HIP events on the handle's stream. */
int admp_profile_enable(admp_handle* h, int on);
/* restrict the bracketing to one kernel label (e.g. "pair_full"); NULL or "" = every kernel */
int admp_profile_filter(admp_handle* h, const char* label);
int admp_profile_reset(admp_handle* h);
/* number of distinct kernel labels seen; label / accumulated ms / launch count of entry idx */
int admp_profile_count(admp_handle* h);
int admp_profile_entry(admp_handle* h, int idx, const char** label, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ADMP_HIP_H */
