// Kernel launchers shared between the translation units of libadmp_hip.
// Every launcher enqueues on `st` and returns immediately; T is float or double.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "frame_math.h"
#include "pme_math.h"
#include "spline_math.h"

namespace admp {

// Per-nbonds scale tables (nbonds 0..15 -> scales[(nbonds-1) mod n], admp/pme.py:681-683) and the
// Fermi weight of the Thole-width switch at that pscale (admp/pme.py:337-348,411).
template <class T>
struct ScaleTab {
  T mm[16];   // mscale - 1
  T p[16];    // pscale
  T w0[16];
};

// i-grouped neighbour table: row i lists every partner of atom i (both directions of each i<j
// pair), entry = partner index | nbonds << 28.
struct NbrTable {
  int* rowptr = nullptr;   // na + 1
  int* col = nullptr;      // 2 * n_half
  int64_t n_half = 0;
  int64_t cap = 0;         // allocated entries of col
  int* order = nullptr;    // na: rows sorted by length inside windows of kRowWindow rows (launch_row_order), or nullptr
  // Site classes the table was compiled with: cls[a] = 1 for a charge-only atom (no dipole/quadrupole, not polarizable).
  // Entries whose PARTNER is charge-only carry kColMono and stand after all the others of their row, so the pair kernel
  // walks each row as two runs of uniform arithmetic.  k_prepare_sites checks cls against the sites of every call
  // (CLS_STALE / CLS_BETTER in the flag word next to E_NACT); a stale table is still a valid table of the general form.
  int* cls = nullptr;      // na
  int* order_plain = nullptr;   // na, only with cls: the row order WITHOUT the class grouping, for the kernels that have no
                                // use for classes (dispersion, Tang-Toennies: grouping costs them locality -- 0.49 -> 0.74 ms
                                // at 1M atoms on a borrowed, class-ordered table)
  // build scratch kept with the table (no allocation / synchronisation per rebuild): the second column buffer that the
  // out-of-place passes (row sort, class partition) write and swap with `col`, and the degree / cursor words
  int* col_alt = nullptr;
  int64_t cap_alt = 0;
  int* deg = nullptr;
  int deg_na = 0;
  // Slab ranks (round 4): a table built from positions holds the rows of the atoms near the rank's slab only (RowFilter);
  // built[i] = 1 for the atoms whose row exists (nullptr: every row).  The ownership pass refuses an evaluation in which an
  // atom without a row has become a home atom (the list is older than its skin allows).
  unsigned char* built = nullptr;
  void free_all() {             // the owner's buffers (a borrowed table is a copy of the lender's struct: never freed)
    for (int** p : {&rowptr, &col, &order, &cls, &order_plain, &col_alt, &deg})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (built) { (void)hipFree(built); built = nullptr; }
    n_half = cap = cap_alt = 0; deg_na = 0;
  }
};
constexpr int kRowWindow = 1024;
constexpr int kColMask = 0x0fffffff;
constexpr int kColMono = (int)0x80000000u;   // partner is charge-only (NbrTable::cls)
constexpr int kNbMask = 7;                   // covalent class in bits 28..30
__host__ __device__ inline int col_nb(int c) { return (c >> 28) & kNbMask; }
enum { CLS_STALE = 1 /* an atom marked charge-only no longer is: ignore kColMono */, CLS_BETTER = 2 /* atoms became charge-only */ };

struct Topology {
  int na = 0;
  int* axis_type = nullptr;   // na
  int* axis_idx = nullptr;    // na * 3
  int* excl_ptr = nullptr;    // na + 1
  int* excl_col = nullptr;
  int* excl_nb = nullptr;
  // inverse frame map: inv_idx[inv_ptr[a] .. inv_ptr[a+1]) = sites whose local frame involves atom a
  int* inv_ptr = nullptr;     // na + 1
  int* inv_idx = nullptr;
  // frame groups: when every connected component of the "takes part in the frame of" graph is a run of <= kMaxGroup
  // consecutive atoms (molecular liquids: one group per molecule), group g = atoms grp_ptr[g] .. grp_ptr[g+1]-1 and the
  // closing kernel evaluates each frame ONCE, by one thread per group (k_finish_groups); else ngroups = 0 (pull form)
  int* grp_ptr = nullptr;     // ngroups + 1
  int ngroups = 0;
  // row form of the closing kernel (k_finish_rows, round 4): one thread per ATOM, a workgroup takes a run of whole groups of at
  // most kFinishBlock atoms -- rows_blk[b] .. rows_blk[b+1]-1 -- so that everything a frame needs of its group mates is
  // exchanged through LDS; grp_of[i] = (first atom of i's group) << 2 | (group size - 1)
  int* rows_blk = nullptr;    // nrowblk + 1
  int nrowblk = 0;
  int* grp_of = nullptr;      // na
  // the same cut into runs of at most kGatherRun atoms: one workgroup of the gather each when the closing work rides in its
  // epilogue (small systems: one dispatch fewer per step)
  int* gath_blk = nullptr;    // ngathblk + 1
  int ngathblk = 0;
};
constexpr int kFinishBlock = 256;
constexpr int kGatherRun = 32;
// closing work in the gather's epilogue (k_gather_staged<.., FIN>): what launch_finish takes besides the gather's arguments
template <class T>
struct FinishArgs {
  const T* pol = nullptr;
  T kappa = 0;
  T* dQlocal = nullptr;
  double* energies = nullptr;   // nullptr = epilogue off
  int want_grad = 1;            // 0: energies only (the frame adjoint is skipped)
};
constexpr int kMaxGroup = 4;

// energies[] slots on the device
enum { E_REAL = 0, E_RECIP = 1, E_SELF = 2, E_PEN = 3, E_NACT = 4 /* int: number of polarizable sites (k_prepare_sites) */,
       E_SCRATCH = 5 /* energy of passes whose energy nobody reads */, E_SCF_RECIP = 6, E_FMAX = 7 /* bit pattern */,
       E_FMAX1 = 8 /* bit patterns: residuals of checks 1 .. E_CHAIN of a chained SCF (engine.hip) */, E_CHAIN = 6,
       E_SLOTS = E_FMAX1 + E_CHAIN,
       E_PARTS = 64 /* partial sums of the atom-side reciprocal energy (k_gather<.., true>): 32k workgroups adding into ONE
                       word serialise at the memory side (0.2 ms at 1M atoms); 64 words take them in parallel */,
       E_RED = E_SLOTS + E_PARTS /* 4 words: (real, recip, self, penalty) packed for the SUM all-reduce of a slab evaluation */,
       E_RPARTS = E_RED + 4 /* E_PARTS partial sums of the real-space energy (k_pair_full: one word per workgroup modulo
                               E_PARTS instead of every workgroup's atomic on E_REAL); the real-space energy is E_REAL + their sum */,
       E_WORDS = E_RPARTS + E_PARTS /* one half of the double-buffered energy block */ };

// optional epilogue of the gather (small systems) or of the closing kernel (large ones), speculative first SCF cycle on
// one rank: total dE/dU and its maximum, exactly what launch_field_finish computes, without a separate dispatch
template <class T>
struct FieldFin {
  const T* pol = nullptr;
  const T* Ucart = nullptr;
  const T* fld_pair = nullptr;
  const T* fld_recip = nullptr;                // closing-kernel variant only (the gather has it in registers)
  T kappa = 0;
  T* field = nullptr;
  unsigned long long* fmax_bits = nullptr;   // nullptr = epilogue off
  const Site<T>* sites = nullptr;            // field-gather variant only: the full site rows (its own rows may be compact)
};
// ---- atom_kernels.hip
template <class T>
void launch_site_classes(hipStream_t st, int na, const Site<T>* sites, int* cls);   // NbrTable::cls <- site_is_mono
template <class T>
void launch_prepare_sites(hipStream_t st, const Topology& top, const T* pos, const T* Qlocal, const T* Ucart,
                          const T* pol, const T* thole, const Box<T>& box, Site<T>* sites,
                          double* zero_next /* E_WORDS doubles cleared for the next evaluation, or nullptr */,
                          const RecipGeom<T>& g, int4* bases /* optional: stencil base indices per atom */,
                          int* act_list = nullptr /* optional: the atoms with pol > 0 (unordered) ... */,
                          int* act_count = nullptr /* ... and their number: an int the caller has zeroed */,
                          const int* cls = nullptr /* NbrTable::cls, checked against the sites: flags OR-ed into ... */,
                          int* cls_flags = nullptr /* ... this word (CLS_STALE / CLS_BETTER) */,
                          RQ4<T>* rq = nullptr /* optional: compact copy (position, charge) of every row */,
                          T* Ucopy = nullptr /* optional: Ucart is read-only; this array receives a copy of it */,
                          const int* list = nullptr /* optional: only the rows of these atoms ... */, int nlist = 0 /* ... this many */);
template <class T>
void launch_local_frames(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, T* out /* (na,3,3) */);
// field (Cartesian dE/dU) = pair (harmonic order) + recip (cartesian) + self + penalty; also max |field| over
// sites with pol > 0.001 (admp/pme.py:130,136) into *fmax_bits (order-preserving bit pattern of a non-negative real)
// Every per-atom / per-row launcher below takes an optional index list (`list`/`rows`, nullptr = atoms 0..n-1):
// with the x-slab decomposition a rank works on its "home" atoms only and n is the list length.
template <class T>
void launch_field_finish(hipStream_t st, int na, const Site<T>* sites, const T* pol, const T* Ucart, const T* fld_pair,
                         const T* fld_recip, T kappa, T* field, unsigned long long* fmax_bits, const int* list,
                         const int* n_dev = nullptr);
// Jacobi step over the polarizable sites (act list): dU = -field pol / D, U += dU (Cartesian array and packed harmonic
// copy), dU (harmonic order) into the pad words of the site row and into the compact row isites[slot] = {r, Q = 0, U = dU}
// that the increment's spread / gather read
template <class T>
void launch_jacobi_delta(hipStream_t st, int n_act, const int* act, const T* pol, const T* field, T* Ucart, Site<T>* sites,
                         Site<T>* isites, const unsigned long long* gate = nullptr /* device word: bit pattern of the residual */,
                         double gate_min = 0.0 /* with gate: a zero step unless residual >= gate_min */);
// self term + polarization penalty energies, self potential, local-frame adjoint, dE/dQ_local
template <class T>
void launch_finish(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, const Site<T>* sites,
                   const T* pol, const T* Ucart, int lpol, T kappa, T* pot, T* grad, T* dQlocal, double* energies,
                   const int* list, int nlist,
                   const FieldFin<T>& ff = FieldFin<T>(),
                   const int* slab_bits = nullptr /* slab rank: per-atom words, kSlabHome marks the rank's own atoms (with a
                                                     home list); the frame-group kernel then serves the groups of home atoms */);
// dispersion: site rows carrying one scalar channel (Q[0] = vals[i*stride+chan], no dipoles/quadrupoles); also
// accumulates coef * sum_i vals^2 into energies[E_SELF] (the self term of admp/disp_pme.py:254-279)
template <class T>
void launch_scalar_sites(hipStream_t st, int na, const T* pos, const T* vals, int stride, int chan, double self_coef,
                         Site<T>* sites, double* energies);
// grad[i] += vals[i*stride+chan] * v[i]   (scalar site: dE/dr = c * Jac . F1)
// nch > 1: grad[i] += sum_b vals[i*stride+chan+b] * v[b][i] with v[b] = v + b * 3 * na
template <class T>
void launch_scale_add(hipStream_t st, int na, const T* vals, int stride, int chan, const T* v, T* grad, int nch = 1);
struct SelfCoefs {
  double c[3] = {0, 0, 0};
};
// channels 0..nch-1 of vals at once: site rows of channel b at sites + b * na
template <class T>
void launch_scalar_sites_batch(hipStream_t st, int na, const T* pos, const T* vals, int stride, int nch,
                               const double* self_coefs, Site<T>* sites, double* energies,
                               const RecipGeom<T>* g = nullptr /* with bases: the mesh the rows will be spread on */,
                               int4* bases = nullptr /* optional: stencil base indices per atom, as launch_prepare_sites */);
// ---- dft_kernels.hip: direct-DFT mesh convolution for mesh sizes rocFFT only does with Bluestein (dft_math.h)
// tw = (cos, sin)(2 pi m / K[d]) tables of the three dimensions back to back (K[0] + K[1] + K[2] complex numbers)
int dft_tile_cols(int N);
// nb independent meshes (batch) are transformed by one launch each: mesh b at mesh + b * mesh_stride, its half spectrum
// at spec + b * spec_stride (strides in reals), its G table tabs.p[b] (dispersion PME: the C6 / C8 / C10 passes)
template <class T>
struct DftTabs {
  const T* p[3] = {nullptr, nullptr, nullptr};
};
// z and y lines of every x plane in one kernel (dft_kernels.hip, round 3); dft_zy_fits: the plane and its spectrum fit the LDS
template <class T>
bool dft_zy_fits(const int K[3]);
// what the forward plane kernel needs to build its planes itself (no spread kernel, no mesh): nb channels of na site rows back to back
template <class T>
struct PlaneSpread {
  int na = 0, lpol = 0;
  const Site<T>* sites = nullptr;
  const int4* bases = nullptr;      // stencil records of the atoms (k_prepare_sites), or nullptr: from the positions
  RecipGeom<T> g;
};
template <class T>
bool dft_zy_spread_fits(const int K[3], int na);
template <class T>
bool launch_dft_zy(hipStream_t st, const int K[3], const T* tw, T* mesh, T* spec, int inverse, int nb = 1, long mesh_stride = 0,
                   long spec_stride = 0, T* accum = nullptr, const PlaneSpread<T>* sp = nullptr /* forward only */);
template <class T>
bool launch_dft_z(hipStream_t st, const int K[3], const T* tw, T* mesh, T* spec, int inverse, int nb = 1,
                  long mesh_stride = 0, long spec_stride = 0,                                     // r2c / c2r along z
                  T* accum = nullptr /* c2r of one mesh: accum += result as well; returns whether that was done */);
template <class T>
void launch_dft_y(hipStream_t st, const int K[3], const T* tw, T* spec, int inverse, int nb = 1,
                  long spec_stride = 0);                                                          // in place along y
// along x: forward, spec *= gtab with energies[slot] += sum w G |S|^2, inverse -- one kernel, in place
template <class T>
void launch_dft_x_conv(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, double* energies,
                       int slot, int nb = 1, long spec_stride = 0);

// ---- pfa_kernels.hip: two-level (Good-Thomas) direct DFT for mesh dimensions N = N1 * N2 > 160 with a hard prime power N2
struct PfaAxis { int N, N1, N2; };            // N1 = 1: plain direct lines
struct PfaPlan {
  PfaAxis ax[3];
  int Khp;                                    // stored z columns: ax[2].N1 * (ax[2].N2 / 2 + 1)
  int tw_off[3];                              // twiddles of axis d at tw + tw_off[d] (complex entries): N2 entries, then N1
  const int* ptab[3];                         // device: index table of axis d (pfa_index_table)
};
bool pfa_split(int N, PfaAxis* out);          // false: no usable split
void pfa_index_table(const PfaAxis& a, int* t);       // t[n1 * N2 + n2] = position | n1 << 16
void pfa_freq_of_slot(const PfaAxis& a, int* f);      // f[slot] = frequency stored at that position of the axis
void pfa_freq_of_zcolumn(const PfaAxis& a, int* f);   // the same for the Khp stored z columns
template <class T>
void launch_pfa_z(hipStream_t st, const PfaPlan& p, const T* tw, T* mesh, T* spec, int inverse, int nb = 1, long mesh_stride = 0,
                  long spec_stride = 0);
template <class T>
void launch_pfa_y(hipStream_t st, const PfaPlan& p, const T* tw, T* spec, int inverse, int nb = 1, long spec_stride = 0);
template <class T>
void launch_pfa_x_conv(hipStream_t st, const PfaPlan& p, const T* tw, T* spec, const DftTabs<T>& tabs, double* energies, int slot,
                       int nb = 1, long spec_stride = 0);

// ---- fftx_kernels.hip: x lines forward * G (+ energy) * x lines inverse in one sweep, power-of-two K[0]
bool fftx_usable(int N);
template <class T>
void launch_fftx_conv(hipStream_t st, const int K[3], const T* tw, T* spec, const T* gtab, double* energies, int slot,
                      int khp = 0 /* row pitch of spec in complex numbers (0: K[2]/2+1) */,
                      int ny = 0 /* y rows held: spec and gtab are [K0][ny][..] (0: K[1]; a slab rank holds its own rows) */);
template <class T>
void launch_fftx_conv_batch(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, int nb,
                            long spec_stride /* reals between the spectra of consecutive channels */, double* energies, int slot,
                            int khp = 0, int ny = 0);

// ---- pair_kernels.hip
template <class T>
void launch_pair_full(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                      const ScaleTab<T>& tab, T kappa, int lpol, T* grad, T* pot, double* energies, const int* rows,
                      T* fld /* optional: also write the real-space dE/dU (speculative SCF pass) */,
                      int use_mono = 0 /* 1: charge-only sites take the reduced pair forms (pme_math.h); pot of such a
                                          ROW then holds only its monopole component: not for dE/dQ_local requests */,
                      const int* cls_flags = nullptr /* the flag word launch_prepare_sites wrote for THIS evaluation */,
                      const RQ4<T>* rq = nullptr /* compact position/charge rows of launch_prepare_sites */,
                      const T* tholes = nullptr /* the caller's per-atom thole array, or nullptr */);
// n_dev (optional): the row count lives on the device (na is then the upper bound the grid is sized for)
template <class T>
void launch_pair_field(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                       const ScaleTab<T>& tab, T kappa, T* fld_pair, const int* rows, const int* n_dev = nullptr,
                       const int* cls_flags = nullptr /* as launch_pair_full: charge-only partners take the short form */,
                       const RQ4<T>* rq = nullptr, const T* tholes = nullptr);
// incremental SCF: fld_pair[row] += sum_j T_ij dU_j over the polarizable partners, dU_j in the pad words of sites[j];
// `it` = the polarizable-polarizable sub-table (rows keyed by atom), `rows` = the n_rows polarizable sites
// Row i holds its kept entries at col[beg[i] .. end[i]) with beg = the neighbour table's own row offsets (borrowed pointer):
// the sub-table is written in ONE pass over the neighbour table -- no count pass, no prefix sum, no host read of the total
// (round 4; the two-pass build cost 0.57 ms per list rebuild at 1M atoms plus a hipMalloc / hipFree pair).  col is as long
// as the neighbour table's column array; for water 1/9 of it is used.
struct IndTable {
  const int* beg = nullptr;   // = NbrTable::rowptr of the table it was built from (not owned)
  int* end = nullptr;         // na
  int* col = nullptr;
  int64_t cap = 0;            // entries col can hold
  int na_cap = 0;             // rows `end` can hold
};
template <class T>
void launch_pair_field_ind(hipStream_t st, int n_rows, const IndTable& it, const Site<T>* sites, const Box<T>& box,
                           const ScaleTab<T>& tab, T kappa, T* fld_pair, const int* rows);
// An SCF field kernel riding in the x pass of a direct-DFT convolution (pair_kernels.hip k_xconv_pair): kind 1 = k_pair_field
// (rowptr / col of the neighbour table), 2 = k_pair_field_ind (rowptr / rowend / col of the polarizable sub-table)
template <class T>
struct FieldRider {
  int kind = 0, na = 0;
  const int* rowptr = nullptr;
  const int* rowend = nullptr;
  const int* col = nullptr;
  const Site<T>* sites = nullptr;
  Box<T> box;
  ScaleTab<T> tab;
  T kappa = 0;
  T* fld = nullptr;
  const int* rows = nullptr;
  unsigned nblocks = 0, grid = 0;
  const int* n_dev = nullptr;
  const int* cls_flags = nullptr;
  const RQ4<T>* rq = nullptr;
  const T* tholes = nullptr;
};
template <class T>
bool field_rider_full(FieldRider<T>& r, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                      const ScaleTab<T>& tab, T kappa, T* fld, const int* rows, const int* n_dev, const int* cls_flags,
                      const RQ4<T>* rq, const T* tholes);
template <class T>
bool field_rider_ind(FieldRider<T>& r, int n_rows, const IndTable& it, const Site<T>* sites, const Box<T>& box,
                     const ScaleTab<T>& tab, T kappa, T* fld, const int* rows);
template <class T>
void launch_dft_x_conv_rider(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, double* energies,
                             int slot, const FieldRider<T>& fr);
// (nbr_kernels.hip) inner table of an MD loop: the entries of `full` whose minimum-image distance is below rc, rows compacted
// in the order of `full`.  hipError_t as int; one host synchronisation (the entry count).
template <class T>
int prune_table(hipStream_t st, int na, const NbrTable& full, const T* pos, const Box<T>& box, double rc, int* rowptr_out,
                int* cnt, int* col_out, void** scratch, size_t* scratch_bytes, int64_t* total);
// (nbr_kernels.hip) ascending in-place sort of n ints; keys_tmp = n ints of scratch.  hipError_t as int.
int sort_ints(hipStream_t st, int* keys, int* keys_tmp, int n, void** scratch, size_t* scratch_bytes);
// (nbr_kernels.hip) it <- the polarizable-polarizable entries of nb; rows keyed by atom, empty for non-polarizable atoms.
// Returns a hipError_t as int; no host synchronisation (allocates only when the neighbour table has grown).
template <class T>
int build_ind_table(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, IndTable& it);
int pair_lanes_per_row(int n_rows);   // 4/8/16 by row count; env ADMP_PAIR_LPR overrides
// sumX[i] = sum_j dE_ij/d ln(au_ij), sumXw[i] = sum_j (same) * d ln(au_ij)/d thole_i  (pme_math.h pair_thole_logderiv)
template <class T>
void launch_thole_sums(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                       const ScaleTab<T>& tab, T* sumX, T* sumXw, const int* rows = nullptr, int n_rows = 0);
// cls16[nb] += sum over the pairs of covalent class nb of d(pair energy)/d(mscale); kind 0: multipolar PME (sites),
// 1: dispersion (pos, par = c6/c8/c10 per atom, pmax), 2: Tang-Toennies (pos, par = a/b/q/c6 per atom)
template <class T>
void launch_mscale_sums(hipStream_t st, int kind, int na, const NbrTable& nb, const Site<T>* sites, const T* pos,
                        const T* par, const Box<T>& box, int pmax, double* cls16,
                        double cutoff = 0.0 /* kinds 1, 2: listed pairs beyond it are skipped (admp_set_cutoff) */,
                        const int* rows = nullptr /* slab rank: its n_rows home rows */, int n_rows = 0);
// cls16[nb] += sum over the pairs of covalent class nb of d(pair energy)/d(pscale) (pme_math.h pair_pscale_deriv)
template <class T>
void launch_pscale_sums(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                        const ScaleTab<T>& tab, double* cls16, const int* rows = nullptr, int n_rows = 0);
// packed rows of the scalar pair kernels: position (3) + up to 4 per-atom parameters + pad, one aligned 8-real row per atom
template <class T>
struct alignas(8 * sizeof(T)) SRow {
  T v[8];
};
template <class T>
void launch_pack_scalar_rows(hipStream_t st, int na, int np /* parameters per atom: 3 (c6, c8, c10) or 4 (a, b, q, c6) */,
                             const T* pos, const T* par, SRow<T>* rows);
template <class T>
void launch_disp_pair(hipStream_t st, int na, const NbrTable& nb, const SRow<T>* srows, const Box<T>& box,
                      const ScaleTab<T>& tab, T kappa, int pmax, T* grad, double* energies,
                      const int* rows = nullptr /* the n_rows rows to evaluate (nullptr: all, in the table's order) */,
                      int n_rows = 0, double cutoff = 0.0 /* > 0: listed pairs beyond it are skipped (admp_set_cutoff) */);
template <class T>
void launch_tt_pair(hipStream_t st, int na, const NbrTable& nb, const SRow<T>* srows, const Box<T>& box,
                    const ScaleTab<T>& tab, T* grad, double* energies, const int* rows = nullptr, int n_rows = 0,
                    double cutoff = 0.0);

// ---- box gradient (dE/dbox at fixed Cartesian positions; on request only).  All sums are double device words.
// vir[9] += 1/2 sum_entries shift (x) dE_pair/dr_I over the pairs whose minimum image crosses the cell boundary
template <class T>
void launch_pair_virial(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                        const ScaleTab<T>& tab, T kappa, int lpol, double* vir,
                        const int* rows = nullptr /* slab rank: its n_rows home rows */, int n_rows = 0);
template <class T>
void launch_scalar_pair_virial(hipStream_t st, int tt, int na, const NbrTable& nb, const T* pos, const T* par,
                               const Box<T>& box, const ScaleTab<T>& tab, T kappa, int pmax, double* vir, double cutoff = 0.0,
                               const int* rows = nullptr /* slab rank: its n_rows home rows */, int n_rows = 0);
// vir[9] += shift (x) dE/d(frame vector) of the local frames whose axis vectors cross the cell boundary
template <class T>
void launch_frame_virial(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, const Site<T>* sites,
                         int lpol, T kappa, const T* pot, double* vir, const int* list = nullptr /* home sites */, int nlist = 0);
// tk[6] += sum_k w dG/dk^2 |S_k|^2 k (x) k (xx, yy, zz, xy, xz, yz), spectrum NOT yet multiplied by G (single rank)
template <class T>
void launch_kspace_virial(hipStream_t st, const int K[3], const double* box_inv, double volume, double kappa, int which,
                          int ref_order, const T* spec, double* tk,
                          int y0 = 0, int ny = 0 /* spec holds the y rows y0 .. y0+ny-1 as [K0][ny][K2/2+1] (0: all K[1] rows) */);
// xw[9] += sum_atoms x (x) dE_recip/dx ; yy[9] += sum_atoms dE_recip/dAop (recip_box_terms)
template <class T>
void launch_gather_virial(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, const T* phi,
                          double* xw, double* yy, const int* list = nullptr /* the na atoms to gather at (nullptr: 0 .. na-1) */);

// Mesh bricks for the LDS-tiled spread: dimension d is cut into nb[d] = ceil(K/16) bricks of 15..16
// (>= 6) points, brick b covering [b*K/nb, (b+1)*K/nb).  An atom is binned by the brick of its
// lowest stencil point; its 6-point stencil then reaches at most the next brick.
struct BrickGrid {
  int nb[3];
  int ncell;
};
__host__ __device__ inline BrickGrid make_bricks(const int K[3]) {
  BrickGrid b;
  for (int d = 0; d < 3; ++d) b.nb[d] = (K[d] + 15) / 16;
  b.ncell = b.nb[0] * b.nb[1] * b.nb[2];
  return b;
}
// brick index of mesh index i on an axis of K points cut into nb bricks [b*K/nb, (b+1)*K/nb)
__host__ __device__ inline int brick_of(int i, int nb, int K) { return ((i + 1) * nb - 1) / K; }
// Packed brick code of a stencil with lowest indices base[3]: 9 bits per axis = brick of `base`, bits 27..29 = the
// stencil (6 points) reaches into the next brick on that axis.  Computed once per atom (the integer divisions are the
// expensive part of binning) by k_prepare_sites, stored in the .w of the atom's int4 base record.
__host__ __device__ inline int brick_code(const int base[3], const int dims[3], const BrickGrid& bg) {
  int code = 0;
  for (int d = 0; d < 3; ++d) {
    const int b = brick_of(base[d], bg.nb[d], dims[d]);
    const int end = ((b + 1) * dims[d]) / bg.nb[d];          // first index past the brick of `base`
    code |= b << (9 * d);
    if (bg.nb[d] > 1 && base[d] + 5 >= end) code |= 1 << (27 + d);
  }
  return code;
}
// below this atom count the spread uses global atomics (8 lanes per atom) instead of binned LDS bricks
// (default 20000; env ADMP_SPREAD_BRICK_MIN overrides, 0 forces the brick path -- used by the parity tests)
int spread_brick_min_atoms();
// launch_spread(na atoms, this mesh) takes the binned brick kernel and leaves its lists in the BinScratch
template <class T>
bool spread_uses_bricks(int na, const RecipGeom<T>& g);
// scratch for the per-call binning: brick offsets cell_start[ncell+1] (+cursor copy) and the
// (atom, brick) entry list sorted[<= 8 na] (a stencil touches at most 2 bricks per axis)
struct BinScratch {
  int* cell_start = nullptr;   // ncell + 1
  int* cursor = nullptr;       // ncell + 1: per-brick entry counts (input of the scan)
  int* fillcur = nullptr;      // ncell + 1: per-brick fill cursors of the second binning pass
  bool counters_zero = false;  // cursor / fillcur are all zero (the brick kernel clears its own words when it is done)
  int* sorted = nullptr;       // 8 * na
  void* scan_tmp = nullptr;
  size_t scan_bytes = 0;
};

// ---- recip_kernels.hip
// LDS-brick spread: bins the atoms, accumulates every brick in LDS, writes each mesh point exactly once
// (no memset, no global atomics).  Returns a hipError_t as int.
template <class T>
int launch_spread(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, BinScratch& bs,
                  T* mesh, const int* list, const int4* bases = nullptr /* from launch_prepare_sites, or recomputed */,
                  int nb = 1 /* batch (scan kernel only): site rows of b at sites + b * na, its mesh at mesh + b * mesh size */,
                  int reuse_bins = 0 /* binned kernel: positions unchanged since the previous call, keep its brick lists */,
                  int tight = 1 /* binned kernel, f32: fixed-point scale from the stencil-window bound (0: entry count; SCF increments) */);
size_t spread_scan_bytes(int ncell);
// k-space layout [K0][ny][K2/2+1] holding mesh rows y0 .. y0+ny-1 (ny = K1, y0 = 0 on one rank).
// which: 1 = electrostatics (Ck_1, gamma point dropped, x DIELECTRIC), 6/8/10 = dispersion kernels
template <class T>
void launch_gtab(hipStream_t st, const int K[3], int y0, int ny, const double* box_inv, double volume, double kappa,
                 int which, T* gtab, int ref_order = 0 /* the reference's k-point table, see k_gtab */,
                 const int* fmap = nullptr /* slot-ordered spectrum (pfa_kernels.hip): device table of the frequency stored
                                              at every x slot (K0), y slot (K1) and z column (nh), back to back */,
                 int nh = 0 /* with fmap: number of stored z columns */);
// spec <- spec * gtab ; energies[slot] += sum_k w_k (gtab/2) |S_k|^2
template <class T>
void launch_kspace(hipStream_t st, const int K[3], int ny, const T* gtab, T* spec /* interleaved complex */,
                   double* energies, int slot);
template <class T>
void launch_gather(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, const T* phi, T* pot,
                   T* grad, const int* list, T* fld_recip /* optional: cartesian reciprocal dE/dU */,
                   const FieldFin<T>& ff = FieldFin<T>(),
                   double* e_recip = nullptr /* optional: += 1/2 sum_i Q_tot,i . dE_recip/dQ_i = the reciprocal energy */,
                   const Topology* top = nullptr /* with fin: the workgroups take the runs of whole frame groups top->gath_blk */,
                   const Box<T>* box = nullptr, const FinishArgs<T>& fin = FinishArgs<T>());
template <class T>
void launch_gather_field(hipStream_t st, int na, const Site<T>* sites, const RecipGeom<T>& g, const T* phi,
                         T* fld_recip, const int* list, int nb = 1 /* batch: phi of b at phi + b * mesh size, fld + b * 3 * na */,
                         const int* n_dev = nullptr /* atom count on the device (na = grid bound) */,
                         const int* add_to = nullptr /* rows are compact: ADD the result to fld_recip[3 * add_to[slot]] */,
                         const FieldFin<T>& ff = FieldFin<T>() /* small systems: total dE/dU and its maximum of the rows' atoms
                                                                 ride along (launch_field_finish's work; nb = 1) */);
// a += b over n mesh points
template <class T>
void launch_mesh_add(hipStream_t st, long n, T* a, const T* b);
// the binning passes of launch_spread on their own (count, scan, fill): leaves the brick lists of the n listed atoms in bs;
// `bases` must hold the stencil records of those atoms (sites may then be nullptr).  hipError_t as int.
template <class T>
int launch_bin_bricks(hipStream_t st, int na, const Site<T>* sites, const RecipGeom<T>& g, BinScratch& bs, const int* list,
                      const int4* bases);
// ---- disp_kernels.hip: the scalar channels of dispersion PME through ONE spread and ONE gather
template <class T>
void launch_atom_bases(hipStream_t st, int na, const T* pos, const RecipGeom<T>& g, int4* bases);
// mesh of channel c (vals[i * stride + c]) at mesh + c * mesh_stride, from the brick lists in bs.  hipError_t as int.
template <class T>
int launch_spread_scalar(hipStream_t st, int nch, const T* pos, const T* vals, int stride, const RecipGeom<T>& g,
                         const BinScratch& bs, T* mesh, long mesh_stride);
// grad[i] += sum_c vals[i][c] Jac . grad phi_c(r_i) for the n listed atoms
template <class T>
void launch_gather_scalar(hipStream_t st, int nch, int na, const T* pos, const T* vals, int stride, const RecipGeom<T>& g,
                          const T* phi, long mesh_stride, T* grad, const int* list,
                          int interleaved = 0 /* phi = [mesh point][nch]: the channels of a point side by side (launch_interleave) */,
                          const int* types = nullptr /* typed form: atom i gathers, with weight 1, from mesh types[i] */);
// typed dispersion meshes (disp_kernels.hip k_spread_bricks_typed): unit sources into mesh types[i]; nt = 1..3
template <class T>
int launch_spread_typed(hipStream_t st, int nt, const T* pos, const int* types, const RecipGeom<T>& g, const BinScratch& bs,
                        T* mesh, long mesh_stride);
// x pass of the typed form (fftx_kernels.hip k_fftx_mix): spec holds nt type meshes; at every k the channel structure
// factors S_p = sum_t ctab[p][t] S_t are formed, E += w G_p |S_p|^2, and the types get back psi_t = sum_p ctab[p][t] G_p S_p
struct MixTab { double c[3][4]; int nch, nt; };      // c[power][type]; columns of unused (padding) types are zero
// direct-DFT meshes (dft_kernels.hip k_dft_x_mix): the same combination between the forward and inverse x lines
template <class T>
void launch_dft_x_mix(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, const MixTab& mix,
                      long spec_stride, double* energies, int slot);
// w[i][t] = (types[i] == t): the typed form through the kernels that take per-atom weights (small systems)
void launch_type_counts(hipStream_t st, int na, const int* types, int* counts4 /* zeroed by the caller */);
template <class T>
void launch_onehot(hipStream_t st, int na, int nt, const int* types, T* w);
template <class T>
void launch_types_check(hipStream_t st, int na, const T* vals, int stride, const int* types, const MixTab& mix, double* bad);
template <class T>
void launch_fftx_mix(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, const MixTab& mix,
                     long spec_stride, double* energies, int slot, int khp);
// dst[i * nch + c] = src[c * stride + i], i < n: the nch meshes of a dispersion call side by side per mesh point, so that the
// gather fetches all channels of a stencil point with ONE load instruction
template <class T>
void launch_interleave(hipStream_t st, int nch, long n, const T* src, long stride, T* dst);
// energies[E_SELF] += sum over the n listed atoms and the nch channels of self_coefs[c] vals[i][c]^2
template <class T>
void launch_scalar_self(hipStream_t st, int nch, int na, const T* vals, int stride, const int* list, const double* self_coefs,
                        double* energies);
// out[i * stride + chan] += phi(r_i) + extra * vals[i * stride + chan]  (mesh potential at the atoms: dE_recip/dc_i)
template <class T>
void launch_gather_value(hipStream_t st, int na, const T* pos, const T* vals, int stride, int chan, const RecipGeom<T>& g,
                         const T* phi, double extra, T* out, const int* list = nullptr /* the na atoms (nullptr: 0 .. na-1) */);
// (pair_kernels.hip) per-atom parameter derivatives of the scalar pair terms: tt = 0 dispersion, out (na,3) = dE/dc6,c8,c10;
// tt = 1 Tang-Toennies, out (na,4) = dE/d(a, b, q, c6)
template <class T>
void launch_scalar_pair_pgrad(hipStream_t st, int tt, int na, const NbrTable& nb, const T* pos, const T* par, const Box<T>& box,
                              const ScaleTab<T>& tab, T kappa, int pmax, T* out, double cutoff = 0.0,
                              const int* rows = nullptr, int n_rows = 0);
// ---- cell_kernels.hip: positions -> half pair list (cell list), two phases so that the caller can size `pairs`
struct CellScratch {
  int n[3] = {0, 0, 0};
  int* start = nullptr; int* cursor = nullptr; int* sorted = nullptr;
  long long* count = nullptr; long long* offs = nullptr;
  void* spos = nullptr;        // cell-sorted packed records (x, y, z, atom id), 4 reals (<= 32 B) per atom
  int* deg4 = nullptr;         // 4 partial row lengths per atom (one per lane of k_cell_rows)
  void* scan_tmp = nullptr; size_t scan_bytes = 0;
  int cap_atoms = 0, cap_cells = 0;
  int ensure(int na, int ncell);
  void release();
};
template <class T>
int cell_count_pairs(hipStream_t st, int na, const T* pos, const Box<T>& box, const double* heights, double rc,
                     CellScratch& cs, long long* n_pairs);
template <class T>
int cell_fill_pairs(hipStream_t st, int na, const T* pos, const Box<T>& box, double rc, CellScratch& cs, int* pairs);
// positions -> the pair kernels' neighbour table directly (both directions, nbonds packed), no pair array in between
// rows to build (slab ranks): the atoms whose approximate stencil base plane floor(frac_x K0) - 2 lies in the periodic
// interval [lo, lo + width) of the K0 planes along x; on == 0: every row
struct RowFilter {
  int on = 0;
  int K0 = 0, lo = 0, width = 0;
};
template <class T>
int cell_build_table(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, const double* heights,
                     double rc, CellScratch& cs, NbrTable& nb, const RowFilter& rf = RowFilter());

// ---- slab_kernels.hip: x-slab decomposition (multi-GPU) ------------------------------------------------------------------
// per-atom word of a rank's view of the decomposition (see slab_kernels.hip)
constexpr int kSlabHome = 1 << 30, kSlabPolar = 1 << 29;
constexpr int kSlabMaxRanks = 28, kSlabMaxCols = 3 + 4 * kSlabMaxRanks;
// columns of the ordered compaction: column c keeps x = seq[c][p] (seq[c] == nullptr: x = p), p < len[c], where
// (bits[x] & mask[c]) == want[c]
struct SlabCols {
  int ncols = 0;
  int len[kSlabMaxCols];
  const int* seq[kSlabMaxCols];
  const int* src[kSlabMaxCols];    // the flag words this column tests (nullptr: the `bits` array of the decomposition)
  int mask[kSlabMaxCols];
  int want[kSlabMaxCols];
  int binned[kSlabMaxCols];        // 1: the column is filled by the one-pass bin compaction (SlabBins), not by its own pass
};
// The per-peer columns of a decomposition -- imports from / exports to / atoms taken over from / given away to every rank --
// plus the home and polarizable-home columns, as BINS of one ordered compaction pass over the atoms (round 3 ran one pass over
// all atoms per column: 7 passes on 2 ranks, 31 on 8 -- the part of a rank's step that GREW with the rank count).  c_*: the
// column (index into SlabCols / totals / lists) a bin writes, -1 = absent.
struct SlabBins {
  int N = 0, me = 0;
  int c_home = -1, c_act = -1;
  int c_imp[kSlabMaxRanks], c_exp[kSlabMaxRanks], c_min[kSlabMaxRanks], c_mout[kSlabMaxRanks];
};
// segments of a joined per-peer list: segment s = column col[s] of the compaction, entries off[s] .. off[s+1]-1 of the result
struct SlabSegs {
  int n = 0;
  int off[kSlabMaxRanks + 1];
  int col[kSlabMaxRanks];
};
// owner[na] / bits[na] of this evaluation and the compacted columns lists[c * na ..]; totals[ncols] stay on the device
// (the caller reads them once).  counts: ncols * slab_compact_blocks(max len) ints of scratch.  hipError_t as int.
// owner_prev (optional): the owners of the previous evaluation; mig[na] then receives, for the atoms that changed hands,
// 1 << previous owner (atoms this rank took over) or kSlabHome | 1 << new owner (atoms it gave away), else 0
// totals[cs.ncols] (one word past the columns' totals) comes back non-zero when a home atom has no row in a filtered table
int launch_slab_decompose(hipStream_t st, int na, const NbrTable& nb, const Topology& top, const int4* bases, const void* pol,
                          int prec, int width, int K0, int X0, int nranks, int me, int* owner, int* bits,
                          const SlabCols& cs, const SlabBins& sb, int* counts, int* totals, int* lists,
                          const int* owner_prev = nullptr, int* mig = nullptr);
int slab_compact_blocks(int maxlen);
void launch_slab_concat(hipStream_t st, const SlabSegs& segs, const int* lists, long col_stride, int* out);
template <class T>
void launch_rows_gather(hipStream_t st, int n, int w, const int* idx, const T* src, T* out);
// mode 0: dst[idx[k]] = in[k]; 1: dst[idx[k]] += in[k]; 2: dst[idx[k]] = 0 (in unused)
template <class T>
void launch_rows_scatter(hipStream_t st, int mode, int n, int w, const int* idx, const T* in, T* dst);
// what 0: Cartesian induced dipoles of the listed atoms; what 1: the last Jacobi step's change (site pad words)
template <class T>
void launch_halo_u_pack(hipStream_t st, int n, int what, const int* idx, const T* Ucart, const Site<T>* sites, T* out);
template <class T>
void launch_halo_u_unpack(hipStream_t st, int n, int what, const int* idx, const T* in, T* Ucart, Site<T>* sites);
// spec[nx][K1][pitch] complex (nh used per row) <-> all-to-all buffer (block of peer t: [nx][ny_t][nh]); dir 0 pack, 1 unpack
template <class T>
void launch_transpose_pack(hipStream_t st, int nx, int K1, int nh, int pitch, int nranks, int dir, T* spec, T* buf);
// out[4] = (real, recip, self, penalty) of the energy block e; recip_slot >= 0: that word, -1: the sum of the E_PARTS words
void launch_energy_pack(hipStream_t st, const double* e, int recip_slot, double* out);

// ---- nbr_kernels.hip
// order[] <- the rows of every window of W <= kRowWindow consecutive rows sorted by neighbour count (ties by index).  The
// pair kernels give each row a fixed group of lanes, so a wavefront runs as long as its longest row: with rows of
// equal length side by side the lanes stay busy (water, rc 4 A: 85 % -> 99 % of the lane-iterations useful), while the
// window keeps the rows' site / output accesses local.
void launch_row_order(hipStream_t st, int na, const int* rowptr, int* order, const int* cls = nullptr);
// re-parts every row of nb by nb.cls (see NbrTable); replaces nb.col.  Returns hipError_t as int.
int launch_class_partition(hipStream_t st, int na, NbrTable& nb);
// builds nb from (n_rows, 2) device pairs; scratch (deg/cursor) is managed inside. Returns hipError_t as int.
int build_neighbour_table(hipStream_t st, const Topology& top, int64_t n_rows, const int* pairs_dev, NbrTable& nb,
                          void** scratch, size_t* scratch_bytes);

// ---- md_kernels.hip: harmonic bonded terms and velocity-Verlet half steps of the MD drivers
template <class T>
void launch_md_bonded(hipStream_t st, int nb, const int* bidx, const T* bpar, int na, const int* aidx, const T* apar, const T* pos,
                      const Box<T>& box, T* grad, double* E);
template <class T>
void launch_md_kick_drift(hipStream_t st, int n, T* pos, T* vel, const T* grad, const T* inv_mass, double half_dt_acc, double dt,
                          double* ekin);

}  // namespace admp
