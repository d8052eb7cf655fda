// Real-space pair kernels (reference admp/pme.py:628-729 pme_real + :479-624 pme_real_kernel,
// admp/disp_pme.py:126-216, admp/pairwise.py:45-113).
//
// Layout: the pair list is compiled (nbr_kernels.hip) into an i-grouped table holding BOTH
// directions of every i<j pair.  LPR lanes of a wavefront share one row: each lane walks every
// LPR-th partner, gathers its 20-real site row with 16-byte loads, evaluates the pair from the
// row atom's side only (energy halved), and the row's 15 accumulators (dE/dr 3, dE/dQ 9, dE/dU 3)
// are folded across the LPR lanes with DPP-class shuffles and written once -- no global atomics,
// bitwise reproducible for a fixed table.
#include <cstdlib>

#include "disp_math.h"
#include "dft_lines.h"
#include "launch.h"
#include "reduce.h"

namespace admp {

constexpr int kPairBlock = 256;
#ifndef ADMP_FULL_BLOCK
#define ADMP_FULL_BLOCK 256
#endif
constexpr int kFullBlock = ADMP_FULL_BLOCK;      // workgroup of k_pair_full (its waves end at one barrier: the energy sum)

template <class T, int LPR>
__device__ __forceinline__ T row_reduce(T v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <class T>
__device__ __forceinline__ void stage_tab(const ScaleTab<T>& tab, T* s) {
  if (threadIdx.x < 16) {
    s[threadIdx.x] = tab.mm[threadIdx.x];
    s[16 + threadIdx.x] = tab.p[threadIdx.x];
    s[32 + threadIdx.x] = tab.w0[threadIdx.x];
  }
  __syncthreads();
}

template <class T, bool LPOL, int LPR, int MINW>
__global__ __launch_bounds__(kFullBlock, MINW) void k_pair_full(int na, const int* __restrict__ rowptr,
                                                          const int* __restrict__ col,
                                                          const Site<T>* __restrict__ sites, Box<T> box,
                                                          ScaleTab<T> tab, T kappa, T* __restrict__ grad,
                                                          T* __restrict__ pot, double* energies,
                                                          const int* __restrict__ rows, T* __restrict__ fld,
                                                          unsigned nblocks, int use_mono,
                                                          const int* __restrict__ cls_flags,
                                                          const RQ4<T>* __restrict__ rq, const T* __restrict__ tholes) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  const long blk = xcd_block(blockIdx.x, nblocks);
  const long t = (blk < 0 ? (long)na * LPR : blk * kFullBlock) + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  // `na` counts the rows this launch owns; with a row list (multi-GPU: the rank's home atoms) slot -> atom
  const int row = slot < na ? (rows ? rows[slot] : slot) : na;
  T g[3] = {0, 0, 0}, P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, F[3] = {0, 0, 0};
  double e = 0.0;
  if (slot < na) {
    const Site<T> I = sites[row];
    const int beg = rowptr[row] + sub, end = rowptr[row + 1];
    T* Fp = (LPOL && fld) ? F : nullptr;
    // Charge-only sites (pme_math.h).  The neighbour table was compiled with the atoms' classes (NbrTable::cls): the entries
    // of a row whose partner is charge-only carry kColMono and stand behind the others, and `rows` groups the rows by class
    // (launch_row_order) -- so each lane walks two runs of uniform arithmetic, and the lanes of a wavefront walk the same
    // kind of run.  The row's own class is read from the site (k_prepare_sites marks it every call, so it is always true);
    // when the table's classes are out of date (CLS_STALE: an atom it has as charge-only is not any more) the marks are
    // ignored and every partner takes the general form -- valid for any site.
    const bool use = use_mono && !(*cls_flags & CLS_STALE);
    const bool imono = use_mono && site_is_mono(I);
    // The loops are software-pipelined: an iteration's col entry (and, in the light loops, the partner's position / charge
    // words) are fetched one or two iterations ahead -- the dependent chain col -> site -> arithmetic otherwise leaves the
    // waves waiting (SQ_WAIT_ANY 71 % of the wave cycles at 4 waves per SIMD once the arithmetic per partner had shrunk).
    // Out-of-range prefetches read entry 0 / site 0: valid addresses, results unused.
    int k = beg;
    if (!imono) {
      {
        int c = k < end ? col[k] : 0;
#pragma unroll 1
        for (; k < end; k += LPR) {
          if (use && c < 0) break;
          const int cn = k + LPR < end ? col[k + LPR] : 0;
          const Site<T> J = sites[c & kColMask];
          const int nb = col_nb(c);
          const PairScales<T> sc = {s_tab[nb], s_tab[16 + nb], s_tab[32 + nb]};
          e += (double)pair_energy_grad<T, LPOL, false>(box, I, J, sc, kappa, g, P, Fp, nullptr, nullptr);
          c = cn;
        }
      }
      if (use) {   // (rq is only there in this mode)
        int c0 = k < end ? col[k] : 0, c1 = k + LPR < end ? col[k + LPR] : 0;
        RQ4<T> q0 = rq[c0 & kColMask];
        T t0 = tholes ? tholes[c0 & kColMask] : T(0);
#pragma unroll 1
        for (; k < end; k += LPR) {
          const int c2 = k + 2 * LPR < end ? col[k + 2 * LPR] : 0;
          const RQ4<T> q1 = rq[c1 & kColMask];
          const T t1 = tholes ? tholes[c1 & kColMask] : T(0);
          const int nb = col_nb(c0);
          const PairScales<T> sc = {s_tab[nb], s_tab[16 + nb], s_tab[32 + nb]};
          e += (double)pair_full_mono<T, LPOL>(box, I, q0.v, q0.v[3], t0, sc, kappa, g, P, Fp);
          c0 = c1; c1 = c2; q0 = q1; t0 = t1;
        }
      }
    } else {
      {
        int c = k < end ? col[k] : 0;
#pragma unroll 1
        for (; k < end; k += LPR) {
          if (use && c < 0) break;
          const int cn = k + LPR < end ? col[k + LPR] : 0;
          const Site<T> J = sites[c & kColMask];
          const int nb = col_nb(c);
          const PairScales<T> sc = {s_tab[nb], s_tab[16 + nb], s_tab[32 + nb]};
          e += (double)pair_mono_full<T, LPOL>(box, I.r, I.Q[0], I.thole, J, sc, kappa, g, P[0]);
          c = cn;
        }
      }
      if (use) {
        int c0 = k < end ? col[k] : 0, c1 = k + LPR < end ? col[k + LPR] : 0;
        RQ4<T> q0 = rq[c0 & kColMask];
#pragma unroll 1
        for (; k < end; k += LPR) {
          const int c2 = k + 2 * LPR < end ? col[k + 2 * LPR] : 0;
          const RQ4<T> q1 = rq[c1 & kColMask];
          e += (double)pair_mono_mono<T>(box, I.r, I.Q[0], q0.v, q0.v[3], s_tab[col_nb(c0)], kappa, g, P[0]);
          c0 = c1; c1 = c2; q0 = q1;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) g[k] = row_reduce<T, LPR>(g[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) P[k] = row_reduce<T, LPR>(P[k]);
  if (LPOL && fld) {   // speculative SCF pass: the real-space dE/dU comes out of the same evaluation
#pragma unroll
    for (int k = 0; k < 3; ++k) F[k] = row_reduce<T, LPR>(F[k]);
  }
  if (slot < na && sub == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) grad[3 * row + k] = g[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) pot[9 * row + k] = P[k];
    if (LPOL && fld) { fld[3 * row] = F[0]; fld[3 * row + 1] = F[1]; fld[3 * row + 2] = F[2]; }
  }
  e = block_reduce_sum<kFullBlock>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[E_RPARTS + (blockIdx.x & (E_PARTS - 1))], 0.5 * e);
}

// (bid: the workgroup's index in a launch of its own -- blockIdx.x of k_pair_field, or its rank among the field workgroups
// that ride in an x pass of the mesh convolution, k_xconv_pair below)
template <class T, int LPR>
__device__ __forceinline__ void pair_field_block(unsigned bid, int na, const int* __restrict__ rowptr,
                                                 const int* __restrict__ col, const Site<T>* __restrict__ sites,
                                                 const Box<T>& box, const ScaleTab<T>& tab, T kappa, T* __restrict__ fld,
                                                 const int* __restrict__ rows, unsigned nblocks, const int* __restrict__ n_dev,
                                                 const int* __restrict__ cls_flags, const RQ4<T>* __restrict__ rq,
                                                 const T* __restrict__ tholes) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  if (n_dev) {                              // row count known on the device only (the polarizable-site list of this call):
    na = min(na, *n_dev);                   // re-derive the XCD block ranges from it, or only the first XCDs would get rows
    nblocks = (unsigned)(((long)na * LPR + kPairBlock - 1) / kPairBlock);
  }
  const long blk = xcd_block(bid, nblocks);
  const long t = (blk < 0 ? (long)na * LPR : blk * kPairBlock) + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  const int row = slot < na ? (rows ? rows[slot] : slot) : na;
  T F[3] = {0, 0, 0};
  if (slot < na) {
    const Site<T> I = sites[row];
    const int end = rowptr[row + 1];
    // two runs per row, as in k_pair_full: partners with higher moments, then (kColMono) the charge-only ones, whose field
    // is one radial coefficient along the pair axis; both loops fetch ahead (see k_pair_full)
    const bool use = cls_flags && !(*cls_flags & CLS_STALE);
    int k = rowptr[row] + sub;
    {
      int c = k < end ? col[k] : 0;
#pragma unroll 1
      for (; k < end; k += LPR) {
        if (use && c < 0) break;
        const int cn = k + LPR < end ? col[k + LPR] : 0;
        const int nb = col_nb(c);
        const Site<T> J = sites[c & kColMask];
        const PairScales<T> sc = {T(0), s_tab[16 + nb], s_tab[32 + nb]};
        pair_field(box, I, J, sc, kappa, F);
        c = cn;
      }
    }
    if (use) {
      int c0 = k < end ? col[k] : 0, c1 = k + LPR < end ? col[k + LPR] : 0;
      RQ4<T> q0 = rq[c0 & kColMask];
      T t0 = tholes ? tholes[c0 & kColMask] : T(0);
#pragma unroll 1
      for (; k < end; k += LPR) {
        const int c2 = k + 2 * LPR < end ? col[k + 2 * LPR] : 0;
        const RQ4<T> q1 = rq[c1 & kColMask];
        const T t1 = tholes ? tholes[c1 & kColMask] : T(0);
        const int nb = col_nb(c0);
        pair_field_mono<T>(box, I.r, I.thole, q0.v, q0.v[3], t0, s_tab[16 + nb], s_tab[32 + nb], kappa, F);
        c0 = c1; c1 = c2; q0 = q1; t0 = t1;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) F[k] = row_reduce<T, LPR>(F[k]);
  if (slot < na && sub == 0) {
    fld[3 * row] = F[0]; fld[3 * row + 1] = F[1]; fld[3 * row + 2] = F[2];
  }
}
template <class T, int LPR>
__global__ __launch_bounds__(kPairBlock) void k_pair_field(int na, const int* __restrict__ rowptr,
                                                           const int* __restrict__ col,
                                                           const Site<T>* __restrict__ sites, Box<T> box,
                                                           ScaleTab<T> tab, T kappa, T* __restrict__ fld,
                                                           const int* __restrict__ rows, unsigned nblocks,
                                                           const int* __restrict__ n_dev,
                                                           const int* __restrict__ cls_flags,
                                                           const RQ4<T>* __restrict__ rq, const T* __restrict__ tholes) {
  pair_field_block<T, LPR>(blockIdx.x, na, rowptr, col, sites, box, tab, kappa, fld, rows, nblocks, n_dev, cls_flags, rq, tholes);
}

// Incremental SCF (engine.hip): fld[row] += sum over the polarizable partners of T_ij . dU_j, rows = the polarizable
// sites.  dU_j (global harmonic order) rides in the pad words of the partner's site row (k_jacobi_delta).  The kernel
// walks the polarizable-polarizable SUB-table (irow / iend / icol, rows keyed by atom, built by build_ind_table when the
// neighbour table or the set of polarizable sites changes): for water the O-O pairs, 1/9 of the directed entries.
template <class T, int LPR>
__device__ __forceinline__ void pair_field_ind_block(unsigned bid, int na, const int* __restrict__ irow,
                                                     const int* __restrict__ iend, const int* __restrict__ icol,
                                                     const Site<T>* __restrict__ sites, const Box<T>& box,
                                                     const ScaleTab<T>& tab, T kappa, T* __restrict__ fld,
                                                     const int* __restrict__ rows) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  const unsigned nblocks = (unsigned)(((long)na * LPR + kPairBlock - 1) / kPairBlock);
  const long blk = xcd_block(bid, nblocks);
  const long t = (blk < 0 ? (long)na * LPR : blk * kPairBlock) + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  const int row = slot < na ? rows[slot] : -1;
  T F[3] = {0, 0, 0};
  if (row >= 0) {
    const T rI[3] = {sites[row].r[0], sites[row].r[1], sites[row].r[2]};
    const T p6I = sites[row].p6, thI = sites[row].thole;
    const int end = iend[row];
#pragma unroll 1
    for (int k = irow[row] + sub; k < end; k += LPR) {
      const int c = icol[k];
      const Site<T>& J = sites[c & kColMask];
      const T rJ[3] = {J.r[0], J.r[1], J.r[2]};
      const T dU[3] = {J.pad[0], J.pad[1], J.pad[2]};
      pair_field_ind<T>(box, rI, p6I, thI, rJ, J.p6, J.thole, dU, s_tab[32 + (col_nb(c))], kappa, F);
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) F[q] = row_reduce<T, LPR>(F[q]);
  if (row >= 0 && sub == 0) { fld[3 * row] += F[0]; fld[3 * row + 1] += F[1]; fld[3 * row + 2] += F[2]; }
}
template <class T, int LPR>
__global__ __launch_bounds__(kPairBlock) void k_pair_field_ind(int na, const int* __restrict__ irow,
                                                               const int* __restrict__ iend,
                                                               const int* __restrict__ icol,
                                                               const Site<T>* __restrict__ sites, Box<T> box,
                                                               ScaleTab<T> tab, T kappa, T* __restrict__ fld,
                                                               const int* __restrict__ rows) {
  pair_field_ind_block<T, LPR>(blockIdx.x, na, irow, iend, icol, sites, box, tab, kappa, fld, rows);
}

// ---- a field kernel riding in the x pass of a direct-DFT convolution (round 4, small systems on one stream) ------------------
// The SCF field kernels (7 us each at 3072 atoms) do not depend on the mesh chain they stand in front of.  Instead of a
// dispatch of their own -- or a side stream, whose fork + join cost more than they hide at this size -- their workgroups are
// appended to the grid of the x pass (same block size; 124 and 134-164 registers): the launch runs both kinds side by side.
// Workgroups with blockIdx.x < nbx are x-pass tiles (dft_lines.h), the others field workgroups number
// (blockIdx.x - nbx) * gridDim.y + blockIdx.y.
template <class T, int LPR>
__global__ __launch_bounds__(kPairBlock) void k_xconv_pair(XConvArgs<T> xa, FieldRider<T> fr, int nbx) {
  static_assert(kPairBlock == kDftBlock, "the two kinds of workgroups share one launch");
  if ((int)blockIdx.x < nbx) {
    dft_x_conv_body<T, 2, 1>(xa, blockIdx.x, blockIdx.y, 0);
    return;
  }
  const unsigned bid = (blockIdx.x - (unsigned)nbx) * gridDim.y + blockIdx.y;
  if (bid >= fr.grid) return;                          // workgroup-uniform
  if (fr.kind == 1)
    pair_field_block<T, LPR>(bid, fr.na, fr.rowptr, fr.col, fr.sites, fr.box, fr.tab, fr.kappa, fr.fld, fr.rows, fr.nblocks,
                             fr.n_dev, fr.cls_flags, fr.rq, fr.tholes);
  else
    pair_field_ind_block<T, LPR>(bid, fr.na, fr.rowptr, fr.rowend, fr.col, fr.sites, fr.box, fr.tab, fr.kappa, fr.fld, fr.rows);
}

// dispersion / Tang-Toennies: scalar pair terms, same row layout.  Position and parameters of an atom are packed into ONE
// 8-real row (k_pack_scalar_rows, per call): a partner costs one 32-byte (f32) fetch instead of two unrelated 12 / 16-byte ones
// from the caller's (Na,3) / (Na,NP) arrays -- the kernels are bound by those dependent fetches, not by their arithmetic.
template <class T>
__global__ __launch_bounds__(256) void k_pack_scalar_rows(int na, int np, const T* __restrict__ pos, const T* __restrict__ par,
                                                          SRow<T>* __restrict__ rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= na) return;
  SRow<T> r;
  r.v[0] = pos[3 * i]; r.v[1] = pos[3 * i + 1]; r.v[2] = pos[3 * i + 2];
#pragma unroll
  for (int k = 0; k < 4; ++k) r.v[3 + k] = k < np ? par[np * i + k] : T(0);
  r.v[7] = T(0);
  rows[i] = r;
}
template <class T>
void launch_pack_scalar_rows(hipStream_t st, int na, int np, const T* pos, const T* par, SRow<T>* rows) {
  if (na > 0) k_pack_scalar_rows<T><<<(na + 255) / 256, 256, 0, st>>>(na, np, pos, par, rows);
}
// CUT (admp_set_cutoff: Verlet lists with a skin): partners beyond the cutoff contribute nothing, so that the result is the
// one of the exact-rc list.  A semantic option, not a faster one: the kernel waits on the chain entry -> partner row, not on
// the pair arithmetic, and the lanes of a wave share the loop (a form in which every lane first walks to its next partner
// inside the cutoff and the lanes then evaluate together measured 0.68 against 0.37 ms for TT at 1M atoms: it gives up the
// two-ahead prefetch).
template <class T, int LPR, bool TT, bool CUT>
__global__ __launch_bounds__(kPairBlock) void k_pair_scalar(int na, const int* __restrict__ rowptr,
                                                            const int* __restrict__ col, const SRow<T>* __restrict__ srows,
                                                            Box<T> box, ScaleTab<T> tab,
                                                            T kappa, int pmax, T* __restrict__ grad, double* energies,
                                                            const int* __restrict__ rows, T rc2) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  const long t = (long)blockIdx.x * kPairBlock + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  // `na` counts the rows of this launch (all atoms in the table's length-sorted order, or a slab rank's home rows)
  const int row = slot < na ? (rows ? rows[slot] : slot) : -1;
  T g[3] = {0, 0, 0};
  double e = 0.0;
  if (row >= 0) {
    const SRow<T> I = srows[row];
    const int end = rowptr[row + 1];
    int k = rowptr[row] + sub;
    {
      // two entries and one partner row are fetched ahead of the arithmetic (the chain col -> row -> pair term is what the
      // kernel waits on); out-of-range prefetches read entry 0 / row 0: valid addresses, results unused
      int c0 = k < end ? col[k] : 0, c1 = k + LPR < end ? col[k + LPR] : 0;
      SRow<T> J = srows[c0 & kColMask];
#pragma unroll 1
      for (; k < end; k += LPR) {
        const int c2 = k + 2 * LPR < end ? col[k + 2 * LPR] : 0;
        const SRow<T> Jn = srows[c1 & kColMask];
        const int nb = col_nb(c0);
        bool in = true;
        if (CUT) {
          T d[3] = {I.v[0] - J.v[0], I.v[1] - J.v[1], I.v[2] - J.v[2]};
          min_image(box, d);
          in = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < rc2;
        }
        if (in) {
          if (TT) e += (double)tt_pair(box, I.v, J.v, I.v + 3, J.v + 3, s_tab[nb] + T(1), g);
          else e += (double)disp_pair(box, I.v, J.v, I.v + 3, J.v + 3, s_tab[nb], kappa, pmax, g);
        }
        c0 = c1; c1 = c2; J = Jn;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) g[k] = row_reduce<T, LPR>(g[k]);
  if (row >= 0 && sub == 0) {
    grad[3 * row] = g[0]; grad[3 * row + 1] = g[1]; grad[3 * row + 2] = g[2];
  }
  e = block_reduce_sum<kPairBlock>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[E_REAL], 0.5 * e);
}

// admp_set_cutoff in the on-request kernels below (parameter gradients, dE/dmScales, box gradient of the scalar pair terms):
// a listed pair beyond the cutoff contributes nothing, exactly as in the energy / gradient kernels -- with a skin list the
// derivatives are then those of the energy the calculator returns (advisor, round 3).  rc2 = 0: every listed pair.
template <class T>
__device__ __forceinline__ bool pair_inside(const Box<T>& box, const T ri[3], const T rj[3], T rc2) {
  if (rc2 <= T(0)) return true;
  T d[3] = {ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]};
  min_image(box, d);
  return d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < rc2;
}

// per-atom parameter derivatives of the scalar pair terms (disp_math.h): out[row][0..NP) = sum over the row's partners
template <class T, bool TT>
__global__ __launch_bounds__(kPairBlock) void k_pair_scalar_pgrad(int na, const int* __restrict__ rowptr,
                                                                  const int* __restrict__ col, const T* __restrict__ pos,
                                                                  const T* __restrict__ par, Box<T> box, ScaleTab<T> tab,
                                                                  T kappa, int pmax, T* __restrict__ out, T rc2,
                                                                  const int* __restrict__ rows) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  constexpr int LPR = 8, NP = TT ? 4 : 3;
  const long t = (long)blockIdx.x * kPairBlock + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  const int row = (rows && slot < na) ? rows[slot] : slot;      // (slab rank: the rows of its home atoms)
  T acc[4] = {0, 0, 0, 0};
  if (slot < na) {
    const T ri[3] = {pos[3 * row], pos[3 * row + 1], pos[3 * row + 2]};
    T pi[4] = {0, 0, 0, 0};
    for (int k = 0; k < NP; ++k) pi[k] = par[NP * row + k];
    const int end = rowptr[row + 1];
#pragma unroll 1
    for (int k = rowptr[row] + sub; k < end; k += LPR) {
      const int c = col[k];
      const int nb = col_nb(c), j = c & kColMask;
      const T rj[3] = {pos[3 * j], pos[3 * j + 1], pos[3 * j + 2]};
      if (!pair_inside(box, ri, rj, rc2)) continue;
      T pj[4] = {0, 0, 0, 0};
      for (int q = 0; q < NP; ++q) pj[q] = par[NP * j + q];
      if (TT) tt_pair_dparams(box, ri, rj, pi, pj, s_tab[nb] + T(1), acc);
      else disp_pair_dc(box, ri, rj, pj, s_tab[nb], kappa, pmax, acc);
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) acc[k] = row_reduce<T, LPR>(acc[k]);
  if (slot < na && sub == 0)
    for (int k = 0; k < NP; ++k) out[NP * row + k] = acc[k];
}
template <class T>
void launch_scalar_pair_pgrad(hipStream_t st, int tt, int na, const NbrTable& nb, const T* pos, const T* par, const Box<T>& box,
                              const ScaleTab<T>& tab, T kappa, int pmax, T* out, double cutoff, const int* rows, int n_rows) {
  if (rows) na = n_rows;
  if (na <= 0) return;
  const unsigned grid = (unsigned)(((long)na * 8 + kPairBlock - 1) / kPairBlock);
  const T rc2 = (T)(cutoff * cutoff);
  if (tt) k_pair_scalar_pgrad<T, true><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, pos, par, box, tab, kappa, pmax, out, rc2, rows);
  else k_pair_scalar_pgrad<T, false><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, pos, par, box, tab, kappa, pmax, out, rc2, rows);
}

// dE/dmScales (the parameter gradient the reference's examples/openmm_api/run.py:41-46 prints): per covalent class
// nb = 0..15 the sum over its pairs of d(pair energy)/d(mscale).  KIND 0: multipolar PME (bare multipole interaction,
// pair_bare_energy), 1: dispersion (sum_p c_p,i c_p,j / r^p), 2: Tang-Toennies (the kernel without its m factor).
// Non-bonded pairs (class 0, almost all of them) accumulate in a register, the bonded classes through LDS atomics.
template <class T, int KIND>
__global__ __launch_bounds__(kPairBlock) void k_pair_mgrad(int na, const int* __restrict__ rowptr,
                                                           const int* __restrict__ col, const Site<T>* __restrict__ sites,
                                                           const T* __restrict__ pos, const T* __restrict__ par, Box<T> box,
                                                           int pmax, double* __restrict__ cls, T rc2,
                                                           const int* __restrict__ rows) {
  __shared__ double s_cls[16];
  if (threadIdx.x < 16) s_cls[threadIdx.x] = 0.0;
  __syncthreads();
  constexpr int LPR = 8;
  const long t = (long)blockIdx.x * kPairBlock + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  const int row = (rows && slot < na) ? rows[slot] : slot;      // (slab rank: its home rows; the class sums are added over the ranks)
  double e0 = 0.0;
  if (slot < na) {
    Site<T> I;
    T ri[3] = {0, 0, 0}, pi[4] = {0, 0, 0, 0};
    constexpr int NP = KIND == 2 ? 4 : 3;
    if (KIND == 0) I = sites[row];
    else {
      ri[0] = pos[3 * row]; ri[1] = pos[3 * row + 1]; ri[2] = pos[3 * row + 2];
      for (int k = 0; k < NP; ++k) pi[k] = par[NP * row + k];
    }
    const int end = rowptr[row + 1];
    for (int k = rowptr[row] + sub; k < end; k += LPR) {
      const int c = col[k];
      const int nb = col_nb(c), j = c & kColMask;
      T v;
      if (KIND == 0) {
        v = pair_bare_energy<T>(box, I, sites[j]);
      } else {
        T rj[3] = {pos[3 * j], pos[3 * j + 1], pos[3 * j + 2]}, pj[4] = {0, 0, 0, 0}, g[3] = {0, 0, 0};
        if (!pair_inside(box, ri, rj, rc2)) continue;      // (the multipolar kernels evaluate every listed pair: KIND 0 has no cutoff)
        for (int q = 0; q < NP; ++q) pj[q] = par[NP * j + q];
        if (KIND == 1) {   // disp_pair is linear in mm: E(mm = 1) - E(mm = 0)
          v = disp_pair(box, ri, rj, pi, pj, T(1), T(0), pmax, g) - disp_pair(box, ri, rj, pi, pj, T(0), T(0), pmax, g);
        } else {
          v = tt_pair(box, ri, rj, pi, pj, T(1), g);
        }
      }
      if (nb == 0) e0 += (double)v;
      else atomicAdd(&s_cls[nb], (double)v);
    }
  }
  e0 = block_reduce_sum<kPairBlock>(e0);
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&cls[0], 0.5 * e0);
  if (threadIdx.x >= 1 && threadIdx.x < 16 && s_cls[threadIdx.x] != 0.0) atomicAdd(&cls[threadIdx.x], 0.5 * s_cls[threadIdx.x]);
}

// dE/dpScales: per covalent class the sum over its pairs of pair_pscale_deriv (polarizable handle, dipoles given)
template <class T>
__global__ __launch_bounds__(kPairBlock) void k_pair_pgrad(int na, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                           const Site<T>* __restrict__ sites, Box<T> box, ScaleTab<T> tab,
                                                           double* __restrict__ cls, const int* __restrict__ rows) {
  __shared__ T s_tab[48];
  __shared__ double s_cls[16];
  stage_tab(tab, s_tab);
  if (threadIdx.x < 16) s_cls[threadIdx.x] = 0.0;
  __syncthreads();
  constexpr int LPR = 8;
  const long t = (long)blockIdx.x * kPairBlock + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  const int row = (rows && slot < na) ? rows[slot] : slot;
  double e0 = 0.0;
  if (slot < na) {
    const Site<T> I = sites[row];
    const int end = rowptr[row + 1];
    for (int k = rowptr[row] + sub; k < end; k += LPR) {
      const int c = col[k];
      const int nb = col_nb(c);
      const PairScales<T> sc = {s_tab[nb], s_tab[16 + nb], s_tab[32 + nb]};
      const T v = pair_pscale_deriv<T>(box, I, sites[c & kColMask], sc);
      if (nb == 0) e0 += (double)v;
      else atomicAdd(&s_cls[nb], (double)v);
    }
  }
  e0 = block_reduce_sum<kPairBlock>(e0);
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&cls[0], 0.5 * e0);
  if (threadIdx.x >= 1 && threadIdx.x < 16 && s_cls[threadIdx.x] != 0.0) atomicAdd(&cls[threadIdx.x], 0.5 * s_cls[threadIdx.x]);
}
template <class T>
void launch_pscale_sums(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                        const ScaleTab<T>& tab, double* cls16, const int* rows, int n_rows) {
  if (rows) na = n_rows;
  if (na <= 0) return;
  const unsigned grid = (unsigned)(((long)na * 8 + kPairBlock - 1) / kPairBlock);
  k_pair_pgrad<T><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, box, tab, cls16, rows);
}
template void launch_pscale_sums<float>(hipStream_t, int, const NbrTable&, const Site<float>*, const Box<float>&,
                                        const ScaleTab<float>&, double*, const int*, int);
template void launch_pscale_sums<double>(hipStream_t, int, const NbrTable&, const Site<double>*, const Box<double>&,
                                         const ScaleTab<double>&, double*, const int*, int);

// per-atom sums of d(pair energy)/d ln(au) (pair_thole_logderiv): sumX[i] = sum_j X_ij, sumXw[i] = sum_j X_ij wth_ij
template <class T>
__global__ __launch_bounds__(kPairBlock) void k_pair_tholegrad(int na, const int* __restrict__ rowptr,
                                                               const int* __restrict__ col,
                                                               const Site<T>* __restrict__ sites, Box<T> box,
                                                               ScaleTab<T> tab, T* __restrict__ sumX, T* __restrict__ sumXw,
                                                               const int* __restrict__ rows) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  constexpr int LPR = 8;
  const long t = (long)blockIdx.x * kPairBlock + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  const int row = (rows && slot < na) ? rows[slot] : slot;      // (slab rank: the sums of its home rows)
  T sx = 0, sw = 0;
  if (slot < na) {
    const Site<T> I = sites[row];
    const int end = rowptr[row + 1];
    for (int k = rowptr[row] + sub; k < end; k += LPR) {
      const int c = col[k];
      const int nb = col_nb(c);
      const PairScales<T> sc = {s_tab[nb], s_tab[16 + nb], s_tab[32 + nb]};
      T wth;
      const T X = pair_thole_logderiv<T>(box, I, sites[c & kColMask], sc, &wth);
      sx += X;
      sw += X * wth;
    }
  }
  sx = row_reduce<T, LPR>(sx);
  sw = row_reduce<T, LPR>(sw);
  if (slot < na && sub == 0) { sumX[row] = sx; sumXw[row] = sw; }
}
template <class T>
void launch_thole_sums(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                       const ScaleTab<T>& tab, T* sumX, T* sumXw, const int* rows, int n_rows) {
  if (rows) na = n_rows;
  if (na <= 0) return;
  const unsigned grid = (unsigned)(((long)na * 8 + kPairBlock - 1) / kPairBlock);
  k_pair_tholegrad<T><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, box, tab, sumX, sumXw, rows);
}

template <class T>
void launch_mscale_sums(hipStream_t st, int kind, int na, const NbrTable& nb, const Site<T>* sites, const T* pos,
                        const T* par, const Box<T>& box, int pmax, double* cls16, double cutoff, const int* rows, int n_rows) {
  if (rows) na = n_rows;
  if (na <= 0) return;
  const unsigned grid = (unsigned)(((long)na * 8 + kPairBlock - 1) / kPairBlock);
  const T rc2 = (T)(cutoff * cutoff);
  if (kind == 0) k_pair_mgrad<T, 0><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, pos, par, box, pmax, cls16, T(0), rows);
  else if (kind == 1) k_pair_mgrad<T, 1><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, pos, par, box, pmax, cls16, rc2, rows);
  else k_pair_mgrad<T, 2><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, pos, par, box, pmax, cls16, rc2, rows);
}

// ---- box gradient, real-space part (jax.grad(get_energy, argnums=1) in the reference) -----------------------
// vir[3c+b] += 1/2 sum over the directed table entries of shift_c (dE_pair/d r_I)_b, shift = the lattice translation of the
// pair's minimum image (image_shift).  Pairs inside the cell are skipped before any arithmetic, so the pass costs the
// boundary-crossing fraction of a pair kernel (2.7 % of the pairs at 1M atoms, 19 % at 3072).  On request only.
template <int BLOCK>
__device__ __forceinline__ void block_add9(double acc[9], double scale, double* out) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double v = block_reduce_sum<BLOCK>(acc[k]);
    if (threadIdx.x == 0 && v != 0.0) atomicAdd(&out[k], scale * v);
  }
}

template <class T, bool LPOL>
__global__ __launch_bounds__(kPairBlock) void k_pair_virial(int na, const int* __restrict__ rowptr,
                                                            const int* __restrict__ col, const Site<T>* __restrict__ sites,
                                                            Box<T> box, ScaleTab<T> tab, T kappa, double* vir,
                                                            const int* __restrict__ rows) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  constexpr int LPR = 8;
  const long t = (long)blockIdx.x * kPairBlock + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (slot < na) {
    const int row = rows ? rows[slot] : slot;           // (slab rank: its home rows)
    const Site<T> I = sites[row];
    const int end = rowptr[row + 1];
#pragma unroll 1
    for (int k = rowptr[row] + sub; k < end; k += LPR) {
      const int c = col[k];
      const int nb = col_nb(c);
      const Site<T> J = sites[c & kColMask];
      const T d[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
      T sh[3];
      if (!image_shift(box, d, sh)) continue;
      const PairScales<T> sc = {s_tab[nb], s_tab[16 + nb], s_tab[32 + nb]};
      T g[3] = {0, 0, 0}, P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      (void)pair_energy_grad<T, LPOL, false>(box, I, J, sc, kappa, g, P, nullptr, nullptr, nullptr);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[3 * a + b] += (double)sh[a] * (double)g[b];
    }
  }
  block_add9<kPairBlock>(acc, 0.5, vir);
}

template <class T, bool TT>
__global__ __launch_bounds__(kPairBlock) void k_pair_scalar_virial(int na, const int* __restrict__ rowptr,
                                                                   const int* __restrict__ col, const T* __restrict__ pos,
                                                                   const T* __restrict__ par, Box<T> box, ScaleTab<T> tab,
                                                                   T kappa, int pmax, double* vir, T rc2,
                                                                   const int* __restrict__ rows) {
  __shared__ T s_tab[48];
  stage_tab(tab, s_tab);
  constexpr int LPR = 8, NP = TT ? 4 : 3;
  const long t = (long)blockIdx.x * kPairBlock + threadIdx.x;
  const int slot = (int)(t / LPR), sub = (int)(t % LPR);
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (slot < na) {
    const int row = rows ? rows[slot] : slot;           // (slab rank: its home rows)
    T ri[3] = {pos[3 * row], pos[3 * row + 1], pos[3 * row + 2]}, pi[4] = {0, 0, 0, 0};
    for (int k = 0; k < NP; ++k) pi[k] = par[NP * row + k];
    const int end = rowptr[row + 1];
#pragma unroll 1
    for (int k = rowptr[row] + sub; k < end; k += LPR) {
      const int c = col[k];
      const int nb = col_nb(c), j = c & kColMask;
      T rj[3] = {pos[3 * j], pos[3 * j + 1], pos[3 * j + 2]}, pj[4] = {0, 0, 0, 0};
      const T d[3] = {ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]};
      T sh[3];
      if (!image_shift(box, d, sh)) continue;
      if (!pair_inside(box, ri, rj, rc2)) continue;
      for (int q = 0; q < NP; ++q) pj[q] = par[NP * j + q];
      T g[3] = {0, 0, 0};
      if (TT) (void)tt_pair(box, ri, rj, pi, pj, s_tab[nb] + T(1), g);
      else (void)disp_pair(box, ri, rj, pi, pj, s_tab[nb], kappa, pmax, g);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[3 * a + b] += (double)sh[a] * (double)g[b];
    }
  }
  block_add9<kPairBlock>(acc, 0.5, vir);
}

template <class T>
void launch_pair_virial(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                        const ScaleTab<T>& tab, T kappa, int lpol, double* vir, const int* rows, int n_rows) {
  if (rows) na = n_rows;
  if (na <= 0) return;
  const unsigned grid = (unsigned)(((long)na * 8 + kPairBlock - 1) / kPairBlock);
  if (lpol) k_pair_virial<T, true><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, box, tab, kappa, vir, rows);
  else k_pair_virial<T, false><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, box, tab, kappa, vir, rows);
}
template <class T>
void launch_scalar_pair_virial(hipStream_t st, int tt, int na, const NbrTable& nb, const T* pos, const T* par,
                               const Box<T>& box, const ScaleTab<T>& tab, T kappa, int pmax, double* vir, double cutoff,
                               const int* rows, int n_rows) {
  if (rows) na = n_rows;
  if (na <= 0) return;
  const unsigned grid = (unsigned)(((long)na * 8 + kPairBlock - 1) / kPairBlock);
  const T rc2 = (T)(cutoff * cutoff);
  if (tt) k_pair_scalar_virial<T, true><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, pos, par, box, tab, kappa, pmax, vir, rc2, rows);
  else k_pair_scalar_virial<T, false><<<grid, kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, pos, par, box, tab, kappa, pmax, vir, rc2, rows);
}

// lanes of a wavefront that share one row.  Few lanes = fewer idle lanes at the row tail and less shuffle
// folding; many lanes = enough wavefronts to fill the chip when there are few rows.  Measured (f32, polarizable):
// 1M rows LPR 1/2/4/8 -> 0.749/0.493/0.502/0.534 ms; 98k rows 2/4/8 -> 0.068/0.064/0.065 ms; 3k rows 4/8/16/32 ->
// 38/23/18/24 us.
int pair_lanes_per_row(int n_rows) {
  static int forced = -2;
  if (forced == -2) {
    const char* s = getenv("ADMP_PAIR_LPR");
    int x = s ? atoi(s) : -1;
    forced = (x == 1 || x == 2 || x == 4 || x == 8 || x == 16 || x == 32) ? x : -1;
  }
  if (forced > 0) return forced;
  // round 2, rows parted by site class (water: a third of the rows and partners carry higher moments), pipelined loops:
  // 1M rows LPR 2/4/8 -> 0.325/0.303/0.422 ms, 98k rows 0.044/0.041/0.052 ms
  return n_rows >= 32768 ? 4 : (n_rows >= 8192 ? 8 : 16);
}
// the field kernels (k_pair_field, k_pair_field_ind): env ADMP_FIELD_LPR overrides
static int field_lanes_per_row(int n_rows, bool ind) {
  static const int forced = [] {
    const char* s = getenv("ADMP_FIELD_LPR");
    const int x = s ? atoi(s) : -1;
    return (x == 1 || x == 2 || x == 4 || x == 8 || x == 16 || x == 32) ? x : -1;
  }();
  if (forced > 0) return forced;
  // light arithmetic per partner, bound by the partner fetches: more lanes per row = more fetches in flight.  350k rows (the
  // polarizable sites of 1M atoms) LPR 2/4/8/16 -> 0.110/0.093/0.083/0.091 ms (k_pair_field), 0.034/0.030/0.033/0.041 ms
  // (k_pair_field_ind); 32k rows: 0.0069/0.0052/0.0051/0.0052 and 0.0019/0.0018/0.0026/0.0017 ms
  return ind ? (n_rows >= 32768 ? 4 : (n_rows >= 8192 ? 8 : 16)) : (n_rows >= 8192 ? 8 : 16);
}

// the scalar pair kernels (dispersion, Tang-Toennies; k_pair_scalar): env ADMP_SCALAR_LPR / ADMP_TT_LPR override
static int scalar_lanes_per_row(int n_rows, bool tt, double avg_row) {
  static const int forced[2] = {
      [] { const char* s = getenv("ADMP_SCALAR_LPR"); const int x = s ? atoi(s) : -1;
           return (x == 1 || x == 2 || x == 4 || x == 8 || x == 16 || x == 32) ? x : -1; }(),
      [] { const char* s = getenv("ADMP_TT_LPR"); const int x = s ? atoi(s) : -1;
           return (x == 1 || x == 2 || x == 4 || x == 8 || x == 16 || x == 32) ? x : -1; }()};
  if (forced[tt ? 1 : 0] > 0) return forced[tt ? 1 : 0];
  // round 4, packed 32-byte partner rows and the two-ahead prefetch: few lanes per row win as soon as the rows fill the chip
  // (a lane walks ~14 partners with its fetches in flight; more lanes only add row_reduce steps and idle tail lanes).
  // 1M rows (f32) LPR 1/2/4/8/16: dispersion 0.251/0.171/0.227/0.422/0.810 ms, TT call 0.383/0.220/0.268/0.457/0.847;
  // 98k rows: 0.033/0.030/0.040/0.053/0.086 and 0.075/0.064/0.069/0.088/0.121; 3072 rows (f64): 0.031/0.020/0.018/0.013/0.014
  // and 0.076/0.058/0.049/0.051/0.045.  (Rounds 2-3 ran dispersion with the field kernels' rule: 8 lanes at 1M rows.)
  // On a Verlet list with a skin the rows are twice as long (rc 4 + 1 A: 55 partners) and the best width doubles (1M rows,
  // skin list, cutoff test, LPR 1/2/4/8: dispersion 0.938/0.407/0.301/0.432 ms, TT call 1.22/0.50/0.37/0.48): the rule is
  // ~14 partners per lane, from the table's average row length.
  if (n_rows < 8192) return tt ? 16 : 8;
  const double lanes = avg_row / 14.0;
  const int lpr = lanes < 1.4 ? 1 : (lanes < 2.8 ? 2 : (lanes < 5.6 ? 4 : 8));
  return n_rows >= 32768 ? lpr : 2 * lpr;
}

// minimum waves per SIMD requested from the register allocator for the polarizable kernel
// (env ADMP_PAIR_MINW: 1 = no constraint, 2 = at most 256 registers).  Defaults from measurement:
//   f32: 2 (242 VGPRs, no spill; S2 0.110 ms vs 0.161 ms at 1 wave/SIMD)
//   f64: 1 (the 256-register cap costs 184 B/lane of scratch: S1 30 us vs 20 us unconstrained)
template <class T>
static int pair_min_waves() {
  static int v = -1;
  if (v < 0) {
    const char* s = getenv("ADMP_PAIR_MINW");
    v = s ? atoi(s) : (sizeof(T) == 4 ? 2 : 1);
  }
  return v;
}

static inline unsigned grid_for(int na, int lpr) { return (unsigned)(((long)na * lpr + kPairBlock - 1) / kPairBlock); }
static inline unsigned grid_full(int na, int lpr) { return (unsigned)(((long)na * lpr + kFullBlock - 1) / kFullBlock); }

#define ADMP_LPR_SWITCH(lpr, CALL) \
  switch (lpr) {                   \
    case 1: CALL(1); break;        \
    case 2: CALL(2); break;        \
    case 4: CALL(4); break;        \
    case 16: CALL(16); break;      \
    case 32: CALL(32); break;      \
    default: CALL(8); break;       \
  }

template <class T>
void launch_pair_full(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                      const ScaleTab<T>& tab, T kappa, int lpol, T* grad, T* pot, double* energies, const int* rows,
                      T* fld, int use_mono, const int* cls_flags, const RQ4<T>* rq, const T* tholes) {
  if (na <= 0) return;
  const int lpr = pair_lanes_per_row(na);
  const int minw = pair_min_waves<T>();
  static const bool mono_off = [] { const char* e = getenv("ADMP_PAIR_MONO"); return e && atoi(e) == 0; }();
  if (mono_off || !cls_flags || !rq) use_mono = 0;
#define CALL(L)                                                                                                        \
  if (lpol && minw >= 2)                                                                                               \
    k_pair_full<T, true, L, 2><<<xcd_grid(grid_full(na, L)), kFullBlock, 0, st>>>(                                      \
        na, nb.rowptr, nb.col, sites, box, tab, kappa, grad, pot, energies, rows, fld, grid_full(na, L), use_mono,     \
        cls_flags, rq, tholes);                                                                                        \
  else if (lpol)                                                                                                       \
    k_pair_full<T, true, L, 1><<<xcd_grid(grid_full(na, L)), kFullBlock, 0, st>>>(                                      \
        na, nb.rowptr, nb.col, sites, box, tab, kappa, grad, pot, energies, rows, fld, grid_full(na, L), use_mono,     \
        cls_flags, rq, tholes);                                                                                        \
  else                                                                                                                 \
    k_pair_full<T, false, L, 2><<<xcd_grid(grid_full(na, L)), kFullBlock, 0, st>>>(                                     \
        na, nb.rowptr, nb.col, sites, box, tab, kappa, grad, pot, energies, rows, fld, grid_full(na, L), use_mono, cls_flags, rq,    \
        tholes)
  ADMP_LPR_SWITCH(lpr, CALL)
#undef CALL
}

template <class T>
void launch_pair_field(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                       const ScaleTab<T>& tab, T kappa, T* fld, const int* rows, const int* n_dev, const int* cls_flags,
                       const RQ4<T>* rq, const T* tholes) {
  static const bool mono_off = [] { const char* e = getenv("ADMP_PAIR_MONO"); return e && atoi(e) == 0; }();
  if (mono_off || !rq) cls_flags = nullptr;
  if (na <= 0) return;
  const int lpr = field_lanes_per_row(na, false);
#define CALL(L)                                                                                              \
  k_pair_field<T, L><<<xcd_grid(grid_for(na, L)), kPairBlock, 0, st>>>(na, nb.rowptr, nb.col, sites, box, tab, kappa, \
                                                                       fld, rows, grid_for(na, L), n_dev, cls_flags, \
                                                                       rq, tholes)
  ADMP_LPR_SWITCH(lpr, CALL)
#undef CALL
}
// the field kernels as riders of an x pass (k_xconv_pair): false = this launch cannot ride (no rows, another lane count)
constexpr int kRiderLpr = 16;
template <class T>
bool field_rider_full(FieldRider<T>& r, int na, const NbrTable& nb, const Site<T>* sites, const Box<T>& box,
                      const ScaleTab<T>& tab, T kappa, T* fld, const int* rows, const int* n_dev, const int* cls_flags,
                      const RQ4<T>* rq, const T* tholes) {
  static const bool mono_off = [] { const char* e = getenv("ADMP_PAIR_MONO"); return e && atoi(e) == 0; }();
  if (mono_off || !rq) cls_flags = nullptr;
  if (na <= 0 || field_lanes_per_row(na, false) != kRiderLpr) return false;
  r.kind = 1; r.na = na; r.rowptr = nb.rowptr; r.rowend = nullptr; r.col = nb.col; r.sites = sites; r.box = box; r.tab = tab;
  r.kappa = kappa; r.fld = fld; r.rows = rows; r.nblocks = grid_for(na, kRiderLpr); r.grid = xcd_grid(r.nblocks);
  r.n_dev = n_dev; r.cls_flags = cls_flags; r.rq = rq; r.tholes = tholes;
  return true;
}
template <class T>
bool field_rider_ind(FieldRider<T>& r, int n_rows, const IndTable& it, const Site<T>* sites, const Box<T>& box,
                     const ScaleTab<T>& tab, T kappa, T* fld, const int* rows) {
  if (n_rows <= 0 || field_lanes_per_row(n_rows, true) != kRiderLpr) return false;
  r.kind = 2; r.na = n_rows; r.rowptr = it.beg; r.rowend = it.end; r.col = it.col; r.sites = sites; r.box = box; r.tab = tab;
  r.kappa = kappa; r.fld = fld; r.rows = rows; r.nblocks = grid_for(n_rows, kRiderLpr); r.grid = xcd_grid(r.nblocks);
  r.n_dev = nullptr; r.cls_flags = nullptr; r.rq = nullptr; r.tholes = nullptr;
  return true;
}
// x pass of a direct-DFT convolution (one mesh) with the rider's workgroups appended to its grid
template <class T>
void launch_dft_x_conv_rider(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, double* energies,
                             int slot, const FieldRider<T>& fr) {
  const int N = K[0], Kh = K[2] / 2 + 1, H = (N - 1) / 2, TK = dft_tasks(N, dft_kq());
  const int NC = dft_cols(N, dft_kq(), sizeof(PairCx<T>) * (size_t)H + sizeof(Cx<T>) * (size_t)(2 + N), sizeof(Cx<T>) * (size_t)N);
  const size_t sh = sizeof(PairCx<T>) * (size_t)(H * NC) + sizeof(Cx<T>) * (size_t)(N + 2 * NC + N * NC);
  const int nbx = (Kh + NC - 1) / NC;
  const unsigned extra = (fr.grid + (unsigned)K[1] - 1) / (unsigned)K[1];
  const XConvArgs<T> xa{N, Kh, NC, TK, (long)K[1] * Kh, (long)Kh, K[2], reinterpret_cast<Cx<T>*>(spec), tabs,
                        reinterpret_cast<const Cx<T>*>(tw), energies, slot, 0};
  k_xconv_pair<T, kRiderLpr><<<dim3((unsigned)nbx + extra, (unsigned)K[1], 1), kPairBlock, sh, st>>>(xa, fr, nbx);
}

template <class T>
void launch_pair_field_ind(hipStream_t st, int n_rows, const IndTable& it, const Site<T>* sites, const Box<T>& box,
                           const ScaleTab<T>& tab, T kappa, T* fld, const int* rows) {
  if (n_rows <= 0) return;
  const int lpr = field_lanes_per_row(n_rows, true);
#define CALL(L)                                                                                                          \
  k_pair_field_ind<T, L><<<xcd_grid(grid_for(n_rows, L)), kPairBlock, 0, st>>>(n_rows, it.beg, it.end, it.col, sites, box, tab, \
                                                                               kappa, fld, rows)
  ADMP_LPR_SWITCH(lpr, CALL)
#undef CALL
}
template <class T>
void launch_disp_pair(hipStream_t st, int na, const NbrTable& nb, const SRow<T>* srows, const Box<T>& box,
                      const ScaleTab<T>& tab, T kappa, int pmax, T* grad, double* energies, const int* rows, int n_rows,
                      double cutoff) {
  if (!rows) { rows = nb.order_plain ? nb.order_plain : nb.order; n_rows = na; }
  if (n_rows <= 0) return;
  const int lpr = scalar_lanes_per_row(n_rows, false, na > 0 ? 2.0 * (double)nb.n_half / na : 0.0);
  const T rc2 = (T)(cutoff * cutoff);
#define CALL(L)                                                                                                        \
  if (cutoff > 0.0)                                                                                                    \
    k_pair_scalar<T, L, false, true><<<grid_for(n_rows, L), kPairBlock, 0, st>>>(n_rows, nb.rowptr, nb.col, srows, box, \
                                                                                 tab, kappa, pmax, grad, energies, rows, rc2); \
  else                                                                                                                 \
    k_pair_scalar<T, L, false, false><<<grid_for(n_rows, L), kPairBlock, 0, st>>>(n_rows, nb.rowptr, nb.col, srows, box, \
                                                                                  tab, kappa, pmax, grad, energies, rows, rc2)
  ADMP_LPR_SWITCH(lpr, CALL)
#undef CALL
}
template <class T>
void launch_tt_pair(hipStream_t st, int na, const NbrTable& nb, const SRow<T>* srows, const Box<T>& box,
                    const ScaleTab<T>& tab, T* grad, double* energies, const int* rows, int n_rows, double cutoff) {
  if (!rows) { rows = nb.order_plain ? nb.order_plain : nb.order; n_rows = na; }
  if (n_rows <= 0) return;
  const int lpr = scalar_lanes_per_row(n_rows, true, na > 0 ? 2.0 * (double)nb.n_half / na : 0.0);
  const T rc2 = (T)(cutoff * cutoff);
#define CALL(L)                                                                                                       \
  if (cutoff > 0.0)                                                                                                   \
    k_pair_scalar<T, L, true, true><<<grid_for(n_rows, L), kPairBlock, 0, st>>>(n_rows, nb.rowptr, nb.col, srows, box,  \
                                                                                tab, T(0), 0, grad, energies, rows, rc2); \
  else                                                                                                                \
    k_pair_scalar<T, L, true, false><<<grid_for(n_rows, L), kPairBlock, 0, st>>>(n_rows, nb.rowptr, nb.col, srows, box, \
                                                                                 tab, T(0), 0, grad, energies, rows, rc2)
  ADMP_LPR_SWITCH(lpr, CALL)
#undef CALL
}

#define INST(T)                                                                                                     \
  template void launch_pair_full<T>(hipStream_t, int, const NbrTable&, const Site<T>*, const Box<T>&,               \
                                    const ScaleTab<T>&, T, int, T*, T*, double*, const int*, T*, int, const int*,   \
                                    const RQ4<T>*, const T*);                                                       \
  template void launch_pair_field<T>(hipStream_t, int, const NbrTable&, const Site<T>*, const Box<T>&,              \
                                     const ScaleTab<T>&, T, T*, const int*, const int*, const int*,                 \
                                     const RQ4<T>*, const T*);                                                      \
  template void launch_pair_field_ind<T>(hipStream_t, int, const IndTable&, const Site<T>*, const Box<T>&,          \
                                         const ScaleTab<T>&, T, T*, const int*);                                    \
  template bool field_rider_full<T>(FieldRider<T>&, int, const NbrTable&, const Site<T>*, const Box<T>&, const ScaleTab<T>&, \
                                    T, T*, const int*, const int*, const int*, const RQ4<T>*, const T*);            \
  template bool field_rider_ind<T>(FieldRider<T>&, int, const IndTable&, const Site<T>*, const Box<T>&,             \
                                   const ScaleTab<T>&, T, T*, const int*);                                          \
  template void launch_dft_x_conv_rider<T>(hipStream_t, const int*, const T*, T*, const DftTabs<T>&, double*, int,  \
                                           const FieldRider<T>&);                                                   \
  template void launch_pack_scalar_rows<T>(hipStream_t, int, int, const T*, const T*, SRow<T>*);                    \
  template void launch_disp_pair<T>(hipStream_t, int, const NbrTable&, const SRow<T>*, const Box<T>&,               \
                                    const ScaleTab<T>&, T, int, T*, double*, const int*, int, double);              \
  template void launch_tt_pair<T>(hipStream_t, int, const NbrTable&, const SRow<T>*, const Box<T>&,                 \
                                  const ScaleTab<T>&, T*, double*, const int*, int, double);                        \
  template void launch_mscale_sums<T>(hipStream_t, int, int, const NbrTable&, const Site<T>*, const T*, const T*,   \
                                      const Box<T>&, int, double*, double, const int*, int);                        \
  template void launch_thole_sums<T>(hipStream_t, int, const NbrTable&, const Site<T>*, const Box<T>&,              \
                                     const ScaleTab<T>&, T*, T*, const int*, int);                                  \
  template void launch_pair_virial<T>(hipStream_t, int, const NbrTable&, const Site<T>*, const Box<T>&,             \
                                      const ScaleTab<T>&, T, int, double*, const int*, int);                        \
  template void launch_scalar_pair_virial<T>(hipStream_t, int, int, const NbrTable&, const T*, const T*,            \
                                             const Box<T>&, const ScaleTab<T>&, T, int, double*, double,            \
                                             const int*, int);                                                      \
  template void launch_scalar_pair_pgrad<T>(hipStream_t, int, int, const NbrTable&, const T*, const T*,             \
                                            const Box<T>&, const ScaleTab<T>&, T, int, T*, double, const int*, int);
INST(float)
INST(double)
#undef INST

}  // namespace admp
