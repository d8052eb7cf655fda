// Per-atom kernels of the PME path: site table construction (local frames + local->global
// rotation, reference admp/pme.py:220-238), SCF field assembly and Jacobi update
// (admp/pme.py:130-138), and the closing kernel that adds the self term (admp/pme.py:738-757),
// the polarization penalty (admp/pme.py:760-774) and the local-frame adjoint.
//
// One thread per atom; 20-real site rows are written as whole 16/32-byte vectors.
#include <cstdlib>

#include <cstring>

#include "launch.h"
#include "reduce.h"

namespace admp {

constexpr int kAtomBlock = 256;

template <class T>
__device__ __forceinline__ void load3(const T* a, int i, T o[3]) {
  o[0] = a[3 * i]; o[1] = a[3 * i + 1]; o[2] = a[3 * i + 2];
}

template <class T>
__device__ __forceinline__ void frame_of(const Topology& top, const T* pos, const Box<T>& box, int i, int& type,
                                         int& iz, int& ix, int& iy, FrameWork<T>& w) {
  type = top.axis_type[i];
  iz = top.axis_idx[3 * i];
  ix = top.axis_idx[3 * i + 1];
  iy = top.axis_idx[3 * i + 2];
  T p[3], pz[3] = {0, 0, 0}, px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
  load3(pos, i, p);
  if (iz >= 0) load3(pos, iz, pz); else if (type != NoAxisType) type = NoAxisType;
  if (ix >= 0) load3(pos, ix, px);
  if (iy >= 0) load3(pos, iy, py);
  local_frame_fwd(type, box, p, pz, px, py, w);
}

template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_prepare_sites(Topology top, const T* __restrict__ pos,
                                                              const T* __restrict__ Qlocal,
                                                              const T* __restrict__ Ucart, const T* __restrict__ pol,
                                                              const T* __restrict__ thole, Box<T> box,
                                                              Site<T>* __restrict__ sites, double* zero_next,
                                                              RecipGeom<T> g, int4* __restrict__ bases,
                                                              int* __restrict__ act_list, int* __restrict__ act_count,
                                                              const int* __restrict__ cls, int* __restrict__ cls_flags,
                                                              RQ4<T>* __restrict__ rq, T* __restrict__ Ucopy,
                                                              const int* __restrict__ list, int nlist) {
  int i = blockIdx.x * kAtomBlock + threadIdx.x;
  if (zero_next && i < E_WORDS) zero_next[i] = 0.0;   // the NEXT evaluation's energy words (engine.hip: Ed_cur)
  if (list) {        // slab rank: the rows of the listed atoms only (its home atoms, or the atoms it reads); no site list here
    if (i >= nlist) return;
    i = list[i];
  }
  if (act_list) {   // kernel-uniform: list of the polarizable sites (pol > 0); ONE counter update per workgroup (every
                    // workgroup of the launch hits the same word: 16k per-wave atomics cost 0.16 ms at 1M atoms), and
                    // a workgroup's sites stay together and in order, which keeps the consumers' gathers local
    __shared__ int wcnt[kAtomBlock / 64], wbase;
    const bool act = i < top.na && pol && pol[i] > T(0);
    const unsigned long long m = __ballot(act);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < kAtomBlock / 64; ++w) { const int c = wcnt[w]; wcnt[w] = tot; tot += c; }
      wbase = tot ? atomicAdd(act_count, tot) : 0;
    }
    __syncthreads();
    if (act) act_list[wbase + wcnt[wave] + __popcll(m & ((1ull << lane) - 1ull))] = i;
  }
  if (i >= top.na) return;
  int type, iz, ix, iy;
  FrameWork<T> w;
  frame_of(top, pos, box, i, type, iz, ix, iy, w);
  T ql[9], cx[3], cy[3], cz[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) ql[k] = Qlocal[9 * i + k];
  Site<T> s;
  frame_cols(w.X, w.Y, w.Z, cx, cy, cz);
  rot_harm(ql, cx, cy, cz, s.Q);   // rot_local2global = rotation with the transposed frame (multipole.py:201)
  load3(pos, i, s.r);
  if (Ucart) {                     // C1_c2h: harmonic order (z, x, y) (admp/pme.py:235)
    s.U[0] = Ucart[3 * i + 2]; s.U[1] = Ucart[3 * i]; s.U[2] = Ucart[3 * i + 1];
    // admp_set_dipole_source: the initial dipoles came from a read-only array; the evaluation's own array starts as a copy
    if (Ucopy) { Ucopy[3 * i] = s.U[1]; Ucopy[3 * i + 1] = s.U[2]; Ucopy[3 * i + 2] = s.U[0]; }
  } else {
    s.U[0] = s.U[1] = s.U[2] = T(0);
  }
  T a = pol ? pol[i] : T(0);
  s.p6 = a > T(0) ? (T)pow((double)a, 1.0 / 6.0) : T(0);
  s.thole = thole ? thole[i] : T(0);
  s.pad[0] = s.pad[1] = s.pad[2] = T(0);
  // charge-only site (pme_math.h site_is_mono): no dipole, no quadrupole, not polarizable, no dipole handed in.  pad[0] is
  // free for the mark: the pad words carry dU only on polarizable sites.  (Q_local zero <=> Q_global zero: orthogonal map.)
  if (s.p6 == T(0) && s.U[0] == T(0) && s.U[1] == T(0) && s.U[2] == T(0)) {
    bool mono = true;
#pragma unroll
    for (int k = 1; k < 9; ++k) mono = mono && ql[k] == T(0);
    s.pad[0] = mono ? T(1) : T(0);
  }
  if (cls_flags) {   // does the neighbour table's idea of the charge-only atoms (NbrTable::cls) still hold?
    const bool was = cls && cls[i] != 0, is = s.pad[0] == T(1) && s.p6 == T(0);
    const int f = was && !is ? CLS_STALE : (!was && is ? CLS_BETTER : 0);
    if (f && !(*(volatile int*)cls_flags & f)) atomicOr(cls_flags, f);
  }
  sites[i] = s;
  if (rq) { RQ4<T> q; q.v[0] = s.r[0]; q.v[1] = s.r[1]; q.v[2] = s.r[2]; q.v[3] = s.Q[0]; rq[i] = q; }
  if (bases) {   // lowest mesh index of the atom's stencil on every axis: what the spread's binning needs, 16 B per atom
    int b[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) grid_ref(g, s.r, d, b[d]);
    const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
    bases[i] = make_int4(b[0], b[1], b[2], brick_code(b, dims, make_bricks(dims)));
  }
}

// NbrTable::cls from the sites of the last evaluation
template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_site_classes(int na, const Site<T>* __restrict__ sites, int* __restrict__ cls) {
  const int i = blockIdx.x * kAtomBlock + threadIdx.x;
  if (i < na) cls[i] = site_is_mono(sites[i]) ? 1 : 0;
}

// Jacobi step of the incremental SCF, over the polarizable sites only (see launch.h)
template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_jacobi_delta(int n_act, const int* __restrict__ act,
                                                             const T* __restrict__ pol, const T* __restrict__ field,
                                                             T* __restrict__ Ucart, Site<T>* __restrict__ sites,
                                                             Site<T>* __restrict__ isites,
                                                             const unsigned long long* __restrict__ gate,
                                                             unsigned long long gate_bits) {
  const int slot = blockIdx.x * kAtomBlock + threadIdx.x;
  if (slot >= n_act) return;
  const int i = act[slot];
  // gate (chained SCF, engine.hip): the step is enqueued before the host has seen the residual of the check it follows; when
  // that check passed (bit patterns of non-negative doubles order like the numbers) the step is a zero step: U keeps its
  // bits and every kernel of the increment that follows adds exact zeros
  const bool on = !gate || *gate >= gate_bits;
  const T s = on ? -pol[i] * T(1.0 / kDielectric) : T(0);
  const T dx = field[3 * i] * s, dy = field[3 * i + 1] * s, dz = field[3 * i + 2] * s;     // admp/pme.py:138
  Ucart[3 * i] += dx; Ucart[3 * i + 1] += dy; Ucart[3 * i + 2] += dz;
  Site<T> r = sites[i];
  r.U[0] += dz; r.U[1] += dx; r.U[2] += dy;                  // harmonic order (z, x, y)
  r.pad[0] = dz; r.pad[1] = dx; r.pad[2] = dy;
  sites[i] = r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.Q[k] = T(0);
  r.U[0] = dz; r.U[1] = dx; r.U[2] = dy;
  isites[slot] = r;
}

// (Na,3,3) local frames, rows x, y, z (admp/spatial.py:76-142): the diagnostic behind ADMPPmeForce.construct_local_frames
template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_local_frames(Topology top, const T* __restrict__ pos, Box<T> box,
                                                             T* __restrict__ out) {
  int i = blockIdx.x * kAtomBlock + threadIdx.x;
  if (i >= top.na) return;
  int type, iz, ix, iy;
  FrameWork<T> w;
  frame_of(top, pos, box, i, type, iz, ix, iy, w);
#pragma unroll
  for (int k = 0; k < 3; ++k) { out[9 * i + k] = w.X[k]; out[9 * i + 3 + k] = w.Y[k]; out[9 * i + 6 + k] = w.Z[k]; }
}
template <class T>
void launch_local_frames(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, T* out) {
  k_local_frames<T><<<(top.na + kAtomBlock - 1) / kAtomBlock, kAtomBlock, 0, st>>>(top, pos, box, out);
}
template void launch_local_frames<float>(hipStream_t, const Topology&, const float*, const Box<float>&, float*);
template void launch_local_frames<double>(hipStream_t, const Topology&, const double*, const Box<double>&, double*);

template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_field_finish(int na, const Site<T>* __restrict__ sites,
                                                             const T* __restrict__ pol, const T* __restrict__ Ucart,
                                                             const T* __restrict__ fld_pair,
                                                             const T* __restrict__ fld_recip, T kappa,
                                                             T* __restrict__ field, unsigned long long* fmax_bits,
                                                             const int* __restrict__ list, const int* __restrict__ n_dev) {
  const int slot = blockIdx.x * kAtomBlock + threadIdx.x;
  if (n_dev) na = min(na, *n_dev);
  double fm = 0.0;
  if (slot < na) {
    const int i = list ? list[slot] : slot;
    const T a = pol[i];
    T fx, fy, fz;
    total_field(sites[i], a, Ucart + 3 * i, fld_pair + 3 * i, fld_recip + 3 * i, kappa, fx, fy, fz);
    field[3 * i] = fx; field[3 * i + 1] = fy; field[3 * i + 2] = fz;
    if (a > T(0.001)) fm = fmax(fabs((double)fx), fmax(fabs((double)fy), fabs((double)fz)));
  }
  fm = block_reduce_max<kAtomBlock>(fm);
  if (threadIdx.x == 0 && fm > 0.0) atomicMax(fmax_bits, nonneg_bits(fm));
}

template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_finish(Topology top, const T* __restrict__ pos, Box<T> box,
                                                       const Site<T>* __restrict__ sites, const T* __restrict__ pol,
                                                       const T* __restrict__ Ucart, int lpol, T kappa,
                                                       T* __restrict__ pot, T* __restrict__ grad,
                                                       T* __restrict__ dQlocal, double* energies,
                                                       const int* __restrict__ list, int nlist) {
  const int slot = blockIdx.x * kAtomBlock + threadIdx.x;
  double eself = 0.0, epen = 0.0;
  if (slot < nlist) {
    const int i = list ? list[slot] : slot;
    T f[3];
    self_factors(kappa, f);
    const Site<T>& s = sites[i];
    T Qt[9], P[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Qt[k] = s.Q[k]; P[k] = pot[9 * i + k]; }
    if (lpol) { Qt[1] += s.U[0]; Qt[2] += s.U[1]; Qt[3] += s.U[2]; }
    // E_self = -D sum_h f_l Qtot_h^2 ; dE/dQ_global = -2 D f_l Qtot_h
    double es = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      T fl = k == 0 ? f[0] : (k < 4 ? f[1] : f[2]);
      es += (double)(fl * Qt[k] * Qt[k]);
      P[k] -= T(2.0 * kDielectric) * fl * Qt[k];
    }
    eself = -kDielectric * es;
    if (lpol) {
      T a = pol[i];
      a = a < T(1e-8) ? T(1e-8) : a;
      double u2 = (double)Ucart[3 * i] * Ucart[3 * i] + (double)Ucart[3 * i + 1] * Ucart[3 * i + 1] +
                  (double)Ucart[3 * i + 2] * Ucart[3 * i + 2];
      epen = kDielectric * 0.5 * u2 / (double)a;
    }
    if (grad) {
      // adjoint of the local frame: torque of the PERMANENT multipoles (the induced dipole is a global
      // Cartesian input held fixed, admp/pme.py:81-85)
      int type, iz, ix, iy;
      FrameWork<T> w;
      frame_of(top, pos, box, i, type, iz, ix, iy, w);
      T tau[3], gp[3], gz[3], gx[3], gy[3];
      multipole_torque(P, s.Q, tau);
      local_frame_bwd(type, w, tau, gp, gz, gx, gy);
      if (type != NoAxisType) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          atomicAdd(&grad[3 * i + k], gp[k]);
          atomicAdd(&grad[3 * iz + k], gz[k]);
          if (type != Zonly) atomicAdd(&grad[3 * ix + k], gx[k]);
          if (type == ZBisect || type == ThreeFold) atomicAdd(&grad[3 * iy + k], gy[k]);
        }
      }
      if (dQlocal) {
        T dl[9];
        rot_harm(P, w.X, w.Y, w.Z, dl);
#pragma unroll
        for (int k = 0; k < 9; ++k) dQlocal[9 * i + k] = dl[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) pot[9 * i + k] = P[k];
  }
  eself = block_reduce_sum<kAtomBlock>(eself);
  epen = block_reduce_sum<kAtomBlock>(epen);
  if (threadIdx.x == 0) {
    atomicAdd(&energies[E_SELF], eself);
    if (lpol) atomicAdd(&energies[E_PEN], epen);
  }
}

// Single-GPU closing kernel without atomics: every atom PULLS the frame-adjoint contributions of the frames it
// takes part in (itself and the sites that use it as z/x/y axis atom, inverse map built by admp_set_topology);
// each frame is re-evaluated by its <= 4 member atoms, which is far cheaper than scattered global float atomics
// (k_finish: 0.27 ms per 1M atoms, this kernel: see profiles/).  pot is read-only here.
template <class T>
__device__ __forceinline__ void total_potential(const Site<T>& s, const T* __restrict__ pot_i, int lpol, const T f[3],
                                                T P[9], double* eself) {
  double es = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    T q = s.Q[k];
    if (lpol && k >= 1 && k <= 3) q += s.U[k - 1];
    const T fl = k == 0 ? f[0] : (k < 4 ? f[1] : f[2]);
    es += (double)(fl * q * q);
    P[k] = pot_i[k] - T(2.0 * kDielectric) * fl * q;
  }
  if (eself) *eself = -kDielectric * es;
}

template <class T, int LANES>
__global__ __launch_bounds__(kAtomBlock) void k_finish_pull(Topology top, const T* __restrict__ pos, Box<T> box,
                                                            const Site<T>* __restrict__ sites,
                                                            const T* __restrict__ pol, const T* __restrict__ Ucart,
                                                            int lpol, T kappa, const T* __restrict__ pot,
                                                            T* __restrict__ grad, T* __restrict__ dQlocal,
                                                            double* energies, FieldFin<T> ff) {
  // LANES = 4 (small systems, latency bound): lane m takes the frames m, m+4, ... of the atom's inverse map (water: 3
  // frames per atom), the partial gradients are folded with two xor shuffles -- a quarter of the dependent chain
  // (3072 atoms: 14.5 -> 9.6 us).  LANES = 1 (large systems, throughput bound: at 1M atoms the 4-lane form is 3x slower).
  const long t = (long)blockIdx.x * kAtomBlock + threadIdx.x;
  const int a = (int)(t / LANES), m = (int)(t % LANES);
  double eself = 0.0, epen = 0.0;
  T g[3] = {0, 0, 0};
  if (a < top.na) {
    T f[3];
    self_factors(kappa, f);
    T P[9];
    total_potential(sites[a], pot + 9 * (size_t)a, lpol, f, P, m == 0 ? &eself : nullptr);
    if (lpol && m == 0) {
      T al = pol[a];
      al = al < T(1e-8) ? T(1e-8) : al;
      double u2 = (double)Ucart[3 * a] * Ucart[3 * a] + (double)Ucart[3 * a + 1] * Ucart[3 * a + 1] +
                  (double)Ucart[3 * a + 2] * Ucart[3 * a + 2];
      epen = kDielectric * 0.5 * u2 / (double)al;
    }
    if (grad) {
      for (int k = top.inv_ptr[a] + m; k < top.inv_ptr[a + 1]; k += LANES) {
        const int i = top.inv_idx[k];
        int type, iz, ix, iy;
        FrameWork<T> w;
        frame_of(top, pos, box, i, type, iz, ix, iy, w);
        if (type == NoAxisType) continue;
        T Pi[9];
        if (i == a) {
#pragma unroll
          for (int q = 0; q < 9; ++q) Pi[q] = P[q];
        } else {
          total_potential(sites[i], pot + 9 * (size_t)i, lpol, f, Pi, nullptr);
        }
        T tau[3], gp[3], gz[3], gx[3], gy[3];
        multipole_torque(Pi, sites[i].Q, tau);
        local_frame_bwd(type, w, tau, gp, gz, gx, gy);
        const bool usex = type != Zonly, usey = (type == ZBisect || type == ThreeFold);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (i == a) g[q] += gp[q];
          if (iz == a) g[q] += gz[q];
          if (usex && ix == a) g[q] += gx[q];
          if (usey && iy == a) g[q] += gy[q];
        }
        if (i == a && dQlocal) {
          T dl[9];
          rot_harm(Pi, w.X, w.Y, w.Z, dl);
#pragma unroll
          for (int q = 0; q < 9; ++q) dQlocal[9 * a + q] = dl[q];
        }
      }
      if (m == 0 && dQlocal && top.axis_type[a] == NoAxisType) {   // identity frame
#pragma unroll
        for (int q = 0; q < 9; ++q) dQlocal[9 * a + q] = P[q];
      }
    }
  }
  if (grad) {   // kernel-uniform
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (LANES > 1) g[q] += __shfl_xor(g[q], 1, 64);
      if (LANES > 2) g[q] += __shfl_xor(g[q], 2, 64);
    }
    if (a < top.na && m == 0) { grad[3 * a] += g[0]; grad[3 * a + 1] += g[1]; grad[3 * a + 2] += g[2]; }
  }
  eself = block_reduce_sum<kAtomBlock>(eself);
  epen = block_reduce_sum<kAtomBlock>(epen);
  if (threadIdx.x == 0) {
    atomicAdd(&energies[E_SELF], eself);
    if (lpol) atomicAdd(&energies[E_PEN], epen);
  }
  if (ff.fmax_bits) {   // kernel-uniform: the SCF residual and its maximum ride along (k_field_finish's work)
    double fm = 0.0;
    if (a < top.na && m == 0) {
      const T al = ff.pol[a];
      T fx, fy, fz;
      total_field(sites[a], al, ff.Ucart + 3 * a, ff.fld_pair + 3 * a, ff.fld_recip + 3 * a, ff.kappa, fx, fy, fz);
      ff.field[3 * a] = fx; ff.field[3 * a + 1] = fy; ff.field[3 * a + 2] = fz;
      if (al > T(0.001)) fm = fmax(fabs((double)fx), fmax(fabs((double)fy), fabs((double)fz)));
    }
    fm = block_reduce_max<kAtomBlock>(fm);
    if (threadIdx.x == 0 && fm > 0.0) atomicMax(ff.fmax_bits, nonneg_bits(fm));
  }
}

// Closing kernel for topologies made of small frame groups (Topology::grp_ptr; one molecule = one group): one thread per
// group evaluates each frame of the group ONCE and keeps the gradient contributions of the group's <= kMaxGroup atoms in
// registers -- the pull form above re-evaluates every frame once per member atom (3x for water) and re-reads its inputs.
template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_finish_groups(Topology top, const T* __restrict__ pos, Box<T> box,
                                                              const Site<T>* __restrict__ sites,
                                                              const T* __restrict__ pol, const T* __restrict__ Ucart,
                                                              int lpol, T kappa, const T* __restrict__ pot,
                                                              T* __restrict__ grad, T* __restrict__ dQlocal,
                                                              double* energies, FieldFin<T> ff,
                                                              const int* __restrict__ slab_bits) {
  const int gidx = blockIdx.x * kAtomBlock + threadIdx.x;
  double eself = 0.0, epen = 0.0, fm = 0.0;
  // slab rank (slab_bits): a group is served by every rank that owns one of its atoms, each for the sites it owns; what a
  // site's frame adds to atoms of other ranks lands in their (zeroed) rows and travels to the owners afterwards
  bool any = gidx < top.ngroups;
  if (any && slab_bits) {
    any = false;
    for (int i = top.grp_ptr[gidx]; i < top.grp_ptr[gidx + 1]; ++i) any = any || (slab_bits[i] & kSlabHome);
  }
  if (any) {
    const int a0 = top.grp_ptr[gidx], n = top.grp_ptr[gidx + 1] - a0;
    T f[3];
    self_factors(kappa, f);
    T acc[kMaxGroup][3];
#pragma unroll
    for (int q = 0; q < kMaxGroup; ++q) acc[q][0] = acc[q][1] = acc[q][2] = T(0);
#pragma unroll
    for (int m = 0; m < kMaxGroup; ++m) {
      if (m >= n) break;
      const int i = a0 + m;
      if (slab_bits && !(slab_bits[i] & kSlabHome)) continue;
      T P[9];
      double es;
      total_potential(sites[i], pot + 9 * (size_t)i, lpol, f, P, &es);
      eself += es;
      if (lpol) {
        T al = pol[i];
        al = al < T(1e-8) ? T(1e-8) : al;
        const double u2 = (double)Ucart[3 * i] * Ucart[3 * i] + (double)Ucart[3 * i + 1] * Ucart[3 * i + 1] +
                          (double)Ucart[3 * i + 2] * Ucart[3 * i + 2];
        epen += kDielectric * 0.5 * u2 / (double)al;
      }
      if (ff.fmax_bits) {
        const T al = ff.pol[i];
        T fx, fy, fz;
        total_field(sites[i], al, ff.Ucart + 3 * i, ff.fld_pair + 3 * i, ff.fld_recip + 3 * i, ff.kappa, fx, fy, fz);
        ff.field[3 * i] = fx; ff.field[3 * i + 1] = fy; ff.field[3 * i + 2] = fz;
        if (al > T(0.001)) fm = fmax(fm, fmax(fabs((double)fx), fmax(fabs((double)fy), fabs((double)fz))));
      }
      if (!grad) continue;
      int type, iz, ix, iy;
      FrameWork<T> w;
      frame_of(top, pos, box, i, type, iz, ix, iy, w);
      if (type == NoAxisType) {
        if (dQlocal) {
#pragma unroll
          for (int q = 0; q < 9; ++q) dQlocal[9 * i + q] = P[q];
        }
        continue;
      }
      T tau[3], gp[3], gz[3], gx[3], gy[3];
      multipole_torque(P, sites[i].Q, tau);
      local_frame_bwd(type, w, tau, gp, gz, gx, gy);
      const int lz = iz - a0, lx = type != Zonly ? ix - a0 : -1, ly = (type == ZBisect || type == ThreeFold) ? iy - a0 : -1;
#pragma unroll
      for (int q = 0; q < kMaxGroup; ++q) {         // register array: compare-select, no dynamic indexing
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          T v = T(0);
          if (q == m) v += gp[c];
          if (q == lz) v += gz[c];
          if (q == lx) v += gx[c];
          if (q == ly) v += gy[c];
          acc[q][c] += v;
        }
      }
      if (dQlocal) {
        T dl[9];
        rot_harm(P, w.X, w.Y, w.Z, dl);
#pragma unroll
        for (int q = 0; q < 9; ++q) dQlocal[9 * i + q] = dl[q];
      }
    }
    if (grad) {
#pragma unroll
      for (int m = 0; m < kMaxGroup; ++m)
        if (m < n && (!slab_bits || slab_bits[a0 + m] != 0)) {      // (slab: home atoms and the atoms the rank reads; others untouched)
          grad[3 * (a0 + m)] += acc[m][0]; grad[3 * (a0 + m) + 1] += acc[m][1]; grad[3 * (a0 + m) + 2] += acc[m][2];
        }
    }
  }
  eself = block_reduce_sum<kAtomBlock>(eself);
  epen = block_reduce_sum<kAtomBlock>(epen);
  if (threadIdx.x == 0) {
    atomicAdd(&energies[E_SELF], eself);
    if (lpol) atomicAdd(&energies[E_PEN], epen);
  }
  if (ff.fmax_bits) {
    fm = block_reduce_max<kAtomBlock>(fm);
    if (threadIdx.x == 0 && fm > 0.0) atomicMax(ff.fmax_bits, nonneg_bits(fm));
  }
}

// Row form of the closing kernel (round 4).  k_finish_groups gives a lane one MOLECULE: its loads walk three 80-byte site
// rows, 27 potential words, positions and gradient rows at a stride of 240 bytes between lanes -- every wave-level load
// touches 64 cache lines, the address unit is busy 59 % of the kernel and the memory side moves 2.1 x the bytes the
// kernel needs (round-3 verdict).  Here a lane takes one ATOM: neighbouring lanes read neighbouring rows (site row, pot,
// gradient, axis record: all contiguous across the wave), every frame is still evaluated exactly once -- by its own site --
// and what a frame needs of its group mates (positions in, gradient contributions out) goes through LDS: a workgroup owns a
// run of WHOLE frame groups (Topology::rows_blk), so no group straddles two workgroups; each atom then pulls the
// contributions addressed to it from the <= kMaxGroup - 1 mates of its group (no atomics).
template <class T>
__global__ __launch_bounds__(kFinishBlock) void k_finish_rows(Topology top, Box<T> box, const Site<T>* __restrict__ sites,
                                                              const T* __restrict__ pol, int lpol, T kappa,
                                                              const T* __restrict__ pot, T* __restrict__ grad,
                                                              T* __restrict__ dQlocal, double* energies, FieldFin<T> ff,
                                                              const int* __restrict__ slab_bits) {
  __shared__ T spos[kFinishBlock][3];
  __shared__ T sg[kFinishBlock][9];          // what the frame of site t adds to its z, x, y atoms
  __shared__ int sidx[kFinishBlock][3];      // ... their indices relative to the run's first atom (-1: none)
  const int t = threadIdx.x;
  double eself = 0.0, epen = 0.0, fm = 0.0;
  // A workgroup walks several runs (grid-stride) and adds its energy sums ONCE: a launch of one workgroup per run put 8 000
  // double-precision atomics on the two energy words at 1M atoms, which serialise at the memory side (8 ns each: the first
  // version of this kernel took 0.116 ms against 0.085 for the form it replaces, all of it waiting there).
  for (int run = blockIdx.x; run < top.nrowblk; run += gridDim.x) {
    const int a0 = top.rows_blk[run], na_blk = top.rows_blk[run + 1] - a0;
    const int i = a0 + t;
    const bool live = t < na_blk;
    // slab rank: a site is served by its owner; the contributions of its frame to atoms of other ranks land in their rows
    // (zeroed by the caller) and travel to the owners afterwards.  bits == 0: an atom this rank neither owns nor reads.
    const int bits = live ? (slab_bits ? slab_bits[i] : kSlabHome) : 0;
    const bool mine = (bits & kSlabHome) != 0;
    if (slab_bits && !__syncthreads_or(mine)) continue;          // (workgroup-uniform) nothing of this rank's in this run
    T gp[3] = {T(0), T(0), T(0)};
    Site<T> s;
    if (live) {
      s = sites[i];
      spos[t][0] = s.r[0]; spos[t][1] = s.r[1]; spos[t][2] = s.r[2];
    }
    sidx[t][0] = sidx[t][1] = sidx[t][2] = -1;
    __syncthreads();
    if (mine) {
      T f[3], P[9];
      double es;
      self_factors(kappa, f);
      total_potential(s, pot + 9 * (size_t)i, lpol, f, P, &es);
      eself += es;
      if (lpol) {
        T al = pol[i];
        al = al < T(1e-8) ? T(1e-8) : al;
        const double u2 = (double)s.U[0] * s.U[0] + (double)s.U[1] * s.U[1] + (double)s.U[2] * s.U[2];   // (harmonic order: same norm)
        epen += kDielectric * 0.5 * u2 / (double)al;
      }
      if (ff.fmax_bits) {
        const T al = ff.pol[i];
        T fx, fy, fz;
        total_field(s, al, ff.Ucart + 3 * i, ff.fld_pair + 3 * i, ff.fld_recip + 3 * i, ff.kappa, fx, fy, fz);
        ff.field[3 * i] = fx; ff.field[3 * i + 1] = fy; ff.field[3 * i + 2] = fz;
        if (al > T(0.001)) fm = fmax(fm, fmax(fabs((double)fx), fmax(fabs((double)fy), fabs((double)fz))));
      }
      if (grad) {
        int type = top.axis_type[i];
        const int iz = top.axis_idx[3 * i], ix = top.axis_idx[3 * i + 1], iy = top.axis_idx[3 * i + 2];
        if (iz < 0) type = NoAxisType;
        if (type == NoAxisType) {
          if (dQlocal) {
#pragma unroll
            for (int q = 0; q < 9; ++q) dQlocal[9 * (size_t)i + q] = P[q];
          }
        } else {
          const bool usex = type != Zonly, usey = (type == ZBisect || type == ThreeFold);
          const int lz = iz - a0, lx = usex && ix >= 0 ? ix - a0 : -1, ly = usey && iy >= 0 ? iy - a0 : -1;
          T pz[3] = {0, 0, 0}, px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
#pragma unroll
          for (int c = 0; c < 3; ++c) {                                  // (group mates: inside this run by construction)
            pz[c] = spos[lz][c];
            if (ix >= 0) px[c] = spos[ix - a0][c];
            if (iy >= 0) py[c] = spos[iy - a0][c];
          }
          FrameWork<T> w;
          local_frame_fwd(type, box, s.r, pz, px, py, w);
          T tau[3], gz[3], gx[3], gy[3];
          multipole_torque(P, s.Q, tau);
          local_frame_bwd(type, w, tau, gp, gz, gx, gy);
#pragma unroll
          for (int c = 0; c < 3; ++c) { sg[t][c] = gz[c]; sg[t][3 + c] = gx[c]; sg[t][6 + c] = gy[c]; }
          sidx[t][0] = lz; sidx[t][1] = lx; sidx[t][2] = ly;
          if (dQlocal) {
            T dl[9];
            rot_harm(P, w.X, w.Y, w.Z, dl);
#pragma unroll
            for (int q = 0; q < 9; ++q) dQlocal[9 * (size_t)i + q] = dl[q];
          }
        }
      }
    }
    if (grad) {                                 // kernel-uniform
      __syncthreads();
      if (bits != 0) {
        const int rec = top.grp_of[i], g0 = (rec >> 2) - a0, gn = (rec & 3) + 1;
        T acc[3] = {gp[0], gp[1], gp[2]};
        for (int m = g0; m < g0 + gn; ++m) {     // the frames of the group's sites (its own included: an axis atom may be the site)
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (sidx[m][k] == t) { acc[0] += sg[m][3 * k]; acc[1] += sg[m][3 * k + 1]; acc[2] += sg[m][3 * k + 2]; }
        }
        grad[3 * (size_t)i] += acc[0]; grad[3 * (size_t)i + 1] += acc[1]; grad[3 * (size_t)i + 2] += acc[2];
      }
    }
    __syncthreads();                            // the LDS rows are rewritten by the next run
  }
  eself = block_reduce_sum<kFinishBlock>(eself);
  epen = block_reduce_sum<kFinishBlock>(epen);
  if (threadIdx.x == 0) {
    if (eself != 0.0) atomicAdd(&energies[E_SELF], eself);
    if (lpol && epen != 0.0) atomicAdd(&energies[E_PEN], epen);
  }
  if (ff.fmax_bits) {
    fm = block_reduce_max<kFinishBlock>(fm);
    if (threadIdx.x == 0 && fm > 0.0) atomicMax(ff.fmax_bits, nonneg_bits(fm));
  }
}

// Box gradient, local-frame part: the frame vectors site -> axis atom go through the same minimum image as the pairs
// (admp/spatial.py:88-101), so a molecule that straddles the cell boundary contributes shift (x) dE/d(vector).
// pot = dE/dQ_global of pair + reciprocal space (the self term is added here, as in the closing kernel).
template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_frame_virial(Topology top, const T* __restrict__ pos, Box<T> box,
                                                             const Site<T>* __restrict__ sites, int lpol, T kappa,
                                                             const T* __restrict__ pot, double* vir,
                                                             const int* __restrict__ list, int nlist) {
  const int slot = blockIdx.x * kAtomBlock + threadIdx.x;
  const int n = list ? nlist : top.na;
  const int i = (list && slot < n) ? list[slot] : slot;      // (slab rank: the frames of its home sites)
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (slot < n) {
    int type, iz, ix, iy;
    FrameWork<T> w;
    frame_of(top, pos, box, i, type, iz, ix, iy, w);
    if (type != NoAxisType) {
      T f[3], P[9], tau[3], gp[3], gz[3], gx[3], gy[3], p[3];
      self_factors(kappa, f);
      total_potential(sites[i], pot + 9 * (size_t)i, lpol, f, P, nullptr);
      multipole_torque(P, sites[i].Q, tau);
      local_frame_bwd(type, w, tau, gp, gz, gx, gy);
      load3(pos, i, p);
      const int member[3] = {iz, type != Zonly ? ix : -1, (type == ZBisect || type == ThreeFold) ? iy : -1};
      const T* gk[3] = {gz, gx, gy};
      for (int m = 0; m < 3; ++m) {
        if (member[m] < 0) continue;
        T q[3], sh[3];
        load3(pos, member[m], q);
        const T d[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
        if (!image_shift(box, d, sh)) continue;
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) acc[3 * a + b] += (double)sh[a] * (double)gk[m][b];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double v = block_reduce_sum<kAtomBlock>(acc[k]);
    if (threadIdx.x == 0 && v != 0.0) atomicAdd(&vir[k], v);
  }
}

template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_scalar_sites(int na, const T* __restrict__ pos, const T* __restrict__ vals,
                                                             int stride, int chan, SelfCoefs self_coefs,
                                                             Site<T>* __restrict__ sites, double* energies, RecipGeom<T> g,
                                                             int4* __restrict__ bases) {
  // blockIdx.y = channel of a batch (the site rows of channel b follow those of channel b-1)
  chan += blockIdx.y;
  sites += (size_t)blockIdx.y * na;
  const double self_coef = self_coefs.c[blockIdx.y];
  const int i = blockIdx.x * kAtomBlock + threadIdx.x;
  double s2 = 0.0;
  if (i < na) {
    Site<T> s;
    load3(pos, i, s.r);
    const T c = vals[(long)stride * i + chan];
    s.Q[0] = c;
#pragma unroll
    for (int k = 1; k < 9; ++k) s.Q[k] = T(0);
    s.U[0] = s.U[1] = s.U[2] = T(0);
    s.p6 = s.thole = T(0);
    s.pad[0] = s.pad[1] = s.pad[2] = T(0);
    sites[i] = s;
    s2 = (double)c * (double)c;
    if (bases && blockIdx.y == 0) {   // stencil records for the spread (as k_prepare_sites writes them): one set for the batch
      int b[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) grid_ref(g, s.r, d, b[d]);
      const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
      bases[i] = make_int4(b[0], b[1], b[2], brick_code(b, dims, make_bricks(dims)));
    }
  }
  s2 = block_reduce_sum<kAtomBlock>(s2);
  if (threadIdx.x == 0) atomicAdd(&energies[E_SELF], self_coef * s2);
}

template <class T>
__global__ __launch_bounds__(kAtomBlock) void k_scale_add(int na, const T* __restrict__ vals, int stride, int chan,
                                                          const T* __restrict__ v, T* __restrict__ grad, int nch) {
  const int i = blockIdx.x * kAtomBlock + threadIdx.x;
  if (i >= na) return;
  T gx = 0, gy = 0, gz = 0;
  for (int b = 0; b < nch; ++b) {            // channel b's vectors follow channel b-1's
    const T c = vals[(long)stride * i + chan + b];
    const T* vb = v + (size_t)b * 3 * na;
    gx += c * vb[3 * i]; gy += c * vb[3 * i + 1]; gz += c * vb[3 * i + 2];
  }
  grad[3 * i] += gx; grad[3 * i + 1] += gy; grad[3 * i + 2] += gz;
}

static inline int nblk(int n) { return (n + kAtomBlock - 1) / kAtomBlock; }

template <class T>
void launch_prepare_sites(hipStream_t st, const Topology& top, const T* pos, const T* Qlocal, const T* Ucart,
                          const T* pol, const T* thole, const Box<T>& box, Site<T>* sites, double* zero_next,
                          const RecipGeom<T>& g, int4* bases, int* act_list, int* act_count, const int* cls,
                          int* cls_flags, RQ4<T>* rq, T* Ucopy, const int* list, int nlist) {
  const int n = list ? nlist : top.na;
  if (n <= 0 && !zero_next) return;
  k_prepare_sites<T><<<nblk(n > E_WORDS || !zero_next ? (n > 0 ? n : 1) : E_WORDS), kAtomBlock, 0, st>>>(
      top, pos, Qlocal, Ucart, pol, thole, box, sites, zero_next, g, bases, list ? nullptr : act_list,
      list ? nullptr : act_count, cls, cls_flags, rq, Ucopy, list, nlist);
}
template <class T>
void launch_site_classes(hipStream_t st, int na, const Site<T>* sites, int* cls) {
  if (na > 0) k_site_classes<T><<<nblk(na), kAtomBlock, 0, st>>>(na, sites, cls);
}
template <class T>
void launch_field_finish(hipStream_t st, int na, const Site<T>* sites, const T* pol, const T* Ucart, const T* fld_pair,
                         const T* fld_recip, T kappa, T* field, unsigned long long* fmax_bits, const int* list,
                         const int* n_dev) {
  k_field_finish<T><<<nblk(na), kAtomBlock, 0, st>>>(na, sites, pol, Ucart, fld_pair, fld_recip, kappa, field, fmax_bits, list,
                                                     n_dev);
}
template <class T>
void launch_jacobi_delta(hipStream_t st, int n_act, const int* act, const T* pol, const T* field, T* Ucart, Site<T>* sites,
                         Site<T>* isites, const unsigned long long* gate, double gate_min) {
  unsigned long long bits;
  std::memcpy(&bits, &gate_min, sizeof(bits));
  if (n_act > 0)
    k_jacobi_delta<T><<<nblk(n_act), kAtomBlock, 0, st>>>(n_act, act, pol, field, Ucart, sites, isites, gate, bits);
}
template <class T>
void launch_finish(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, const Site<T>* sites,
                   const T* pol, const T* Ucart, int lpol, T kappa, T* pot, T* grad, T* dQlocal, double* energies,
                   const int* list, int nlist, const FieldFin<T>& ff, const int* slab_bits) {
  // molecular liquid at scale: one thread per frame group (below ~8k atoms the 4-lane pull form is faster: latency bound).
  // ADMP_FINISH_GROUPS_MIN overrides the threshold; read per call so that the parity tests can force either form.
  const char* gmin_env = getenv("ADMP_FINISH_GROUPS_MIN");
  const int gmin = gmin_env ? atoi(gmin_env) : 8192;
  if ((!list || slab_bits) && top.ngroups > 0 && top.na > gmin) {
    // ADMP_FINISH_ROWS=0: the one-thread-per-group form of round 2 (A/B, tests)
    const char* rows_env = getenv("ADMP_FINISH_ROWS");
    if (top.rows_blk && !(rows_env && atoi(rows_env) == 0))
      k_finish_rows<T><<<top.nrowblk < 1024 ? top.nrowblk : 1024, kFinishBlock, 0, st>>>(top, box, sites, pol, lpol, kappa, pot, grad,
                                                                                         dQlocal, energies, ff,
                                                                                         list ? slab_bits : nullptr);
    else
      k_finish_groups<T><<<nblk(top.ngroups), kAtomBlock, 0, st>>>(top, pos, box, sites, pol, Ucart, lpol, kappa, pot, grad,
                                                                   dQlocal, energies, ff, list ? slab_bits : nullptr);
    return;
  }
  if (!list && top.inv_ptr)   // single GPU: pull formulation, no atomics
  {
    static const int lanes4_max = [] { const char* e = getenv("ADMP_FINISH4_MAX"); return e ? atoi(e) : 8192; }();
    if (top.na <= lanes4_max)   // 3072 atoms: 9.6 vs 14.5 us; 30k atoms: 19.9 vs 14.2 us; 98k: 47 vs 21 us
      k_finish_pull<T, 4><<<nblk(4 * top.na), kAtomBlock, 0, st>>>(top, pos, box, sites, pol, Ucart, lpol, kappa, pot, grad,
                                                                   dQlocal, energies, ff);
    else
      k_finish_pull<T, 1><<<nblk(top.na), kAtomBlock, 0, st>>>(top, pos, box, sites, pol, Ucart, lpol, kappa, pot, grad,
                                                               dQlocal, energies, ff);
  }
  else
    k_finish<T><<<nblk(nlist), kAtomBlock, 0, st>>>(top, pos, box, sites, pol, Ucart, lpol, kappa, pot, grad, dQlocal,
                                                    energies, list, nlist);
}

template <class T>
void launch_frame_virial(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, const Site<T>* sites,
                         int lpol, T kappa, const T* pot, double* vir, const int* list, int nlist) {
  const int n = list ? nlist : top.na;
  if (n > 0) k_frame_virial<T><<<nblk(n), kAtomBlock, 0, st>>>(top, pos, box, sites, lpol, kappa, pot, vir, list, nlist);
}
template void launch_frame_virial<float>(hipStream_t, const Topology&, const float*, const Box<float>&, const Site<float>*,
                                         int, float, const float*, double*, const int*, int);
template void launch_frame_virial<double>(hipStream_t, const Topology&, const double*, const Box<double>&,
                                          const Site<double>*, int, double, const double*, double*, const int*, int);

template <class T>
void launch_scalar_sites(hipStream_t st, int na, const T* pos, const T* vals, int stride, int chan, double self_coef,
                         Site<T>* sites, double* energies) {
  SelfCoefs sc;
  sc.c[0] = self_coef;
  k_scalar_sites<T><<<nblk(na), kAtomBlock, 0, st>>>(na, pos, vals, stride, chan, sc, sites, energies, RecipGeom<T>(), nullptr);
}
template <class T>
void launch_scalar_sites_batch(hipStream_t st, int na, const T* pos, const T* vals, int stride, int nch,
                               const double* self_coefs, Site<T>* sites, double* energies, const RecipGeom<T>* g, int4* bases) {
  SelfCoefs sc;
  for (int b = 0; b < nch && b < 3; ++b) sc.c[b] = self_coefs[b];
  k_scalar_sites<T><<<dim3(nblk(na), nch), kAtomBlock, 0, st>>>(na, pos, vals, stride, 0, sc, sites, energies,
                                                                 g ? *g : RecipGeom<T>(), g ? bases : nullptr);
}
template <class T>
void launch_scale_add(hipStream_t st, int na, const T* vals, int stride, int chan, const T* v, T* grad, int nch) {
  k_scale_add<T><<<nblk(na), kAtomBlock, 0, st>>>(na, vals, stride, chan, v, grad, nch);
}

#define INST(T)                                                                                                        \
  template void launch_scalar_sites<T>(hipStream_t, int, const T*, const T*, int, int, double, Site<T>*, double*);      \
  template void launch_scale_add<T>(hipStream_t, int, const T*, int, int, const T*, T*, int);                           \
  template void launch_scalar_sites_batch<T>(hipStream_t, int, const T*, const T*, int, int, const double*, Site<T>*,   \
                                             double*, const RecipGeom<T>*, int4*);                                      \
  template void launch_prepare_sites<T>(hipStream_t, const Topology&, const T*, const T*, const T*, const T*, const T*, \
                                        const Box<T>&, Site<T>*, double*, const RecipGeom<T>&, int4*, int*, int*,       \
                                        const int*, int*, RQ4<T>*, T*, const int*, int);                                \
  template void launch_site_classes<T>(hipStream_t, int, const Site<T>*, int*);                                         \
  template void launch_field_finish<T>(hipStream_t, int, const Site<T>*, const T*, const T*, const T*, const T*, T, T*,  \
                                       unsigned long long*, const int*, const int*);                                    \
  template void launch_jacobi_delta<T>(hipStream_t, int, const int*, const T*, const T*, T*, Site<T>*, Site<T>*,        \
                                       const unsigned long long*, double);                                            \
  template void launch_finish<T>(hipStream_t, const Topology&, const T*, const Box<T>&, const Site<T>*, const T*,        \
                                 const T*, int, T, T*, T*, T*, double*, const int*, int, const FieldFin<T>&, const int*);
INST(float)
INST(double)
#undef INST

}  // namespace admp
