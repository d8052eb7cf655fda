// Order-6 cardinal B-spline pieces of the reciprocal-space path (reference admp/recip.py).
//
//   u_reference      recip.py:56-78     m_u0 = ceil(K frac), f = m_u0 - K frac in [0,1); the six
//                                       stencil points p = 0..5 sit at grid index m_u0 + p - 3 with
//                                       spline argument u = f + p.
//   bspline(+', '')  recip.py:80-137    here by the Cox-de Boor recursion (algebraically the same
//                                       piecewise polynomials), which also yields M''' -- needed by
//                                       the explicit position-gradient of quadrupoles that the
//                                       reference gets from autodiff.
//   sph_harmonics_GO recip.py:215-275   the harmonic operators are applied in "u space": the site's
//   Q_m_peratom      recip.py:278-310   multipoles are folded once per atom into 10 coefficients of
//                                       the separable products M^(a) M^(b) M^(c), a+b+c <= 2.
#pragma once
#include "pme_math.h"

namespace admp {

// M_6, M_6', M_6'', M_6''' at u = f + p, p = 0..5
template <class T>
ADMP_HD void bspline6(T f, T M[6], T D1[6], T D2[6], T D3[6]) {
  T a2[6] = {f, T(1) - f, 0, 0, 0, 0};
  T a3[6], a4[6], a5[6];
  // M_n(u) = [u M_{n-1}(u) + (n-u) M_{n-1}(u-1)]/(n-1)
#define ADMP_RAISE(src, dst, n)                                                          \
  for (int p = 0; p < 6; ++p) {                                                          \
    T lo = p > 0 ? src[p - 1] : T(0);                                                    \
    T hi = p < (n) - 1 ? src[p] : T(0);                                                  \
    dst[p] = p < (n) ? ((f + T(p)) * hi + (T(n) - f - T(p)) * lo) * T(1.0 / ((n) - 1)) : T(0); \
  }
  ADMP_RAISE(a2, a3, 3)
  ADMP_RAISE(a3, a4, 4)
  ADMP_RAISE(a4, a5, 5)
  ADMP_RAISE(a5, M, 6)
#undef ADMP_RAISE
  for (int p = 0; p < 6; ++p) {
    T m5_0 = a5[p], m5_1 = p > 0 ? a5[p - 1] : T(0);
    D1[p] = m5_0 - m5_1;
    T m4_0 = a4[p], m4_1 = p > 0 ? a4[p - 1] : T(0), m4_2 = p > 1 ? a4[p - 2] : T(0);
    D2[p] = m4_0 - T(2) * m4_1 + m4_2;
    T m3_0 = a3[p], m3_1 = p > 0 ? a3[p - 1] : T(0), m3_2 = p > 1 ? a3[p - 2] : T(0), m3_3 = p > 2 ? a3[p - 3] : T(0);
    D3[p] = m3_0 - T(3) * m3_1 + T(3) * m3_2 - m3_3;
  }
}

// Constants of one reciprocal-space setup.
template <class T>
struct RecipGeom {
  int K[3];
  T hinv[9];   // box^-1
  T Aop[9];    // d/dx_i = sum_j Aop[i][j] d/du_j as the reference's operators use it:
               //   Aop = -(N * box^-1)^T indexed exactly as admp/recip.py:52,177,212
  T Jac[9];    // true Jacobian du_j/dx_i = Jac[i][j] (what autodiff follows through u0, recip.py:75-77)
  // x-slab view of the mesh (multi-GPU): the local array holds planes xoff .. xoff+nloc0-1 (mod K[0]) of the
  // global mesh; wrap0 = K[0] when the whole axis is local (periodic wrap inside the stencil), else "never".
  int xoff, nloc0, wrap0;
  ADMP_HD int dim(int d) const { return d == 0 ? nloc0 : K[d]; }
  ADMP_HD void whole_mesh() { xoff = 0; nloc0 = K[0]; wrap0 = K[0]; }
};

// fractional grid coordinate of an atom along dimension d: returns f, writes the base index m_u0 - 3 (wrapped)
template <class T>
ADMP_HD T grid_ref(const RecipGeom<T>& g, const T r[3], int d, int& base) {
  T frac = r[0] * g.hinv[0 + d] + r[1] * g.hinv[3 + d] + r[2] * g.hinv[6 + d];
  T rm = frac * T(g.K[d]);
  T c = m_ceil(rm);
  int m = (int)c - 3 - (d == 0 ? g.xoff : 0);
  m %= g.K[d];
  if (m < 0) m += g.K[d];
  base = m;
  return c - rm;
}

// Fold a site's multipoles (global harmonics, quadrupoles to be divided by 3: recip.py:305) into
// u-space coefficients: value(grid) = q th + sum_j c1[j] D_j + sum_{m<=n} c2[mn] DD_mn,
// c2 order (00, 11, 22, 01, 02, 12), off-diagonals already doubled.
template <class T>
ADMP_HD void fold_multipole(const RecipGeom<T>& g, const T Q[9], T c1[3], T c2[6]) {
  const T* A = g.Aop;
  T dx = Q[2], dy = Q[3], dz = Q[1];
  for (int j = 0; j < 3; ++j) c1[j] = dx * A[0 + j] + dy * A[3 + j] + dz * A[6 + j];
  const T h = T(0.5 * kSqrt3 / 3.0), third = T(1.0 / 3.0);
  T tzz = Q[4] * third;
  T txx = T(0.5) * (-Q[4] + T(kSqrt3) * Q[7]) * third;
  T tyy = T(0.5) * (-Q[4] - T(kSqrt3) * Q[7]) * third;
  T txz = h * Q[5], tyz = h * Q[6], txy = h * Q[8];
  // W = Theta/3 . A   (3x3),  c2 = A^T W
  T W[9];
  for (int j = 0; j < 3; ++j) {
    W[0 + j] = txx * A[0 + j] + txy * A[3 + j] + txz * A[6 + j];
    W[3 + j] = txy * A[0 + j] + tyy * A[3 + j] + tyz * A[6 + j];
    W[6 + j] = txz * A[0 + j] + tyz * A[3 + j] + tzz * A[6 + j];
  }
  auto at = [&](int m, int n) { return A[0 + m] * W[0 + n] + A[3 + m] * W[3 + n] + A[6 + m] * W[6 + n]; };
  c2[0] = at(0, 0);
  c2[1] = at(1, 1);
  c2[2] = at(2, 2);
  c2[3] = T(2) * at(0, 1);
  c2[4] = T(2) * at(0, 2);
  c2[5] = T(2) * at(1, 2);
}

// The 20 u-space derivative sums of the potential grid at one atom:
// F[abc] = sum_grid phi * M0^(a) M1^(b) M2^(c), a+b+c <= 3, in this order.
enum { F000, F100, F010, F001, F200, F020, F002, F110, F101, F011, F300, F030, F003, F210, F201, F120, F021, F102,
       F012, F111, NF };

// dE/dQ (global harmonics) and dE/dr from the F sums.  pot/grad are ADDED to.
template <class T>
ADMP_HD void unfold_potential(const RecipGeom<T>& g, const T Q[9], const T* F, T pot[9], T grad[3]) {
  const T* A = g.Aop;
  pot[0] += F[F000];
  const T F1[3] = {F[F100], F[F010], F[F001]};
  T gxyz[3];
  for (int i = 0; i < 3; ++i) gxyz[i] = A[3 * i + 0] * F1[0] + A[3 * i + 1] * F1[1] + A[3 * i + 2] * F1[2];
  pot[1] += gxyz[2];
  pot[2] += gxyz[0];
  pot[3] += gxyz[1];
  const T F2[9] = {F[F200], F[F110], F[F101], F[F110], F[F020], F[F011], F[F101], F[F011], F[F002]};
  // G = A F2 A^T
  T W[9], G[9];
  for (int i = 0; i < 3; ++i)
    for (int n = 0; n < 3; ++n) W[3 * i + n] = A[3 * i + 0] * F2[0 + n] + A[3 * i + 1] * F2[3 + n] + A[3 * i + 2] * F2[6 + n];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) G[3 * i + k] = W[3 * i + 0] * A[3 * k + 0] + W[3 * i + 1] * A[3 * k + 1] + W[3 * i + 2] * A[3 * k + 2];
  const T third = T(1.0 / 3.0), r3 = T(kSqrt3);
  T tr = G[0] + G[4] + G[8];
  pot[4] += third * T(0.5) * (T(3) * G[8] - tr);
  pot[5] += third * r3 * G[2];
  pot[6] += third * r3 * G[5];
  pot[7] += third * T(0.5) * r3 * (G[0] - G[4]);
  pot[8] += third * r3 * G[1];
  // force: dE/du_j = q F1_j + sum_k c1_k F2_kj + sum_{mn} c2_mn F3_mnj ; dE/dx_i = sum_j Jac[i][j] dE/du_j
  T c1[3], c2[6];
  fold_multipole(g, Q, c1, c2);
  T fu[3];
  fu[0] = Q[0] * F1[0] + c1[0] * F[F200] + c1[1] * F[F110] + c1[2] * F[F101] + c2[0] * F[F300] + c2[1] * F[F120] +
          c2[2] * F[F102] + c2[3] * F[F210] + c2[4] * F[F201] + c2[5] * F[F111];
  fu[1] = Q[0] * F1[1] + c1[0] * F[F110] + c1[1] * F[F020] + c1[2] * F[F011] + c2[0] * F[F210] + c2[1] * F[F030] +
          c2[2] * F[F012] + c2[3] * F[F120] + c2[4] * F[F111] + c2[5] * F[F021];
  fu[2] = Q[0] * F1[2] + c1[0] * F[F101] + c1[1] * F[F011] + c1[2] * F[F002] + c2[0] * F[F201] + c2[1] * F[F021] +
          c2[2] * F[F003] + c2[3] * F[F111] + c2[4] * F[F102] + c2[5] * F[F012];
  for (int i = 0; i < 3; ++i) grad[i] += g.Jac[3 * i + 0] * fu[0] + g.Jac[3 * i + 1] * fu[1] + g.Jac[3 * i + 2] * fu[2];
}

// Box-gradient terms of one site from its F sums (reciprocal space, positions fixed).  The mesh energy depends on the box
// (a) through the fractional coordinates u_j = -K_j (x . box^-1)_j: d/d(box^-1[c][j]) = x_c (-K_j) dE/du_j, which in terms
//     of the Cartesian reciprocal gradient g = Jac . dE/du is the outer product x (x) g           -> xw[3c+b] += x_c g_b
// (b) through the multipole operators Aop (fold_multipole): E = sum_j c1_j F1_j + sum_mn (Aop^T Th Aop)_mn F2_mn with
//     c1 = d . Aop, Th = Theta/3                                    -> y[3i+j] += d_i F1_j + 2 (Th Aop F2)_ij = dE/dAop_ij
// (the k vectors and the volume are handled in k space).  The host turns the sums into dE/dbox (engine.hip).
template <class T>
ADMP_HD void recip_box_terms(const RecipGeom<T>& g, const T r[3], const T Q[9], const T* F, double xw[9], double y[9]) {
  const T* A = g.Aop;
  T pot[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gr[3] = {0, 0, 0};
  unfold_potential(g, Q, F, pot, gr);
  for (int c = 0; c < 3; ++c)
    for (int b = 0; b < 3; ++b) xw[3 * c + b] += (double)r[c] * (double)gr[b];
  const T F1[3] = {F[F100], F[F010], F[F001]};
  const T F2[9] = {F[F200], F[F110], F[F101], F[F110], F[F020], F[F011], F[F101], F[F011], F[F002]};
  const T d[3] = {Q[2], Q[3], Q[1]};                       // harmonic (z, x, y) -> cartesian
  const T h = T(0.5 * kSqrt3 / 3.0), third = T(1.0 / 3.0);
  const T tzz = Q[4] * third, txx = T(0.5) * (-Q[4] + T(kSqrt3) * Q[7]) * third, tyy = T(0.5) * (-Q[4] - T(kSqrt3) * Q[7]) * third;
  const T txz = h * Q[5], tyz = h * Q[6], txy = h * Q[8];
  const T Th[9] = {txx, txy, txz, txy, tyy, tyz, txz, tyz, tzz};
  T AF[9];                                                 // Aop . F2
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) AF[3 * i + j] = A[3 * i + 0] * F2[0 + j] + A[3 * i + 1] * F2[3 + j] + A[3 * i + 2] * F2[6 + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      y[3 * i + j] += (double)(d[i] * F1[j] + T(2) * (Th[3 * i + 0] * AF[0 + j] + Th[3 * i + 1] * AF[3 + j] + Th[3 * i + 2] * AF[6 + j]));
}

// All stencil weights of one atom.
template <class T>
struct Stencil {
  int base[3];
  T M[3][6], D1[3][6], D2[3][6], D3[3][6];
  ADMP_HD void init(const RecipGeom<T>& g, const T r[3]) {
    for (int d = 0; d < 3; ++d) {
      T f = grid_ref(g, r, d, base[d]);
      bspline6(f, M[d], D1[d], D2[d], D3[d]);
    }
  }
};

ADMP_HD int wrap_add(int base, int p, int K) {
  int i = base + p;
  return i >= K ? i - K : i;
}

// Spread one site (admp/recip.py:313-329 Q_mesh_on_m for one atom): add(index, value) is called
// for its 216 mesh points, index = (i0*K1 + i1)*K2 + i2.
template <class T, class AddF>
ADMP_HD void spread_atom(const RecipGeom<T>& g, const T r[3], const T Q[9], AddF add) {
  Stencil<T> st;
  st.init(g, r);
  T c1[3], c2[6];
  fold_multipole(g, Q, c1, c2);
  const T q = Q[0];
  for (int a = 0; a < 6; ++a) {
    const int ia = wrap_add(st.base[0], a, g.wrap0);
    const T m0 = st.M[0][a], d0 = st.D1[0][a], e0 = st.D2[0][a];
    for (int b = 0; b < 6; ++b) {
      const int ib = wrap_add(st.base[1], b, g.K[1]);
      const T m1 = st.M[1][b], d1 = st.D1[1][b], e1 = st.D2[1][b];
      const T mm = m0 * m1;
      const T P0 = q * mm + c1[0] * d0 * m1 + c1[1] * m0 * d1 + c2[0] * e0 * m1 + c2[1] * m0 * e1 + c2[3] * d0 * d1;
      const T P1 = c1[2] * mm + c2[4] * d0 * m1 + c2[5] * m0 * d1;
      const T P2 = c2[2] * mm;
      const long row = ((long)ia * g.K[1] + ib) * g.K[2];
      for (int c = 0; c < 6; ++c) {
        const int ic = wrap_add(st.base[2], c, g.K[2]);
        add(row + ic, P0 * st.M[2][c] + P1 * st.D1[2][c] + P2 * st.D2[2][c]);
      }
    }
  }
}

// Gather the 20 derivative sums of phi at one site; phi(index) loads a mesh value.
template <class T, class LoadF>
ADMP_HD void gather_atom(const RecipGeom<T>& g, const T r[3], LoadF phi, T* F) {
  Stencil<T> st;
  st.init(g, r);
  for (int k = 0; k < NF; ++k) F[k] = T(0);
  for (int a = 0; a < 6; ++a) {
    const int ia = wrap_add(st.base[0], a, g.wrap0);
    T s00 = 0, s10 = 0, s20 = 0, s30 = 0, s01 = 0, s11 = 0, s21 = 0, s02 = 0, s12 = 0, s03 = 0;
    for (int b = 0; b < 6; ++b) {
      const int ib = wrap_add(st.base[1], b, g.K[1]);
      const long row = ((long)ia * g.K[1] + ib) * g.K[2];
      T t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      for (int c = 0; c < 6; ++c) {
        const T v = phi(row + wrap_add(st.base[2], c, g.K[2]));
        t0 += v * st.M[2][c];
        t1 += v * st.D1[2][c];
        t2 += v * st.D2[2][c];
        t3 += v * st.D3[2][c];
      }
      const T m1 = st.M[1][b], d1 = st.D1[1][b], e1 = st.D2[1][b], g1 = st.D3[1][b];
      s00 += t0 * m1; s10 += t0 * d1; s20 += t0 * e1; s30 += t0 * g1;
      s01 += t1 * m1; s11 += t1 * d1; s21 += t1 * e1;
      s02 += t2 * m1; s12 += t2 * d1;
      s03 += t3 * m1;
    }
    const T m0 = st.M[0][a], d0 = st.D1[0][a], e0 = st.D2[0][a], g0 = st.D3[0][a];
    F[F000] += m0 * s00; F[F100] += d0 * s00; F[F200] += e0 * s00; F[F300] += g0 * s00;
    F[F010] += m0 * s10; F[F110] += d0 * s10; F[F210] += e0 * s10;
    F[F020] += m0 * s20; F[F120] += d0 * s20;
    F[F030] += m0 * s30;
    F[F001] += m0 * s01; F[F101] += d0 * s01; F[F201] += e0 * s01;
    F[F011] += m0 * s11; F[F111] += d0 * s11;
    F[F021] += m0 * s21;
    F[F002] += m0 * s02; F[F102] += d0 * s02;
    F[F012] += m0 * s12;
    F[F003] += m0 * s03;
  }
}

// One x-plane (stencil index a along dimension 0) of gather_atom: the kernels give each of 6 lanes of an
// 8-lane group one plane and fold the 20 sums across the group.  w0[4] = (M, M1, M2, M3): the spline and
// its first three derivatives of dimension 0 at that plane.
template <class T, class LoadF>
ADMP_HD void gather_plane(const RecipGeom<T>& g, const Stencil<T>& st, int ia, const T w0[4], LoadF phi, T* F) {
  T s00 = 0, s10 = 0, s20 = 0, s30 = 0, s01 = 0, s11 = 0, s21 = 0, s02 = 0, s12 = 0, s03 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int b = 0; b < 6; ++b) {
    const int ib = wrap_add(st.base[1], b, g.K[1]);
    const long row = ((long)ia * g.K[1] + ib) * g.K[2];
    T t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < 6; ++c) {
      const T v = phi(row + wrap_add(st.base[2], c, g.K[2]));
      t0 += v * st.M[2][c];
      t1 += v * st.D1[2][c];
      t2 += v * st.D2[2][c];
      t3 += v * st.D3[2][c];
    }
    const T m1 = st.M[1][b], d1 = st.D1[1][b], e1 = st.D2[1][b], g1 = st.D3[1][b];
    s00 += t0 * m1; s10 += t0 * d1; s20 += t0 * e1; s30 += t0 * g1;
    s01 += t1 * m1; s11 += t1 * d1; s21 += t1 * e1;
    s02 += t2 * m1; s12 += t2 * d1;
    s03 += t3 * m1;
  }
  const T m0 = w0[0], d0 = w0[1], e0 = w0[2], g0 = w0[3];
  F[F000] = m0 * s00; F[F100] = d0 * s00; F[F200] = e0 * s00; F[F300] = g0 * s00;
  F[F010] = m0 * s10; F[F110] = d0 * s10; F[F210] = e0 * s10;
  F[F020] = m0 * s20; F[F120] = d0 * s20;
  F[F030] = m0 * s30;
  F[F001] = m0 * s01; F[F101] = d0 * s01; F[F201] = e0 * s01;
  F[F011] = m0 * s11; F[F111] = d0 * s11;
  F[F021] = m0 * s21;
  F[F002] = m0 * s02; F[F102] = d0 * s02;
  F[F012] = m0 * s12;
  F[F003] = m0 * s03;
}

// One z-index (stencil index c along dimension 2) of gather_atom: lane c of a 6-lane set reads, for every (a, b), the
// mesh word next to its neighbours' -- the six loads of one (a, b) fall into one or two cache lines, where the
// per-plane split above touches six.  wz[4] = the dimension-2 spline and its first three derivatives at that index.
template <class T, class LoadF>
ADMP_HD void gather_zcol(const RecipGeom<T>& g, const Stencil<T>& st, int ic, const T wz[4], LoadF phi, T* F) {
  T u00 = 0, u10 = 0, u20 = 0, u30 = 0, u01 = 0, u11 = 0, u21 = 0, u02 = 0, u12 = 0, u03 = 0;   // u[i along x][j along y]
  // mesh index = ia K1 K2 + (ib K2 + ic): the y-z part is formed once for the six b (integer multiplies run at a quarter of
  // the FMA rate: one 64-bit multiply-add per (a, b) was a third of this function's issue time)
  long rb[6];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int b = 0; b < 6; ++b) rb[b] = (long)wrap_add(st.base[1], b, g.K[1]) * g.K[2] + ic;
  const long k12 = (long)g.K[1] * g.K[2];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int a = 0; a < 6; ++a) {
    const long ra = (long)wrap_add(st.base[0], a, g.wrap0) * k12;
    T t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 6; ++b) {
      const T v = phi(ra + rb[b]);
      t0 += v * st.M[1][b];
      t1 += v * st.D1[1][b];
      t2 += v * st.D2[1][b];
      t3 += v * st.D3[1][b];
    }
    const T m0 = st.M[0][a], d0 = st.D1[0][a], e0 = st.D2[0][a], g0 = st.D3[0][a];
    u00 += m0 * t0; u10 += d0 * t0; u20 += e0 * t0; u30 += g0 * t0;
    u01 += m0 * t1; u11 += d0 * t1; u21 += e0 * t1;
    u02 += m0 * t2; u12 += d0 * t2;
    u03 += m0 * t3;
  }
  const T m2 = wz[0], d2 = wz[1], e2 = wz[2], g2 = wz[3];
  F[F000] = m2 * u00; F[F100] = m2 * u10; F[F200] = m2 * u20; F[F300] = m2 * u30;
  F[F010] = m2 * u01; F[F110] = m2 * u11; F[F210] = m2 * u21;
  F[F020] = m2 * u02; F[F120] = m2 * u12;
  F[F030] = m2 * u03;
  F[F001] = d2 * u00; F[F101] = d2 * u10; F[F201] = d2 * u20;
  F[F011] = d2 * u01; F[F111] = d2 * u11;
  F[F021] = d2 * u02;
  F[F002] = e2 * u00; F[F102] = e2 * u10;
  F[F012] = e2 * u01;
  F[F003] = g2 * u00;
}
// The same z-index sums with the spline weights taken from staged rows (the gather's workgroups evaluate every spline once
// and keep it in LDS: W4 = the four orders of one stencil index side by side, one 16-byte read).  wx / wy: the six rows of
// dimensions 0 / 1, wz: the lane's own row of dimension 2.  Mesh offsets advance by additions (no multiplies in the loops).
template <class T>
struct alignas(4 * sizeof(T)) W4 {
  T m, d1, d2, d3;
};

// load(a, b) returns the mesh word of stencil point (a, b) at the lane's z index
template <class T, class LoadAB>
ADMP_HD void gather_zcol_core(const W4<T>* wx, const W4<T>* wy, const W4<T>& wz, LoadAB load, T* F) {
  T u00 = 0, u10 = 0, u20 = 0, u30 = 0, u01 = 0, u11 = 0, u21 = 0, u02 = 0, u12 = 0, u03 = 0;   // u[i along x][j along y]
  W4<T> y[6];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int b = 0; b < 6; ++b) y[b] = wy[b];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int a = 0; a < 6; ++a) {
    T t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 6; ++b) {
      const T v = load(a, b);
      t0 += v * y[b].m;
      t1 += v * y[b].d1;
      t2 += v * y[b].d2;
      t3 += v * y[b].d3;
    }
    const W4<T> x = wx[a];
    u00 += x.m * t0; u10 += x.d1 * t0; u20 += x.d2 * t0; u30 += x.d3 * t0;
    u01 += x.m * t1; u11 += x.d1 * t1; u21 += x.d2 * t1;
    u02 += x.m * t2; u12 += x.d1 * t2;
    u03 += x.m * t3;
  }
  const T m2 = wz.m, d2 = wz.d1, e2 = wz.d2, g2 = wz.d3;
  F[F000] = m2 * u00; F[F100] = m2 * u10; F[F200] = m2 * u20; F[F300] = m2 * u30;
  F[F010] = m2 * u01; F[F110] = m2 * u11; F[F210] = m2 * u21;
  F[F020] = m2 * u02; F[F120] = m2 * u12;
  F[F030] = m2 * u03;
  F[F001] = d2 * u00; F[F101] = d2 * u10; F[F201] = d2 * u20;
  F[F011] = d2 * u01; F[F111] = d2 * u11;
  F[F021] = d2 * u02;
  F[F002] = e2 * u00; F[F102] = e2 * u10;
  F[F012] = e2 * u01;
  F[F003] = g2 * u00;
}
template <class T, class LoadF>
ADMP_HD void gather_zcol_w(const RecipGeom<T>& g, const int base[3], const W4<T>* wx, const W4<T>* wy, const W4<T>& wz, int ic,
                           LoadF phi, T* F) {
  T u00 = 0, u10 = 0, u20 = 0, u30 = 0, u01 = 0, u11 = 0, u21 = 0, u02 = 0, u12 = 0, u03 = 0;   // u[i along x][j along y]
  W4<T> y[6];
  long rb[6];
  {
    int ib = base[1];
    long r = (long)ib * g.K[2] + ic;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 6; ++b) {
      y[b] = wy[b];
      rb[b] = r;
      r += g.K[2];
      if (++ib == g.K[1]) { ib = 0; r = ic; }
    }
  }
  const long k12 = (long)g.K[1] * g.K[2];
  int ia = base[0];
  long ra = (long)ia * k12;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int a = 0; a < 6; ++a) {
    T t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 6; ++b) {
      const T v = phi(ra + rb[b]);
      t0 += v * y[b].m;
      t1 += v * y[b].d1;
      t2 += v * y[b].d2;
      t3 += v * y[b].d3;
    }
    const W4<T> x = wx[a];
    u00 += x.m * t0; u10 += x.d1 * t0; u20 += x.d2 * t0; u30 += x.d3 * t0;
    u01 += x.m * t1; u11 += x.d1 * t1; u21 += x.d2 * t1;
    u02 += x.m * t2; u12 += x.d1 * t2;
    u03 += x.m * t3;
    ra += k12;
    if (++ia == g.wrap0) { ia = 0; ra = 0; }
  }
  const T m2 = wz.m, d2 = wz.d1, e2 = wz.d2, g2 = wz.d3;
  F[F000] = m2 * u00; F[F100] = m2 * u10; F[F200] = m2 * u20; F[F300] = m2 * u30;
  F[F010] = m2 * u01; F[F110] = m2 * u11; F[F210] = m2 * u21;
  F[F020] = m2 * u02; F[F120] = m2 * u12;
  F[F030] = m2 * u03;
  F[F001] = d2 * u00; F[F101] = d2 * u10; F[F201] = d2 * u20;
  F[F011] = d2 * u01; F[F111] = d2 * u11;
  F[F021] = d2 * u02;
  F[F002] = e2 * u00; F[F102] = e2 * u10;
  F[F012] = e2 * u01;
  F[F003] = g2 * u00;
}
// ... and its first-derivative part (SCF field, scalar dispersion channels)
template <class T, class LoadF>
ADMP_HD void gather_zcol_field_w(const RecipGeom<T>& g, const int base[3], const W4<T>* wx, const W4<T>* wy, const W4<T>& wz,
                                 int ic, LoadF phi, T f[3]) {
  T u00 = 0, u10 = 0, u01 = 0;
  T ym[6], yd[6];
  long rb[6];
  {
    int ib = base[1];
    long r = (long)ib * g.K[2] + ic;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 6; ++b) {
      ym[b] = wy[b].m; yd[b] = wy[b].d1;
      rb[b] = r;
      r += g.K[2];
      if (++ib == g.K[1]) { ib = 0; r = ic; }
    }
  }
  const long k12 = (long)g.K[1] * g.K[2];
  int ia = base[0];
  long ra = (long)ia * k12;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int a = 0; a < 6; ++a) {
    T t0 = 0, t1 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 6; ++b) {
      const T v = phi(ra + rb[b]);
      t0 += v * ym[b];
      t1 += v * yd[b];
    }
    const T xm = wx[a].m, xd = wx[a].d1;
    u00 += xm * t0;
    u10 += xd * t0;
    u01 += xm * t1;
    ra += k12;
    if (++ia == g.wrap0) { ia = 0; ra = 0; }
  }
  f[0] = wz.m * u10;
  f[1] = wz.m * u01;
  f[2] = wz.d1 * u00;
}
// first-derivative part of one z-index: f[3] = (F100, F010, F001) contributions
template <class T, class LoadF>
ADMP_HD void gather_zcol_field(const RecipGeom<T>& g, const Stencil<T>& st, int ic, T m2, T d2, LoadF phi, T f[3]) {
  T u00 = 0, u10 = 0, u01 = 0;
  long rb[6];                                          // see gather_zcol
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int b = 0; b < 6; ++b) rb[b] = (long)wrap_add(st.base[1], b, g.K[1]) * g.K[2] + ic;
  const long k12 = (long)g.K[1] * g.K[2];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int a = 0; a < 6; ++a) {
    const long ra = (long)wrap_add(st.base[0], a, g.wrap0) * k12;
    T t0 = 0, t1 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 6; ++b) {
      const T v = phi(ra + rb[b]);
      t0 += v * st.M[1][b];
      t1 += v * st.D1[1][b];
    }
    u00 += st.M[0][a] * t0;
    u10 += st.D1[0][a] * t0;
    u01 += st.M[0][a] * t1;
  }
  f[0] = m2 * u10;
  f[1] = m2 * u01;
  f[2] = d2 * u00;
}

// first-derivative part of one plane: f[3] = (F100, F010, F001) contributions
template <class T, class LoadF>
ADMP_HD void gather_plane_field(const RecipGeom<T>& g, const Stencil<T>& st, int ia, T m0, T d0, LoadF phi, T f[3]) {
  T s00 = 0, s10 = 0, s01 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int b = 0; b < 6; ++b) {
    const int ib = wrap_add(st.base[1], b, g.K[1]);
    const long row = ((long)ia * g.K[1] + ib) * g.K[2];
    T t0 = 0, t1 = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < 6; ++c) {
      const T v = phi(row + wrap_add(st.base[2], c, g.K[2]));
      t0 += v * st.M[2][c];
      t1 += v * st.D1[2][c];
    }
    s00 += t0 * st.M[1][b];
    s10 += t0 * st.D1[1][b];
    s01 += t1 * st.M[1][b];
  }
  f[0] = d0 * s00;
  f[1] = m0 * s10;
  f[2] = m0 * s01;
}

// First-derivative sums only (the SCF needs just dE/d(dipole)): returns cartesian-operator gradient
// gxyz[i] = sum_j Aop[i][j] F1_j.
template <class T, class LoadF>
ADMP_HD void gather_atom_field(const RecipGeom<T>& g, const T r[3], LoadF phi, T gxyz[3]) {
  Stencil<T> st;
  st.init(g, r);
  T f0 = 0, f1 = 0, f2 = 0;
  for (int a = 0; a < 6; ++a) {
    const int ia = wrap_add(st.base[0], a, g.wrap0);
    T s00 = 0, s10 = 0, s01 = 0;
    for (int b = 0; b < 6; ++b) {
      const int ib = wrap_add(st.base[1], b, g.K[1]);
      const long row = ((long)ia * g.K[1] + ib) * g.K[2];
      T t0 = 0, t1 = 0;
      for (int c = 0; c < 6; ++c) {
        const T v = phi(row + wrap_add(st.base[2], c, g.K[2]));
        t0 += v * st.M[2][c];
        t1 += v * st.D1[2][c];
      }
      s00 += t0 * st.M[1][b];
      s10 += t0 * st.D1[1][b];
      s01 += t1 * st.M[1][b];
    }
    f0 += st.D1[0][a] * s00;
    f1 += st.M[0][a] * s10;
    f2 += st.M[0][a] * s01;
  }
  const T* A = g.Aop;
  for (int i = 0; i < 3; ++i) gxyz[i] = A[3 * i + 0] * f0 + A[3 * i + 1] * f1 + A[3 * i + 2] * f2;
}

// theta_k factor of one dimension (admp/recip.py:400-408): sum_{m=-2..2} M6(m+3) cos(2 pi m k / K)
ADMP_HD double theta_k_1d(int k, int K) {
  const double w = 6.283185307179586 * (double)k / (double)K;
  return (66.0 + 52.0 * cos(w) + 2.0 * cos(2.0 * w)) / 120.0;
}

// FFT-ordered signed frequency of index i on a K-point axis (admp/recip.py:332-341)
ADMP_HD int signed_freq(int i, int K) { return i <= (K - 1) / 2 ? i : i - K; }

}  // namespace admp
