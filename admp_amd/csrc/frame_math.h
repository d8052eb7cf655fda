// Per-atom local frames (reference admp/spatial.py:44-147) and their hand-coded adjoint.
//
// Forward: frame rows (x,y,z) from the site and its z/x/y axis atoms under the five
// MPID/AMOEBA axis rules; Q_global = rot_local2global(Q_local, frame) (admp/pme.py:221-222).
// Backward (what jax.grad does implicitly, admp/pme.py:108): the cotangent of Q_global is
// folded into a torque tau (rotation generators of pme_math.h); a rotation d(omega) of an
// orthonormal frame moves row k by d(omega) x row_k, so dE = tau . d(omega) gives the row
// cotangents c_k = (tau x row_k)/2, which are then pulled back through normalise / bisect /
// Gram-Schmidt / cross by ordinary reverse-mode rules onto the four atoms' positions.
#pragma once
#include "pme_math.h"

namespace admp {

enum AxisType { ZThenX = 0, Bisector = 1, ZBisect = 2, ThreeFold = 3, Zonly = 4, NoAxisType = 5 };

template <class T> ADMP_HD T dot3(const T a[3], const T b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> ADMP_HD void cross3(const T a[3], const T b[3], T c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
// n = v/|v| ; returns 1/|v|
template <class T> ADMP_HD T unit3(const T v[3], T n[3]) {
  T inv = T(1) / m_sqrt(dot3(v, v));
  n[0] = v[0] * inv; n[1] = v[1] * inv; n[2] = v[2] * inv;
  return inv;
}
// cotangent of v given cotangent nb of n = v/|v|
template <class T> ADMP_HD void unit3_bwd(const T n[3], T inv, const T nb[3], T vb[3]) {
  T s = dot3(n, nb);
  vb[0] = (nb[0] - n[0] * s) * inv;
  vb[1] = (nb[1] - n[1] * s) * inv;
  vb[2] = (nb[2] - n[2] * s) * inv;
}

// Intermediates of one frame, kept between forward and backward.
template <class T>
struct FrameWork {
  T vz0[3], vx0[3], vy0[3];   // unit vectors site -> z/x/y atom
  T iz, ix, iy;               // 1/|displacement|
  T vz1[3], vx1[3];           // after the bisector / z-bisect / threefold step
  T isz, isx;                 // 1/|sum| of those steps
  T iw;                       // 1/|x - z (x.z)|
  T X[3], Y[3], Z[3];         // the frame rows
};

template <class T>
ADMP_HD void local_frame_fwd(int type, const Box<T>& box, const T p[3], const T pz[3], const T px[3], const T py[3],
                             FrameWork<T>& w) {
  if (type == NoAxisType) {   // no axis atoms: identity frame (only charges may sit on such a site)
    w.X[0] = 1; w.X[1] = 0; w.X[2] = 0;
    w.Y[0] = 0; w.Y[1] = 1; w.Y[2] = 0;
    w.Z[0] = 0; w.Z[1] = 0; w.Z[2] = 1;
    return;
  }
  T d[3] = {pz[0] - p[0], pz[1] - p[1], pz[2] - p[2]};
  min_image(box, d);
  w.iz = unit3(d, w.vz0);
  if (type == Zonly) {   // admp/spatial.py:103-105
    T xz = m_floor(m_abs(w.vz0[0]) + T(0.5));
    w.vx0[0] = T(1) - xz; w.vx0[1] = xz; w.vx0[2] = T(0);
    w.ix = T(0);
  } else {
    T e[3] = {px[0] - p[0], px[1] - p[1], px[2] - p[2]};
    min_image(box, e);
    w.ix = unit3(e, w.vx0);
  }
  w.iy = T(0);
  if (type == ZBisect || type == ThreeFold) {
    T f[3] = {py[0] - p[0], py[1] - p[1], py[2] - p[2]};
    min_image(box, f);
    w.iy = unit3(f, w.vy0);
  }
  for (int k = 0; k < 3; ++k) { w.vz1[k] = w.vz0[k]; w.vx1[k] = w.vx0[k]; }
  w.isz = T(0); w.isx = T(0);
  if (type == Bisector) {          // spatial.py:112-114
    T s[3] = {w.vz0[0] + w.vx0[0], w.vz0[1] + w.vx0[1], w.vz0[2] + w.vx0[2]};
    w.isz = unit3(s, w.vz1);
  } else if (type == ZBisect) {    // spatial.py:116-121
    T s[3] = {w.vx0[0] + w.vy0[0], w.vx0[1] + w.vy0[1], w.vx0[2] + w.vy0[2]};
    w.isx = unit3(s, w.vx1);
  } else if (type == ThreeFold) {  // spatial.py:123-134
    T s[3] = {w.vz0[0] + w.vx0[0] + w.vy0[0], w.vz0[1] + w.vx0[1] + w.vy0[1], w.vz0[2] + w.vx0[2] + w.vy0[2]};
    w.isz = unit3(s, w.vz1);
  }
  // Gram-Schmidt and y = z cross x (spatial.py:137-142)
  T proj = dot3(w.vx1, w.vz1);
  T u[3] = {w.vx1[0] - w.vz1[0] * proj, w.vx1[1] - w.vz1[1] * proj, w.vx1[2] - w.vz1[2] * proj};
  w.iw = unit3(u, w.X);
  for (int k = 0; k < 3; ++k) w.Z[k] = w.vz1[k];
  cross3(w.Z, w.X, w.Y);
}

// Torque (dE/d(rotation vector), global axes) of a multipole Q with potential P = dE/dQ,
// both in global harmonics.
template <class T>
ADMP_HD void multipole_torque(const T P[9], const T Q[9], T tau[3]) {
  tau[0] = gen_toward_y(P, Q);    // object rotation about +x == frame turning z -> y
  tau[1] = -gen_toward_x(P, Q);   // object rotation about +y == frame turning x -> z
  tau[2] = gen_about_z(P, Q);
}

// Reverse pass: tau -> dE/d(position) of the site (gp), its z atom (gz), x atom (gx), y atom (gy).
// Outputs are overwritten.
template <class T>
ADMP_HD void local_frame_bwd(int type, const FrameWork<T>& w, const T tau[3], T gp[3], T gz[3], T gx[3], T gy[3]) {
  for (int k = 0; k < 3; ++k) { gp[k] = T(0); gz[k] = T(0); gx[k] = T(0); gy[k] = T(0); }
  if (type == NoAxisType) return;
  T cX[3], cY[3], cZ[3], t[3];
  cross3(tau, w.X, cX);
  cross3(tau, w.Y, cY);
  cross3(tau, w.Z, cZ);
  for (int k = 0; k < 3; ++k) { cX[k] *= T(0.5); cY[k] *= T(0.5); cZ[k] *= T(0.5); }
  // Y = Z x X
  cross3(w.X, cY, t);
  for (int k = 0; k < 3; ++k) cZ[k] += t[k];
  cross3(cY, w.Z, t);
  for (int k = 0; k < 3; ++k) cX[k] += t[k];
  // X = unit(u), u = vx1 - vz1 (vx1.vz1)
  T ub[3];
  unit3_bwd(w.X, w.iw, cX, ub);
  T proj = dot3(w.vx1, w.vz1);
  T pb = -dot3(w.vz1, ub);
  T vx1b[3], vz1b[3];
  for (int k = 0; k < 3; ++k) {
    vx1b[k] = ub[k] + pb * w.vz1[k];
    vz1b[k] = cZ[k] - proj * ub[k] + pb * w.vx1[k];
  }
  T vz0b[3], vx0b[3], vy0b[3] = {T(0), T(0), T(0)};
  if (type == Bisector) {
    T sb[3];
    unit3_bwd(w.vz1, w.isz, vz1b, sb);
    for (int k = 0; k < 3; ++k) { vz0b[k] = sb[k]; vx0b[k] = vx1b[k] + sb[k]; }
  } else if (type == ThreeFold) {
    T sb[3];
    unit3_bwd(w.vz1, w.isz, vz1b, sb);
    for (int k = 0; k < 3; ++k) { vz0b[k] = sb[k]; vx0b[k] = vx1b[k] + sb[k]; vy0b[k] = sb[k]; }
  } else if (type == ZBisect) {
    T sb[3];
    unit3_bwd(w.vx1, w.isx, vx1b, sb);
    for (int k = 0; k < 3; ++k) { vz0b[k] = vz1b[k]; vx0b[k] = sb[k]; vy0b[k] = sb[k]; }
  } else {
    for (int k = 0; k < 3; ++k) { vz0b[k] = vz1b[k]; vx0b[k] = vx1b[k]; }
  }
  // unit displacement vectors -> positions (min-image shift has unit Jacobian)
  unit3_bwd(w.vz0, w.iz, vz0b, t);
  for (int k = 0; k < 3; ++k) { gz[k] = t[k]; gp[k] -= t[k]; }
  if (type != Zonly) {
    unit3_bwd(w.vx0, w.ix, vx0b, t);
    for (int k = 0; k < 3; ++k) { gx[k] = t[k]; gp[k] -= t[k]; }
  }
  if (type == ZBisect || type == ThreeFold) {
    unit3_bwd(w.vy0, w.iy, vy0b, t);
    for (int k = 0; k < 3; ++k) { gy[k] = t[k]; gp[k] -= t[k]; }
  }
}

// PME self term (admp/pme.py:738-757): E = -D sum_h f_l Q_h^2, f_l = kappa/sqrt(pi) (2 kappa^2)^l/(2l+1)!!
template <class T>
ADMP_HD void self_factors(T kappa, T f[3]) {
  T k2 = T(2) * kappa * kappa;
  f[0] = kappa * T(0.5 * kTwoOverSqrtPi);
  f[1] = f[0] * k2 / T(3);
  f[2] = f[0] * k2 * k2 / T(15);
}

// Cartesian dE/dU of one atom (admp/pme.py:111-143: the SCF residual): real-space part (harmonic order z,x,y) + reciprocal
// part (cartesian) + self term + polarization penalty D U / max(pol, 1e-8).
template <class T>
ADMP_HD void total_field(const Site<T>& s, T a, const T* U, const T* fld_pair, const T* fld_recip, T kappa, T& fx, T& fy,
                         T& fz) {
  T f[3];
  self_factors(kappa, f);
  const T twoDf1 = T(2.0 * kDielectric) * f[1];
  const T ainv = T(kDielectric) / (a < T(1e-8) ? T(1e-8) : a);   // d/dU of D U^2 / (2 max(pol, 1e-8))
  const T hz = fld_pair[0] - twoDf1 * (s.Q[1] + s.U[0]);
  const T hx = fld_pair[1] - twoDf1 * (s.Q[2] + s.U[1]);
  const T hy = fld_pair[2] - twoDf1 * (s.Q[3] + s.U[2]);
  fx = hx + fld_recip[0] + ainv * U[0];
  fy = hy + fld_recip[1] + ainv * U[1];
  fz = hz + fld_recip[2] + ainv * U[2];
}

}  // namespace admp
