// Dispersion PME real-space pair term (reference admp/disp_pme.py:179-251), its k-space kernels
// (admp/recip.py:437-462) and the Tang-Toennies damping pair kernel (admp/pairwise.py:94-113),
// each with the explicit radial derivative that the reference leaves to autodiff.
#pragma once
#include "pme_math.h"

namespace admp {

// E = sum_{p=6,8,10} (m + g_p(kappa^2 r^2) - 1) c_p,i c_p,j / r^p ; adds dE/dr_i to gi (dE/dr_j = -gi).
// mm = mscale - 1.
template <class T>
ADMP_HD T disp_pair(const Box<T>& box, const T ri[3], const T rj[3], const T ci[3], const T cj[3], T mm, T kappa,
                    int pmax, T gi[3]) {
  T d[3] = {ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]};
  min_image(box, d);
  T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  T k2 = kappa * kappa;
  T x2 = k2 * r2, x4 = x2 * x2;
  T ex = m_exp(-x2);
  T ir2 = T(1) / r2, ir6 = ir2 * ir2 * ir2;
  // g_p and d g_p / d(x^2) (disp_pme.py:219-251): g6 = (1 + x2 + x4/2) e, g8 = g6 + x6/6 e, g10 = g8 + x8/24 e
  T g = (T(1) + x2 + T(0.5) * x4) * ex;
  T dg = -T(0.5) * x4 * ex;
  T cc = ci[0] * cj[0];
  T e = (mm + g) * cc * ir6;
  T de = (k2 * dg - T(3) * (mm + g) * ir2) * cc * ir6;      // dE/d(r^2)
  if (pmax >= 8) {
    T x6e = x4 * x2 * ex;
    g += x6e * T(1.0 / 6.0);
    dg = -x6e * T(1.0 / 6.0);
    T ir8 = ir6 * ir2;
    cc = ci[1] * cj[1];
    e += (mm + g) * cc * ir8;
    de += (k2 * dg - T(4) * (mm + g) * ir2) * cc * ir8;
    if (pmax >= 10) {
      T x8e = x4 * x4 * ex;
      g += x8e * T(1.0 / 24.0);
      dg = -x8e * T(1.0 / 24.0);
      T ir10 = ir8 * ir2;
      cc = ci[2] * cj[2];
      e += (mm + g) * cc * ir10;
      de += (k2 * dg - T(5) * (mm + g) * ir2) * cc * ir10;
    }
  }
  gi[0] += T(2) * de * d[0];
  gi[1] += T(2) * de * d[1];
  gi[2] += T(2) * de * d[2];
  return e;
}

// Tang-Toennies damped exchange / charge penetration / C6 (pairwise.py:94-113).
// per-atom parameter quadruple (a, b, q, c6); m = mscale.
template <class T>
ADMP_HD T tt_pair(const Box<T>& box, const T ri[3], const T rj[3], const T pi[4], const T pj[4], T m, T gi[3]) {
  T d[3] = {ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]};
  min_image(box, d);
  T rr = m_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  typedef Dual<T> S;
  S r(rr, T(1));
  T a = m_sqrt(pi[0] * pj[0]);
  T b = m_sqrt(pi[1] * pj[1]);
  T q = pi[2] * pj[2];
  T c = pi[3] * pj[3];
  S br = r * (b * T(1.889726878));
  S ebr = m_exp(-br);
  S br2 = br * br, br3 = br2 * br;
  S poly = br + T(1) + br2 * T(0.5) + br3 * T(1.0 / 6.0) + br2 * br2 * T(1.0 / 24.0) + br2 * br3 * T(1.0 / 120.0) +
           br3 * br3 * T(1.0 / 720.0);
  S ir = recip(r);
  S ir2 = ir * ir;
  S ir6 = ir2 * ir2 * ir2;
  S f = ebr * (T(2625.5) * a) + ebr * (br + T(1)) * recip(br) * (T(-2625.5) * q) + ebr * poly * ir6 * c;
  T s = m * f.d / rr;
  gi[0] += s * d[0];
  gi[1] += s * d[1];
  gi[2] += s * d[2];
  return m * f.v;
}

// Parameter derivatives of the two scalar pair terms (what jax.grad(pot_disp, argnums=3) differentiates in the reference's
// examples/openmm_api/run.py:41-43 through admp/api.py:183-199).
// Dispersion: E_pair = sum_p (mm + g_p) c_p,i c_p,j / r^p is linear in c_p,i: out[p] += dE_pair/dc_p,i.
template <class T>
ADMP_HD void disp_pair_dc(const Box<T>& box, const T ri[3], const T rj[3], const T cj[3], T mm, T kappa, int pmax, T out[3]) {
  T d[3] = {ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T x2 = kappa * kappa * r2, x4 = x2 * x2, ex = m_exp(-x2);
  const T ir2 = T(1) / r2, ir6 = ir2 * ir2 * ir2;
  T g = (T(1) + x2 + T(0.5) * x4) * ex;
  out[0] += (mm + g) * cj[0] * ir6;
  if (pmax >= 8) {
    g += x4 * x2 * ex * T(1.0 / 6.0);
    out[1] += (mm + g) * cj[1] * ir6 * ir2;
    if (pmax >= 10) {
      g += x4 * x4 * ex * T(1.0 / 24.0);
      out[2] += (mm + g) * cj[2] * ir6 * ir2 * ir2;
    }
  }
}
// Tang-Toennies: E_pair = m f(r; a, b, q, c) with a = sqrt(a_i a_j), b = sqrt(b_i b_j), q = q_i q_j, c = c_i c_j
// (pairwise.py:94-113): out[0..3] += dE_pair/d(a_i, b_i, q_i, c_i).  d/db at fixed r through x = 1.889726878 r b by a dual
// number.  A zero a_i or b_i has an infinite one-sided derivative (sqrt at 0; NaN in the reference's autodiff): 0 is returned.
template <class T>
ADMP_HD void tt_pair_dparams(const Box<T>& box, const T ri[3], const T rj[3], const T pi[4], const T pj[4], T m, T out[4]) {
  T d[3] = {ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]};
  min_image(box, d);
  const T rr = m_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  typedef Dual<T> S;
  const T a = m_sqrt(pi[0] * pj[0]), b = m_sqrt(pi[1] * pj[1]), q = pi[2] * pj[2], c = pi[3] * pj[3];
  const S x(rr * b * T(1.889726878), T(1));
  const S ex = m_exp(-x);
  const S x2 = x * x, x3 = x2 * x;
  const S poly = x + T(1) + x2 * T(0.5) + x3 * T(1.0 / 6.0) + x2 * x2 * T(1.0 / 24.0) + x2 * x3 * T(1.0 / 120.0) +
                 x3 * x3 * T(1.0 / 720.0);
  const T ir2 = T(1) / (rr * rr), ir6 = ir2 * ir2 * ir2;
  const S fq = ex * (x + T(1)) * recip(x) * T(-2625.5);      // df/dq
  const S fc = ex * poly * ir6;                             // df/dc
  const S f = ex * (T(2625.5) * a) + fq * q + fc * c;
  if (pi[0] != T(0)) out[0] += m * T(2625.5) * ex.v * a / (T(2) * pi[0]);
  if (pi[1] != T(0)) out[1] += m * f.d * x.v / (T(2) * pi[1]);          // df/db = f'(x) x / b ; db/db_i = b / (2 b_i)
  out[2] += m * fq.v * pj[2];
  out[3] += m * fc.v * pj[3];
}

// k-space dispersion kernels Ck_6/8/10 (recip.py:437-462); which = 6, 8, 10.
ADMP_HD double disp_ck(int which, double ksq, double kappa, double V) {
  const double sqrt_pi = 1.7724538509055159, pi = 3.141592653589793;
  double x2 = ksq / 4.0 / (kappa * kappa);
  double x = sqrt(x2);
  double ex = exp(-x2);
  double erfcx = erfc(x);
  double k3 = kappa * kappa * kappa;
  if (which == 6) {
    double f = (1 - 2 * x2) * ex + 2 * x2 * x * sqrt_pi * erfcx;
    return sqrt_pi * pi / 2 / V * k3 * f / 3;
  } else if (which == 8) {
    double x4 = x2 * x2;
    double f = (3 - 2 * x2 + 4 * x4) * ex - 4 * x4 * x * sqrt_pi * erfcx;
    return sqrt_pi * pi / 2 / V * k3 * kappa * kappa * f / 45;
  } else {
    double x4 = x2 * x2, x6 = x4 * x2;
    double f = (15 - 6 * x2 + 4 * x4 - 8 * x6) * ex + 8 * x6 * x * sqrt_pi * erfcx;
    return sqrt_pi * pi / 2 / V * k3 * k3 * kappa * f / 1260;
  }
}

// d Ck_p / d(k^2) (box gradient: the k vectors move with the box).  With y = k^2 / 4 kappa^2:
// f6' = 3 sqrt(pi) x erfc(x) - 3 e^-y,  f8' = 5 [(2y - 1) e^-y - 2 sqrt(pi) y^(3/2) erfc(x)],
// f10' = 7 [(-4 y^2 + 2y - 3) e^-y + 4 sqrt(pi) y^(5/2) erfc(x)];  dC/dk^2 = C0 f'(y) / (4 kappa^2).
ADMP_HD double disp_ck_dksq(int which, double ksq, double kappa, double V) {
  const double sqrt_pi = 1.7724538509055159, pi = 3.141592653589793;
  const double y = ksq / 4.0 / (kappa * kappa), x = sqrt(y), ex = exp(-y), ec = erfc(x);
  const double k3 = kappa * kappa * kappa, pre = sqrt_pi * pi / 2 / V / (4.0 * kappa * kappa);
  if (which == 6) return pre * k3 * (3 * sqrt_pi * x * ec - 3 * ex) / 3;
  if (which == 8) return pre * k3 * kappa * kappa * 5 * ((2 * y - 1) * ex - 2 * sqrt_pi * y * x * ec) / 45;
  return pre * k3 * k3 * kappa * 7 * ((-4 * y * y + 2 * y - 3) * ex + 4 * sqrt_pi * y * y * x * ec) / 1260;
}

}  // namespace admp
