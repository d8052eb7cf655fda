// Per-pair arithmetic of the real-space multipolar PME kernel and its hand-coded adjoint.
//
// One translation of the reference's quasi-internal-frame pair energy
// (reference admp/pme.py:258-334 calc_e_perm, :379-475 calc_e_ind, :479-624 pme_real_kernel,
//  admp/spatial.py:149-178 build_quasi_internal, admp/multipole.py:80-179 rotations)
// written as plain inline functions so that the HIP kernels (pair_kernels.hip) and the
// host-compiled test shim (tests/hostshim) execute the very same arithmetic.
//
// What the reference obtains by jax.value_and_grad (admp/pme.py:108) is coded explicitly:
//   * d(coef)/dr by forward-mode dual numbers through the coefficient formulas
//     (same branch semantics as reverse-mode AD: clamped branches carry zero derivative);
//   * dE/d(multipole components) from the bilinear form;
//   * the two transverse gradient components from rotational invariance: displacing dr
//     by eps along the pair-frame x (y) axis turns the frame about y (x) by eps/r, so
//     g_x = (1/r) sum_sites P . (G_y q),  g_y = (1/r) sum_sites P . (G_x q)
//     with G the l<=2 real-harmonic rotation generators written out below.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ADMP_HD __host__ __device__ __forceinline__
#else
#define ADMP_HD inline
#endif

namespace admp {

constexpr double kDielectric = 1389.35455846;   // admp/pme.py:16
constexpr double kDefaultTholeWidth = 0.3;      // admp/pme.py:17
constexpr double kSqrt3 = 1.7320508075688772;
constexpr double kTwoOverSqrtPi = 1.1283791670955126;

// ---------------------------------------------------------------- dual numbers (value, d/dr)
template <class T>
struct Dual {
  T v, d;
  ADMP_HD Dual() {}
  ADMP_HD Dual(T v_) : v(v_), d(T(0)) {}
  ADMP_HD Dual(T v_, T d_) : v(v_), d(d_) {}
};
template <class T> ADMP_HD Dual<T> operator+(Dual<T> a, Dual<T> b) { return {a.v + b.v, a.d + b.d}; }
template <class T> ADMP_HD Dual<T> operator-(Dual<T> a, Dual<T> b) { return {a.v - b.v, a.d - b.d}; }
template <class T> ADMP_HD Dual<T> operator*(Dual<T> a, Dual<T> b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
template <class T> ADMP_HD Dual<T> operator+(Dual<T> a, T b) { return {a.v + b, a.d}; }
template <class T> ADMP_HD Dual<T> operator+(T b, Dual<T> a) { return {a.v + b, a.d}; }
template <class T> ADMP_HD Dual<T> operator-(Dual<T> a, T b) { return {a.v - b, a.d}; }
template <class T> ADMP_HD Dual<T> operator-(T b, Dual<T> a) { return {b - a.v, -a.d}; }
template <class T> ADMP_HD Dual<T> operator*(Dual<T> a, T b) { return {a.v * b, a.d * b}; }
template <class T> ADMP_HD Dual<T> operator*(T b, Dual<T> a) { return {a.v * b, a.d * b}; }
template <class T> ADMP_HD Dual<T> operator-(Dual<T> a) { return {-a.v, -a.d}; }

template <class T> ADMP_HD T val(T a) { return a; }
template <class T> ADMP_HD T val(Dual<T> a) { return a.v; }

#if !defined(ADMP_LIBM_EXP) && defined(__HIP_DEVICE_COMPILE__)
// f32 device code: exp2(x log2 e) on the transcendental unit (relative error ~1e-7 |x|; arguments here are -x^2 >= -10
// and -au > -50 with results below 1e-20 beyond that).  3 exponentials per pair: 1M-atom pair kernel 0.432 -> 0.416 ms.
// -DADMP_LIBM_EXP selects expf.
ADMP_HD float m_exp(float x) { return __expf(x); }
#else
ADMP_HD float m_exp(float x) { return expf(x); }
#endif
ADMP_HD double m_exp(double x) { return exp(x); }
#if !defined(ADMP_LIBM_ERFC) && defined(__HIP_DEVICE_COMPILE__)
// erfc for x >= 0 (the only use: x = kappa r): t exp(-x^2 + P(t)), t = 1/(1 + x/2), fractional error < 1.2e-7
// (Numerical Recipes erfcc); ~20 instructions against ~55 of the library routine: pair kernel at 1M atoms 0.469 -> 0.457 ms.
// -DADMP_LIBM_ERFC selects erfcf.
ADMP_HD float m_erfc(float x) {
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.5f * x);
  const float p = -1.26551223f + t * (1.00002368f + t * (0.37409196f + t * (0.09678418f + t * (-0.18628806f + t * (0.27886807f +
                  t * (-1.13520398f + t * (1.48851587f + t * (-0.82215223f + t * 0.17087277f))))))));
  return t * m_exp(p - x * x);
}
#else
ADMP_HD float m_erfc(float x) { return erfcf(x); }
#endif
ADMP_HD double m_erfc(double x) { return erfc(x); }
ADMP_HD float m_sqrt(float x) { return sqrtf(x); }
ADMP_HD double m_sqrt(double x) { return sqrt(x); }
ADMP_HD float m_floor(float x) { return floorf(x); }
ADMP_HD double m_floor(double x) { return floor(x); }
ADMP_HD float m_ceil(float x) { return ceilf(x); }
ADMP_HD double m_ceil(double x) { return ceil(x); }
ADMP_HD float m_abs(float x) { return fabsf(x); }
ADMP_HD double m_abs(double x) { return fabs(x); }
// reciprocal / reciprocal square root: hardware 1-ulp forms for f32 on the device, exact elsewhere
#if defined(__HIP_DEVICE_COMPILE__)
ADMP_HD float m_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
ADMP_HD float m_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#define ADMP_PHASE() __builtin_amdgcn_sched_barrier(0)
// "register fence": an empty volatile asm that reads and writes the given values.  Volatile asms keep their
// program order, so everything that produces these values is scheduled before the fence and everything that
// consumes them after it -- the only way to stop the machine scheduler from interleaving the phases of the
// pair function (which costs ~100 extra live registers); emits no instruction.
#define ADMP_FENCE3(a) asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]))
#define ADMP_FENCE9(a) asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]), "+v"((a)[4]), \
                                         "+v"((a)[5]), "+v"((a)[6]), "+v"((a)[7]), "+v"((a)[8]))
#define ADMP_FENCE2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#else
ADMP_HD float m_rcp(float x) { return 1.0f / x; }
ADMP_HD float m_rsqrt(float x) { return 1.0f / sqrtf(x); }
#define ADMP_PHASE() ((void)0)
#define ADMP_FENCE3(a) ((void)0)
#define ADMP_FENCE9(a) ((void)0)
#define ADMP_FENCE2(a, b) ((void)0)
#endif
ADMP_HD double m_rcp(double x) { return 1.0 / x; }
ADMP_HD double m_rsqrt(double x) { return 1.0 / sqrt(x); }
template <class T> ADMP_HD Dual<T> m_exp(Dual<T> a) { T e = m_exp(a.v); return {e, e * a.d}; }
// erfc'(x) = -2/sqrt(pi) exp(-x^2)
template <class T> ADMP_HD Dual<T> m_erfc(Dual<T> a) {
  return {m_erfc(a.v), -T(kTwoOverSqrtPi) * m_exp(-a.v * a.v) * a.d};
}
template <class T> ADMP_HD Dual<T> recip(Dual<T> a) { T i = m_rcp(a.v); return {i, -a.d * i * i}; }
template <class T> ADMP_HD T recip(T a) { return m_rcp(a); }
// trim_val_infty (admp/pme.py:365-376): x >= thresh -> constant thresh
template <class T> ADMP_HD T trim_inf(T a, T th) { return a < th ? a : th; }
template <class T> ADMP_HD Dual<T> trim_inf(Dual<T> a, T th) { return a.v < th ? a : Dual<T>(th, T(0)); }

// ---------------------------------------------------------------- geometry
template <class T>
struct Box {
  T h[9];     // lattice vectors in rows (admp/pme.py:159)
  T hinv[9];  // inverse
};

// admp/spatial.py:13-32: ds = dr . box^-1 ; ds -= floor(ds + 1/2) ; dr = ds . box
template <class T>
ADMP_HD void min_image(const Box<T>& b, T d[3]) {
  T s0 = d[0] * b.hinv[0] + d[1] * b.hinv[3] + d[2] * b.hinv[6];
  T s1 = d[0] * b.hinv[1] + d[1] * b.hinv[4] + d[2] * b.hinv[7];
  T s2 = d[0] * b.hinv[2] + d[1] * b.hinv[5] + d[2] * b.hinv[8];
  s0 -= m_floor(s0 + T(0.5));
  s1 -= m_floor(s1 + T(0.5));
  s2 -= m_floor(s2 + T(0.5));
  d[0] = s0 * b.h[0] + s1 * b.h[3] + s2 * b.h[6];
  d[1] = s0 * b.h[1] + s1 * b.h[4] + s2 * b.h[7];
  d[2] = s0 * b.h[2] + s1 * b.h[5] + s2 * b.h[8];
}

// The lattice translation min_image removes from d: sh = n . box with n = floor(d . box^-1 + 1/2) (integers).  Returns
// false when n = 0.  Box gradient (jax.grad of the reference with respect to `box`, positions fixed): pbc_shift's floor
// has zero derivative and dr.box^-1.box = dr, so d(shifted dr)/d(box[a][b]) = -n_a e_b -- only pairs (and local-frame
// vectors) that cross the cell boundary contribute, with -n (x) dE/d(dr).
template <class T>
ADMP_HD bool image_shift(const Box<T>& b, const T d[3], T sh[3]) {
  const T n0 = m_floor(d[0] * b.hinv[0] + d[1] * b.hinv[3] + d[2] * b.hinv[6] + T(0.5));
  const T n1 = m_floor(d[0] * b.hinv[1] + d[1] * b.hinv[4] + d[2] * b.hinv[7] + T(0.5));
  const T n2 = m_floor(d[0] * b.hinv[2] + d[1] * b.hinv[5] + d[2] * b.hinv[8] + T(0.5));
  sh[0] = n0 * b.h[0] + n1 * b.h[3] + n2 * b.h[6];
  sh[1] = n0 * b.h[1] + n1 * b.h[4] + n2 * b.h[7];
  sh[2] = n0 * b.h[2] + n1 * b.h[5] + n2 * b.h[8];
  return n0 != T(0) || n1 != T(0) || n2 != T(0);
}

// Pair frame with z || dr (admp/spatial.py:149-178).  The reference picks the helper axis
// e_x unless the raw y and z coordinates coincide (spatial.py:172); the pair energy does not
// depend on that choice (it is invariant under rotation about z), so the helper is chosen
// here by conditioning instead: e_x unless z is within ~37 deg of it.
template <class T>
ADMP_HD void qi_frame(const T z[3], T x[3], T y[3]) {
  T hx = T(1), hy = T(0);
  if (m_abs(z[0]) > T(0.8)) { hx = T(0); hy = T(1); }
  T dot = z[0] * hx + z[1] * hy;
  x[0] = hx - z[0] * dot;
  x[1] = hy - z[1] * dot;
  x[2] = -z[2] * dot;
  T inv = m_rsqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  x[0] *= inv; x[1] *= inv; x[2] *= inv;
  y[0] = z[1] * x[2] - z[2] * x[1];
  y[1] = z[2] * x[0] - z[0] * x[2];
  y[2] = z[0] * x[1] - z[1] * x[0];
}

// Components, in the frame with rows (x,y,z), of a harmonic multipole given in the global
// frame.  Same map as rot_global2local (admp/multipole.py:92-179), evaluated through the
// Cartesian traceless tensor instead of the 25-entry matrix: Theta' = R Theta R^T.
// Harmonic order [00, 10, 11c, 11s, 20, 21c, 21s, 22c, 22s]; dipole (10,11c,11s) = (z,x,y).
// The map is orthogonal, so the adjoint (cotangent local -> global) is the same routine
// with the transposed frame: pass (cx,cy,cz) = columns.
template <class T>
ADMP_HD void rot_dip(const T* q, const T x[3], const T y[3], const T z[3], T* o) {
  // q = (dz, dx, dy)
  T dx = q[1], dy = q[2], dz = q[0];
  o[0] = z[0] * dx + z[1] * dy + z[2] * dz;
  o[1] = x[0] * dx + x[1] * dy + x[2] * dz;
  o[2] = y[0] * dx + y[1] * dy + y[2] * dz;
}

template <class T>
ADMP_HD void rot_quad(const T* q, const T x[3], const T y[3], const T z[3], T* o) {
  const T h = T(0.5 * kSqrt3);
  T tzz = q[0];
  T txx = T(0.5) * (-q[0] + T(kSqrt3) * q[3]);
  T tyy = T(0.5) * (-q[0] - T(kSqrt3) * q[3]);
  T txz = h * q[1], tyz = h * q[2], txy = h * q[4];
  // t_k = Theta . r_k
  T ax = txx * x[0] + txy * x[1] + txz * x[2];
  T ay = txy * x[0] + tyy * x[1] + tyz * x[2];
  T az = txz * x[0] + tyz * x[1] + tzz * x[2];
  T bx = txx * y[0] + txy * y[1] + txz * y[2];
  T by = txy * y[0] + tyy * y[1] + tyz * y[2];
  T bz = txz * y[0] + tyz * y[1] + tzz * y[2];
  T cx = txx * z[0] + txy * z[1] + txz * z[2];
  T cy = txy * z[0] + tyy * z[1] + tyz * z[2];
  T cz = txz * z[0] + tyz * z[1] + tzz * z[2];
  T nzz = z[0] * cx + z[1] * cy + z[2] * cz;
  T nxz = x[0] * cx + x[1] * cy + x[2] * cz;
  T nyz = y[0] * cx + y[1] * cy + y[2] * cz;
  T nxx = x[0] * ax + x[1] * ay + x[2] * az;
  T nyy = y[0] * bx + y[1] * by + y[2] * bz;
  T nxy = x[0] * bx + x[1] * by + x[2] * bz;
  const T g = T(2.0 / kSqrt3);
  o[0] = nzz;
  o[1] = g * nxz;
  o[2] = g * nyz;
  o[3] = T(1.0 / kSqrt3) * (nxx - nyy);
  o[4] = g * nxy;
}

template <class T>
ADMP_HD void rot_harm(const T Q[9], const T x[3], const T y[3], const T z[3], T o[9]) {
  o[0] = Q[0];
  rot_dip(Q + 1, x, y, z, o + 1);
  rot_quad(Q + 4, x, y, z, o + 4);
}

// transposed frame helper: columns of the (x,y,z)-row matrix
template <class T>
ADMP_HD void frame_cols(const T x[3], const T y[3], const T z[3], T cx[3], T cy[3], T cz[3]) {
  cx[0] = x[0]; cx[1] = y[0]; cx[2] = z[0];
  cy[0] = x[1]; cy[1] = y[1]; cy[2] = z[1];
  cz[0] = x[2]; cz[1] = y[2]; cz[2] = z[2];
}

// Rotation generators in the real-harmonic basis (derivation in the file header).
// gen_toward_x: d(components)/d(theta) when the frame turns about its y axis (z -> x).
template <class T>
ADMP_HD T gen_toward_x(const T* P, const T* q) {   // 9-component sets
  return P[1] * q[2] - P[2] * q[1] + T(kSqrt3) * (P[4] * q[5] - P[5] * q[4]) + P[5] * q[7] - P[7] * q[5] +
         P[6] * q[8] - P[8] * q[6];
}
// gen_toward_y: frame turns z -> y.
template <class T>
ADMP_HD T gen_toward_y(const T* P, const T* q) {
  return P[1] * q[3] - P[3] * q[1] + T(kSqrt3) * (P[4] * q[6] - P[6] * q[4]) + P[5] * q[8] - P[8] * q[5] -
         P[6] * q[7] + P[7] * q[6];
}
// rotation of the object about the third axis of the basis the components are given in
template <class T>
ADMP_HD T gen_about_z(const T* P, const T* q) {
  return -P[2] * q[3] + P[3] * q[2] - P[5] * q[6] + P[6] * q[5] + T(2) * (P[8] * q[7] - P[7] * q[8]);
}

// ---------------------------------------------------------------- radial coefficients
// Shared Ewald pieces (admp/pme.py:283-300).  bVec is kept in the complementary form
// Bn = 1 + b_n (so that m + b_n = (m - 1) + Bn): B1 = erfc(x), B2 = B1 + xX, ...
// which avoids the 1 - erf cancellation in single precision.
template <class S, class T>
struct Radial {
  S R1, R2, R3, R4, R5;   // DIELECTRIC * r^-n
  S x2, x3X, x5X, xX;     // powers of x = kappa r times X = 2 exp(-x^2)/sqrt(pi)
  S B2, B3, B4;
  ADMP_HD void init(S r, T kappa) {
    S ri = recip(r);
    R1 = ri * T(kDielectric);
    R2 = R1 * ri; R3 = R2 * ri; R4 = R3 * ri; R5 = R4 * ri;
    S x = r * kappa;
    x2 = x * x;
    S X = m_exp(-x2) * T(kTwoOverSqrtPi);
    xX = x * X;
    x3X = xX * x2;
    x5X = x3X * x2;
    S B1 = m_erfc(x);
    B2 = B1 + xX;
    B3 = B2 + x3X * T(2.0 / 3.0);
    B4 = B3 + x5X * T(4.0 / 15.0);
  }
  ADMP_HD void fence() {
    ADMP_FENCE2(R1.v, R1.d); ADMP_FENCE2(R2.v, R2.d); ADMP_FENCE2(R3.v, R3.d); ADMP_FENCE2(R4.v, R4.d);
    ADMP_FENCE2(R5.v, R5.d); ADMP_FENCE2(x2.v, x2.d); ADMP_FENCE2(x3X.v, x3X.d); ADMP_FENCE2(x5X.v, x5X.d);
    ADMP_FENCE2(xX.v, xX.d); ADMP_FENCE2(B2.v, B2.d); ADMP_FENCE2(B3.v, B3.d); ADMP_FENCE2(B4.v, B4.d);
  }
};

// d(pair energy)/d(mscale): the real-space pair energy is linear in mm = mscale - 1 (every permanent coefficient is
// R_n (mm + B_k) x const, admp/pme.py:303-324; the induced ones carry pscale instead), so the derivative is the bare
// (undamped, unscaled) multipole interaction of the two sites in their quasi-internal frame.
template <class T, class SiteT>
ADMP_HD T pair_bare_energy(const Box<T>& box, const SiteT& I, const SiteT& J) {
  T d[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = T(1) / m_sqrt(r2);
  T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv}, x[3], y[3];
  qi_frame(z, x, y);
  T A[9], B[9];
  rot_harm(I.Q, x, y, z, A);
  rot_harm(J.Q, x, y, z, B);
  const T R1 = T(kDielectric) * rinv, R2 = R1 * rinv, R3 = R2 * rinv, R4 = R3 * rinv, R5 = R4 * rinv;
  return R1 * (A[0] * B[0]) + R2 * (A[0] * B[1] - A[1] * B[0]) +
         R3 * ((A[0] * B[4] + A[4] * B[0]) - T(2) * A[1] * B[1] + (A[2] * B[2] + A[3] * B[3])) +
         R4 * (T(3) * (A[4] * B[1] - A[1] * B[4]) - T(kSqrt3) * (A[5] * B[2] + A[6] * B[3] - A[2] * B[5] - A[3] * B[6])) +
         R5 * (T(6) * A[4] * B[4] - T(4) * (A[5] * B[5] + A[6] * B[6]) + (A[7] * B[7] + A[8] * B[8]));
}

// coefficient slots (the ten permanent ones of admp/pme.py:303-324 are streamed inside pair_energy_grad)
enum { CC, CD, DD0, DD1, CQ, DQ0, DQ1, QQ0, QQ1, QQ2, CUD, DUD0, DUD1, UDQ0, UDQ1, UDUD0, UDUD1, NCOEF };

// seven induced coefficients (admp/pme.py:408-475).
//   aw   : Thole width after the Fermi switch on pscale (pme.py:411; r-independent)
//   dmp  : (pol_i pol_j)^(1/6), floored at 1e-8 by the caller (pme.py:413)
//   pm   : pscale ; the Ewald part enters as (p*thole + b_n) = p*thole - 1 + Bn
template <class S, class T>
ADMP_HD void ind_coefs(const Radial<S, T>& a, S r, T aw, T dmp, T p, S* c) {
  S u = trim_inf(r * (T(1) / dmp), T(1e8));
  S au = u * aw;
  S expau = (val(au) < T(50)) ? m_exp(-au) : S(T(0));
  S au2 = trim_inf(au * au, T(1e8));
  S au3 = trim_inf(au2 * au, T(1e8));
  S au4 = trim_inf(au3 * au, T(1e8));
  S base = au + T(1) + au2 * T(0.5);
  S th_c = T(1) - expau * base;
  S th_d0 = T(1) - expau * (base + au3 * T(0.25));
  S th_q1 = T(1) - expau * (base + au3 * T(1.0 / 6.0));
  S th_q0 = T(1) - expau * (base + au3 * T(1.0 / 6.0) + au4 * T(1.0 / 18.0));
  // th_d1 == th_c (pme.py:430)
  c[CUD] = a.R2 * (th_c * p - T(1) + a.B2) * T(2);
  c[DUD0] = a.R3 * ((th_d0 * p - T(1) + a.B3) * T(3) + a.x3X) * T(-4.0 / 3.0);
  c[DUD1] = a.R3 * (th_c * p - T(1) + a.B3 - a.x3X * T(2.0 / 3.0)) * T(2);
  c[UDQ0] = a.R4 * ((th_q0 * p - T(1) + a.B3) * T(3) + a.x5X * T(4.0 / 3.0)) * T(2);
  c[UDQ1] = a.R4 * (th_q1 * p - T(1) + a.B3) * T(-2.0 * kSqrt3);
  // uscales = 1 (pme.py:472): dscales is ignored by the reference
  c[UDUD0] = a.R3 * ((th_d0 - T(1) + a.B3) * T(3) + a.x3X) * T(-2.0 / 3.0);
  c[UDUD1] = a.R3 * (th_c - T(1) + a.B3 - a.x3X * T(2.0 / 3.0));
}

// ---------------------------------------------------------------- the pair
// Per-site data as the kernels keep it (global frame).
// 20 reals = five 16-byte (f32) / 32-byte (f64) vectors: one gathered row of the site table.
#ifndef ADMP_SITE_ALIGN
#define ADMP_SITE_ALIGN 16
#endif
template <class T>
struct alignas(ADMP_SITE_ALIGN) Site {
  T r[3];
  T Q[9];    // permanent multipoles, global harmonics
  T U[3];    // induced dipole, global harmonic order (z,x,y)  (admp/pme.py:235)
  T p6;      // pol^(1/6)  (dmp = p6_i p6_j, admp/pme.py:732-735)
  T thole;
  T pad[3];
};

// Position and charge of a site, 4 words: the compact copy of the rows that the pair kernels read for charge-only partners
// (k_prepare_sites writes it next to the Site rows).  Four sites per 64-B line instead of 80 B per site, and the
// hydrogens of a molecule share their line -- the loops over charge-only partners are bound by L2 -> L1 traffic.
template <class T>
struct alignas(4 * sizeof(T)) RQ4 { T v[4]; };

template <class T>
struct PairScales {
  T mm;    // mscale - 1
  T p;     // pscale
  T w0;    // Fermi weight of DEFAULT_THOLE_WIDTH at this pscale (admp/pme.py:337-348, 411)
};

// Full evaluation, centre = site I (dr = r_I - r_J, reference ordering admp/pme.py:713).
//   returns the pair energy; adds to gI (dE/dr_I = -dE/dr_J), potI (dE/dQ_I global),
//   fldI (dE/dU_I global harmonic).  If WITH_J also adds site J's potJ / fldJ.
template <class T, bool LPOL, bool WITH_J>
ADMP_HD T pair_energy_grad(const Box<T>& box, const Site<T>& I, const Site<T>& J, const PairScales<T>& sc, T kappa,
                           T gI[3], T potI[9], T fldI[3], T potJ[9], T fldJ[3]) {
  // ---- phase 1: geometry, pair frame, moments rotated into it
  T d[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = m_rsqrt(r2);
  const T r = r2 * rinv;
  T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv};
  T x[3], y[3];
  qi_frame(z, x, y);
  T A[9], B[9], UA[3] = {T(0), T(0), T(0)}, UB[3] = {T(0), T(0), T(0)};
  rot_harm(I.Q, x, y, z, A);
  rot_harm(J.Q, x, y, z, B);
  T aw = T(0), dmp = T(1);
  if (LPOL) {
    rot_dip(I.U, x, y, z, UA);
    rot_dip(J.U, x, y, z, UB);
    aw = sc.w0 * T(kDefaultTholeWidth) + (T(1) - sc.w0) * (I.thole + J.thole);
    dmp = I.p6 * J.p6;
    dmp = dmp < T(1e-8) ? T(1e-8) : dmp;
  }
  ADMP_FENCE9(A); ADMP_FENCE9(B);
  if (LPOL) { ADMP_FENCE3(UA); ADMP_FENCE3(UB); }

  // ---- phase 2: permanent coefficients (admp/pme.py:283-324) with hand-derived radial derivatives.
  // Every coefficient is c = R_n f with R_n = D r^-n and f a linear form in  mm + B_k  and  x^m X
  // (x = kappa r, X = 2 exp(-x^2)/sqrt(pi), B_1 = erfc x, B_k = B_{k-1} + 2^{k-2} x^{2k-3} X / (2k-3)!!).
  // With g = r df/dr one has  dc/dr = R_n (g - n f) / r, and the building blocks obey
  //   r d(x^m X)/dr = x^m X (m - 2 x^2),  r dB_1/dr = -xX,  r dB_2/dr = -2 x^2 xX,
  //   r dB_3/dr = -(4/3) x^2 x^3X,        r dB_4/dr = -(8/15) x^2 x^5X
  // -- the same numbers reverse-mode AD of the reference's formulas produces (checked against the oracle).
  const T R1 = T(kDielectric) * rinv, R2 = R1 * rinv, R3 = R2 * rinv, R4 = R3 * rinv, R5 = R4 * rinv;
  const T xk = kappa * r, x2 = xk * xk;
  const T X = T(kTwoOverSqrtPi) * m_exp(-x2);
  const T xX = xk * X, x3X = x2 * xX, x5X = x2 * x3X;
  const T B1 = m_erfc(xk);
  const T B2 = B1 + xX, B3 = B2 + T(2.0 / 3.0) * x3X, B4 = B3 + T(4.0 / 15.0) * x5X;
  const T b2p = T(-2) * x2 * xX, b3p = T(-4.0 / 3.0) * x2 * x3X, b4p = T(-8.0 / 15.0) * x2 * x5X;
  const T x3Xp = x3X * (T(3) - T(2) * x2), x5Xp = x5X * (T(5) - T(2) * x2);
  T e = T(0), dedr = T(0);
  T PA[9], PB[9], FA[3] = {T(0), T(0), T(0)}, FB[3] = {T(0), T(0), T(0)};
  const T mm = sc.mm;
  // cv = coefficient value, accumulates energy and dE/dr for the bilinear shape s
#define ADMP_COEF(Rn, n, f, g, shape)                                          \
  const T cv = (Rn) * (f);                                                     \
  { const T s_ = (shape); e += cv * s_; dedr += (Rn) * rinv * ((g) - T(n) * (f)) * s_; }
  {  // cc : f = mm + B2 - xX = mm + erfc(x)
    const T f = mm + B1, g = -xX;
    ADMP_COEF(R1, 1, f, g, A[0] * B[0])
    PA[0] = cv * B[0]; PB[0] = cv * A[0];
  }
  {  // cd
    const T f = mm + B2;
    ADMP_COEF(R2, 2, f, b2p, A[0] * B[1] - A[1] * B[0])
    PA[0] += cv * B[1]; PA[1] = -cv * B[0]; PB[0] -= cv * A[1]; PB[1] = cv * A[0];
  }
  const T f3 = mm + B3, f4 = mm + B4;
  {  // cq
    ADMP_COEF(R3, 3, f3, b3p, A[0] * B[4] + A[4] * B[0])
    PA[0] += cv * B[4]; PA[4] = cv * B[0]; PB[0] += cv * A[4]; PB[4] = cv * A[0];
  }
  {  // dd_m0
    const T f = T(-2) * f3 - T(2.0 / 3.0) * x3X, g = T(-2) * b3p - T(2.0 / 3.0) * x3Xp;
    ADMP_COEF(R3, 3, f, g, A[1] * B[1])
    PA[1] += cv * B[1]; PB[1] += cv * A[1];
  }
  {  // dd_m1
    const T f = f3 - T(2.0 / 3.0) * x3X, g = b3p - T(2.0 / 3.0) * x3Xp;
    ADMP_COEF(R3, 3, f, g, A[2] * B[2] + A[3] * B[3])
    PA[2] = cv * B[2]; PA[3] = cv * B[3]; PB[2] = cv * A[2]; PB[3] = cv * A[3];
  }
  {  // dq_m0
    const T f = T(3) * f3 + T(4.0 / 3.0) * x5X, g = T(3) * b3p + T(4.0 / 3.0) * x5Xp;
    ADMP_COEF(R4, 4, f, g, A[4] * B[1] - A[1] * B[4])
    PA[1] -= cv * B[4]; PA[4] += cv * B[1]; PB[1] += cv * A[4]; PB[4] -= cv * A[1];
  }
  {  // dq_m1
    const T f = T(-kSqrt3) * f3, g = T(-kSqrt3) * b3p;
    ADMP_COEF(R4, 4, f, g, A[5] * B[2] + A[6] * B[3] - A[2] * B[5] - A[3] * B[6])
    PA[2] -= cv * B[5]; PA[3] -= cv * B[6]; PA[5] = cv * B[2]; PA[6] = cv * B[3];
    PB[2] += cv * A[5]; PB[3] += cv * A[6]; PB[5] = -cv * A[2]; PB[6] = -cv * A[3];
  }
  {  // qq_m0 : (4/45)(10 x^2 - 3) x^5X, r d/dr of it = (4/45)[(10 x^2 - 3) x^5X' + 20 x^2 x^5X]
    const T w = T(10) * x2 - T(3);
    const T f = T(6) * f4 + T(4.0 / 45.0) * w * x5X;
    const T g = T(6) * b4p + T(4.0 / 45.0) * (w * x5Xp + T(20) * x2 * x5X);
    ADMP_COEF(R5, 5, f, g, A[4] * B[4])
    PA[4] += cv * B[4]; PB[4] += cv * A[4];
  }
  {  // qq_m1
    const T f = T(-4) * f4 - T(4.0 / 15.0) * x5X, g = T(-4) * b4p - T(4.0 / 15.0) * x5Xp;
    ADMP_COEF(R5, 5, f, g, A[5] * B[5] + A[6] * B[6])
    PA[5] += cv * B[5]; PA[6] += cv * B[6]; PB[5] += cv * A[5]; PB[6] += cv * A[6];
  }
  {  // qq_m2
    const T f = f4 - T(4.0 / 15.0) * x5X, g = b4p - T(4.0 / 15.0) * x5Xp;
    ADMP_COEF(R5, 5, f, g, A[7] * B[7] + A[8] * B[8])
    PA[7] = cv * B[7]; PA[8] = cv * B[8]; PB[7] = cv * A[7]; PB[8] = cv * A[8];
  }
  ADMP_FENCE9(PA); ADMP_FENCE9(PB); ADMP_FENCE2(e, dedr);

  // ---- phase 3: induced coefficients (admp/pme.py:408-475).  Thole factors th = 1 - exp(-au) poly(au) with
  // au = a r / dmp: r d(th)/dr = au d(th)/d(au) (zero where the reference's clamps are active: au >= 50 kills
  // the exponential, a clamped u = r/dmp has no r dependence).
  if (LPOL) {
    const T uraw = r * m_rcp(dmp);
    const bool ucap = !(uraw < T(1e8));
    const T au = (ucap ? T(1e8) : uraw) * aw;
    const bool live = au < T(50);
    const T Ex = live ? m_exp(-au) : T(0);
    const T au2 = au * au, au3 = au2 * au, au4 = au3 * au;      // only used where Ex != 0, i.e. au < 50: no clamp active
    const T base = T(1) + au + T(0.5) * au2;
    const T th_c = T(1) - Ex * base;                            // == thole_d1 (pme.py:430)
    const T th_d0 = T(1) - Ex * (base + T(0.25) * au3);
    const T th_q1 = T(1) - Ex * (base + T(1.0 / 6.0) * au3);
    const T th_q0 = T(1) - Ex * (base + T(1.0 / 6.0) * au3 + T(1.0 / 18.0) * au4);
    const T Ed = ucap ? T(0) : Ex;
    const T tcp = T(0.5) * Ed * au3;                            // r d(th_c)/dr
    const T td0p = T(0.25) * Ed * au3 * (au - T(1));
    const T tq1p = T(1.0 / 6.0) * Ed * au4;
    const T tq0p = T(1.0 / 18.0) * Ed * au4 * (au - T(1));
    const T p = sc.p, hf = T(0.5);
    {  // cud
      const T f = T(2) * (p * th_c - T(1) + B2), g = T(2) * (p * tcp + b2p);
      ADMP_COEF(R2, 2, f, g, hf * (A[0] * UB[0] - B[0] * UA[0]))
      const T h = hf * cv;
      PA[0] += h * UB[0]; PB[0] -= h * UA[0]; FA[0] = -h * B[0]; FB[0] = h * A[0];
    }
    {  // dud_m0
      const T f = T(-4.0 / 3.0) * (T(3) * (p * th_d0 - T(1) + B3) + x3X), g = T(-4.0 / 3.0) * (T(3) * (p * td0p + b3p) + x3Xp);
      ADMP_COEF(R3, 3, f, g, hf * (B[1] * UA[0] + A[1] * UB[0]))
      const T h = hf * cv;
      PA[1] += h * UB[0]; PB[1] += h * UA[0]; FA[0] += h * B[1]; FB[0] += h * A[1];
    }
    {  // dud_m1
      const T f = T(2) * (p * th_c - T(1) + B3 - T(2.0 / 3.0) * x3X), g = T(2) * (p * tcp + b3p - T(2.0 / 3.0) * x3Xp);
      ADMP_COEF(R3, 3, f, g, hf * (B[2] * UA[1] + B[3] * UA[2] + A[2] * UB[1] + A[3] * UB[2]))
      const T h = hf * cv;
      PA[2] += h * UB[1]; PA[3] += h * UB[2]; PB[2] += h * UA[1]; PB[3] += h * UA[2];
      FA[1] = h * B[2]; FA[2] = h * B[3]; FB[1] = h * A[2]; FB[2] = h * A[3];
    }
    {  // udq_m0
      const T f = T(2) * (T(3) * (p * th_q0 - T(1) + B3) + T(4.0 / 3.0) * x5X);
      const T g = T(2) * (T(3) * (p * tq0p + b3p) + T(4.0 / 3.0) * x5Xp);
      ADMP_COEF(R4, 4, f, g, hf * (A[4] * UB[0] - B[4] * UA[0]))
      const T h = hf * cv;
      PA[4] += h * UB[0]; PB[4] -= h * UA[0]; FA[0] -= h * B[4]; FB[0] += h * A[4];
    }
    {  // udq_m1
      const T f = T(-2.0 * kSqrt3) * (p * th_q1 - T(1) + B3), g = T(-2.0 * kSqrt3) * (p * tq1p + b3p);
      ADMP_COEF(R4, 4, f, g, hf * (A[5] * UB[1] + A[6] * UB[2] - B[5] * UA[1] - B[6] * UA[2]))
      const T h = hf * cv;
      PA[5] += h * UB[1]; PA[6] += h * UB[2]; PB[5] -= h * UA[1]; PB[6] -= h * UA[2];
      FA[1] -= h * B[5]; FA[2] -= h * B[6]; FB[1] += h * A[5]; FB[2] += h * A[6];
    }
    {  // udud_m0 (uscales = 1, pme.py:472)
      const T f = T(-2.0 / 3.0) * (T(3) * (th_d0 - T(1) + B3) + x3X), g = T(-2.0 / 3.0) * (T(3) * (td0p + b3p) + x3Xp);
      ADMP_COEF(R3, 3, f, g, UA[0] * UB[0])
      FA[0] += cv * UB[0]; FB[0] += cv * UA[0];
    }
    {  // udud_m1
      const T f = th_c - T(1) + B3 - T(2.0 / 3.0) * x3X, g = tcp + b3p - T(2.0 / 3.0) * x3Xp;
      ADMP_COEF(R3, 3, f, g, UA[1] * UB[1] + UA[2] * UB[2])
      FA[1] += cv * UB[1]; FA[2] += cv * UB[2]; FB[1] += cv * UA[1]; FB[2] += cv * UA[2];
    }
    ADMP_FENCE9(PA); ADMP_FENCE9(PB); ADMP_FENCE3(FA); ADMP_FENCE3(FB); ADMP_FENCE2(e, dedr);
  }
#undef ADMP_COEF

  // ---- phase 4: transverse gradient from rotational invariance, back to the global frame
  T gx = gen_toward_x(PA, A) + gen_toward_x(PB, B);
  T gy = gen_toward_y(PA, A) + gen_toward_y(PB, B);
  if (LPOL) {
    gx += FA[0] * UA[1] - FA[1] * UA[0] + FB[0] * UB[1] - FB[1] * UB[0];
    gy += FA[0] * UA[2] - FA[2] * UA[0] + FB[0] * UB[2] - FB[2] * UB[0];
  }
  gx *= rinv;
  gy *= rinv;
  gI[0] += x[0] * gx + y[0] * gy + z[0] * dedr;
  gI[1] += x[1] * gx + y[1] * gy + z[1] * dedr;
  gI[2] += x[2] * gx + y[2] * gy + z[2] * dedr;
  ADMP_FENCE3(gI); ADMP_FENCE9(PA);

  // adjoint of an orthogonal map = the map with the transposed frame
  T cx[3], cy[3], cz[3], t[9];
  frame_cols(x, y, z, cx, cy, cz);
  rot_harm(PA, cx, cy, cz, t);
#pragma unroll
  for (int k = 0; k < 9; ++k) potI[k] += t[k];
  if (LPOL && fldI) {
    rot_dip(FA, cx, cy, cz, t);
    fldI[0] += t[0]; fldI[1] += t[1]; fldI[2] += t[2];
  }
  if (WITH_J) {
    rot_harm(PB, cx, cy, cz, t);
#pragma unroll
    for (int k = 0; k < 9; ++k) potJ[k] += t[k];
    if (LPOL) {
      rot_dip(FB, cx, cy, cz, t);
      fldJ[0] += t[0]; fldJ[1] += t[1]; fldJ[2] += t[2];
    }
  }
  return e;
}

// ---------------------------------------------------------------- charge-only sites
// Most sites of a typical model carry a charge and nothing else (the hydrogens of MPID water: no dipole, no quadrupole,
// no polarizability).  Against such a partner every coefficient block of pair_energy_grad except cc, cd, cq and cud has
// a zero bilinear shape, the partner's moments need no rotation and nothing of it needs the Thole exponential's
// derivative beyond cud.  The three functions below are pair_energy_grad with those zeros folded in by hand (IEEE
// arithmetic lets no compiler drop `x * 0`): same formulas, same results, ~1/3 and ~1/10 of the instructions.  A site
// is charge-only iff Q[1..8] == 0 and pol == 0 (k_prepare_sites marks it: p6 == 0 and pad[0] == 1).
template <class T>
ADMP_HD bool site_is_mono(const Site<T>& s) { return s.p6 == T(0) && s.pad[0] == T(1); }

// shared radial pieces of the surviving blocks
template <class T>
struct MonoRadial {
  T R1, R2, R3, rinv, B1, B2, B3, xX, b2p, b3p;
  ADMP_HD void init(T r, T rinv_, T kappa) {
    rinv = rinv_;
    R1 = T(kDielectric) * rinv; R2 = R1 * rinv; R3 = R2 * rinv;
    const T xk = kappa * r, x2 = xk * xk;
    const T X = T(kTwoOverSqrtPi) * m_exp(-x2);
    xX = xk * X;
    const T x3X = x2 * xX;
    B1 = m_erfc(xk);
    B2 = B1 + xX;
    B3 = B2 + T(2.0 / 3.0) * x3X;
    b2p = T(-2) * x2 * xX;
    b3p = T(-4.0 / 3.0) * x2 * x3X;
  }
};
// cud coefficient (value, r d/dr numerator g) for a pair one of whose sites is not polarizable: dmp sits at its floor
template <class T>
ADMP_HD void mono_cud(const MonoRadial<T>& m, T r, T aw, T p, T& f, T& g) {
  const T uraw = r * T(1e8);                         // r / max(dmp, 1e-8) with dmp = 0
  const bool ucap = !(uraw < T(1e8));
  const T au = (ucap ? T(1e8) : uraw) * aw;
  const bool live = au < T(50);
  const T Ex = live ? m_exp(-au) : T(0);
  const T au2 = au * au, au3 = au2 * au;
  const T th_c = T(1) - Ex * (T(1) + au + T(0.5) * au2);
  const T tcp = T(0.5) * (ucap ? T(0) : Ex) * au3;
  f = T(2) * (p * th_c - T(1) + m.B2);
  g = T(2) * (p * tcp + m.b2p);
}

// row site I with all its moments, partner J charge-only
template <class T, bool LPOL>
ADMP_HD T pair_full_mono(const Box<T>& box, const Site<T>& I, const T rJ[3], T qJ, T tholeJ, const PairScales<T>& sc,
                         T kappa, T gI[3], T potI[9], T fldI[3]) {
  T d[3] = {I.r[0] - rJ[0], I.r[1] - rJ[1], I.r[2] - rJ[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = m_rsqrt(r2);
  const T r = r2 * rinv;
  T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv}, x[3], y[3];
  qi_frame(z, x, y);
  T A[9], UA[3] = {T(0), T(0), T(0)};
  rot_harm(I.Q, x, y, z, A);
  if (LPOL) rot_dip(I.U, x, y, z, UA);
  MonoRadial<T> m;
  m.init(r, rinv, kappa);
  const T mm = sc.mm, B0 = qJ;
  T e, dedr, PA[9] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)}, FA0 = T(0);
  {  // cc : A0 B0
    const T f = mm + m.B1, cv = m.R1 * f, s_ = A[0] * B0;
    e = cv * s_; dedr = m.R1 * rinv * (-m.xX - f) * s_;
    PA[0] = cv * B0;
  }
  {  // cd : -A1 B0
    const T f = mm + m.B2, cv = m.R2 * f, s_ = -A[1] * B0;
    e += cv * s_; dedr += m.R2 * rinv * (m.b2p - T(2) * f) * s_;
    PA[1] = -cv * B0;
  }
  {  // cq : A4 B0
    const T f = mm + m.B3, cv = m.R3 * f, s_ = A[4] * B0;
    e += cv * s_; dedr += m.R3 * rinv * (m.b3p - T(3) * f) * s_;
    PA[4] = cv * B0;
  }
  if (LPOL) {  // cud : -(1/2) B0 UA0
    T f, g;
    mono_cud(m, r, sc.w0 * T(kDefaultTholeWidth) + (T(1) - sc.w0) * (I.thole + tholeJ), sc.p, f, g);
    const T cv = m.R2 * f, s_ = T(-0.5) * B0 * UA[0];
    e += cv * s_; dedr += m.R2 * rinv * (g - T(2) * f) * s_;
    FA0 = T(-0.5) * cv * B0;
  }
  // transverse gradient: generators acting on (PA, A) with PA in slots 0, 1, 4 only; induced part FA x UA
  T gx = PA[1] * A[2] + T(kSqrt3) * PA[4] * A[5];
  T gy = PA[1] * A[3] + T(kSqrt3) * PA[4] * A[6];
  if (LPOL) { gx += FA0 * UA[1]; gy += FA0 * UA[2]; }
  gx *= rinv; gy *= rinv;
  gI[0] += x[0] * gx + y[0] * gy + z[0] * dedr;
  gI[1] += x[1] * gx + y[1] * gy + z[1] * dedr;
  gI[2] += x[2] * gx + y[2] * gy + z[2] * dedr;
  T cx[3], cy[3], cz[3], t[9];
  frame_cols(x, y, z, cx, cy, cz);
  rot_harm(PA, cx, cy, cz, t);
#pragma unroll
  for (int k = 0; k < 9; ++k) potI[k] += t[k];
  if (LPOL && fldI) {
    const T FA[3] = {FA0, T(0), T(0)};
    rot_dip(FA, cx, cy, cz, t);
    fldI[0] += t[0]; fldI[1] += t[1]; fldI[2] += t[2];
  }
  return e;
}

// row site I charge-only, partner J with all its moments (only potI[0] is produced: a charge-only site has no torque)
template <class T, bool LPOL>
ADMP_HD T pair_mono_full(const Box<T>& box, const T rI[3], T qI, T tholeI, const Site<T>& J, const PairScales<T>& sc,
                         T kappa, T gI[3], T& potI0) {
  T d[3] = {rI[0] - J.r[0], rI[1] - J.r[1], rI[2] - J.r[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = m_rsqrt(r2);
  const T r = r2 * rinv;
  T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv}, x[3], y[3];
  qi_frame(z, x, y);
  T B[9], UB[3] = {T(0), T(0), T(0)};
  rot_harm(J.Q, x, y, z, B);
  if (LPOL) rot_dip(J.U, x, y, z, UB);
  MonoRadial<T> m;
  m.init(r, rinv, kappa);
  const T mm = sc.mm, A0 = qI;
  T e, dedr, PB1, PB4, FB0 = T(0);
  {  // cc : A0 B0
    const T f = mm + m.B1, cv = m.R1 * f, s_ = A0 * B[0];
    e = cv * s_; dedr = m.R1 * rinv * (-m.xX - f) * s_;
    potI0 += cv * B[0];
  }
  {  // cd : A0 B1
    const T f = mm + m.B2, cv = m.R2 * f, s_ = A0 * B[1];
    e += cv * s_; dedr += m.R2 * rinv * (m.b2p - T(2) * f) * s_;
    potI0 += cv * B[1]; PB1 = cv * A0;
  }
  {  // cq : A0 B4
    const T f = mm + m.B3, cv = m.R3 * f, s_ = A0 * B[4];
    e += cv * s_; dedr += m.R3 * rinv * (m.b3p - T(3) * f) * s_;
    potI0 += cv * B[4]; PB4 = cv * A0;
  }
  if (LPOL) {  // cud : (1/2) A0 UB0
    T f, g;
    mono_cud(m, r, sc.w0 * T(kDefaultTholeWidth) + (T(1) - sc.w0) * (tholeI + J.thole), sc.p, f, g);
    const T cv = m.R2 * f, s_ = T(0.5) * A0 * UB[0];
    e += cv * s_; dedr += m.R2 * rinv * (g - T(2) * f) * s_;
    potI0 += T(0.5) * cv * UB[0]; FB0 = T(0.5) * cv * A0;
  }
  T gx = PB1 * B[2] + T(kSqrt3) * PB4 * B[5];
  T gy = PB1 * B[3] + T(kSqrt3) * PB4 * B[6];
  if (LPOL) { gx += FB0 * UB[1]; gy += FB0 * UB[2]; }
  gx *= rinv; gy *= rinv;
  gI[0] += x[0] * gx + y[0] * gy + z[0] * dedr;
  gI[1] += x[1] * gx + y[1] * gy + z[1] * dedr;
  gI[2] += x[2] * gx + y[2] * gy + z[2] * dedr;
  return e;
}

// both charge-only: the screened Coulomb term alone
template <class T>
ADMP_HD T pair_mono_mono(const Box<T>& box, const T rI[3], T qI, const T rJ[3], T qJ, T mm, T kappa, T gI[3], T& potI0) {
  T d[3] = {rI[0] - rJ[0], rI[1] - rJ[1], rI[2] - rJ[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = m_rsqrt(r2);
  const T r = r2 * rinv;
  const T xk = kappa * r;
  const T xX = xk * T(kTwoOverSqrtPi) * m_exp(-xk * xk);
  const T R1 = T(kDielectric) * rinv, f = mm + m_erfc(xk), cv = R1 * f;
  const T s = R1 * rinv * rinv * (-xX - f) * qI * qJ;      // dE/dr / r
  gI[0] += s * d[0]; gI[1] += s * d[1]; gI[2] += s * d[2];
  potI0 += cv * qJ;
  return cv * qI * qJ;
}

// d(pair energy)/d ln(au), au = a_w r / dmp the Thole argument (admp/pme.py:408-432): the only way the polarizabilities
// (dmp = (alpha_i alpha_j)^(1/6)) and the Thole parameters (a_w) enter the pair energy.  The Thole factors depend on au
// alone, so this is the "au d(th)/d(au)" part of the induced coefficients' radial derivative in pair_energy_grad,
// contracted with the same bilinear shapes.  Returns X; *wth receives d ln(au) / d(thole_i) = (1 - w0) / a_w.
// dE/dalpha_i = -(1/(6 alpha_i)) sum_j X_ij (+ the penalty term), dE/dthole_i = sum_j X_ij wth_ij.
template <class T>
ADMP_HD T pair_thole_logderiv(const Box<T>& box, const Site<T>& I, const Site<T>& J, const PairScales<T>& sc, T* wth) {
  T d[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = T(1) / m_sqrt(r2);
  const T r = r2 * rinv;
  T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv}, x[3], y[3];
  qi_frame(z, x, y);
  T A[9], B[9], UA[3], UB[3];
  rot_harm(I.Q, x, y, z, A);
  rot_harm(J.Q, x, y, z, B);
  rot_dip(I.U, x, y, z, UA);
  rot_dip(J.U, x, y, z, UB);
  const T aw = sc.w0 * T(kDefaultTholeWidth) + (T(1) - sc.w0) * (I.thole + J.thole);
  T dmp = I.p6 * J.p6;
  const bool dmp_floor = dmp < T(1e-8);
  dmp = dmp_floor ? T(1e-8) : dmp;
  const T uraw = r / dmp;
  *wth = aw > T(0) ? (T(1) - sc.w0) / aw : T(0);           // a_w = 0 (both tholes 0): X ~ au^3 vanishes with it
  if (!(uraw < T(1e8))) return T(0);                      // u clamped: no dependence left
  const T au = uraw * aw;
  if (!(au < T(50))) return T(0);                         // exponential switched off
  const T Ex = m_exp(-au);
  const T au3 = au * au * au, au4 = au3 * au;
  const T tcp = T(0.5) * Ex * au3;                        // au d(th_c)/d(au)
  const T td0p = T(0.25) * Ex * au3 * (au - T(1));
  const T tq1p = T(1.0 / 6.0) * Ex * au4;
  const T tq0p = T(1.0 / 18.0) * Ex * au4 * (au - T(1));
  const T R2 = T(kDielectric) * rinv * rinv, R3 = R2 * rinv, R4 = R3 * rinv;
  const T p = sc.p, hf = T(0.5);
  T X = R2 * (T(2) * p * tcp) * hf * (A[0] * UB[0] - B[0] * UA[0]);
  X += R3 * (T(-4) * p * td0p) * hf * (B[1] * UA[0] + A[1] * UB[0]);
  X += R3 * (T(2) * p * tcp) * hf * (B[2] * UA[1] + B[3] * UA[2] + A[2] * UB[1] + A[3] * UB[2]);
  X += R4 * (T(6) * p * tq0p) * hf * (A[4] * UB[0] - B[4] * UA[0]);
  X += R4 * (T(-2.0 * kSqrt3) * p * tq1p) * hf * (A[5] * UB[1] + A[6] * UB[2] - B[5] * UA[1] - B[6] * UA[2]);
  X += R3 * (T(-2) * td0p) * (UA[0] * UB[0]);
  X += R3 * tcp * (UA[1] * UB[1] + UA[2] * UB[2]);
  if (dmp_floor) { /* dmp pinned at its floor: X still drives dE/dthole, the caller drops the alpha part via p6 == 0 */ }
  return X;
}

// d(pair energy)/d(pscale) at fixed Thole width: pscale multiplies the Thole factor of the five permanent-induced
// coefficients (cud, dud_m0, dud_m1, udq_m0, udq_m1: admp/pme.py:455-470), so the derivative is R_n x const x thole_k
// contracted with the same bilinear shapes as in pair_thole_logderiv.  pscale also enters the Fermi switch of the Thole
// width (pme.py:337-348, 411); that weight is flat to < 1e-38 at every pscale the model uses (0 or 1), and in the reference
// its autodiff derivative is NaN as soon as pscale > ~0.008 (exp overflow); the analytic limit, 0, is used here.
template <class T>
ADMP_HD T pair_pscale_deriv(const Box<T>& box, const Site<T>& I, const Site<T>& J, const PairScales<T>& sc) {
  T d[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = T(1) / m_sqrt(r2);
  const T r = r2 * rinv;
  T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv}, x[3], y[3];
  qi_frame(z, x, y);
  T A[9], B[9], UA[3], UB[3];
  rot_harm(I.Q, x, y, z, A);
  rot_harm(J.Q, x, y, z, B);
  rot_dip(I.U, x, y, z, UA);
  rot_dip(J.U, x, y, z, UB);
  const T aw = sc.w0 * T(kDefaultTholeWidth) + (T(1) - sc.w0) * (I.thole + J.thole);
  T dmp = I.p6 * J.p6;
  dmp = dmp < T(1e-8) ? T(1e-8) : dmp;
  const T u = trim_inf(r * (T(1) / dmp), T(1e8));
  const T au = u * aw;
  const T expau = au < T(50) ? m_exp(-au) : T(0);
  const T au2 = trim_inf(au * au, T(1e8)), au3 = trim_inf(au2 * au, T(1e8)), au4 = trim_inf(au3 * au, T(1e8));
  const T base = au + T(1) + au2 * T(0.5);
  const T th_c = T(1) - expau * base;
  const T th_d0 = T(1) - expau * (base + au3 * T(0.25));
  const T th_q1 = T(1) - expau * (base + au3 * T(1.0 / 6.0));
  const T th_q0 = T(1) - expau * (base + au3 * T(1.0 / 6.0) + au4 * T(1.0 / 18.0));
  const T R2 = T(kDielectric) * rinv * rinv, R3 = R2 * rinv, R4 = R3 * rinv, hf = T(0.5);
  T X = R2 * (T(2) * th_c) * hf * (A[0] * UB[0] - B[0] * UA[0]);
  X += R3 * (T(-4) * th_d0) * hf * (B[1] * UA[0] + A[1] * UB[0]);
  X += R3 * (T(2) * th_c) * hf * (B[2] * UA[1] + B[3] * UA[2] + A[2] * UB[1] + A[3] * UB[2]);
  X += R4 * (T(6) * th_q0) * hf * (A[4] * UB[0] - B[4] * UA[0]);
  X += R4 * (T(-2.0 * kSqrt3) * th_q1) * hf * (A[5] * UB[1] + A[6] * UB[2] - B[5] * UA[1] - B[6] * UA[2]);
  return X;
}

// Field-only evaluation for the SCF (dE/dU_I of the real-space term, admp/pme.py:133):
// only the seven induced coefficients, no radial derivative, no torque.
template <class T>
ADMP_HD void pair_field(const Box<T>& box, const Site<T>& I, const Site<T>& J, const PairScales<T>& sc, T kappa,
                        T fldI[3]) {
  T d[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
  min_image(box, d);
  T r = m_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  T rinv = T(1) / r;
  T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv};
  T x[3], y[3];
  qi_frame(z, x, y);
  T B[9], UB[3];
  rot_harm(J.Q, x, y, z, B);
  rot_dip(J.U, x, y, z, UB);
  T c[NCOEF];
  Radial<T, T> rad;
  rad.init(r, kappa);
  T aw = sc.w0 * T(kDefaultTholeWidth) + (T(1) - sc.w0) * (I.thole + J.thole);
  T dmp = I.p6 * J.p6;
  dmp = dmp < T(1e-8) ? T(1e-8) : dmp;
  ind_coefs(rad, r, aw, dmp, sc.p, c);
  const T hf = T(0.5);
  T FA[3];
  FA[0] = hf * (-c[CUD] * B[0] + c[DUD0] * B[1] - c[UDQ0] * B[4]) + c[UDUD0] * UB[0];
  FA[1] = hf * (c[DUD1] * B[2] - c[UDQ1] * B[5]) + c[UDUD1] * UB[1];
  FA[2] = hf * (c[DUD1] * B[3] - c[UDQ1] * B[6]) + c[UDUD1] * UB[2];
  T cx[3], cy[3], cz[3], t[3];
  frame_cols(x, y, z, cx, cy, cz);
  rot_dip(FA, cx, cy, cz, t);
  fldI[0] += t[0]; fldI[1] += t[1]; fldI[2] += t[2];
}

// pair_field for a charge-only partner J (site_is_mono): of the seven coefficients only cud meets a non-zero component
// (B0 = qJ), and the field points along the pair axis.
template <class T>
ADMP_HD void pair_field_mono(const Box<T>& box, const T rI[3], T tholeI, const T rJ[3], T qJ, T tholeJ, T p, T w0, T kappa,
                             T fldI[3]) {
  T d[3] = {rI[0] - rJ[0], rI[1] - rJ[1], rI[2] - rJ[2]};
  min_image(box, d);
  const T r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const T rinv = m_rsqrt(r2);
  const T r = r2 * rinv;
  MonoRadial<T> m;
  m.init(r, rinv, kappa);
  T f, g;
  mono_cud(m, r, w0 * T(kDefaultTholeWidth) + (T(1) - w0) * (tholeI + tholeJ), p, f, g);
  const T s = T(-0.5) * m.R2 * f * qJ * rinv;          // FA0 / r: the field is FA0 z, z = d / r, harmonic order (z, x, y)
  fldI[0] += s * d[2]; fldI[1] += s * d[0]; fldI[2] += s * d[1];
}

// Change of the real-space dE/dU_I when the partner's induced dipole changes by dUJ (global harmonic order z, x, y): two
// induced dipoles couple only through udud_m0 / udud_m1 (admp/pme.py:472-474, uscales = 1) -- the field is linear in the
// dipoles, so the SCF cycles after the first only need this increment over the polarizable-polarizable pairs
// (engine.hip, "incremental SCF").  Positions / damping data are read from the two site rows.
template <class T>
ADMP_HD void pair_field_ind(const Box<T>& box, const T rI[3], T p6I, T thI, const T rJ[3], T p6J, T thJ, const T dUJ[3],
                            T w0, T kappa, T fldI[3]) {
  T d[3] = {rI[0] - rJ[0], rI[1] - rJ[1], rI[2] - rJ[2]};
  min_image(box, d);
  const T r = m_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const T rinv = T(1) / r;
  const T z[3] = {d[0] * rinv, d[1] * rinv, d[2] * rinv};
  T x[3], y[3], UB[3];
  qi_frame(z, x, y);
  rot_dip(dUJ, x, y, z, UB);
  T c[NCOEF];
  Radial<T, T> rad;
  rad.init(r, kappa);
  const T aw = w0 * T(kDefaultTholeWidth) + (T(1) - w0) * (thI + thJ);
  T dmp = p6I * p6J;
  dmp = dmp < T(1e-8) ? T(1e-8) : dmp;
  ind_coefs(rad, r, aw, dmp, T(0), c);      // only the two induced-induced coefficients are used (the rest is dead code)
  const T FA[3] = {c[UDUD0] * UB[0], c[UDUD1] * UB[1], c[UDUD1] * UB[2]};
  T cx[3], cy[3], cz[3], t[3];
  frame_cols(x, y, z, cx, cy, cz);
  rot_dip(FA, cx, cy, cz, t);
  fldI[0] += t[0]; fldI[1] += t[1]; fldI[2] += t[2];
}

}  // namespace admp
