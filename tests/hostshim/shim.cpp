// Host-compiled harness around the SAME per-pair / per-atom arithmetic the HIP kernels run
// (admp_amd/csrc/*_math.h).  TEST INFRASTRUCTURE: lets the CPU test-suite check the hand-coded
// adjoints against the oracle without a GPU.  It is built by tests/ only and is never loaded by
// the admp_amd package (the product path has no CPU fallback).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../admp_amd/csrc/frame_math.h"
#include "../../admp_amd/csrc/pme_math.h"
#include "../../admp_amd/csrc/spline_math.h"
#include "../../admp_amd/csrc/disp_math.h"
#include "../../admp_amd/csrc/dft_math.h"

using namespace admp;

template <class T>
static Box<T> make_box(const double* h) {
  Box<T> b;
  double d = h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) + h[2] * (h[3] * h[7] - h[4] * h[6]);
  double inv[9] = {(h[4] * h[8] - h[5] * h[7]) / d, (h[2] * h[7] - h[1] * h[8]) / d, (h[1] * h[5] - h[2] * h[4]) / d,
                   (h[5] * h[6] - h[3] * h[8]) / d, (h[0] * h[8] - h[2] * h[6]) / d, (h[2] * h[3] - h[0] * h[5]) / d,
                   (h[3] * h[7] - h[4] * h[6]) / d, (h[1] * h[6] - h[0] * h[7]) / d, (h[0] * h[4] - h[1] * h[3]) / d};
  for (int k = 0; k < 9; ++k) { b.h[k] = (T)h[k]; b.hinv[k] = (T)inv[k]; }
  return b;
}

template <class T>
static Site<T> load_site(int i, const double* pos, const double* Q, const double* U, const double* p6, const double* th) {
  Site<T> s;
  for (int k = 0; k < 3; ++k) s.r[k] = (T)pos[3 * i + k];
  for (int k = 0; k < 9; ++k) s.Q[k] = (T)Q[9 * i + k];
  for (int k = 0; k < 3; ++k) s.U[k] = U ? (T)U[3 * i + k] : T(0);
  s.p6 = p6 ? (T)p6[i] : T(0);
  s.thole = th ? (T)th[i] : T(0);
  s.pad[0] = s.pad[1] = s.pad[2] = T(0);
  // the charge-only mark of k_prepare_sites (atom_kernels.hip)
  bool mono = s.p6 == T(0) && s.U[0] == T(0) && s.U[1] == T(0) && s.U[2] == T(0);
  for (int k = 1; k < 9; ++k) mono = mono && s.Q[k] == T(0);
  s.pad[0] = mono ? T(1) : T(0);
  return s;
}

// mode 0: half list, both sites updated from one evaluation; mode 1: every pair evaluated from
// both ends, centre-only accumulation (what the GPU kernel does); energy halves summed;
// mode 2: as mode 1 with the charge-only dispatch of k_pair_full (pair_full_mono / pair_mono_full / pair_mono_mono).
template <class T>
static double pair_real(int na, const double* pos, const double* Q, const double* U, const double* p6, const double* th,
                        const double* boxh, long np, const int32_t* pairs, const int32_t* nb, const double* mtab,
                        const double* ptab, const double* w0tab, double kappa, int lpol, int mode, double* grad,
                        double* pot, double* fld) {
  Box<T> box = make_box<T>(boxh);
  double e = 0;
  for (long p = 0; p < np; ++p) {
    int i = pairs[2 * p], j = pairs[2 * p + 1];
    PairScales<T> sc = {(T)(mtab[nb[p]] - 1.0), (T)ptab[nb[p]], (T)w0tab[nb[p]]};
    Site<T> I = load_site<T>(i, pos, Q, U, p6, th), J = load_site<T>(j, pos, Q, U, p6, th);
    for (int side = 0; side < (mode ? 2 : 1); ++side) {
      T g[3] = {0, 0, 0}, pi[9] = {0}, fi[3] = {0, 0, 0}, pj[9] = {0}, fj[3] = {0, 0, 0};
      const Site<T>& C = side ? J : I;
      const Site<T>& P = side ? I : J;
      int ci = side ? j : i, pi_ = side ? i : j;
      T ep;
      if (mode == 0) {
        ep = lpol ? pair_energy_grad<T, true, true>(box, C, P, sc, (T)kappa, g, pi, fi, pj, fj)
                  : pair_energy_grad<T, false, true>(box, C, P, sc, (T)kappa, g, pi, fi, pj, fj);
        e += ep;
        for (int k = 0; k < 3; ++k) { grad[3 * ci + k] += g[k]; grad[3 * pi_ + k] -= g[k]; }
        for (int k = 0; k < 9; ++k) { pot[9 * ci + k] += pi[k]; pot[9 * pi_ + k] += pj[k]; }
        for (int k = 0; k < 3; ++k) { fld[3 * ci + k] += fi[k]; fld[3 * pi_ + k] += fj[k]; }
      } else {
        const bool cm = mode == 2 && site_is_mono(C), pm = mode == 2 && site_is_mono(P);
        if (cm && pm) ep = pair_mono_mono<T>(box, C.r, C.Q[0], P.r, P.Q[0], sc.mm, (T)kappa, g, pi[0]);
        else if (cm) ep = lpol ? pair_mono_full<T, true>(box, C.r, C.Q[0], C.thole, P, sc, (T)kappa, g, pi[0])
                               : pair_mono_full<T, false>(box, C.r, C.Q[0], C.thole, P, sc, (T)kappa, g, pi[0]);
        else if (pm) ep = lpol ? pair_full_mono<T, true>(box, C, P.r, P.Q[0], P.thole, sc, (T)kappa, g, pi, fi)
                               : pair_full_mono<T, false>(box, C, P.r, P.Q[0], P.thole, sc, (T)kappa, g, pi, fi);
        else
        ep = lpol ? pair_energy_grad<T, true, false>(box, C, P, sc, (T)kappa, g, pi, fi, pj, fj)
                  : pair_energy_grad<T, false, false>(box, C, P, sc, (T)kappa, g, pi, fi, pj, fj);
        e += 0.5 * ep;
        for (int k = 0; k < 3; ++k) grad[3 * ci + k] += g[k];
        for (int k = 0; k < 9; ++k) pot[9 * ci + k] += pi[k];
        for (int k = 0; k < 3; ++k) fld[3 * ci + k] += fi[k];
      }
    }
  }
  return e;
}

template <class T>
static void pair_field_all(int na, const double* pos, const double* Q, const double* U, const double* p6, const double* th,
                           const double* boxh, long np, const int32_t* pairs, const int32_t* nb, const double* ptab,
                           const double* w0tab, double kappa, double* fld, int mono = 0) {
  Box<T> box = make_box<T>(boxh);
  for (long p = 0; p < np; ++p) {
    int i = pairs[2 * p], j = pairs[2 * p + 1];
    PairScales<T> sc = {T(0), (T)ptab[nb[p]], (T)w0tab[nb[p]]};
    Site<T> I = load_site<T>(i, pos, Q, U, p6, th), J = load_site<T>(j, pos, Q, U, p6, th);
    T fi[3] = {0, 0, 0}, fj[3] = {0, 0, 0};
    // mono: the dispatch of k_pair_field (charge-only partner -> pair_field_mono)
    if (mono && site_is_mono(J)) pair_field_mono<T>(box, I.r, I.thole, J.r, J.Q[0], J.thole, sc.p, sc.w0, (T)kappa, fi);
    else pair_field(box, I, J, sc, (T)kappa, fi);
    if (mono && site_is_mono(I)) pair_field_mono<T>(box, J.r, J.thole, I.r, I.Q[0], I.thole, sc.p, sc.w0, (T)kappa, fj);
    else
    pair_field(box, J, I, sc, (T)kappa, fj);
    for (int k = 0; k < 3; ++k) { fld[3 * i + k] += fi[k]; fld[3 * j + k] += fj[k]; }
  }
}

template <class T>
static void frames(int na, const double* pos, const double* boxh, const int32_t* atype, const int32_t* aidx,
                   const double* Qlocal, const double* pot, double* frames_out, double* Qglobal, double* grad,
                   double* dQlocal) {
  Box<T> box = make_box<T>(boxh);
  for (int i = 0; i < na; ++i) {
    int t = atype[i];
    T p[3], pz[3] = {0, 0, 0}, px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
    int iz = aidx[3 * i], ix = aidx[3 * i + 1], iy = aidx[3 * i + 2];
    for (int k = 0; k < 3; ++k) {
      p[k] = (T)pos[3 * i + k];
      if (iz >= 0) pz[k] = (T)pos[3 * iz + k];
      if (ix >= 0) px[k] = (T)pos[3 * ix + k];
      if (iy >= 0) py[k] = (T)pos[3 * iy + k];
    }
    FrameWork<T> w;
    local_frame_fwd(t, box, p, pz, px, py, w);
    T ql[9], qg[9], cx[3], cy[3], cz[3];
    for (int k = 0; k < 9; ++k) ql[k] = (T)Qlocal[9 * i + k];
    frame_cols(w.X, w.Y, w.Z, cx, cy, cz);
    rot_harm(ql, cx, cy, cz, qg);   // local -> global = rotation with the transposed frame (multipole.py:201)
    for (int k = 0; k < 3; ++k) { frames_out[9 * i + k] = w.X[k]; frames_out[9 * i + 3 + k] = w.Y[k]; frames_out[9 * i + 6 + k] = w.Z[k]; }
    for (int k = 0; k < 9; ++k) Qglobal[9 * i + k] = qg[k];
    if (pot) {
      T P[9], tau[3], gp[3], gz[3], gx[3], gy[3], dl[9];
      for (int k = 0; k < 9; ++k) P[k] = (T)pot[9 * i + k];
      multipole_torque(P, qg, tau);
      local_frame_bwd(t, w, tau, gp, gz, gx, gy);
      for (int k = 0; k < 3; ++k) {
        grad[3 * i + k] += gp[k];
        if (iz >= 0) grad[3 * iz + k] += gz[k];
        if (ix >= 0) grad[3 * ix + k] += gx[k];
        if (iy >= 0) grad[3 * iy + k] += gy[k];
      }
      rot_harm(P, w.X, w.Y, w.Z, dl);
      for (int k = 0; k < 9; ++k) dQlocal[9 * i + k] = dl[k];
    }
  }
}

template <class T>
static RecipGeom<T> make_geom(const double* boxh, const int* K) {
  Box<double> b = make_box<double>(boxh);
  RecipGeom<T> g;
  for (int d = 0; d < 3; ++d) g.K[d] = K[d];
  for (int k = 0; k < 9; ++k) g.hinv[k] = (T)b.hinv[k];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      g.Aop[3 * i + j] = (T)(-(double)K[i] * b.hinv[3 * j + i]);   // -Nstar[i][j], Nstar[a][b] = N[a] inv[b][a]
      g.Jac[3 * i + j] = (T)(-(double)K[j] * b.hinv[3 * i + j]);   // du_j/dx_i
    }
  g.whole_mesh();
  return g;
}

template <class T>
static void spread(int na, const double* pos, const double* Q, const double* boxh, const int* K, double* mesh) {
  RecipGeom<T> g = make_geom<T>(boxh, K);
  for (int i = 0; i < na; ++i) {
    T r[3], q[9];
    for (int k = 0; k < 3; ++k) r[k] = (T)pos[3 * i + k];
    for (int k = 0; k < 9; ++k) q[k] = (T)Q[9 * i + k];
    spread_atom(g, r, q, [&](long idx, T v) { mesh[idx] += (double)v; });
  }
}

template <class T>
static void gather(int na, const double* pos, const double* Q, const double* boxh, const int* K, const double* phi,
                   double* pot, double* grad, double* fieldonly) {
  RecipGeom<T> g = make_geom<T>(boxh, K);
  for (int i = 0; i < na; ++i) {
    T r[3], q[9], F[NF], P[9] = {0}, gr[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) r[k] = (T)pos[3 * i + k];
    for (int k = 0; k < 9; ++k) q[k] = (T)Q[9 * i + k];
    gather_atom(g, r, [&](long idx) { return (T)phi[idx]; }, F);
    unfold_potential(g, q, F, P, gr);
    for (int k = 0; k < 9; ++k) pot[9 * i + k] += P[k];
    for (int k = 0; k < 3; ++k) grad[3 * i + k] += gr[k];
    if (fieldonly) {
      T gx[3];
      gather_atom_field(g, r, [&](long idx) { return (T)phi[idx]; }, gx);
      for (int k = 0; k < 3; ++k) fieldonly[3 * i + k] = gx[k];
    }
  }
}

template <class T>
static double disp_real(int na, const double* pos, const double* c, const double* boxh, long np, const int32_t* pairs,
                        const int32_t* nb, const double* mtab, double kappa, int pmax, double* grad) {
  Box<T> box = make_box<T>(boxh);
  double e = 0;
  for (long p = 0; p < np; ++p) {
    int i = pairs[2 * p], j = pairs[2 * p + 1];
    T ri[3], rj[3], ci[3], cj[3], g[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) { ri[k] = (T)pos[3 * i + k]; rj[k] = (T)pos[3 * j + k]; ci[k] = (T)c[3 * i + k]; cj[k] = (T)c[3 * j + k]; }
    e += disp_pair(box, ri, rj, ci, cj, (T)(mtab[nb[p]] - 1.0), (T)kappa, pmax, g);
    for (int k = 0; k < 3; ++k) { grad[3 * i + k] += g[k]; grad[3 * j + k] -= g[k]; }
  }
  return e;
}

template <class T>
static double tt_real(int na, const double* pos, const double* abqc, const double* boxh, long np, const int32_t* pairs,
                      const int32_t* nb, const double* mtab, double* grad) {
  Box<T> box = make_box<T>(boxh);
  double e = 0;
  for (long p = 0; p < np; ++p) {
    int i = pairs[2 * p], j = pairs[2 * p + 1];
    T ri[3], rj[3], pi[4], pj[4], g[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) { ri[k] = (T)pos[3 * i + k]; rj[k] = (T)pos[3 * j + k]; }
    for (int k = 0; k < 4; ++k) { pi[k] = (T)abqc[4 * i + k]; pj[k] = (T)abqc[4 * j + k]; }
    e += tt_pair(box, ri, rj, pi, pj, (T)mtab[nb[p]], g);
    for (int k = 0; k < 3; ++k) { grad[3 * i + k] += g[k]; grad[3 * j + k] -= g[k]; }
  }
  return e;
}

// ---- direct DFT lines (dft_math.h), driven like dft_kernels.hip drives them: KQ outputs per call, k = g + q * TK
template <class T, int KQ>
static void dft_line(int N, int sign, const double* in, double* out) {
  const int H = (N - 1) / 2, Kh = N / 2 + 1, TK = (Kh + KQ - 1) / KQ;
  std::vector<Cx<T>> tw(N);
  for (int m = 0; m < N; ++m) tw[m] = Cx<T>{(T)std::cos(2.0 * M_PI * m / N), (T)std::sin(2.0 * M_PI * m / N)};
  std::vector<PairCx<T>> ab(H > 0 ? H : 1);
  for (int j = 1; j <= H; ++j)
    ab[j - 1] = PairCx<T>{(T)(in[2 * j] + in[2 * (N - j)]), (T)(in[2 * j + 1] + in[2 * (N - j) + 1]),
                          (T)(in[2 * j] - in[2 * (N - j)]), (T)(in[2 * j + 1] - in[2 * (N - j) + 1])};
  Cx<T> x0{(T)in[0], (T)in[1]}, xn{T(0), T(0)};
  if (N % 2 == 0) xn = Cx<T>{(T)in[N], (T)in[N + 1]};
  for (int g = 0; g < TK; ++g) {
    int k[KQ];
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK < Kh) ? g + q * TK : 0;
    Cx<T> a[KQ], b[KQ];
    if (sign < 0) dft_pair_outputs<T, -1, KQ>(N, k, 1, ab.data(), x0, xn, tw.data(), a, b);
    else dft_pair_outputs<T, +1, KQ>(N, k, 1, ab.data(), x0, xn, tw.data(), a, b);
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK;
      if (kq >= Kh) continue;
      out[2 * kq] = a[q].re; out[2 * kq + 1] = a[q].im;
      if (kq != 0 && 2 * kq != N) { out[2 * (N - kq)] = b[q].re; out[2 * (N - kq) + 1] = b[q].im; }
    }
  }
}
template <class T, int KQ>
static void rdft_line(int N, const double* in, double* out) {
  const int H = (N - 1) / 2, Kh = N / 2 + 1, TK = (Kh + KQ - 1) / KQ;
  std::vector<Cx<T>> tw(N), p(H > 0 ? H : 1);
  for (int m = 0; m < N; ++m) tw[m] = Cx<T>{(T)std::cos(2.0 * M_PI * m / N), (T)std::sin(2.0 * M_PI * m / N)};
  for (int j = 1; j <= H; ++j) p[j - 1] = Cx<T>{(T)(in[j] + in[N - j]), (T)(in[j] - in[N - j])};
  for (int g = 0; g < TK; ++g) {
    int k[KQ];
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK < Kh) ? g + q * TK : 0;
    Cx<T> X[KQ];
    rdft_outputs<T, KQ>(N, k, 1, p.data(), (T)in[0], (N % 2 == 0) ? (T)in[N / 2] : T(0), tw.data(), X);
    for (int q = 0; q < KQ; ++q)
      if (g + q * TK < Kh) { out[2 * (g + q * TK)] = X[q].re; out[2 * (g + q * TK) + 1] = X[q].im; }
  }
}
template <class T, int KQ>
static void irdft_line(int N, const double* in, double* out) {
  const int H = (N - 1) / 2, Kh = N / 2 + 1, TK = (Kh + KQ - 1) / KQ;
  std::vector<Cx<T>> tw(N), p(H > 0 ? H : 1);
  for (int m = 0; m < N; ++m) tw[m] = Cx<T>{(T)std::cos(2.0 * M_PI * m / N), (T)std::sin(2.0 * M_PI * m / N)};
  for (int k = 1; k <= H; ++k) p[k - 1] = Cx<T>{(T)in[2 * k], (T)in[2 * k + 1]};
  for (int g = 0; g < TK; ++g) {
    int j[KQ];
    for (int q = 0; q < KQ; ++q) j[q] = (g + q * TK < Kh) ? g + q * TK : 0;
    T a[KQ], b[KQ];
    irdft_pair_outputs<T, KQ>(N, j, 1, p.data(), (T)in[0], (N % 2 == 0) ? (T)in[N] : T(0), tw.data(), a, b);
    for (int q = 0; q < KQ; ++q) {
      const int jq = g + q * TK;
      if (jq >= Kh) continue;
      out[jq] = a[q];
      if (jq != 0 && 2 * jq != N) out[N - jq] = b[q];
    }
  }
}
template <class T, int KQ>
static void dft_any(int kind, int N, int sign, const double* in, double* out) {
  if (kind == 0) dft_line<T, KQ>(N, sign, in, out);
  else if (kind == 1) rdft_line<T, KQ>(N, in, out);
  else irdft_line<T, KQ>(N, in, out);
}

extern "C" {
double shim_pair_real(int prec, int na, const double* pos, const double* Q, const double* U, const double* p6,
                      const double* th, const double* boxh, long np, const int32_t* pairs, const int32_t* nb,
                      const double* mtab, const double* ptab, const double* w0tab, double kappa, int lpol, int mode,
                      double* grad, double* pot, double* fld) {
  return prec == 4 ? pair_real<float>(na, pos, Q, U, p6, th, boxh, np, pairs, nb, mtab, ptab, w0tab, kappa, lpol, mode, grad, pot, fld)
                   : pair_real<double>(na, pos, Q, U, p6, th, boxh, np, pairs, nb, mtab, ptab, w0tab, kappa, lpol, mode, grad, pot, fld);
}
void shim_pair_field(int prec, int na, const double* pos, const double* Q, const double* U, const double* p6,
                     const double* th, const double* boxh, long np, const int32_t* pairs, const int32_t* nb,
                     const double* ptab, const double* w0tab, double kappa, double* fld) {
  if (prec == 4) pair_field_all<float>(na, pos, Q, U, p6, th, boxh, np, pairs, nb, ptab, w0tab, kappa, fld);
  else pair_field_all<double>(na, pos, Q, U, p6, th, boxh, np, pairs, nb, ptab, w0tab, kappa, fld);
}
void shim_pair_field_mono(int prec, int na, const double* pos, const double* Q, const double* U, const double* p6,
                          const double* th, const double* boxh, long np, const int32_t* pairs, const int32_t* nb,
                          const double* ptab, const double* w0tab, double kappa, double* fld) {
  if (prec == 4) pair_field_all<float>(na, pos, Q, U, p6, th, boxh, np, pairs, nb, ptab, w0tab, kappa, fld, 1);
  else pair_field_all<double>(na, pos, Q, U, p6, th, boxh, np, pairs, nb, ptab, w0tab, kappa, fld, 1);
}
void shim_frames(int prec, int na, const double* pos, const double* boxh, const int32_t* atype, const int32_t* aidx,
                 const double* Qlocal, const double* pot, double* frames_out, double* Qglobal, double* grad,
                 double* dQlocal) {
  if (prec == 4) frames<float>(na, pos, boxh, atype, aidx, Qlocal, pot, frames_out, Qglobal, grad, dQlocal);
  else frames<double>(na, pos, boxh, atype, aidx, Qlocal, pot, frames_out, Qglobal, grad, dQlocal);
}
void shim_spread(int prec, int na, const double* pos, const double* Q, const double* boxh, const int* K, double* mesh) {
  if (prec == 4) spread<float>(na, pos, Q, boxh, K, mesh);
  else spread<double>(na, pos, Q, boxh, K, mesh);
}
void shim_gather(int prec, int na, const double* pos, const double* Q, const double* boxh, const int* K,
                 const double* phi, double* pot, double* grad, double* fieldonly) {
  if (prec == 4) gather<float>(na, pos, Q, boxh, K, phi, pot, grad, fieldonly);
  else gather<double>(na, pos, Q, boxh, K, phi, pot, grad, fieldonly);
}
void shim_bspline6(double f, double* out24) {
  bspline6<double>(f, out24, out24 + 6, out24 + 12, out24 + 18);
}
double shim_disp_real(int prec, int na, const double* pos, const double* c, const double* boxh, long np,
                      const int32_t* pairs, const int32_t* nb, const double* mtab, double kappa, int pmax, double* grad) {
  return prec == 4 ? disp_real<float>(na, pos, c, boxh, np, pairs, nb, mtab, kappa, pmax, grad)
                   : disp_real<double>(na, pos, c, boxh, np, pairs, nb, mtab, kappa, pmax, grad);
}
double shim_tt_real(int prec, int na, const double* pos, const double* abqc, const double* boxh, long np,
                    const int32_t* pairs, const int32_t* nb, const double* mtab, double* grad) {
  return prec == 4 ? tt_real<float>(na, pos, abqc, boxh, np, pairs, nb, mtab, grad)
                   : tt_real<double>(na, pos, abqc, boxh, np, pairs, nb, mtab, grad);
}
double shim_disp_ck(int which, double ksq, double kappa, double V) { return disp_ck(which, ksq, kappa, V); }
// kind 0: complex line (sign -1 / +1), 1: r2c line, 2: c2r line of a half spectrum; kq = outputs per call (1, 2, 4)
void shim_dft_line(int prec, int kq, int kind, int N, int sign, const double* in, double* out) {
  if (prec == 4) {
    if (kq == 1) dft_any<float, 1>(kind, N, sign, in, out);
    else if (kq == 2) dft_any<float, 2>(kind, N, sign, in, out);
    else dft_any<float, 4>(kind, N, sign, in, out);
  } else {
    if (kq == 1) dft_any<double, 1>(kind, N, sign, in, out);
    else if (kq == 2) dft_any<double, 2>(kind, N, sign, in, out);
    else dft_any<double, 4>(kind, N, sign, in, out);
  }
}
int shim_largest_prime_factor(int n) { return largest_prime_factor(n); }
}
