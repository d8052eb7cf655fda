// Direct (O(N^2) per line) discrete Fourier transforms for PME meshes whose dimensions rocFFT can only do with
// Bluestein's algorithm (a prime factor above 13 -- e.g. the 97^3 mesh the reference's setup_ewald_parameters,
// admp/pme.py:717-744, gives for examples/water_pol_1024).  Replaces jnp.fft.fftn / ifftn of the reference
// (admp/recip.py:410,414) exactly like the rocFFT path does (unnormalised both ways).
//
// Every line transform uses the j <-> N-j symmetry of the twiddles, w^(jk) = c_jk + i s_jk with c even and s odd in j:
//     X[k], X[N-k] = x_0 + sum_{j=1..H} (x_j + x_{N-j}) c_jk  +-  sgn i sum_{j=1..H} (x_j - x_{N-j}) s_jk   (+ x_{N/2} (-1)^k, N even)
// with H = (N-1)/2: one pass over the H pair sums yields two outputs, a quarter of the multiplications of the plain sum.
// The functions here are the per-output-pair arithmetic, shared by dft_kernels.hip and the host-compiled test shim.
#pragma once
#include "pme_math.h"

namespace admp {

template <class T>
struct alignas(2 * sizeof(T)) Cx {
  T re, im;
};
// pair sums of one line position: a = x_j + x_{N-j}, b = x_j - x_{N-j}
template <class T>
struct alignas(4 * sizeof(T)) PairCx {
  T are, aim, bre, bim;
};

// complex line, direction SIGN (-1 forward, +1 inverse): outputs X[k] and X[N-k] (k = 0 .. N/2)
//   ab[(j-1)*stride], j = 1..H   pair sums;  x0 = x_0;  xn = x_{N/2} (used when N is even);  tw[m] = (cos, sin)(2 pi m / N)
template <class T, int SIGN>
ADMP_HD void dft_pair_outputs(int N, int k, int stride, const PairCx<T>* ab, Cx<T> x0, Cx<T> xn, const Cx<T>* tw,
                              Cx<T>& Xk, Cx<T>& Xnk) {
  const int H = (N - 1) / 2;
  T Are = 0, Aim = 0, Bre = 0, Bim = 0;
  int m = 0;
  for (int j = 0; j < H; ++j) {
    m += k;
    if (m >= N) m -= N;
    const Cx<T> w = tw[m];
    const PairCx<T> p = ab[j * stride];
    Are += p.are * w.re;
    Aim += p.aim * w.re;
    Bre += p.bre * w.im;
    Bim += p.bim * w.im;
  }
  T bre = x0.re + Are, bim = x0.im + Aim;
  if ((N & 1) == 0) {
    const T s = (k & 1) ? T(-1) : T(1);
    bre += s * xn.re;
    bim += s * xn.im;
  }
  // sgn * i * B = sgn * (-Bim, Bre)
  Xk.re = bre - T(SIGN) * Bim;
  Xk.im = bim + T(SIGN) * Bre;
  Xnk.re = bre + T(SIGN) * Bim;
  Xnk.im = bim - T(SIGN) * Bre;
}

// real pair sums shared by the r2c and c2r lines: P = sum_j p_j.re c_jk, R = sum_j p_j.im s_jk
template <class T>
ADMP_HD void real_pair_sums(int N, int k, int stride, const Cx<T>* p, const Cx<T>* tw, T& P, T& R) {
  const int H = (N - 1) / 2;
  T sp = 0, sr = 0;
  int m = 0;
  for (int j = 0; j < H; ++j) {
    m += k;
    if (m >= N) m -= N;
    const Cx<T> w = tw[m];
    const Cx<T> v = p[j * stride];
    sp += v.re * w.re;
    sr += v.im * w.im;
  }
  P = sp;
  R = sr;
}

// r2c line: p_j = (x_j + x_{N-j}, x_j - x_{N-j}) real pair sums; X[k] = x0 + P - i R  (+ xn (-1)^k), k = 0 .. N/2
template <class T>
ADMP_HD Cx<T> rdft_output(int N, int k, int stride, const Cx<T>* p, T x0, T xn, const Cx<T>* tw) {
  T P, R;
  real_pair_sums(N, k, stride, p, tw, P, R);
  Cx<T> X;
  X.re = x0 + P;
  if ((N & 1) == 0) X.re += (k & 1) ? -xn : xn;
  X.im = -R;
  return X;
}

// c2r line of a Hermitian half spectrum: p_k = X[k], k = 1..H; x_j = X0.re + 2(P - R), x_{N-j} = X0.re + 2(P + R)
// (+ X[N/2].re (-1)^j); the imaginary parts of X[0] and X[N/2] are ignored like rocFFT's c2r does.
template <class T>
ADMP_HD void irdft_pair_outputs(int N, int j, int stride, const Cx<T>* p, T X0re, T Xnre, const Cx<T>* tw, T& xj, T& xnj) {
  T P, R;
  real_pair_sums(N, j, stride, p, tw, P, R);
  T base = X0re + T(2) * P;
  if ((N & 1) == 0) base += (j & 1) ? -Xnre : Xnre;
  xj = base - T(2) * R;
  xnj = base + T(2) * R;
}

// largest prime factor (host): rocFFT has radix kernels for 2, 3, 5, 7, 11, 13 and falls back to Bluestein above
inline int largest_prime_factor(int n) {
  int best = 1;
  for (int p = 2; (long)p * p <= n; ++p)
    while (n % p == 0) {
      best = p;
      n /= p;
    }
  return n > 1 ? (n > best ? n : best) : best;
}

}  // namespace admp
