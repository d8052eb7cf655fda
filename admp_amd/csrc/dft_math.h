// Direct (O(N^2) per line) discrete Fourier transforms for PME meshes whose dimensions rocFFT can only do with
// Bluestein's algorithm (a prime factor above 13 -- e.g. the 97^3 mesh the reference's setup_ewald_parameters,
// admp/pme.py:717-744, gives for examples/water_pol_1024).  Replaces jnp.fft.fftn / ifftn of the reference
// (admp/recip.py:410,414) exactly like the rocFFT path does (unnormalised both ways).
//
// Every line transform uses the j <-> N-j symmetry of the twiddles, w^(jk) = c_jk + i s_jk with c even and s odd in j:
//     X[k], X[N-k] = x_0 + sum_{j=1..H} (x_j + x_{N-j}) c_jk  +-  sgn i sum_{j=1..H} (x_j - x_{N-j}) s_jk   (+ x_{N/2} (-1)^k, N even)
// with H = (N-1)/2: one pass over the H pair sums yields two outputs, a quarter of the multiplications of the plain sum.
// The kernels are bound by LDS reads, not by arithmetic, so one thread produces KQ output pairs from one read of the
// pair sums and carries the twiddles by rotation instead of fetching them.
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

// Each call produces KQ output pairs of one line (k[0..KQ)): the pair sums are read once for all of them, and the
// twiddle of output k is advanced by the rotation w^k from j to j+1, re-seeded from the exact table every kDftReseed
// steps (so the recurrence never runs longer than that; entries of k outside 0..N/2 may be passed as 0 and ignored).
constexpr int kDftReseed = 8;

template <class T>
ADMP_HD Cx<T> cx_mul(Cx<T> a, Cx<T> b) {
  return Cx<T>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// Twiddle of output k carried from line position j to j + 1.  f32: the rotation w <- w * w^k (4 operations).  f64, where an
// operation costs twice as much and the 97^3 headline mesh spends 40 % of its step in these loops: the three-term recurrence
// w_{j+1} = 2 cos(theta) w_j - w_{j-1}, one FMA per component; its error grows with j^2, which the re-seeding from the exact
// table every kDftReseed steps bounds at ~64 ulp of f64 (in f32 that would show, so f32 keeps the rotation).
template <class T>
struct TwStep {
  Cx<T> w, aux;     // f32: aux = w^k (the rotation); f64: aux = w_{j-1}
  T t;              // f64: 2 cos(theta)
  ADMP_HD void seed(const Cx<T>* tw, int m, int k, int N) {
    w = tw[m];
    if (sizeof(T) == 8) {
      int mp = m - k;
      if (mp < 0) mp += N;
      aux = tw[mp];
      t = T(2) * tw[k].re;
    } else {
      aux = tw[k];
      t = T(0);
    }
  }
  ADMP_HD void next() {
    if (sizeof(T) == 8) {
      const Cx<T> n{t * w.re - aux.re, t * w.im - aux.im};
      aux = w;
      w = n;
    } else {
      w = cx_mul(w, aux);
    }
  }
};

// complex line, direction SIGN (-1 forward, +1 inverse): outputs X[k] and X[N-k] (k = 0 .. N/2)
//   load(j), j = 0..H-1, returns the pair sums of line positions j+1 / N-1-j;  x0 = x_0;  xn = x_{N/2} (used when N is
//   even);  tw[m] = (cos, sin)(2 pi m / N)
// partial sums over the pair positions j = jb0 .. je0-1 (0-based: line positions j+1 / N-1-j), ADDED to Are .. Bim;
// jb0 must be a multiple of kDftReseed steps away from nothing in particular: the twiddle is seeded exactly at jb0
template <class T, int KQ, class LoadAB>
ADMP_HD void dft_pair_partial(int N, const int* k, LoadAB load, const Cx<T>* tw, int jb0, int je0, T* Are, T* Aim, T* Bre,
                              T* Bim) {
  TwStep<T> w[KQ];
  int m[KQ], step[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    m[q] = jb0 == 0 ? k[q] : (int)(((long)k[q] * (jb0 + 1)) % N);
    step[q] = (kDftReseed * k[q]) % N;
  }
  for (int jb = jb0; jb < je0; jb += kDftReseed) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) w[q].seed(tw, m[q], k[q], N);
    const int je = (je0 - jb) < kDftReseed ? (je0 - jb) : kDftReseed;
    if (je == kDftReseed) {
#pragma unroll
      for (int jj = 0; jj < kDftReseed; ++jj) {
        const PairCx<T> p = load(jb + jj);
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          Are[q] += p.are * w[q].w.re;
          Aim[q] += p.aim * w[q].w.re;
          Bre[q] += p.bre * w[q].w.im;
          Bim[q] += p.bim * w[q].w.im;
          w[q].next();
        }
      }
    } else {
      for (int jj = 0; jj < je; ++jj) {
        const PairCx<T> p = load(jb + jj);
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          Are[q] += p.are * w[q].w.re;
          Aim[q] += p.aim * w[q].w.re;
          Bre[q] += p.bre * w[q].w.im;
          Bim[q] += p.bim * w[q].w.im;
          w[q].next();
        }
      }
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      m[q] += step[q];
      if (m[q] >= N) m[q] -= N;
    }
  }
}
// outputs from the complete sums
template <class T, int SIGN, int KQ>
ADMP_HD void dft_pair_finish(int N, const int* k, Cx<T> x0, Cx<T> xn, const T* Are, const T* Aim, const T* Bre, const T* Bim,
                             Cx<T>* Xk, Cx<T>* Xnk) {
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    T bre = x0.re + Are[q], bim = x0.im + Aim[q];
    if ((N & 1) == 0) {
      const T s = (k[q] & 1) ? T(-1) : T(1);
      bre += s * xn.re;
      bim += s * xn.im;
    }
    // sgn * i * B = sgn * (-Bim, Bre)
    Xk[q].re = bre - T(SIGN) * Bim[q];
    Xk[q].im = bim + T(SIGN) * Bre[q];
    Xnk[q].re = bre + T(SIGN) * Bim[q];
    Xnk[q].im = bim - T(SIGN) * Bre[q];
  }
}
template <class T, int SIGN, int KQ, class LoadAB>
ADMP_HD void dft_pair_core(int N, const int* k, LoadAB load, Cx<T> x0, Cx<T> xn, const Cx<T>* tw, Cx<T>* Xk, Cx<T>* Xnk) {
  T Are[KQ], Aim[KQ], Bre[KQ], Bim[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) Are[q] = Aim[q] = Bre[q] = Bim[q] = T(0);
  dft_pair_partial<T, KQ>(N, k, load, tw, 0, (N - 1) / 2, Are, Aim, Bre, Bim);
  dft_pair_finish<T, SIGN, KQ>(N, k, x0, xn, Are, Aim, Bre, Bim, Xk, Xnk);
}
// where a line is split between two threads: the first takes positions 0 .. dft_split(N)-1 (whole re-seed blocks)
ADMP_HD int dft_split(int N) {
  const int H = (N - 1) / 2, mid = ((H / 2 + kDftReseed - 1) / kDftReseed) * kDftReseed;
  return mid < H ? mid : H;
}
// pair sums stored as one record per position: ab[(j-1)*stride], j = 1..H
template <class T, int SIGN, int KQ>
ADMP_HD void dft_pair_outputs(int N, const int* k, int stride, const PairCx<T>* ab, Cx<T> x0, Cx<T> xn, const Cx<T>* tw,
                              Cx<T>* Xk, Cx<T>* Xnk) {
  dft_pair_core<T, SIGN, KQ>(N, k, [=](int j) { return ab[j * stride]; }, x0, xn, tw, Xk, Xnk);
}
// pair sums formed in place in a [N][stride] array of line values: row j holds x_j + x_{N-j}, row N-j holds x_j - x_{N-j}
template <class T, int SIGN, int KQ>
ADMP_HD void dft_pair_outputs_rows(int N, const int* k, int stride, const Cx<T>* col, const Cx<T>* tw, Cx<T>* Xk,
                                   Cx<T>* Xnk) {
  const Cx<T> x0 = col[0];
  const Cx<T> xn = (N & 1) ? Cx<T>{T(0), T(0)} : col[(N / 2) * stride];
  dft_pair_core<T, SIGN, KQ>(N, k, [=](int j) {
    const Cx<T> a = col[(1 + j) * stride], b = col[(N - 1 - j) * stride];
    return PairCx<T>{a.re, a.im, b.re, b.im};
  }, x0, xn, tw, Xk, Xnk);
}

// real pair sums shared by the r2c and c2r lines: P = sum_j p_j.re c_jk, R = sum_j p_j.im s_jk
template <class T, int KQ>
ADMP_HD void real_pair_partial(int N, const int* k, int stride, const Cx<T>* p, const Cx<T>* tw, int jb0, int je0, T* P, T* R) {
  TwStep<T> w[KQ];
  int m[KQ], step[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    m[q] = jb0 == 0 ? k[q] : (int)(((long)k[q] * (jb0 + 1)) % N);
    step[q] = (kDftReseed * k[q]) % N;
  }
  for (int jb = jb0; jb < je0; jb += kDftReseed) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) w[q].seed(tw, m[q], k[q], N);
    const int je = (je0 - jb) < kDftReseed ? (je0 - jb) : kDftReseed;
    if (je == kDftReseed) {
#pragma unroll
      for (int jj = 0; jj < kDftReseed; ++jj) {
        const Cx<T> v = p[(jb + jj) * stride];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          P[q] += v.re * w[q].w.re;
          R[q] += v.im * w[q].w.im;
          w[q].next();
        }
      }
    } else {
      for (int jj = 0; jj < je; ++jj) {
        const Cx<T> v = p[(jb + jj) * stride];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          P[q] += v.re * w[q].w.re;
          R[q] += v.im * w[q].w.im;
          w[q].next();
        }
      }
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      m[q] += step[q];
      if (m[q] >= N) m[q] -= N;
    }
  }
}
template <class T, int KQ>
ADMP_HD void real_pair_sums(int N, const int* k, int stride, const Cx<T>* p, const Cx<T>* tw, T* P, T* R) {
#pragma unroll
  for (int q = 0; q < KQ; ++q) P[q] = R[q] = T(0);
  real_pair_partial<T, KQ>(N, k, stride, p, tw, 0, (N - 1) / 2, P, R);
}

// r2c line: p_j = (x_j + x_{N-j}, x_j - x_{N-j}) real pair sums; X[k] = x0 + P - i R  (+ xn (-1)^k), k = 0 .. N/2
template <class T, int KQ>
ADMP_HD void rdft_outputs(int N, const int* k, int stride, const Cx<T>* p, T x0, T xn, const Cx<T>* tw, Cx<T>* X) {
  T P[KQ], R[KQ];
  real_pair_sums<T, KQ>(N, k, stride, p, tw, P, R);
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    X[q].re = x0 + P[q];
    if ((N & 1) == 0) X[q].re += (k[q] & 1) ? -xn : xn;
    X[q].im = -R[q];
  }
}

// c2r line of a Hermitian half spectrum: p_k = X[k], k = 1..H; x_j = X0.re + 2(P - R), x_{N-j} = X0.re + 2(P + R)
// (+ X[N/2].re (-1)^j); the imaginary parts of X[0] and X[N/2] are ignored like rocFFT's c2r does.
template <class T, int KQ>
ADMP_HD void irdft_pair_outputs(int N, const int* j, int stride, const Cx<T>* p, T X0re, T Xnre, const Cx<T>* tw, T* xj,
                                T* xnj) {
  T P[KQ], R[KQ];
  real_pair_sums<T, KQ>(N, j, stride, p, tw, P, R);
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    T base = X0re + T(2) * P[q];
    if ((N & 1) == 0) base += (j[q] & 1) ? -Xnre : Xnre;
    xj[q] = base - T(2) * R[q];
    xnj[q] = base + T(2) * R[q];
  }
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
