// Fused x pass of the mesh convolution for power-of-two meshes: x lines forward, times G (+ the reciprocal energy), x lines
// inverse -- ONE sweep over the half spectrum where rocFFT's 3-D plans + k_kspace make three (the x pass of the r2c plan,
// the G multiply, the x pass of the c2r plan; 58 + 43 + 58 us at 256^3 f32).  rocFFT keeps the y-z planes (batched 2-D
// r2c / c2r plans, engine.hip); replaces fftn / ifftn + the k-space product of admp/recip.py:395-426 exactly like they do
// (unnormalised both ways).
//
// A workgroup holds NC neighbouring columns (y, kz) of all N x positions in LDS.  Forward: radix-2 decimation in frequency,
// natural order in, bit-reversed order out; the G multiply works on that order (row p holds frequency bitrev(p));
// inverse: radix-2 decimation in time, bit-reversed in, natural out -- in place, no reordering pass, one twiddle table.
#include "dft_math.h"
#include "launch.h"
#include "reduce.h"

namespace admp {

constexpr int kFftxBlock = 256;
extern __shared__ __align__(32) unsigned char fftx_smem[];

constexpr int kFftxGMax = 16;      // G values a thread keeps in registers: N * NC / kFftxBlock <= 16 (else fetched late)
template <class T>
__global__ __launch_bounds__(kFftxBlock) void k_fftx_conv(int N, int logN, int ncols, int nfix, int NC, long jstride, long fixstride,
                                                          long gjstride, long gfixstride, int K3, Cx<T>* __restrict__ spec,
                                                          DftTabs<T> tabs, long spec_bstride, const Cx<T>* __restrict__ twg,
                                                          double* energies, int slot) {
  // blockIdx.y = channel of a batch (dispersion PME: the C6 / C8 / C10 meshes, one G table each)
  const T* __restrict__ gtab = tabs.p[blockIdx.y];
  spec += (long)blockIdx.y * spec_bstride;
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(fftx_smem);      // [N / 2]: (cos, sin)(2 pi k / N)
  Cx<T>* D = tw + N / 2;                                 // [N][NC]
  // tiles of one row of columns share their 128-B lines: neighbours on the same XCD (same L2)
  const int ntile = (ncols + NC - 1) / NC;
  const long L = xcd_block(blockIdx.x, (unsigned)(ntile * nfix));
  if (L < 0) return;
  const int col0 = (int)(L % ntile) * NC, nca = min(NC, ncols - col0);
  const long base = (L / ntile) * fixstride + col0;
  const long gbase = (L / ntile) * gfixstride + col0;      // the G table keeps the unpadded rows
  const int sh = 31 - __clz(NC);                         // NC is a power of two
  for (int t = threadIdx.x; t < N / 2; t += kFftxBlock) tw[t] = twg[t];
#pragma unroll 4
  for (int t = threadIdx.x; t < N * NC; t += kFftxBlock) {
    const int j = t >> sh, c = t & (NC - 1);
    D[t] = c < nca ? spec[base + (long)j * jstride + c] : Cx<T>{T(0), T(0)};
  }
  // the G factors this thread applies after the forward transform: fetched now, their latency hides behind the transform
  const bool gpre = N * NC <= kFftxGMax * kFftxBlock;
  T Gr[kFftxGMax];
#pragma unroll
  for (int u = 0; u < kFftxGMax; ++u) {
    const int t = threadIdx.x + u * kFftxBlock;
    const int p = t >> sh, c = t & (NC - 1);
    Gr[u] = (gpre && t < N * NC && c < nca) ? gtab[gbase + (long)(__brev((unsigned)p) >> (32 - logN)) * gjstride + c] : T(0);
  }
  __syncthreads();
  const int nbf = (N / 2) * NC, nbq = (N / 4) * NC;
  auto cmul = [](Cx<T> a, Cx<T> b) { return Cx<T>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; };
  auto cmulc = [](Cx<T> a, Cx<T> b) { return Cx<T>{a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; };   // a conj(b)
  // forward, decimation in frequency: spans N/2, N/4, .. 1; W = exp(-2 pi i / N).  Two stages (spans 2q, q) per LDS round
  // trip: same arithmetic and data order as two radix-2 stages, half the LDS traffic and barriers.
  int s = logN - 1;
  for (; s >= 1; s -= 2) {
    const int q = 1 << (s - 1);                     // spans 2q (stage s) and q (stage s - 1)
    for (int t = threadIdx.x; t < nbq; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NC - 1);
      const int k = b & (q - 1), i0 = ((b >> (s - 1)) << (s + 1)) | k;
      const Cx<T> w1 = tw[k << (logN - 1 - s)], w2 = tw[k << (logN - s)];      // W_{4q}^k, W_{2q}^k
      Cx<T>* p = D + ((size_t)i0 << sh) + c;
      const int st = q << sh;
      const Cx<T> x0 = p[0], x1 = p[st], x2 = p[2 * st], x3 = p[3 * st];
      const Cx<T> t0{x0.re + x2.re, x0.im + x2.im}, t1{x1.re + x3.re, x1.im + x3.im};
      const Cx<T> t2 = cmulc(Cx<T>{x0.re - x2.re, x0.im - x2.im}, w1);
      const Cx<T> d13 = cmulc(Cx<T>{x1.re - x3.re, x1.im - x3.im}, w1);
      const Cx<T> t3{d13.im, -d13.re};                                            // times -i (W_{4q}^q)
      p[0] = Cx<T>{t0.re + t1.re, t0.im + t1.im};
      p[st] = cmulc(Cx<T>{t0.re - t1.re, t0.im - t1.im}, w2);
      p[2 * st] = Cx<T>{t2.re + t3.re, t2.im + t3.im};
      p[3 * st] = cmulc(Cx<T>{t2.re - t3.re, t2.im - t3.im}, w2);
    }
    __syncthreads();
  }
  for (; s >= 0; --s) {                             // odd log2 N: one plain stage (span 1) is left
    const int half = 1 << s;
    for (int t = threadIdx.x; t < nbf; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NC - 1);
      const int k = b & (half - 1), i = ((b >> s) << (s + 1)) | k;
      const Cx<T> w = tw[k << (logN - 1 - s)];
      const Cx<T> u = D[(i << sh) + c], v = D[((i + half) << sh) + c];
      const T dr = u.re - v.re, di = u.im - v.im;
      D[(i << sh) + c] = Cx<T>{u.re + v.re, u.im + v.im};
      D[((i + half) << sh) + c] = Cx<T>{dr * w.re + di * w.im, di * w.re - dr * w.im};      // (u - v) conj(w)
    }
    __syncthreads();
  }
  // times G; row p holds frequency bitrev(p)
  double e = 0.0;
#pragma unroll
  for (int u = 0; u < kFftxGMax; ++u) {
    const int t = threadIdx.x + u * kFftxBlock;
    if (!gpre || t >= N * NC) break;
    const int c = t & (NC - 1);
    if (c < nca) {
      const T G = Gr[u];
      const Cx<T> X = D[t];
      const int kz = col0 + c;
      const double w = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
      e += w * (double)G * ((double)X.re * X.re + (double)X.im * X.im);
      D[t] = Cx<T>{G * X.re, G * X.im};
    }
  }
  for (int t = threadIdx.x; !gpre && t < N * NC; t += kFftxBlock) {
    const int p = t >> sh, c = t & (NC - 1);
    if (c < nca) {
      const int k = (int)(__brev((unsigned)p) >> (32 - logN));
      const T G = gtab[gbase + (long)k * gjstride + c];
      const Cx<T> X = D[t];
      const int kz = col0 + c;
      const double w = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
      e += w * (double)G * ((double)X.re * X.re + (double)X.im * X.im);
      D[t] = Cx<T>{G * X.re, G * X.im};
    }
  }
  __syncthreads();
  // inverse, decimation in time: spans 1, 2, .. N/2; W = exp(+2 pi i / N); again two stages (spans q, 2q) per round trip,
  // after the single plain stage of an odd log2 N (the mirror image of the forward order)
  s = 0;
  for (; s < (logN & 1); ++s) {
    const int half = 1 << s;
    for (int t = threadIdx.x; t < nbf; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NC - 1);
      const int k = b & (half - 1), i = ((b >> s) << (s + 1)) | k;
      const Cx<T> w = tw[k << (logN - 1 - s)];
      const Cx<T> u = D[(i << sh) + c], y = D[((i + half) << sh) + c];
      const Cx<T> v{y.re * w.re - y.im * w.im, y.re * w.im + y.im * w.re};                    // y w
      D[(i << sh) + c] = Cx<T>{u.re + v.re, u.im + v.im};
      D[((i + half) << sh) + c] = Cx<T>{u.re - v.re, u.im - v.im};
    }
    __syncthreads();
  }
  for (; s + 1 < logN; s += 2) {
    const int q = 1 << s;                           // spans q (stage s) and 2q (stage s + 1)
    for (int t = threadIdx.x; t < nbq; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NC - 1);
      const int k = b & (q - 1), i0 = ((b >> s) << (s + 2)) | k;
      const Cx<T> w1 = tw[k << (logN - 2 - s)], w2 = tw[k << (logN - 1 - s)];  // W_{4q}^k, W_{2q}^k
      Cx<T>* p = D + ((size_t)i0 << sh) + c;
      const int st = q << sh;
      const Cx<T> x0 = p[0], x2 = p[2 * st];
      const Cx<T> y1 = cmul(p[st], w2), y3 = cmul(p[3 * st], w2);
      const Cx<T> t0{x0.re + y1.re, x0.im + y1.im}, t1{x0.re - y1.re, x0.im - y1.im};
      const Cx<T> t2{x2.re + y3.re, x2.im + y3.im}, t3{x2.re - y3.re, x2.im - y3.im};
      const Cx<T> z2 = cmul(t2, w1), z3w = cmul(t3, w1);
      const Cx<T> z3{-z3w.im, z3w.re};                                            // times +i (conj W_{4q}^q)
      p[0] = Cx<T>{t0.re + z2.re, t0.im + z2.im};
      p[2 * st] = Cx<T>{t0.re - z2.re, t0.im - z2.im};
      p[st] = Cx<T>{t1.re + z3.re, t1.im + z3.im};
      p[3 * st] = Cx<T>{t1.re - z3.re, t1.im - z3.im};
    }
    __syncthreads();
  }
#pragma unroll 4
  for (int t = threadIdx.x; t < N * NC; t += kFftxBlock) {
    const int j = t >> sh, c = t & (NC - 1);
    if (c < nca) spec[base + (long)j * jstride + c] = D[t];
  }
  e = block_reduce_sum<kFftxBlock>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[slot], e);
}

// ---- typed dispersion (round 4): the x pass that also combines the type meshes (see disp_kernels.hip) -----------------------
// A workgroup holds NC columns of ALL NTP (padded: 1, 2 or 4) type spectra side by side: column t * NC + c.  The transforms
// along x treat them as NC * NTP independent columns; between them, at every (frequency, column):
//   S_p = sum_t c[p][t] X_t,   E += w G_p |S_p|^2,   X_t <- sum_p c[p][t] G_p S_p.
template <class T, int NTP>
__global__ __launch_bounds__(kFftxBlock) void k_fftx_mix(int N, int logN, int ncols, int nfix, int NC, long jstride, long fixstride,
                                                         long gjstride, long gfixstride, int K3, Cx<T>* __restrict__ spec,
                                                         DftTabs<T> tabs, MixTab mix, long spec_tstride,
                                                         const Cx<T>* __restrict__ twg, double* energies, int slot) {
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(fftx_smem);      // [N / 2]
  Cx<T>* D = tw + N / 2;                                 // [N][NC * NTP]
  const int NCT = NC * NTP;
  const int ntile = (ncols + NC - 1) / NC;
  const long L = xcd_block(blockIdx.x, (unsigned)(ntile * nfix));
  if (L < 0) return;
  const int col0 = (int)(L % ntile) * NC, nca = min(NC, ncols - col0);
  const long base = (L / ntile) * fixstride + col0;
  const long gbase = (L / ntile) * gfixstride + col0;
  const int sh = 31 - __clz(NCT), shc = 31 - __clz(NC);
  for (int t = threadIdx.x; t < N / 2; t += kFftxBlock) tw[t] = twg[t];
#pragma unroll 4
  for (int t = threadIdx.x; t < N * NCT; t += kFftxBlock) {
    const int j = t >> sh, cc = t & (NCT - 1), ty = cc >> shc, c = cc & (NC - 1);
    D[t] = (c < nca && ty < mix.nt) ? spec[(long)ty * spec_tstride + base + (long)j * jstride + c] : Cx<T>{T(0), T(0)};
  }
  // the G factors of the (frequency, column) points this thread combines: fetched now, behind the forward transform
  constexpr int kMixG = 8;                               // points per thread kept in registers: N * NC <= 8 * 256
  const bool gpre = N * NC <= kMixG * kFftxBlock;
  T Gr[3][kMixG];
#pragma unroll
  for (int u = 0; u < kMixG; ++u) {
    const int t = threadIdx.x + u * kFftxBlock;
    const int p = t >> shc, c = t & (NC - 1);
    const bool in = gpre && t < N * NC && c < nca;
    const long gi = gbase + (long)(__brev((unsigned)p) >> (32 - logN)) * gjstride + c;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) Gr[ch][u] = (in && ch < mix.nch) ? tabs.p[ch][gi] : T(0);
  }
  __syncthreads();
  const int nbf = (N / 2) * NCT, nbq = (N / 4) * NCT;
  auto cmul = [](Cx<T> a, Cx<T> b) { return Cx<T>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; };
  auto cmulc = [](Cx<T> a, Cx<T> b) { return Cx<T>{a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; };   // a conj(b)
  int s = logN - 1;
  for (; s >= 1; s -= 2) {                          // forward, two radix-2 stages per LDS round trip (k_fftx_conv)
    const int q = 1 << (s - 1);
    for (int t = threadIdx.x; t < nbq; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NCT - 1);
      const int k = b & (q - 1), i0 = ((b >> (s - 1)) << (s + 1)) | k;
      const Cx<T> w1 = tw[k << (logN - 1 - s)], w2 = tw[k << (logN - s)];
      Cx<T>* p = D + ((size_t)i0 << sh) + c;
      const int st = q << sh;
      const Cx<T> x0 = p[0], x1 = p[st], x2 = p[2 * st], x3 = p[3 * st];
      const Cx<T> t0{x0.re + x2.re, x0.im + x2.im}, t1{x1.re + x3.re, x1.im + x3.im};
      const Cx<T> t2 = cmulc(Cx<T>{x0.re - x2.re, x0.im - x2.im}, w1);
      const Cx<T> d13 = cmulc(Cx<T>{x1.re - x3.re, x1.im - x3.im}, w1);
      const Cx<T> t3{d13.im, -d13.re};
      p[0] = Cx<T>{t0.re + t1.re, t0.im + t1.im};
      p[st] = cmulc(Cx<T>{t0.re - t1.re, t0.im - t1.im}, w2);
      p[2 * st] = Cx<T>{t2.re + t3.re, t2.im + t3.im};
      p[3 * st] = cmulc(Cx<T>{t2.re - t3.re, t2.im - t3.im}, w2);
    }
    __syncthreads();
  }
  for (; s >= 0; --s) {
    const int half = 1 << s;
    for (int t = threadIdx.x; t < nbf; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NCT - 1);
      const int k = b & (half - 1), i = ((b >> s) << (s + 1)) | k;
      const Cx<T> w = tw[k << (logN - 1 - s)];
      const Cx<T> u = D[(i << sh) + c], v = D[((i + half) << sh) + c];
      const T dr = u.re - v.re, di = u.im - v.im;
      D[(i << sh) + c] = Cx<T>{u.re + v.re, u.im + v.im};
      D[((i + half) << sh) + c] = Cx<T>{dr * w.re + di * w.im, di * w.re - dr * w.im};
    }
    __syncthreads();
  }
  // combine the types at every (row p = frequency bitrev(p), column c): one thread per (p, c), all types of the point
  double e = 0.0;
  auto combine = [&](int t, T G0, T G1, T G2) {
    const int p = t >> shc, c = t & (NC - 1);
    Cx<T> X[NTP], Y[NTP];
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty) { X[ty] = D[((size_t)p << sh) + ty * NC + c]; Y[ty] = Cx<T>{T(0), T(0)}; }
    const int kz = col0 + c;
    const double w = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
    const T G[3] = {G0, G1, G2};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (ch >= mix.nch) break;
      Cx<T> S{T(0), T(0)};
#pragma unroll
      for (int ty = 0; ty < NTP; ++ty) { const T cc = (T)mix.c[ch][ty]; S.re += cc * X[ty].re; S.im += cc * X[ty].im; }
      e += w * (double)G[ch] * ((double)S.re * S.re + (double)S.im * S.im);
#pragma unroll
      for (int ty = 0; ty < NTP; ++ty) { const T cg = (T)mix.c[ch][ty] * G[ch]; Y[ty].re += cg * S.re; Y[ty].im += cg * S.im; }
    }
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty) D[((size_t)p << sh) + ty * NC + c] = Y[ty];
  };
  if (gpre) {
#pragma unroll
    for (int u = 0; u < kMixG; ++u) {
      const int t = threadIdx.x + u * kFftxBlock;
      if (t < N * NC && (t & (NC - 1)) < nca) combine(t, Gr[0][u], Gr[1][u], Gr[2][u]);
    }
  } else {
    for (int t = threadIdx.x; t < N * NC; t += kFftxBlock) {
      const int p = t >> shc, c = t & (NC - 1);
      if (c >= nca) continue;
      const long gi = gbase + (long)(__brev((unsigned)p) >> (32 - logN)) * gjstride + c;
      combine(t, tabs.p[0][gi], mix.nch > 1 ? tabs.p[1][gi] : T(0), mix.nch > 2 ? tabs.p[2][gi] : T(0));
    }
  }
  __syncthreads();
  s = 0;
  for (; s < (logN & 1); ++s) {                     // inverse (k_fftx_conv)
    const int half = 1 << s;
    for (int t = threadIdx.x; t < nbf; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NCT - 1);
      const int k = b & (half - 1), i = ((b >> s) << (s + 1)) | k;
      const Cx<T> w = tw[k << (logN - 1 - s)];
      const Cx<T> u = D[(i << sh) + c], y = D[((i + half) << sh) + c];
      const Cx<T> v{y.re * w.re - y.im * w.im, y.re * w.im + y.im * w.re};
      D[(i << sh) + c] = Cx<T>{u.re + v.re, u.im + v.im};
      D[((i + half) << sh) + c] = Cx<T>{u.re - v.re, u.im - v.im};
    }
    __syncthreads();
  }
  for (; s + 1 < logN; s += 2) {
    const int q = 1 << s;
    for (int t = threadIdx.x; t < nbq; t += kFftxBlock) {
      const int b = t >> sh, c = t & (NCT - 1);
      const int k = b & (q - 1), i0 = ((b >> s) << (s + 2)) | k;
      const Cx<T> w1 = tw[k << (logN - 2 - s)], w2 = tw[k << (logN - 1 - s)];
      Cx<T>* p = D + ((size_t)i0 << sh) + c;
      const int st = q << sh;
      const Cx<T> x0 = p[0], x2 = p[2 * st];
      const Cx<T> y1 = cmul(p[st], w2), y3 = cmul(p[3 * st], w2);
      const Cx<T> t0{x0.re + y1.re, x0.im + y1.im}, t1{x0.re - y1.re, x0.im - y1.im};
      const Cx<T> t2{x2.re + y3.re, x2.im + y3.im}, t3{x2.re - y3.re, x2.im - y3.im};
      const Cx<T> z2 = cmul(t2, w1), z3w = cmul(t3, w1);
      const Cx<T> z3{-z3w.im, z3w.re};
      p[0] = Cx<T>{t0.re + z2.re, t0.im + z2.im};
      p[2 * st] = Cx<T>{t0.re - z2.re, t0.im - z2.im};
      p[st] = Cx<T>{t1.re + z3.re, t1.im + z3.im};
      p[3 * st] = Cx<T>{t1.re - z3.re, t1.im - z3.im};
    }
    __syncthreads();
  }
#pragma unroll 4
  for (int t = threadIdx.x; t < N * NCT; t += kFftxBlock) {
    const int j = t >> sh, cc = t & (NCT - 1), ty = cc >> shc, c = cc & (NC - 1);
    if (c < nca && ty < mix.nt) spec[(long)ty * spec_tstride + base + (long)j * jstride + c] = D[t];
  }
  e = block_reduce_sum<kFftxBlock>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[slot], e);
}

template <class T>
void launch_fftx_mix(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, const MixTab& mix,
                     long spec_stride, double* energies, int slot, int khp) {
  const int N = K[0], Kh = K[2] / 2 + 1, ny = K[1];
  if (khp < Kh) khp = Kh;
  int logN = 0;
  while ((1 << logN) < N) ++logN;
  const int ntp = mix.nt <= 1 ? 1 : (mix.nt == 2 ? 2 : 4);
  int NC = (int)(128 / (2 * sizeof(T)));                 // one 128-B line of columns per type ...
  static const int lds_kb = [] { const char* e = getenv("ADMP_MIX_LDS_KB"); return e ? atoi(e) : 34; }();
  while (NC > 1 && sizeof(Cx<T>) * ((size_t)N / 2 + (size_t)N * NC * ntp) > (size_t)lds_kb * 1024) NC >>= 1;     // ... while the tile fits
  const size_t sh = sizeof(Cx<T>) * ((size_t)N / 2 + (size_t)N * NC * ntp);
  const int ntile = (Kh + NC - 1) / NC;
  const dim3 grid(xcd_grid((unsigned)(ntile * ny)));
#define MIX_LAUNCH(NTP)                                                                                                     \
  {                                                                                                                          \
    auto kern = k_fftx_mix<T, NTP>;                                                                                          \
    static size_t attr = 0;                                                                                                  \
    if (sh > 48 * 1024 && attr < sh) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = sh; } \
    kern<<<grid, kFftxBlock, sh, st>>>(N, logN, Kh, ny, NC, (long)ny * khp, (long)khp, (long)ny * Kh, (long)Kh, K[2],          \
                                       reinterpret_cast<Cx<T>*>(spec), tabs, mix, spec_stride / 2,                           \
                                       reinterpret_cast<const Cx<T>*>(tw), energies, slot);                                  \
  }
  if (ntp == 1) MIX_LAUNCH(1) else if (ntp == 2) MIX_LAUNCH(2) else MIX_LAUNCH(4)
#undef MIX_LAUNCH
}

bool fftx_usable(int N) { return N >= 32 && N <= 1024 && (N & (N - 1)) == 0; }

// spec = [K0][K1][khp] complex (khp >= K2/2+1: rows padded to whole 128-byte lines, engine.hip) after the batched 2-D r2c of
// the y-z planes; gtab = [K0][K1][K2/2+1]; tw = (cos, sin)(2 pi k / K0), k < K0 / 2
template <class T>
void launch_fftx_conv(hipStream_t st, const int K[3], const T* tw, T* spec, const T* gtab, double* energies, int slot, int khp,
                      int ny) {
  DftTabs<T> tabs;
  tabs.p[0] = gtab;
  launch_fftx_conv_batch<T>(st, K, tw, spec, tabs, 1, 0, energies, slot, khp, ny);
}
// nb channels: spec of channel b at spec + b * spec_stride (in reals), its table tabs.p[b]; the energies of all channels
// are summed into energies[slot]
template <class T>
void launch_fftx_conv_batch(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, int nb, long spec_stride,
                            double* energies, int slot, int khp, int ny) {
  const int N = K[0], Kh = K[2] / 2 + 1;
  if (khp < Kh) khp = Kh;
  if (ny <= 0) ny = K[1];
  int logN = 0;
  while ((1 << logN) < N) ++logN;
  int NC = (int)(128 / (2 * sizeof(T)));                 // one 128-B line of columns
  while (NC > 1 && sizeof(Cx<T>) * ((size_t)N / 2 + (size_t)N * NC) > 60 * 1024) NC >>= 1;
  const size_t sh = sizeof(Cx<T>) * ((size_t)N / 2 + (size_t)N * NC);
  const int ntile = (Kh + NC - 1) / NC;
  k_fftx_conv<T><<<dim3(xcd_grid((unsigned)(ntile * ny)), (unsigned)nb), kFftxBlock, sh, st>>>(
      N, logN, Kh, ny, NC, (long)ny * khp, (long)khp, (long)ny * Kh, (long)Kh, K[2], reinterpret_cast<Cx<T>*>(spec), tabs,
      spec_stride / 2, reinterpret_cast<const Cx<T>*>(tw), energies, slot);
}
#define INST(T)                                                                                                            \
  template void launch_fftx_mix<T>(hipStream_t, const int*, const T*, T*, const DftTabs<T>&, const MixTab&, long, double*,   \
                                   int, int);                                                                               \
  template void launch_fftx_conv<T>(hipStream_t, const int*, const T*, T*, const T*, double*, int, int, int);               \
  template void launch_fftx_conv_batch<T>(hipStream_t, const int*, const T*, T*, const DftTabs<T>&, int, long, double*, int, \
                                          int, int);
INST(float)
INST(double)
#undef INST

}  // namespace admp
