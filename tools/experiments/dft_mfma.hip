// Direct-DFT mesh convolution on the matrix cores (the five passes of dft_kernels.hip, same data layout, same results to
// round-off).  A line transform of a "hard" length N (97 for the reference's water_pol_1024 mesh, admp/pme.py:717-744) is
// a product with the H x H cosine and sine matrices of the pair sums (dft_math.h), H = (N-1)/2:
//     P_k = sum_jj cos(2 pi k (1+jj) / N) s_jj ,   R_k = sum_jj sin(2 pi k (1+jj) / N) d_jj ,   k = 1..H, jj = 0..H-1
//     X[k], X[N-k] = x_0 + P_k  +-  sgn i R_k
// -- GEMM-shaped work: [H x H] x [H x (lines)] twice per pass.  v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 tiles:
// one wavefront per 16 outputs k (tile mt), two data tiles per wavefront (real and imaginary parts of 16 complex lines,
// or 2 x 16 real lines in the z passes), K-loop over jj in steps of 4.  The vector form in dft_kernels.hip is bound by its
// LDS reads (one 32-B record + twiddle rotation per 8 flops); here a 16x16x4 tile needs one operand word per lane.
// k = 0 and (N even) k = N/2 are plain sums, done by a few lanes on the side.
// Status: validated against rocFFT (tests/test_gpu_parity.py::test_direct_dft_convolution_vs_rocfft) but NOT faster than
// the vector forms on MI355X for f64 -- see dftm_enabled() below; opt-in.
//
// Operand maps (cdna_hip_programming.md "Fragment layout"): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15];
// C/D column = lane & 15, row = (lane >> 4) + 4 r for f64, 4 (lane >> 4) + r for f32, r = 0..3.
#include "dft_math.h"
#include "launch.h"
#include "mfma.h"
#include "reduce.h"

namespace admp {

constexpr int kMfmaMaxTiles = 8;     // H <= 128

// phase probes for tools/ubench/dftm_phases.hip (100 MHz wall clock of thread 0 of every block); compiled out otherwise
#ifdef ADMP_DFTM_TRACE
__device__ long long* g_dftm_trace;
#define DFTM_TRACE(slot) do { if (threadIdx.x == 0) g_dftm_trace[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DFTM_TRACE(slot) do {} while (0)
#endif

extern __shared__ __align__(32) unsigned char dftm_smem[];


// ---- z lines (contiguous): real mesh [nlines][N] -> half spectrum [nlines][N/2+1].  32 lines per block; A = data, B = twiddle
template <class T>
__global__ __launch_bounds__(64 * kMfmaMaxTiles) void k_dftm_z_r2c(int N, int nlines, int RS, const T* __restrict__ mesh,
                                                                  Cx<T>* __restrict__ spec, const Cx<T>* __restrict__ twg,
                                                                  long mesh_stride, long spec_stride) {
  typedef typename Mfma<T>::Acc Acc;
  mesh += blockIdx.y * mesh_stride;
  spec += blockIdx.y * spec_stride;
  const int H = (N - 1) / 2, Kh = N / 2 + 1, nthr = blockDim.x;
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dftm_smem);   // [N]
  T* raw = reinterpret_cast<T*>(tw + N);              // [32][RS], RS = 2 mod 32: conflict-free operand reads
  const int line0 = blockIdx.x * 32;
  const int nl = min(32, nlines - line0);
  for (int t = threadIdx.x; t < N; t += nthr) tw[t] = twg[t];
  {
    const T* chunk = mesh + (long)line0 * N;
    const int nvalid = nl * N;
#pragma unroll 8
    for (int t = threadIdx.x; t < 32 * N; t += nthr) {        // independent loads: several in flight per thread
      const int l = t / N, j = t - l * N;
      raw[l * RS + j] = t < nvalid ? chunk[t] : T(0);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
  const int i = 16 * mt + lo;                       // this lane's twiddle column: output k = 1 + i
  TwIdx ti(i < H ? i : 0, hi, N);
  Acc P[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, R[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  // k = 0 and k = N/2 are plain (alternating) sums of the line: every thread sums a share, a few finish after the products
  T* zp = raw + 32 * RS;                       // [2][nthr / 32][32]
  {
    const int l = threadIdx.x & 31, part = threadIdx.x >> 5, np = nthr >> 5;
    const T* x = raw + l * RS;
    T s0 = T(0), s1 = T(0);
    for (int j = part; j < N; j += np) { s0 += x[j]; s1 += (j & 1) ? -x[j] : x[j]; }
    zp[part * 32 + l] = s0;
    zp[(np + part) * 32 + l] = s1;
  }
  // two operand sets in flight (see strided_mma)
  struct Ops { Cx<T> w; T a[2], b[2]; };
  const int KP = (H + 3) & ~3;
  auto fetch = [&](Ops& o, int kk) {
    o.w = tw[ti.m];
    ti.step();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const T* x = raw + (16 * dt + lo) * RS;
      o.a[dt] = kk < H ? x[1 + kk] : T(0);
      o.b[dt] = kk < H ? x[N - 1 - kk] : T(0);
    }
  };
  auto mul = [&](const Ops& o) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      P[dt] = Mfma<T>::mma(o.a[dt] + o.b[dt], o.w.re, P[dt]);
      R[dt] = Mfma<T>::mma(o.a[dt] - o.b[dt], o.w.im, R[dt]);
    }
  };
  {
    Ops A, B;
    fetch(A, hi);
    for (int kk = hi; kk < KP; kk += 8) {
      const bool two = kk + 4 < KP;
      if (two) fetch(B, kk + 4);
      mul(A);
      if (kk + 8 < KP) fetch(A, kk + 8);
      if (two) mul(B);
    }
  }
  __syncthreads();                             // zp complete
  const int k = 1 + i;
  if (k <= H) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int l = 16 * dt + Mfma<T>::row(lane, r);
        if (l < nl) {
          T re = raw[l * RS] + P[dt][r];
          if ((N & 1) == 0) re += (k & 1) ? -raw[l * RS + N / 2] : raw[l * RS + N / 2];
          spec[(long)(line0 + l) * Kh + k] = Cx<T>{re, -R[dt][r]};
        }
      }
  }
  if (threadIdx.x < 32 && (int)threadIdx.x < nl) {
    const int np = nthr >> 5;
    T s0 = T(0), s1 = T(0);
    for (int p = 0; p < np; ++p) { s0 += zp[p * 32 + threadIdx.x]; s1 += zp[(np + p) * 32 + threadIdx.x]; }
    spec[(long)(line0 + threadIdx.x) * Kh] = Cx<T>{s0, T(0)};
    if ((N & 1) == 0) spec[(long)(line0 + threadIdx.x) * Kh + N / 2] = Cx<T>{s1, T(0)};
  }
}

// ---- z lines back: half spectrum -> real mesh.  32 lines per block; A = data (X[k] re / im), B = twiddle
template <class T>
__global__ __launch_bounds__(64 * kMfmaMaxTiles) void k_dftm_z_c2r(int N, int nlines, int RC, const Cx<T>* __restrict__ spec,
                                                                  T* __restrict__ mesh, const Cx<T>* __restrict__ twg,
                                                                  long mesh_stride, long spec_stride) {
  typedef typename Mfma<T>::Acc Acc;
  mesh += blockIdx.y * mesh_stride;
  spec += blockIdx.y * spec_stride;
  const int H = (N - 1) / 2, Kh = N / 2 + 1, nthr = blockDim.x;
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dftm_smem);   // [N]
  Cx<T>* raw = tw + N;                                // [32][RC], RC odd
  const int line0 = blockIdx.x * 32;
  const int nl = min(32, nlines - line0);
  for (int t = threadIdx.x; t < N; t += nthr) tw[t] = twg[t];
  {
    const Cx<T>* chunk = spec + (long)line0 * Kh;
    const int nvalid = nl * Kh;
#pragma unroll 8
    for (int t = threadIdx.x; t < 32 * Kh; t += nthr) {
      const int l = t / Kh, k = t - l * Kh;
      raw[l * RC + k] = t < nvalid ? chunk[t] : Cx<T>{T(0), T(0)};
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
  const int i = 16 * mt + lo;                       // output position j = 1 + i
  TwIdx ti(i < H ? i : 0, hi, N);
  Acc P[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, R[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  // j = 0 and j = N/2 need the (alternating) sums of Re X[k], k = 1..H: shares here, finished after the products
  T* zp = reinterpret_cast<T*>(raw + 32 * RC);   // [2][nthr / 32][32]
  {
    const int l = threadIdx.x & 31, part = threadIdx.x >> 5, np = nthr >> 5;
    const Cx<T>* X = raw + l * RC;
    T s0 = T(0), s1 = T(0);
    for (int k = 1 + part; k <= H; k += np) { s0 += X[k].re; s1 += (k & 1) ? -X[k].re : X[k].re; }
    zp[part * 32 + l] = s0;
    zp[(np + part) * 32 + l] = s1;
  }
  struct Ops { Cx<T> w; Cx<T> v[2]; };
  const int KP = (H + 3) & ~3;
  auto fetch = [&](Ops& o, int kk) {
    o.w = tw[ti.m];
    ti.step();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) o.v[dt] = kk < H ? raw[(16 * dt + lo) * RC + 1 + kk] : Cx<T>{T(0), T(0)};
  };
  auto mul = [&](const Ops& o) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      P[dt] = Mfma<T>::mma(o.v[dt].re, o.w.re, P[dt]);
      R[dt] = Mfma<T>::mma(o.v[dt].im, o.w.im, R[dt]);
    }
  };
  {
    Ops A, B;
    fetch(A, hi);
    for (int kk = hi; kk < KP; kk += 8) {
      const bool two = kk + 4 < KP;
      if (two) fetch(B, kk + 4);
      mul(A);
      if (kk + 8 < KP) fetch(A, kk + 8);
      if (two) mul(B);
    }
  }
  __syncthreads();                               // zp complete
  const int j = 1 + i;
  if (j <= H) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int l = 16 * dt + Mfma<T>::row(lane, r);
        if (l < nl) {
          T base = raw[l * RC].re + T(2) * P[dt][r];
          if ((N & 1) == 0) base += (j & 1) ? -raw[l * RC + N / 2].re : raw[l * RC + N / 2].re;
          T* x = mesh + (long)(line0 + l) * N;
          x[j] = base - T(2) * R[dt][r];
          x[N - j] = base + T(2) * R[dt][r];
        }
      }
  }
  if (threadIdx.x < 32 && (int)threadIdx.x < nl) {
    const Cx<T>* X = raw + threadIdx.x * RC;
    const int np = nthr >> 5;
    T s0 = T(0), s1 = T(0);
    for (int p = 0; p < np; ++p) { s0 += zp[p * 32 + threadIdx.x]; s1 += zp[(np + p) * 32 + threadIdx.x]; }
    T* x = mesh + (long)(line0 + threadIdx.x) * N;
    T x0 = X[0].re + T(2) * s0, xh = X[0].re + T(2) * s1;
    if ((N & 1) == 0) {
      x0 += X[N / 2].re;
      xh += ((N / 2) & 1) ? -X[N / 2].re : X[N / 2].re;
      x[N / 2] = xh;
    }
    x[0] = x0;
  }
}

// Shared pieces of the strided complex passes: 8 complex lines per block (the 16 data columns of an MFMA tile are their
// real parts, then their imaginary parts: column lo = c + 8 comp), flattened line index L = fix * ncols + c, element j of
// line L at spec[(L / ncols) * fixstride + L % ncols + j * jstride].  One wavefront per 16 outputs k: 2 accumulators
// (P, R), 2 MFMAs per K-step; a lane ends up with ONE component of X[k] and of X[N-k] and takes the other accumulator's
// partner component from lane ^ 8.
// LDS: tw[N] | planes S, D [KP][16] | x0[16], xn[16] (by column lo) | part[2][blockDim / 8][16]
constexpr int kLinesPerTile = 8;
template <class T>
struct StridedTile {
  Cx<T>* tw;
  T* S;
  T* D;
  T* x0;
  T* xn;
  T* part;        // per 8-thread group partial column sums of S and of (-1)^(1+jj) S
  int KP;
  __device__ __forceinline__ void put_partials(int g, int lo, T a, T ar) const {
    const int ng = blockDim.x >> 3;
    part[g * 16 + lo] = a;
    part[(ng + g) * 16 + lo] = ar;
  }
};
template <class T>
__device__ __forceinline__ StridedTile<T> strided_tile(int N) {
  StridedTile<T> s;
  const int H = (N - 1) / 2;
  s.KP = (H + 3) & ~3;
  s.tw = reinterpret_cast<Cx<T>*>(dftm_smem);
  s.S = reinterpret_cast<T*>(s.tw + N);
  s.D = s.S + (size_t)s.KP * 16;
  s.x0 = s.D + (size_t)s.KP * 16;
  s.xn = s.x0 + 16;
  s.part = s.xn + 16;
  return s;
}
static size_t strided_tile_bytes(int N, size_t w, int nthreads) {   // w = sizeof(T)
  const int H = (N - 1) / 2, KP = (H + 3) & ~3;
  return 2 * w * N + 2 * w * KP * 16 + 2 * w * 16 + 2 * w * (nthreads / 8) * 16;
}

template <class T>
__device__ __forceinline__ void strided_load(int N, const StridedTile<T>& s, const Cx<T>* __restrict__ spec, int L0, int nlines,
                                             int ncols, long jstride, long fixstride, const Cx<T>* __restrict__ twg) {
  const int H = (N - 1) / 2, nthr = blockDim.x;
  for (int t = threadIdx.x; t < N; t += nthr) s.tw[t] = twg[t];
  // the thread's line is the same in every iteration (nthr is a multiple of 8): one division, and the loads of the
  // iterations are independent -- unrolled so that they are in flight together
  const int c = threadIdx.x & 7, Lc = L0 + c;
  const bool livec = Lc < nlines;
  const long basec = livec ? (long)(Lc / ncols) * fixstride + Lc % ncols : 0;
  T sa[2] = {T(0), T(0)}, sar[2] = {T(0), T(0)};     // this thread's share of the column sums (k = 0, N/2 outputs)
#pragma unroll 2
  for (int t = threadIdx.x; t < s.KP * 8; t += nthr) {
    const int jj = t >> 3;
    Cx<T> a{T(0), T(0)}, b{T(0), T(0)};
    if (jj < H && livec) {
      a = spec[basec + (long)(1 + jj) * jstride];
      b = spec[basec + (long)(N - 1 - jj) * jstride];
    }
    const T u = a.re + b.re, v = a.im + b.im;
    s.S[jj * 16 + c] = u; s.S[jj * 16 + 8 + c] = v;
    s.D[jj * 16 + c] = a.re - b.re; s.D[jj * 16 + 8 + c] = a.im - b.im;
    sa[0] += u; sa[1] += v;
    sar[0] += (jj & 1) ? u : -u;      // (-1)^(1+jj)
    sar[1] += (jj & 1) ? v : -v;
  }
  s.put_partials(threadIdx.x >> 3, c, sa[0], sar[0]);
  s.put_partials(threadIdx.x >> 3, 8 + c, sa[1], sar[1]);
  if (threadIdx.x < 8) {
    Cx<T> a{T(0), T(0)}, b{T(0), T(0)};
    if (livec) {
      a = spec[basec];
      if ((N & 1) == 0) b = spec[basec + (long)(N / 2) * jstride];
    }
    s.x0[c] = a.re; s.x0[8 + c] = a.im;
    s.xn[c] = b.re; s.xn[8 + c] = b.im;
  }
}

// the two matrix products of one wavefront (tile mt): A = twiddle, B = data planes
template <class T>
__device__ __forceinline__ void strided_mma(int N, const StridedTile<T>& s, int mt, typename Mfma<T>::Acc& P,
                                            typename Mfma<T>::Acc& R) {
  const int H = (N - 1) / 2, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int i = 16 * mt + lo;
  TwIdx ti(i < H ? i : 0, hi, N);
  const T* sp = s.S + lo; const T* dp = s.D + lo;
  // two operand sets in flight: the LDS reads of a step are issued before the products of the step before it
  struct Ops { Cx<T> w; T b[2]; };
  auto fetch = [&](Ops& o, int kk) {
    o.w = s.tw[ti.m];
    ti.step();
    o.b[0] = sp[kk * 16]; o.b[1] = dp[kk * 16];
  };
  auto mul = [&](const Ops& o) {
    P = Mfma<T>::mma(o.w.re, o.b[0], P);
    R = Mfma<T>::mma(o.w.im, o.b[1], R);
  };
  Ops A, B;
  fetch(A, hi);
  for (int kk = hi; kk < s.KP; kk += 8) {
    const bool two = kk + 4 < s.KP;           // (uniform: KP is a multiple of 4)
    if (two) fetch(B, kk + 4);
    mul(A);
    if (kk + 8 < s.KP) fetch(A, kk + 8);
    if (two) mul(B);
  }
}
// component lo of X[0] and (N even) X[N/2] from the partial column sums
template <class T>
__device__ __forceinline__ void strided_special(int N, const StridedTile<T>& s, int lo, T& X0, T& Xh) {
  const int ng = blockDim.x >> 3;
  T a = T(0), ar = T(0);
  for (int g = 0; g < ng; ++g) {       // (independent reads)
    a += s.part[g * 16 + lo];
    ar += s.part[(ng + g) * 16 + lo];
  }
  X0 = s.x0[lo] + a;
  Xh = T(0);
  if ((N & 1) == 0) {
    X0 += s.xn[lo];
    Xh = s.x0[lo] + (((N / 2) & 1) ? -s.xn[lo] : s.xn[lo]) + ar;
  }
}
// component lo (re of line lo, or im of line lo - 8) of the outputs k and N - k: X = x0 + P + sgn i R, i R = (-R.im, R.re)
template <class T, int SIGN>
__device__ __forceinline__ void strided_outputs(int N, const StridedTile<T>& s, int k, int lo, T p, T r_other, T& Xk, T& Xnk) {
  T base = s.x0[lo] + p;
  if ((N & 1) == 0) base += (k & 1) ? -s.xn[lo] : s.xn[lo];
  const T q = (lo & 8) ? T(SIGN) * r_other : -T(SIGN) * r_other;     // im: + sgn R.re ; re: - sgn R.im
  Xk = base + q;
  Xnk = base - q;
}

// ---- strided complex lines, in place (y lines)
template <class T, int SIGN>
__global__ __launch_bounds__(64 * kMfmaMaxTiles) void k_dftm_strided(int N, int nlines, int ncols, long jstride, long fixstride,
                                                                    Cx<T>* __restrict__ spec, const Cx<T>* __restrict__ twg,
                                                                    long spec_stride) {
  typedef typename Mfma<T>::Acc Acc;
  spec += blockIdx.y * spec_stride;
  const int H = (N - 1) / 2;
  const StridedTile<T> s = strided_tile<T>(N);
  const int L0 = blockIdx.x * kLinesPerTile;
  DFTM_TRACE(0);
  strided_load<T>(N, s, spec, L0, nlines, ncols, jstride, fixstride, twg);
  __syncthreads();
  DFTM_TRACE(1);
  const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6, lo = lane & 15;
  Acc P = {0, 0, 0, 0}, R = {0, 0, 0, 0};
  strided_mma<T>(N, s, mt, P, R);
  DFTM_TRACE(2);
  const int L = L0 + (lo & 7);
  T* out = reinterpret_cast<T*>(spec) + (lo >> 3);                 // this lane's component of the complex words
  const long base = L < nlines ? (long)(L / ncols) * fixstride + L % ncols : 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 1 + 16 * mt + Mfma<T>::row(lane, r);
    const T other = __shfl_xor(R[r], 8, 64);
    if (L < nlines && k <= H) {
      T Xk, Xnk;
      strided_outputs<T, SIGN>(N, s, k, lo, P[r], other, Xk, Xnk);
      out[2 * (base + (long)k * jstride)] = Xk;
      out[2 * (base + (long)(N - k) * jstride)] = Xnk;
    }
  }
  DFTM_TRACE(3);
  if (threadIdx.x < 16 && L < nlines) {     // (threads 0..15: lane = column lo)
    T X0, Xh;
    strided_special<T>(N, s, lo, X0, Xh);
    out[2 * base] = X0;
    if ((N & 1) == 0) out[2 * (base + (long)(N / 2) * jstride)] = Xh;
  }
  DFTM_TRACE(4);
}

// ---- x lines: forward, multiply by G (accumulating sum w G |S|^2, recip.py:400-414 / pme.py:240), inverse; in place.
// Lines: L = y * Kh + kz, contiguous in memory (fixstride = ncols = Kh), element j at spec[L + j * jstride].
template <class T>
__global__ __launch_bounds__(64 * kMfmaMaxTiles) void k_dftm_x_conv(int N, int nlines, int Kh, long jstride, int K3,
                                                                   Cx<T>* __restrict__ spec, DftTabs<T> tabs,
                                                                   const Cx<T>* __restrict__ twg, double* energies, int slot,
                                                                   long spec_stride) {
  typedef typename Mfma<T>::Acc Acc;
  spec += blockIdx.y * spec_stride;
  const T* __restrict__ gtab = tabs.p[blockIdx.y];
  const int H = (N - 1) / 2;
  const StridedTile<T> s = strided_tile<T>(N);
  const int L0 = blockIdx.x * kLinesPerTile;
  const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6, lo = lane & 15;
  const int L = L0 + (lo & 7);
  const bool live = L < nlines;
  T* out = reinterpret_cast<T*>(spec) + (lo >> 3);
  // G of this lane's outputs, fetched before the transform so that the latency hides behind it
  T Gk[4], Gnk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 1 + 16 * mt + Mfma<T>::row(lane, r);
    const bool ok = live && k <= H;
    Gk[r] = ok ? gtab[L + (long)k * jstride] : T(0);
    Gnk[r] = ok ? gtab[L + (long)(N - k) * jstride] : T(0);
  }
  T G0 = T(0), Gh = T(0);
  if (threadIdx.x < 16 && live) {
    G0 = gtab[L];
    if ((N & 1) == 0) Gh = gtab[L + (long)(N / 2) * jstride];
  }
  strided_load<T>(N, s, spec, L0, nlines, Kh, jstride, (long)Kh, twg);
  __syncthreads();
  Acc P = {0, 0, 0, 0}, R = {0, 0, 0, 0};
  strided_mma<T>(N, s, mt, P, R);
  const int kz = L % Kh;
  const double wz = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
  double e = 0.0;
  T Y0 = T(0), Yh = T(0);
  if (threadIdx.x < 16 && live) {
    T X0, Xh;
    strided_special<T>(N, s, lo, X0, Xh);
    e += wz * ((double)G0 * (double)X0 * X0 + (double)Gh * (double)Xh * Xh);
    Y0 = G0 * X0;
    Yh = Gh * Xh;
  }
  T vs[4], vd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 1 + 16 * mt + Mfma<T>::row(lane, r);
    const T other = __shfl_xor(R[r], 8, 64);
    T Xk, Xnk;
    strided_outputs<T, -1>(N, s, k, lo, P[r], other, Xk, Xnk);
    e += wz * ((double)Gk[r] * (double)Xk * Xk + (double)Gnk[r] * (double)Xnk * Xnk);    // (G is 0 on dead lanes)
    const T a = Gk[r] * Xk, b = Gnk[r] * Xnk;
    vs[r] = a + b; vd[r] = a - b;
  }
  __syncthreads();          // every wavefront is done reading the forward pair sums (and their partial column sums)
  {
    T sa = T(0), sar = T(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 1 + 16 * mt + Mfma<T>::row(lane, r);
      if (k <= H) {                       // position jj = k - 1
        s.S[(k - 1) * 16 + lo] = vs[r];
        s.D[(k - 1) * 16 + lo] = vd[r];
        sa += vs[r];
        sar += (k & 1) ? -vs[r] : vs[r];
      }
    }
    s.put_partials(4 * mt + (lane >> 4), lo, sa, sar);     // 4 MT groups write here; the rest of the 8 MT rows stay ...
    s.put_partials(4 * (blockDim.x >> 6) + 4 * mt + (lane >> 4), lo, T(0), T(0));       // ... zero
  }
  if (threadIdx.x < 16) { s.x0[lo] = Y0; s.xn[lo] = Yh; }
  __syncthreads();
  Acc Q = {0, 0, 0, 0}, Z = {0, 0, 0, 0};
  strided_mma<T>(N, s, mt, Q, Z);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 1 + 16 * mt + Mfma<T>::row(lane, r);
    const T other = __shfl_xor(Z[r], 8, 64);
    if (live && k <= H) {
      T Xk, Xnk;
      strided_outputs<T, +1>(N, s, k, lo, Q[r], other, Xk, Xnk);
      out[2 * (L + (long)k * jstride)] = Xk;
      out[2 * (L + (long)(N - k) * jstride)] = Xnk;
    }
  }
  if (threadIdx.x < 16 && live) {
    T X0, Xh;
    strided_special<T>(N, s, lo, X0, Xh);
    out[2 * (long)L] = X0;
    if ((N & 1) == 0) out[2 * (L + (long)(N / 2) * jstride)] = Xh;
  }
  // block sum of the energy: blockDim is a multiple of 64, not of kDftBlock -- wave sums, then one atomic per wavefront
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
  if (lane == 0 && e != 0.0) atomicAdd(&energies[slot], e);
}

// ---- host side
// Opt-in (ADMP_DFT_MFMA=1).  Measured on MI355X at 97^3 f64 (profiles/README.md, tools/ubench/mfma_f64_rate.hip,
// tools/ubench/dftm_phases.hip): v_mfma_f64_16x16x4_f64 issues every 45-61 ns per SIMD (= 34-46 TFLOP/s chip-wide, under
// the vector FMA rate), and a pass is latency-bound either way (launch + end-of-kernel write-back 5-6 us, loads 2.4 us,
// products 1.5-2.7 us, stores 0.8 us for ONE block per CU): z 15.4 / 11.6, y 13.6-14.1, x 24.7-33 us against 11.3 / 11.0,
// 13.3-14.0, 23.5 us of the vector forms in dft_kernels.hip -- which therefore stay the default.
bool dftm_enabled(int N) {
  static const bool on = [] { const char* e = getenv("ADMP_DFT_MFMA"); return e && atoi(e) == 1; }();
  const int H = (N - 1) / 2;
  return on && N >= 9 && (H + 15) / 16 <= kMfmaMaxTiles;
}
static int dftm_tiles(int N) { return ((N - 1) / 2 + 15) / 16; }

template <class T>
void launch_dftm_z(hipStream_t st, const int K[3], const T* tw, T* mesh, T* spec, int inverse, int nb, long mesh_stride,
                   long spec_stride) {
  const int N = K[2], nlines = K[0] * K[1], Kh = N / 2 + 1, MT = dftm_tiles(N);
  const Cx<T>* t2 = reinterpret_cast<const Cx<T>*>(tw) + K[0] + K[1];
  const dim3 grid((nlines + 31) / 32, nb);
  const long ss = spec_stride / 2;
  if (inverse) {
    const int RC = Kh | 1;
    const size_t sh = sizeof(Cx<T>) * (size_t)(N + 32 * RC) + sizeof(T) * (size_t)(2 * (2 * MT) * 32);
    k_dftm_z_c2r<T><<<grid, 64 * MT, sh, st>>>(N, nlines, RC, reinterpret_cast<const Cx<T>*>(spec), mesh, t2, mesh_stride, ss);
  } else {
    const int RS = ((N - 2 + 31) / 32) * 32 + 2;
    const size_t sh = sizeof(Cx<T>) * (size_t)N + sizeof(T) * (size_t)(32 * RS + 2 * (2 * MT) * 32);
    k_dftm_z_r2c<T><<<grid, 64 * MT, sh, st>>>(N, nlines, RS, mesh, reinterpret_cast<Cx<T>*>(spec), t2, mesh_stride, ss);
  }
}
template <class T>
void launch_dftm_y(hipStream_t st, const int K[3], const T* tw, T* spec, int inverse, int nb, long spec_stride) {
  const int N = K[1], Kh = K[2] / 2 + 1, MT = dftm_tiles(N);
  const Cx<T>* t1 = reinterpret_cast<const Cx<T>*>(tw) + K[0];
  const int nlines = K[0] * Kh;
  const dim3 grid((unsigned)((nlines + kLinesPerTile - 1) / kLinesPerTile), nb);
  const size_t sh = strided_tile_bytes(N, sizeof(T), 64 * MT);
  Cx<T>* sp = reinterpret_cast<Cx<T>*>(spec);
  if (inverse)
    k_dftm_strided<T, +1><<<grid, 64 * MT, sh, st>>>(N, nlines, Kh, (long)Kh, (long)K[1] * Kh, sp, t1, spec_stride / 2);
  else
    k_dftm_strided<T, -1><<<grid, 64 * MT, sh, st>>>(N, nlines, Kh, (long)Kh, (long)K[1] * Kh, sp, t1, spec_stride / 2);
}
template <class T>
void launch_dftm_x_conv(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, double* energies,
                        int slot, int nb, long spec_stride) {
  const int N = K[0], Kh = K[2] / 2 + 1, MT = dftm_tiles(N);
  const int nlines = K[1] * Kh;
  const dim3 grid((unsigned)((nlines + kLinesPerTile - 1) / kLinesPerTile), nb);
  const size_t sh = strided_tile_bytes(N, sizeof(T), 64 * MT);
  k_dftm_x_conv<T><<<grid, 64 * MT, sh, st>>>(N, nlines, Kh, (long)K[1] * Kh, K[2], reinterpret_cast<Cx<T>*>(spec), tabs,
                                              reinterpret_cast<const Cx<T>*>(tw), energies, slot, spec_stride / 2);
}
#define INST(T)                                                                                    \
  template void launch_dftm_z<T>(hipStream_t, const int*, const T*, T*, T*, int, int, long, long); \
  template void launch_dftm_y<T>(hipStream_t, const int*, const T*, T*, int, int, long);           \
  template void launch_dftm_x_conv<T>(hipStream_t, const int*, const T*, T*, const DftTabs<T>&, double*, int, int, long);
INST(float)
INST(double)
#undef INST

}  // namespace admp
