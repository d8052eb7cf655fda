// Two-level direct-DFT mesh convolution for mesh dimensions that rocFFT can only do with Bluestein's algorithm AND that are
// too long for the plain O(N^2) lines of dft_kernels.hip -- e.g. the 305 = 5 * 61 mesh the reference's
// setup_ewald_parameters (admp/pme.py:146-172) gives for the 98 304-atom water box.
//
// Good-Thomas (prime-factor) split of every axis: N = N1 * N2 with gcd(N1, N2) = 1, N2 = the power of the largest prime in
// N (the "hard" part, <= 160), N1 the smooth cofactor (<= 32).  With the index maps
//     position  n  <->  (n1, n2):  n = (N2 n1 + N1 n2) mod N           frequency  k  <->  (k1, k2):  k = k1 (mod N1), k = k2 (mod N2)
// the length-N transform IS the two-dimensional N1 x N2 transform -- no twiddles between the stages:
//     X(k1, k2) = sum_n1 W_N1^(n1 k1)  sum_n2 W_N2^(n2 k2)  x(n1, n2)
// Stage A: N2-point lines (the pair-symmetric direct sums of dft_math.h), stage B: N1-point lines (plain sums), both on a
// tile of lines held in LDS, one kernel per mesh axis like dft_kernels.hip:
//     z lines r2c -> y lines -> x lines forward * G (+ energy) x lines inverse -> y lines -> z lines c2r
// The spectrum is stored in SLOT order: X(k1, k2) of an axis sits where x(n1 = k1, n2 = k2) sat, i.e. at position
// (N2 k1 + N1 k2) mod N.  Nothing ever needs the natural order: the G table is generated in slot order (launch_gtab with
// the frequency-of-slot maps), the inverse transform reads slots and writes positions.
// z axis (real input): stage A is a real N2-point transform (outputs k2 = 0 .. N2/2), stage B complex over n1; the stored
// half is the N1 x (N2/2 + 1) set of (k1, k2 <= N2/2), column cz = k2 * N1 + k1 -- Khp = N1 (N2/2 + 1) columns instead of
// N/2 + 1.  The Hermitian partner of (k1, k2) is (-k1, -k2): inside the set only for k2 = 0 (and k2 = N2/2, N2 even), whose
// columns therefore weigh 1/2 in the energy sum -- the role kz = 0 / Nyquist play in the natural layout (recip.py:400-414).
#include <type_traits>

#include "dft_math.h"
#include "launch.h"
#include "mfma.h"
#include "reduce.h"

namespace admp {

constexpr int kPfaBlock = 256;
extern __shared__ __align__(32) unsigned char pfa_smem[];

__host__ __device__ inline int pfa_pos(const PfaAxis& a, int n1, int n2) { return (a.N2 * n1 + a.N1 * n2) % a.N; }

// sum_n1 y[n1 * stride] w1^(sign n1 k1), k1 fixed: the short stage (N1 <= 32)
template <class T>
__device__ __forceinline__ Cx<T> short_dft(int N1, int k1, int sign, const Cx<T>* y, int stride, const Cx<T>* tw1) {
  Cx<T> acc{T(0), T(0)};
  int m = 0;
#pragma unroll 4
  for (int n1 = 0; n1 < N1; ++n1) {
    const Cx<T> v = y[n1 * stride];
    const T c = tw1[m].re, s = sign < 0 ? -tw1[m].im : tw1[m].im;
    acc.re += v.re * c - v.im * s;
    acc.im += v.re * s + v.im * c;
    m += k1;
    if (m >= N1) m -= N1;
  }
  return acc;
}

// The short stage for a whole column at once, N1 known at compile time: the N1 inputs and the N1 twiddles are read once
// (not once per output), the index arithmetic folds away.  out[k1] = sum_n1 y[n1 * stride] w1^(sign n1 k1).
template <class T, int N1>
__device__ __forceinline__ void short_dft_col(int sign, const Cx<T>* y, int stride, const Cx<T>* tw1, Cx<T> (&out)[N1]) {
  Cx<T> v[N1], w[N1];
#pragma unroll
  for (int n = 0; n < N1; ++n) {
    v[n] = y[n * stride];
    w[n] = Cx<T>{tw1[n].re, sign < 0 ? -tw1[n].im : tw1[n].im};
  }
#pragma unroll
  for (int k = 0; k < N1; ++k) {
    Cx<T> acc = v[0];
#pragma unroll
    for (int n = 1; n < N1; ++n) {
      const int m = (n * k) % N1;                    // compile time after unrolling
      acc.re += v[n].re * w[m].re - v[n].im * w[m].im;
      acc.im += v[n].re * w[m].im + v[n].im * w[m].re;
    }
    out[k] = acc;
  }
}
// run `body(std::integral_constant<int, N1>)` for the compiled short lengths; false: not one of them (generic loops instead)
template <class F>
__device__ __forceinline__ bool pfa_n1_dispatch(int N1, F&& body) {
  switch (N1) {
    case 1: body(std::integral_constant<int, 1>()); return true;
    case 2: body(std::integral_constant<int, 2>()); return true;
    case 3: body(std::integral_constant<int, 3>()); return true;
    case 4: body(std::integral_constant<int, 4>()); return true;
    case 5: body(std::integral_constant<int, 5>()); return true;
    case 6: body(std::integral_constant<int, 6>()); return true;
    case 8: body(std::integral_constant<int, 8>()); return true;
    default: return false;
  }
}

// real line, rows layout (x[j * stride], j = 0..N-1): X[k] = x0 + P - i R (+ xn (-1)^k)
template <class T>
__device__ __forceinline__ Cx<T> rdft_rows(int N, int k, int stride, const T* x, const Cx<T>* tw) {
  const int H = (N - 1) / 2;
  T P = T(0), R = T(0);
  int m = k;
  for (int j = 1; j <= H; ++j) {
    const T a = x[j * stride], b = x[(N - j) * stride];
    P += (a + b) * tw[m].re;
    R += (a - b) * tw[m].im;
    m += k;
    if (m >= N) m -= N;
  }
  T re = x[0] + P;
  if ((N & 1) == 0) re += (k & 1) ? -x[(N / 2) * stride] : x[(N / 2) * stride];
  return Cx<T>{re, -R};
}
// Hermitian half line, rows layout (X[k * stride], k = 0 .. N/2): x_j and x_{N-j}
template <class T>
__device__ __forceinline__ void irdft_rows(int N, int j, int stride, const Cx<T>* X, const Cx<T>* tw, T& xj, T& xnj) {
  const int H = (N - 1) / 2;
  T P = T(0), R = T(0);
  int m = j;
  for (int k = 1; k <= H; ++k) {
    const Cx<T> v = X[k * stride];
    P += v.re * tw[m].re;
    R += v.im * tw[m].im;
    m += j;
    if (m >= N) m -= N;
  }
  T base = X[0].re + T(2) * P;
  if ((N & 1) == 0) base += (j & 1) ? -X[(N / 2) * stride].re : X[(N / 2) * stride].re;
  xj = base - T(2) * R;
  xnj = base + T(2) * R;
}

// LDS layout of the strided passes: tw2[N2] | tw1[N1] | D[N][NC] | Y[N][NC]   (row index = n1 * N2 + n2, resp. k1 * N2 + k2)
template <class T>
struct PfaTile {
  Cx<T>* tw2; Cx<T>* tw1; Cx<T>* D; Cx<T>* Y;
  int* ptab;        // [N]: row idx = n1 * N2 + n2 -> position (N2 n1 + N1 n2) mod N in the low 16 bits, n1 above
};
// the index table spares every element of every sweep two integer divisions and a modulo; it is built once on the host
// (pfa_index_table) -- computing it per workgroup cost ~300 VALU instructions per thread, a sixth of a kernel's issue time
__device__ __forceinline__ void pfa_fill_ptab(const PfaAxis& a, int* ptab, const int* __restrict__ ptabg) {
  for (int idx = threadIdx.x; idx < a.N; idx += kPfaBlock) ptab[idx] = ptabg[idx];
}
template <class T>
__device__ __forceinline__ PfaTile<T> pfa_tile(const PfaAxis& a, int NC) {
  PfaTile<T> t;
  t.tw2 = reinterpret_cast<Cx<T>*>(pfa_smem);
  t.tw1 = t.tw2 + a.N2;
  t.D = t.tw1 + a.N1;
  t.Y = t.D + (size_t)a.N * NC;
  t.ptab = reinterpret_cast<int*>(t.Y + (size_t)a.N * NC);
  return t;
}
static size_t pfa_tile_bytes(const PfaAxis& a, int NC, size_t w) {
  return 2 * w * ((size_t)a.N2 + a.N1 + 2 * (size_t)a.N * NC) + sizeof(int) * (size_t)a.N;
}
// columns per block: as many as fit the budget, at most 16 (64-B segments of f32 complex at 8)
static size_t pfa_lds_budget() {
  static const size_t b = [] { const char* e = getenv("ADMP_PFA_LDS_KB"); return (size_t)(e ? atoi(e) : 60) * 1024; }();
  return b;
}
static int pfa_cols(const PfaAxis& a, size_t w) {
  int nc = 8;        // 8 complex columns = the 16 words of an MFMA tile row (pfa_stage_a_mfma); 16 measured slower (occupancy)
  while (nc > 1 && pfa_tile_bytes(a, nc, w) > pfa_lds_budget()) nc >>= 1;
  return nc;
}

template <class T>
__device__ __forceinline__ void pfa_load(const PfaAxis& a, const PfaTile<T>& s, int NC, int nca, const Cx<T>* __restrict__ spec,
                                         long base, long jstride, const Cx<T>* __restrict__ tw2g, const Cx<T>* __restrict__ tw1g,
                                         const int* __restrict__ ptabg) {
  for (int t = threadIdx.x; t < a.N2; t += kPfaBlock) s.tw2[t] = tw2g[t];
  for (int t = threadIdx.x; t < a.N1; t += kPfaBlock) s.tw1[t] = tw1g[t];
  pfa_fill_ptab(a, s.ptab, ptabg);
  __syncthreads();
  const int sh = 31 - __clz(NC);                         // NC is a power of two
#pragma unroll 4
  for (int t = threadIdx.x; t < a.N * NC; t += kPfaBlock) {
    const int idx = t >> sh, c = t & (NC - 1);
    s.D[t] = c < nca ? spec[base + (long)(s.ptab[idx] & 0xffff) * jstride + c] : Cx<T>{T(0), T(0)};
  }
}
// stage A: N2-point lines over n2 for every (n1, column): in[(n1 N2 + n2) NC + c] -> out[(n1 N2 + k2) NC + c]
template <class T, int SIGN>
__device__ __forceinline__ void pfa_stage_a(const PfaAxis& a, int NC, const Cx<T>* in, Cx<T>* out, const Cx<T>* tw2) {
  constexpr int KQ = 2;
  const int Kh2 = a.N2 / 2 + 1, TK = (Kh2 + KQ - 1) / KQ;
  for (int t = threadIdx.x; t < a.N1 * TK * NC; t += kPfaBlock) {
    const int c = t % NC, g = (t / NC) % TK, n1 = t / (NC * TK);
    int k[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK < Kh2) ? g + q * TK : 0;
    Cx<T> Xk[KQ], Xnk[KQ];
    {   // (dft_pair_outputs_rows of dft_math.h wants the pair sums already formed in the rows; these rows hold the values)
      const Cx<T>* col = in + (size_t)n1 * a.N2 * NC + c;
      const int N2 = a.N2;
      const Cx<T> x0 = col[0];
      const Cx<T> xn = (N2 & 1) ? Cx<T>{T(0), T(0)} : col[(N2 / 2) * NC];
      dft_pair_core<T, SIGN, KQ>(N2, k, [=](int j) {
        const Cx<T> u = col[(1 + j) * NC], v = col[(N2 - 1 - j) * NC];
        return PairCx<T>{u.re + v.re, u.im + v.im, u.re - v.re, u.im - v.im};
      }, x0, xn, tw2, Xk, Xnk);
    }
    Cx<T>* o = out + (size_t)n1 * a.N2 * NC + c;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK;
      if (kq < Kh2) {
        o[kq * NC] = Xk[q];
        if (kq != 0 && 2 * kq != a.N2) o[(a.N2 - kq) * NC] = Xnk[q];
      }
    }
  }
}

// Stage A on the matrix cores (N2 odd, NC = 8): the N2-point sums of a sub-line are products with the H x H cosine / sine
// matrices of its pair sums (dft_mfma.hip has the derivation), H = (N2 - 1) / 2.  With 8 complex columns per tile the 16
// data columns of one MFMA tile are exactly the 16 words (re, im of 8 columns) of ONE row of sub-line n1 -- contiguous in
// LDS, conflict-free.  Work items (16 outputs k, sub-line n1) are dealt round-robin to the four wavefronts; a lane ends
// up with one component of X[k] and X[N2-k] and takes the partner component of the sine product from lane ^ 1.
// f32: v_mfma_f32_16x16x4_f32 issues every 20 ns per SIMD (tools/ubench/mfma_f64_rate.hip) -- the vector form of stage A
// (dft_pair_core) was 3x over the memory time of a pass at 305^3.
template <class T, int SIGN>
__device__ __forceinline__ void pfa_stage_a_mfma(const PfaAxis& a, const Cx<T>* in, Cx<T>* out, const Cx<T>* tw2) {
  typedef typename Mfma<T>::Acc Acc;
  const int N2 = a.N2, H = (N2 - 1) / 2, MT = (H + 15) / 16, KP = (H + 3) & ~3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
  const T* inr = reinterpret_cast<const T*>(in);
  T* outr = reinterpret_cast<T*>(out);
  for (int item = wave; item < MT * a.N1; item += kPfaBlock / 64) {
    const int mt = item % MT, n1 = item / MT;
    const int i = 16 * mt + lo;
    TwIdx ti(i < H ? i : 0, hi, N2);
    const T* col = inr + (size_t)n1 * N2 * 16 + lo;          // word lo of row n2 at col[n2 * 16]
    Acc P = {0, 0, 0, 0}, R = {0, 0, 0, 0};
    // two operand sets in flight: the LDS reads of a step are issued before the products of the step before it
    struct Ops { Cx<T> w; T u, v; };
    auto fetch = [&](Ops& o, int kk) {
      o.w = tw2[ti.m];
      ti.step();
      o.u = kk < H ? col[(1 + kk) * 16] : T(0);
      o.v = kk < H ? col[(N2 - 1 - kk) * 16] : T(0);
    };
    auto mul = [&](const Ops& o) {
      P = Mfma<T>::mma(o.w.re, o.u + o.v, P);
      R = Mfma<T>::mma(o.w.im, o.u - o.v, R);
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
    const T x0 = col[0];
    T* o = outr + (size_t)n1 * N2 * 16 + lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 1 + 16 * mt + Mfma<T>::row(lane, r);
      const T other = __shfl_xor(R[r], 1, 64);               // sgn i R = sgn (-R.im, R.re)
      if (k <= H) {
        const T base = x0 + P[r];
        const T q = (lo & 1) ? T(SIGN) * other : -T(SIGN) * other;
        o[k * 16] = base + q;
        o[(N2 - k) * 16] = base - q;
      }
    }
  }
  for (int t = threadIdx.x; t < a.N1 * 16; t += kPfaBlock) {   // k = 0: plain sums
    const int n1 = t >> 4, w = t & 15;
    const T* col = inr + (size_t)n1 * N2 * 16 + w;
    T sum = T(0);
#pragma unroll 8
    for (int n2 = 0; n2 < N2; ++n2) sum += col[n2 * 16];
    outr[(size_t)n1 * N2 * 16 + w] = sum;
  }
}
static bool pfa_use_mfma(const PfaAxis& a, int NC) {
  static const bool off = [] { const char* e = getenv("ADMP_PFA_MFMA"); return e && atoi(e) == 0; }();
  return !off && NC == 8 && (a.N2 & 1) && (a.N2 - 1) / 2 <= 128;
}
template <class T, int SIGN>
__device__ __forceinline__ void pfa_stage_a_any(const PfaAxis& a, int NC, int mfma, const Cx<T>* in, Cx<T>* out, const Cx<T>* tw2) {
  if (mfma) pfa_stage_a_mfma<T, SIGN>(a, in, out, tw2);
  else pfa_stage_a<T, SIGN>(a, NC, in, out, tw2);
}

// ---- strided complex lines, in place (y lines)
template <class T, int SIGN>
__global__ __launch_bounds__(kPfaBlock) void k_pfa_strided(PfaAxis a, int ncols, int nfix, int NC, int mfma, long jstride, long fixstride,
                                                          Cx<T>* __restrict__ spec, const Cx<T>* __restrict__ tw2g,
                                                          const Cx<T>* __restrict__ tw1g, const int* __restrict__ ptabg,
                                                          long spec_stride) {
  spec += blockIdx.y * spec_stride;
  const PfaTile<T> s = pfa_tile<T>(a, NC);
  // tiles of one row of columns share their 128-B lines (8 f32 complex columns = 64 B): the XCD remap puts neighbours on
  // the same XCD, i.e. behind the same L2
  const int ntile = (ncols + NC - 1) / NC;
  const long L = xcd_block(blockIdx.x, (unsigned)(ntile * nfix));
  if (L < 0) return;
  const int col0 = (int)(L % ntile) * NC, nca = min(NC, ncols - col0);
  const long base = (L / ntile) * fixstride + col0;
  pfa_load<T>(a, s, NC, nca, spec, base, jstride, tw2g, tw1g, ptabg);
  __syncthreads();
  pfa_stage_a_any<T, SIGN>(a, NC, mfma, s.D, s.Y, s.tw2);
  __syncthreads();
  // stage B straight to memory: X(k1, k2) -> slot (N2 k1 + N1 k2) mod N
  const int sh = 31 - __clz(NC);
  const bool done = pfa_n1_dispatch(a.N1, [&](auto n1c) {
    constexpr int N1 = decltype(n1c)::value;
    for (int t = threadIdx.x; t < a.N2 * NC; t += kPfaBlock) {       // one column (k2, c) per task
      const int k2 = t >> sh, c = t & (NC - 1);
      if (c >= nca) continue;
      Cx<T> X[N1];
      short_dft_col<T, N1>(SIGN, s.Y + (size_t)k2 * NC + c, a.N2 * NC, s.tw1, X);
#pragma unroll
      for (int k1 = 0; k1 < N1; ++k1) spec[base + (long)(s.ptab[k1 * a.N2 + k2] & 0xffff) * jstride + c] = X[k1];
    }
  });
  if (done) return;
  for (int t = threadIdx.x; t < a.N * NC; t += kPfaBlock) {
    const int idx = t >> sh, c = t & (NC - 1);
    const int pk = s.ptab[idx], k1 = pk >> 16, k2 = idx - k1 * a.N2;
    if (c < nca) spec[base + (long)(pk & 0xffff) * jstride + c] = short_dft<T>(a.N1, k1, SIGN, s.Y + (size_t)k2 * NC + c, a.N2 * NC, s.tw1);
  }
}

// ---- x lines: forward, multiply by G (accumulating sum w G |S|^2), inverse; in place.  az = the z axis (column weights)
template <class T>
__global__ __launch_bounds__(kPfaBlock) void k_pfa_x_conv(PfaAxis a, PfaAxis az, int ncols, int nfix, int NC, int mfma, long jstride, long fixstride,
                                                         Cx<T>* __restrict__ spec, DftTabs<T> tabs, const Cx<T>* __restrict__ tw2g,
                                                         const Cx<T>* __restrict__ tw1g, const int* __restrict__ ptabg,
                                                         double* energies, int slot, long spec_stride) {
  spec += blockIdx.y * spec_stride;
  const T* __restrict__ gtab = tabs.p[blockIdx.y];
  const PfaTile<T> s = pfa_tile<T>(a, NC);
  const int ntile = (ncols + NC - 1) / NC;
  const long L = xcd_block(blockIdx.x, (unsigned)(ntile * nfix));   // (see k_pfa_strided)
  if (L < 0) return;
  const int col0 = (int)(L % ntile) * NC, nca = min(NC, ncols - col0);
  const long base = (L / ntile) * fixstride + col0;
  pfa_load<T>(a, s, NC, nca, spec, base, jstride, tw2g, tw1g, ptabg);
  // the G factors this thread applies in stage B: fetched now (the index table is complete after pfa_load's barrier), so
  // that their latency hides behind stage A
  constexpr int kGMax = 16;
  const int sh = 31 - __clz(NC);
  const bool n1_compiled = a.N1 <= 6 || a.N1 == 8;      // pfa_n1_dispatch: those read G inside the fused short stage
  const bool gpre = !n1_compiled && a.N * NC <= kGMax * kPfaBlock;
  T Gr[kGMax];
#pragma unroll
  for (int u = 0; u < kGMax; ++u) {
    const int t = threadIdx.x + u * kPfaBlock;
    const int idx = t >> sh, c = t & (NC - 1);
    Gr[u] = (gpre && t < a.N * NC && c < nca) ? gtab[base + (long)(s.ptab[idx] & 0xffff) * jstride + c] : T(0);
  }
  __syncthreads();
  pfa_stage_a_any<T, -1>(a, NC, mfma, s.D, s.Y, s.tw2);
  __syncthreads();
  double e = 0.0;
  auto stage_b_g = [&](int t, T Gpre, bool have) {                    // stage B, times G: Y -> D
    const int idx = t >> sh, c = t & (NC - 1);
    const int pk = s.ptab[idx], k1 = pk >> 16, k2 = idx - k1 * a.N2;
    Cx<T> X{T(0), T(0)};
    if (c < nca) {
      const T G = have ? Gpre : gtab[base + (long)(pk & 0xffff) * jstride + c];
      X = short_dft<T>(a.N1, k1, -1, s.Y + (size_t)k2 * NC + c, a.N2 * NC, s.tw1);
      const int k2z = (col0 + c) / az.N1;
      const double w = (k2z == 0 || ((az.N2 & 1) == 0 && k2z == az.N2 / 2)) ? 0.5 : 1.0;
      e += w * (double)G * ((double)X.re * X.re + (double)X.im * X.im);
      X.re *= G; X.im *= G;
    }
    s.D[t] = X;
  };
  // compiled short lengths: a column (k2, c) per task -- forward short stage, G, inverse short stage all in registers
  // (Y -> Y; the LDS round trip through D and one barrier fall away)
  const bool fused_b = pfa_n1_dispatch(a.N1, [&](auto n1c) {
    constexpr int N1 = decltype(n1c)::value;
    for (int t = threadIdx.x; t < a.N2 * NC; t += kPfaBlock) {
      const int k2 = t >> sh, c = t & (NC - 1);
      if (c >= nca) continue;
      Cx<T> X[N1], Z[N1];
      short_dft_col<T, N1>(-1, s.Y + (size_t)k2 * NC + c, a.N2 * NC, s.tw1, X);
      const int k2z = (col0 + c) / az.N1;
      const double w = (k2z == 0 || ((az.N2 & 1) == 0 && k2z == az.N2 / 2)) ? 0.5 : 1.0;
#pragma unroll
      for (int k1 = 0; k1 < N1; ++k1) {
        const T G = gtab[base + (long)(s.ptab[k1 * a.N2 + k2] & 0xffff) * jstride + c];
        e += w * (double)G * ((double)X[k1].re * X[k1].re + (double)X[k1].im * X[k1].im);
        X[k1].re *= G; X[k1].im *= G;
      }
      short_dft_col<T, N1>(+1, X, 1, s.tw1, Z);
#pragma unroll
      for (int n1 = 0; n1 < N1; ++n1) s.Y[((size_t)n1 * a.N2 + k2) * NC + c] = Z[n1];
    }
  });
  if (!fused_b) {
    if (gpre) {
#pragma unroll
      for (int u = 0; u < kGMax; ++u) {
        const int t = threadIdx.x + u * kPfaBlock;
        if (t < a.N * NC) stage_b_g(t, Gr[u], true);
      }
    } else {
      for (int t = threadIdx.x; t < a.N * NC; t += kPfaBlock) stage_b_g(t, T(0), false);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < a.N * NC; t += kPfaBlock) {          // inverse stage B: D(k1, k2) -> Y(n1, k2)
      const int idx = t >> sh, c = t & (NC - 1);
      const int n1 = s.ptab[idx] >> 16, k2 = idx - n1 * a.N2;
      s.Y[t] = short_dft<T>(a.N1, n1, +1, s.D + (size_t)k2 * NC + c, a.N2 * NC, s.tw1);
    }
  }
  __syncthreads();
  pfa_stage_a_any<T, +1>(a, NC, mfma, s.Y, s.D, s.tw2);              // inverse stage A: Y(n1, k2) -> D(n1, n2)
  __syncthreads();
  for (int t = threadIdx.x; t < a.N * NC; t += kPfaBlock) {
    const int idx = t >> sh, c = t & (NC - 1);
    if (c < nca) spec[base + (long)(s.ptab[idx] & 0xffff) * jstride + c] = s.D[t];
  }
  e = block_reduce_sum<kPfaBlock>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[slot], e);
}

// ---- z lines (contiguous, real): mesh [nlines][N] -> spec [nlines][Khp], column cz = k2 * N1 + k1, k2 <= N2/2.
// LDS: tw2[N2] | tw1[N1] | Dr[N][NL] (reals) | Y[N1 * Kh2][NL]
template <class T>
__global__ __launch_bounds__(kPfaBlock) void k_pfa_z_r2c(PfaAxis a, int nlines, int NL, int mfma, const T* __restrict__ mesh,
                                                        Cx<T>* __restrict__ spec, const Cx<T>* __restrict__ tw2g,
                                                        const Cx<T>* __restrict__ tw1g, const int* __restrict__ ptabg,
                                                        long mesh_stride, long spec_stride) {
  mesh += blockIdx.y * mesh_stride;
  spec += blockIdx.y * spec_stride;
  const int Kh2 = a.N2 / 2 + 1, Khp = a.N1 * Kh2;
  Cx<T>* tw2 = reinterpret_cast<Cx<T>*>(pfa_smem);
  Cx<T>* tw1 = tw2 + a.N2;
  Cx<T>* Y = tw1 + a.N1;                                   // [N1 * Kh2][NL]
  T* Dr = reinterpret_cast<T*>(Y + (size_t)Khp * NL);      // [N][NL]
  int* ptab = reinterpret_cast<int*>(Dr + (size_t)a.N * NL);   // [N] (pfa_fill_ptab)
  const int line0 = blockIdx.x * NL, nl = min(NL, nlines - line0);
  for (int t = threadIdx.x; t < a.N2; t += kPfaBlock) tw2[t] = tw2g[t];
  for (int t = threadIdx.x; t < a.N1; t += kPfaBlock) tw1[t] = tw1g[t];
  pfa_fill_ptab(a, ptab, ptabg);
  __syncthreads();
  const int shl = 31 - __clz(NL);                          // NL is a power of two
#pragma unroll 4
  for (int t = threadIdx.x; t < a.N * NL; t += kPfaBlock) {          // line index fastest: conflict-free LDS writes; the
    const int idx = t >> shl, l = t & (NL - 1);                      // NL lines of the block stay in L1 while it sweeps them
    Dr[t] = l < nl ? mesh[(long)(line0 + l) * a.N + (ptab[idx] & 0xffff)] : T(0);
  }
  __syncthreads();
  if (mfma) {
    // stage A on the matrix cores (NL = 16: the 16 data columns of a tile are the 16 lines of ONE sub-line n1)
    typedef typename Mfma<T>::Acc Acc;
    const int N2 = a.N2, H = (N2 - 1) / 2, MT = (H + 15) / 16, KP = (H + 3) & ~3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
    for (int item = wave; item < MT * a.N1; item += kPfaBlock / 64) {
      const int mt = item % MT, n1 = item / MT;
      const int i = 16 * mt + lo;
      TwIdx ti(i < H ? i : 0, hi, N2);
      const T* col = Dr + (size_t)n1 * N2 * 16 + lo;
      Acc P = {0, 0, 0, 0}, R = {0, 0, 0, 0};
      for (int kk = hi; kk < KP; kk += 4) {
        const Cx<T> w = tw2[ti.m];
        ti.step();
        T u = T(0), v = T(0);
        if (kk < H) { u = col[(1 + kk) * 16]; v = col[(N2 - 1 - kk) * 16]; }
        P = Mfma<T>::mma(w.re, u + v, P);
        R = Mfma<T>::mma(w.im, u - v, R);
      }
      const T x0 = col[0];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 1 + 16 * mt + Mfma<T>::row(lane, r);
        if (k <= H) Y[(n1 * Kh2 + k) * 16 + lo] = Cx<T>{x0 + P[r], -R[r]};
      }
    }
    for (int t = threadIdx.x; t < a.N1 * 16; t += kPfaBlock) {        // k2 = 0: plain sums
      const int n1 = t >> 4, l = t & 15;
      const T* col = Dr + (size_t)n1 * N2 * 16 + l;
      T sum = T(0);
#pragma unroll 8
      for (int n2 = 0; n2 < N2; ++n2) sum += col[n2 * 16];
      Y[(n1 * Kh2) * 16 + l] = Cx<T>{sum, T(0)};
    }
  } else {
    for (int t = threadIdx.x; t < a.N1 * Kh2 * NL; t += kPfaBlock) {   // stage A: real N2-point lines
      const int l = t % NL, k2 = (t / NL) % Kh2, n1 = t / (NL * Kh2);
      Y[(n1 * Kh2 + k2) * NL + l] = rdft_rows<T>(a.N2, k2, NL, Dr + (size_t)n1 * a.N2 * NL + l, tw2);
    }
  }
  __syncthreads();
  const bool done = pfa_n1_dispatch(a.N1, [&](auto n1c) {          // compiled short lengths: a column (k2, line) per task
    constexpr int N1 = decltype(n1c)::value;
    const int shl = 31 - __clz(NL);
    for (int t = threadIdx.x; t < Kh2 * NL; t += kPfaBlock) {
      const int k2 = t >> shl, l = t & (NL - 1);
      if (l >= nl) continue;
      Cx<T> X[N1];
      short_dft_col<T, N1>(-1, Y + (size_t)k2 * NL + l, Kh2 * NL, tw1, X);
      Cx<T>* o = spec + (long)(line0 + l) * Khp + k2 * N1;
#pragma unroll
      for (int k1 = 0; k1 < N1; ++k1) o[k1] = X[k1];
    }
  });
  if (done) return;
  const float invK = 1.0f / (float)Khp, inv1 = 1.0f / (float)a.N1;
  for (int t = threadIdx.x; t < Khp * NL; t += kPfaBlock) {          // stage B over n1, straight to memory
    const int l = fast_div(t, Khp, invK), cz = t - l * Khp;
    const int k2 = fast_div(cz, a.N1, inv1), k1 = cz - k2 * a.N1;
    if (l < nl) spec[(long)(line0 + l) * Khp + cz] = short_dft<T>(a.N1, k1, -1, Y + (size_t)k2 * NL + l, Kh2 * NL, tw1);
  }
}

// ---- z lines back: spec [nlines][Khp] -> mesh [nlines][N]
template <class T>
__global__ __launch_bounds__(kPfaBlock) void k_pfa_z_c2r(PfaAxis a, int nlines, int NL, int mfma, const Cx<T>* __restrict__ spec,
                                                        T* __restrict__ mesh, const Cx<T>* __restrict__ tw2g,
                                                        const Cx<T>* __restrict__ tw1g, const int* __restrict__ ptabg,
                                                        long mesh_stride, long spec_stride) {
  mesh += blockIdx.y * mesh_stride;
  spec += blockIdx.y * spec_stride;
  const int Kh2 = a.N2 / 2 + 1, Khp = a.N1 * Kh2;
  Cx<T>* tw2 = reinterpret_cast<Cx<T>*>(pfa_smem);
  Cx<T>* tw1 = tw2 + a.N2;
  const int KhpP = Khp | 1;                                // odd pitch: conflict-free both ways
  Cx<T>* X = tw1 + a.N1;                                   // [NL][KhpP], column cz = k2 * N1 + k1
  Cx<T>* Y = X + (size_t)KhpP * NL;                        // [N1 * Kh2][NL], row n1 * Kh2 + k2
  int* ptab = reinterpret_cast<int*>(Y + (size_t)Khp * NL);    // [N] (pfa_fill_ptab)
  const int line0 = blockIdx.x * NL, nl = min(NL, nlines - line0);
  for (int t = threadIdx.x; t < a.N2; t += kPfaBlock) tw2[t] = tw2g[t];
  for (int t = threadIdx.x; t < a.N1; t += kPfaBlock) tw1[t] = tw1g[t];
  pfa_fill_ptab(a, ptab, ptabg);
  const float invK = 1.0f / (float)Khp;
#pragma unroll 4
  for (int t = threadIdx.x; t < Khp * NL; t += kPfaBlock) {
    const int l = fast_div(t, Khp, invK), cz = t - l * Khp;
    X[l * KhpP + cz] = l < nl ? spec[(long)(line0 + l) * Khp + cz] : Cx<T>{T(0), T(0)};
  }
  __syncthreads();
  const int shl = 31 - __clz(NL);                          // NL is a power of two
  const bool done_b = pfa_n1_dispatch(a.N1, [&](auto n1c) {
    constexpr int N1 = decltype(n1c)::value;
    for (int t = threadIdx.x; t < Kh2 * NL; t += kPfaBlock) {        // a column (k2, line) per task
      const int k2 = t >> shl, l = t & (NL - 1);
      Cx<T> Z[N1];
      short_dft_col<T, N1>(+1, X + (size_t)l * KhpP + k2 * N1, 1, tw1, Z);
#pragma unroll
      for (int n1 = 0; n1 < N1; ++n1) Y[(n1 * Kh2 + k2) * NL + l] = Z[n1];
    }
  });
  if (!done_b) {
    const float invH = 1.0f / (float)Kh2;
    for (int t = threadIdx.x; t < Khp * NL; t += kPfaBlock) {          // inverse stage B: X(k1, k2) -> Y(n1, k2)
      const int l = t & (NL - 1), r = t >> shl, n1 = fast_div(r, Kh2, invH), k2 = r - n1 * Kh2;
      Y[(n1 * Kh2 + k2) * NL + l] = short_dft<T>(a.N1, n1, +1, X + (size_t)l * KhpP + k2 * a.N1, 1, tw1);
    }
  }
  __syncthreads();
  if (mfma) {
    // inverse stage A on the matrix cores (NL = 16): x_j, x_{N2-j} = Y0.re + 2 (P -+ R), P = sum Re Y_k cos, R = sum Im Y_k sin;
    // the reals go to LDS first (X is free now) so that the lines leave in order
    typedef typename Mfma<T>::Acc Acc;
    T* Dr = reinterpret_cast<T*>(X);                         // [N][16]
    const int N2 = a.N2, H = (N2 - 1) / 2, MT = (H + 15) / 16, KP = (H + 3) & ~3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
    for (int item = wave; item < MT * a.N1; item += kPfaBlock / 64) {
      const int mt = item % MT, n1 = item / MT;
      const int i = 16 * mt + lo;
      TwIdx ti(i < H ? i : 0, hi, N2);
      const Cx<T>* col = Y + (size_t)n1 * Kh2 * 16 + lo;
      Acc P = {0, 0, 0, 0}, R = {0, 0, 0, 0};
      for (int kk = hi; kk < KP; kk += 4) {
        const Cx<T> w = tw2[ti.m];
        ti.step();
        const Cx<T> v = kk < H ? col[(1 + kk) * 16] : Cx<T>{T(0), T(0)};
        P = Mfma<T>::mma(w.re, v.re, P);
        R = Mfma<T>::mma(w.im, v.im, R);
      }
      const T y0 = col[0].re;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 1 + 16 * mt + Mfma<T>::row(lane, r);
        if (j <= H) {
          Dr[(n1 * N2 + j) * 16 + lo] = y0 + T(2) * (P[r] - R[r]);
          Dr[(n1 * N2 + N2 - j) * 16 + lo] = y0 + T(2) * (P[r] + R[r]);
        }
      }
    }
    for (int t = threadIdx.x; t < a.N1 * 16; t += kPfaBlock) {        // j = 0
      const int n1 = t >> 4, l = t & 15;
      const Cx<T>* col = Y + (size_t)n1 * Kh2 * 16 + l;
      T sum = T(0);
#pragma unroll 8
      for (int k = 1; k <= H; ++k) sum += col[k * 16].re;
      Dr[(n1 * N2) * 16 + l] = col[0].re + T(2) * sum;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < a.N * NL; t += kPfaBlock) {
      const int idx = t >> 4, l = t & 15;
      if (l < nl) mesh[(long)(line0 + l) * a.N + (ptab[idx] & 0xffff)] = Dr[t];
    }
    return;
  }
  for (int t = threadIdx.x; t < a.N1 * Kh2 * NL; t += kPfaBlock) {   // inverse stage A: Hermitian N2-point lines -> reals
    const int l = t % NL, j = (t / NL) % Kh2, n1 = t / (NL * Kh2);
    if (l >= nl) continue;
    T xj, xnj;
    irdft_rows<T>(a.N2, j, NL, Y + (size_t)n1 * Kh2 * NL + l, tw2, xj, xnj);
    T* x = mesh + (long)(line0 + l) * a.N;
    x[pfa_pos(a, n1, j)] = xj;
    if (j != 0 && 2 * j != a.N2) x[pfa_pos(a, n1, a.N2 - j)] = xnj;
  }
}

// ---- host side
// split of one axis; false: this length neither fits the plain lines nor has a usable coprime split
bool pfa_split(int N, PfaAxis* out) {
  int p = largest_prime_factor(N), N2 = 1, m = N;
  while (m % p == 0) { N2 *= p; m /= p; }
  PfaAxis a;
  a.N = N;
  static const int plain_max = [] { const char* e = getenv("ADMP_PFA_MIN"); return e ? atoi(e) : 160; }();   // tests: 0 splits whatever it can
  if (N <= plain_max || m == 1) { a.N1 = 1; a.N2 = N; }
  else { a.N1 = m; a.N2 = N2; }
  if (a.N2 > 160 || a.N1 > 32 || a.N2 < 2) return false;
  *out = a;
  return true;
}
void pfa_index_table(const PfaAxis& a, int* t) {      // t[n1 * N2 + n2] = position (low 16 bits) | n1 << 16
  for (int n1 = 0; n1 < a.N1; ++n1)
    for (int n2 = 0; n2 < a.N2; ++n2) t[n1 * a.N2 + n2] = pfa_pos(a, n1, n2) | (n1 << 16);
}
void pfa_freq_of_slot(const PfaAxis& a, int* f) {      // f[slot] = the frequency stored there
  for (int k1 = 0; k1 < a.N1; ++k1)
    for (int k2 = 0; k2 < a.N2; ++k2) {
      int k = k2;
      while (k % a.N1 != k1) k += a.N2;               // CRT by search (N1 <= 32 steps)
      f[pfa_pos(a, k1, k2)] = k;
    }
}
void pfa_freq_of_zcolumn(const PfaAxis& a, int* f) {   // f[cz] for the stored z half, cz = k2 * N1 + k1
  const int Kh2 = a.N2 / 2 + 1;
  for (int k2 = 0; k2 < Kh2; ++k2)
    for (int k1 = 0; k1 < a.N1; ++k1) {
      int k = k2;
      while (k % a.N1 != k1) k += a.N2;
      f[k2 * a.N1 + k1] = k;
    }
}

template <class T>
void launch_pfa_z(hipStream_t st, const PfaPlan& p, const T* tw, T* mesh, T* spec, int inverse, int nb, long mesh_stride,
                  long spec_stride) {
  const PfaAxis& a = p.ax[2];
  const int nlines = p.ax[0].N * p.ax[1].N, Khp = p.Khp;
  const Cx<T>* tw2 = reinterpret_cast<const Cx<T>*>(tw) + p.tw_off[2];
  const Cx<T>* tw1 = tw2 + a.N2;
  // lines per block: LDS = tw + Khp * NL complex + max(N reals, Khp complex) * NL
  const size_t per_line = sizeof(Cx<T>) * (size_t)Khp + (inverse ? sizeof(Cx<T>) * (size_t)(Khp | 1) : sizeof(T) * (size_t)a.N);
  int NL = 16;        // 16 lines = the 16 data columns of an MFMA tile (one sub-line n1)
  while (NL > 1 && sizeof(Cx<T>) * (size_t)(a.N1 + a.N2) + per_line * NL + sizeof(int) * (size_t)a.N > pfa_lds_budget()) NL >>= 1;
  const int mf = (NL == 16 && pfa_use_mfma(a, 8)) ? 1 : 0;
  const size_t sh = sizeof(Cx<T>) * (size_t)(a.N1 + a.N2) + per_line * NL + sizeof(int) * (size_t)a.N;
  const dim3 grid((nlines + NL - 1) / NL, nb);
  if (inverse)
    k_pfa_z_c2r<T><<<grid, kPfaBlock, sh, st>>>(a, nlines, NL, mf, reinterpret_cast<const Cx<T>*>(spec), mesh, tw2, tw1, p.ptab[2], mesh_stride, spec_stride / 2);
  else
    k_pfa_z_r2c<T><<<grid, kPfaBlock, sh, st>>>(a, nlines, NL, mf, mesh, reinterpret_cast<Cx<T>*>(spec), tw2, tw1, p.ptab[2], mesh_stride, spec_stride / 2);
}
template <class T>
void launch_pfa_y(hipStream_t st, const PfaPlan& p, const T* tw, T* spec, int inverse, int nb, long spec_stride) {
  const PfaAxis& a = p.ax[1];
  const int Khp = p.Khp, NC = pfa_cols(a, sizeof(T)), mf = pfa_use_mfma(a, NC) ? 1 : 0;
  const Cx<T>* tw2 = reinterpret_cast<const Cx<T>*>(tw) + p.tw_off[1];
  const Cx<T>* tw1 = tw2 + a.N2;
  const int nfix = p.ax[0].N;
  const dim3 grid(xcd_grid((unsigned)(((Khp + NC - 1) / NC) * nfix)), nb);
  const size_t sh = pfa_tile_bytes(a, NC, sizeof(T));
  Cx<T>* sp = reinterpret_cast<Cx<T>*>(spec);
  if (inverse)
    k_pfa_strided<T, +1><<<grid, kPfaBlock, sh, st>>>(a, Khp, nfix, NC, mf, (long)Khp, (long)a.N * Khp, sp, tw2, tw1, p.ptab[1], spec_stride / 2);
  else
    k_pfa_strided<T, -1><<<grid, kPfaBlock, sh, st>>>(a, Khp, nfix, NC, mf, (long)Khp, (long)a.N * Khp, sp, tw2, tw1, p.ptab[1], spec_stride / 2);
}
template <class T>
void launch_pfa_x_conv(hipStream_t st, const PfaPlan& p, const T* tw, T* spec, const DftTabs<T>& tabs, double* energies,
                       int slot, int nb, long spec_stride) {
  const PfaAxis& a = p.ax[0];
  const int Khp = p.Khp, NC = pfa_cols(a, sizeof(T)), mf = pfa_use_mfma(a, NC) ? 1 : 0;
  const Cx<T>* tw2 = reinterpret_cast<const Cx<T>*>(tw) + p.tw_off[0];
  const Cx<T>* tw1 = tw2 + a.N2;
  const int nfix = p.ax[1].N;
  const dim3 grid(xcd_grid((unsigned)(((Khp + NC - 1) / NC) * nfix)), nb);
  const size_t sh = pfa_tile_bytes(a, NC, sizeof(T));
  k_pfa_x_conv<T><<<grid, kPfaBlock, sh, st>>>(a, p.ax[2], Khp, nfix, NC, mf, (long)p.ax[1].N * Khp, (long)Khp, reinterpret_cast<Cx<T>*>(spec),
                                              tabs, tw2, tw1, p.ptab[0], energies, slot, spec_stride / 2);
}
#define INST(T)                                                                                          \
  template void launch_pfa_z<T>(hipStream_t, const PfaPlan&, const T*, T*, T*, int, int, long, long);    \
  template void launch_pfa_y<T>(hipStream_t, const PfaPlan&, const T*, T*, int, int, long);              \
  template void launch_pfa_x_conv<T>(hipStream_t, const PfaPlan&, const T*, T*, const DftTabs<T>&, double*, int, int, long);
INST(float)
INST(double)
#undef INST

}  // namespace admp
