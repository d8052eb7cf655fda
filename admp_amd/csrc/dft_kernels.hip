// Direct-DFT mesh convolution for "hard" PME mesh sizes (see dft_math.h): five kernels replace
// rocFFT r2c (Bluestein) -> k_kspace -> rocFFT c2r on one GPU:
//     z lines r2c  ->  y lines forward  ->  x lines forward * G (+ reciprocal energy) x lines inverse  ->  y lines inverse  ->  z lines c2r
// Every block stages a tile of lines in LDS as j <-> N-j pair sums, then one thread produces one output pair.
// At 97^3 (f64) the whole chain is ~0.9 GFLOP of f64 FMAs; the lines come out of L2 / Infinity Cache (7 MB spectrum).
#include "dft_lines.h"

namespace admp {

int dft_tile_cols(int N) { return dft_cols(N, dft_kq(), 0, 0); }
// ---- z lines (contiguous): real mesh [nlines][N] -> half spectrum [nlines][N/2+1]
template <class T, int KQ, int JS>
__global__ __launch_bounds__(kDftBlock) void k_dft_z_r2c(int N, int nlines, int NL, int TK, const T* __restrict__ mesh,
                                                        Cx<T>* __restrict__ spec, const Cx<T>* __restrict__ twg,
                                                        long mesh_stride, long spec_stride) {
  mesh += blockIdx.y * mesh_stride;      // batch of independent meshes (dispersion: C6, C8, C10)
  spec += blockIdx.y * spec_stride;
  const int H = (N - 1) / 2, Kh = N / 2 + 1;
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_smem);
  Cx<T>* p = tw + N;                              // [H][NL]
  T* x0 = reinterpret_cast<T*>(p + H * NL);       // [NL]
  T* xn = x0 + NL;                                // [NL]
  const int line0 = blockIdx.x * NL;
  const int nl = min(NL, nlines - line0);
  for (int t = threadIdx.x; t < N; t += kDftBlock) tw[t] = twg[t];
  for (int t = threadIdx.x; t < H * nl; t += kDftBlock) {
    const int l = t / H, jj = t - l * H;
    const T* x = mesh + (long)(line0 + l) * N;
    const T a = x[1 + jj], b = x[N - 1 - jj];
    p[jj * NL + l] = Cx<T>{a + b, a - b};
  }
  if (threadIdx.x < nl) {
    const T* x = mesh + (long)(line0 + threadIdx.x) * N;
    x0[threadIdx.x] = x[0];
    xn[threadIdx.x] = (N & 1) ? T(0) : x[N / 2];
  }
  __syncthreads();
  int tid, half;
  dft_task_of_thread<JS>(tid, half);
  const int l = tid / TK, g = tid - l * TK;
  if (l < nl) {
    int k[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK < Kh) ? g + q * TK : 0;
    T P[KQ], R[KQ];
    real_pair_sums_js<T, KQ, JS>(N, k, NL, p + l, tw, half, P, R);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      if (g + q * TK < Kh && (JS == 1 || half == (q & 1))) {        // the two lanes of a task share the stores
        Cx<T> X;
        X.re = x0[l] + P[q];
        if ((N & 1) == 0) X.re += (k[q] & 1) ? -xn[l] : xn[l];
        X.im = -R[q];
        spec[(long)(line0 + l) * Kh + g + q * TK] = X;
      }
    }
  }
}

// ---- z lines back: half spectrum -> real mesh
template <class T, int KQ, int JS>
__global__ __launch_bounds__(kDftBlock) void k_dft_z_c2r(int N, int nlines, int NL, int TK, const Cx<T>* __restrict__ spec,
                                                        T* __restrict__ mesh, const Cx<T>* __restrict__ twg,
                                                        long mesh_stride, long spec_stride, T* __restrict__ accum) {
  mesh += blockIdx.y * mesh_stride;
  spec += blockIdx.y * spec_stride;
  const int H = (N - 1) / 2, Kh = N / 2 + 1;
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_smem);
  Cx<T>* p = tw + N;
  T* X0 = reinterpret_cast<T*>(p + H * NL);
  T* Xn = X0 + NL;
  const int line0 = blockIdx.x * NL;
  const int nl = min(NL, nlines - line0);
  for (int t = threadIdx.x; t < N; t += kDftBlock) tw[t] = twg[t];
  for (int t = threadIdx.x; t < Kh * nl; t += kDftBlock) {
    const int l = t / Kh, k = t - l * Kh;
    const Cx<T> v = spec[(long)(line0 + l) * Kh + k];
    if (k == 0) X0[l] = v.re;
    else if (2 * k == N) Xn[l] = v.re;
    else p[(k - 1) * NL + l] = v;
  }
  if ((N & 1) && threadIdx.x < nl) Xn[threadIdx.x] = T(0);
  __syncthreads();
  int tid, half;
  dft_task_of_thread<JS>(tid, half);
  const int l = tid / TK, g = tid - l * TK;
  if (l < nl) {
    int j[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) j[q] = (g + q * TK < Kh) ? g + q * TK : 0;
    T P[KQ], R[KQ];
    real_pair_sums_js<T, KQ, JS>(N, j, NL, p + l, tw, half, P, R);
    T* x = mesh + (long)(line0 + l) * N;
    T* acc = accum ? accum + (long)(line0 + l) * N : nullptr;   // SCF increment: phi += this mesh in the same pass
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int jq = g + q * TK;
      if (jq < Kh) {
        T base = X0[l] + T(2) * P[q];
        if ((N & 1) == 0) base += (j[q] & 1) ? -Xn[l] : Xn[l];
        if (JS == 1 || half == 0) {                                  // lane 0 of the task stores x_j, lane 1 x_{N-j}
          const T v = base - T(2) * R[q];
          x[jq] = v;
          if (acc) acc[jq] += v;
        }
        if ((JS == 1 || half == 1) && jq != 0 && 2 * jq != N) {
          const T v = base + T(2) * R[q];
          x[N - jq] = v;
          if (acc) acc[N - jq] += v;
        }
      }
    }
  }
}

// ---- strided complex lines, in place (y lines: fix = x plane; x lines: fix = y row)
template <class T, int SIGN, int KQ, int JS>
__global__ __launch_bounds__(kDftBlock) void k_dft_strided(int N, int ncols, int NC, int TK, long jstride, long fixstride,
                                                          Cx<T>* __restrict__ spec, const Cx<T>* __restrict__ twg,
                                                          long spec_stride) {
  spec += blockIdx.z * spec_stride;
  const int H = (N - 1) / 2, Kh = N / 2 + 1;
  PairCx<T>* ab = reinterpret_cast<PairCx<T>*>(dft_smem);   // [H][NC]
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(ab + H * NC);          // [N]
  Cx<T>* x0 = tw + N;                                         // [NC]
  Cx<T>* xn = x0 + NC;                                        // [NC]
  const int col0 = blockIdx.x * NC;
  const int nca = min(NC, ncols - col0);
  const long base = (long)blockIdx.y * fixstride + col0;
  for (int t = threadIdx.x; t < N; t += kDftBlock) tw[t] = twg[t];
  load_pairs<T>(N, NC, nca, spec, base, jstride, ab, x0, xn);
  __syncthreads();
  int tid, half;
  dft_task_of_thread<JS>(tid, half);
  const int g = tid / NC, c = tid - g * NC;
  if (g < TK && c < nca) {
    int k[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK < Kh) ? g + q * TK : 0;
    Cx<T> Xk[KQ], Xnk[KQ];
    dft_pair_outputs_js<T, SIGN, KQ, JS>(N, k, NC, ab + c, x0[c], xn[c], tw, half, Xk, Xnk);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK;
      if (kq < Kh) {
        if (JS == 1 || half == 0) spec[base + (long)kq * jstride + c] = Xk[q];
        if ((JS == 1 || half == 1) && kq != 0 && 2 * kq != N) spec[base + (long)(N - kq) * jstride + c] = Xnk[q];
      }
    }
  }
}

// ---- x lines: forward, multiply by G (accumulating sum w G |S|^2, recip.py:400-414 / pme.py:240), inverse; in place
// (dft_lines.h dft_x_conv_body)
template <class T, int KQ, int JS>
__global__ __launch_bounds__(kDftBlock) void k_dft_x_conv(XConvArgs<T> xa) {
  dft_x_conv_body<T, KQ, JS>(xa, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- x lines of the TYPED dispersion meshes (round 4; see disp_kernels.hip): the workgroup's NC "columns" are NC / NTP
// physical columns of each of NTP (padded: 1, 2 or 4) type spectra, column t * NCt + c.  The transforms treat them as NC
// independent columns; between them the types are combined per (frequency, physical column):
//   S_p = sum_t c[p][t] X_t,   E += w G_p |S_p|^2,   X_t <- sum_p c[p][t] G_p S_p.
template <class T, int KQ, int JS, int NTP>
__global__ __launch_bounds__(kDftBlock) void k_dft_x_mix(int N, int ncols, int NC, int TK, long jstride, long fixstride,
                                                        int K3, Cx<T>* __restrict__ spec, DftTabs<T> tabs, MixTab mix,
                                                        const Cx<T>* __restrict__ twg, double* energies, int slot,
                                                        long spec_tstride) {
  const int H = (N - 1) / 2, Kh = N / 2 + 1, NCt = NC / NTP;
  PairCx<T>* ab = reinterpret_cast<PairCx<T>*>(dft_smem);   // [H][NC]
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(ab + H * NC);          // [N]
  Cx<T>* x0 = tw + N;                                         // [NC]
  Cx<T>* xn = x0 + NC;                                        // [NC]
  Cx<T>* S = xn + NC;                                         // [N][NC]
  const int col0 = blockIdx.x * NCt;
  const int ncp = min(NCt, ncols - col0);                     // physical columns of this workgroup
  const long base = (long)blockIdx.y * fixstride + col0;
  for (int t = threadIdx.x; t < N; t += kDftBlock) tw[t] = twg[t];
  // the G values this thread applies after the forward transform (every channel's, both frequencies of its pair)
  T Gp[KQ][3][2], G0[3] = {T(0), T(0), T(0)}, Gn[3] = {T(0), T(0), T(0)};
#pragma unroll
  for (int u = 0; u < KQ; ++u) {
    const int t = threadIdx.x + u * kDftBlock;
    const int jj = t / NCt, cc = t - jj * NCt;
    const bool in = t < H * NCt && cc < ncp;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const bool on = in && ch < mix.nch;
      Gp[u][ch][0] = on ? tabs.p[ch][base + (long)(1 + jj) * jstride + cc] : T(0);
      Gp[u][ch][1] = on ? tabs.p[ch][base + (long)(N - 1 - jj) * jstride + cc] : T(0);
    }
  }
  if ((int)threadIdx.x < ncp) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      if (ch < mix.nch) {
        G0[ch] = tabs.p[ch][base + threadIdx.x];
        if ((N & 1) == 0) Gn[ch] = tabs.p[ch][base + (long)(N / 2) * jstride + threadIdx.x];
      }
  }
  // pair sums of the columns: virtual column vc = type * NCt + physical column
  for (int t = threadIdx.x; t < H * NC; t += kDftBlock) {
    const int jj = t / NC, vc = t - jj * NC, ty = vc / NCt, c = vc - ty * NCt;
    PairCx<T> v{T(0), T(0), T(0), T(0)};
    if (c < ncp && ty < mix.nt) {
      const Cx<T>* sp = spec + (long)ty * spec_tstride + base + c;
      const Cx<T> a = sp[(long)(1 + jj) * jstride], b = sp[(long)(N - 1 - jj) * jstride];
      v = PairCx<T>{a.re + b.re, a.im + b.im, a.re - b.re, a.im - b.im};
    }
    ab[t] = v;
  }
  if ((int)threadIdx.x < NC) {
    const int vc = threadIdx.x, ty = vc / NCt, c = vc - ty * NCt;
    Cx<T> a{T(0), T(0)}, b{T(0), T(0)};
    if (c < ncp && ty < mix.nt) {
      const Cx<T>* sp = spec + (long)ty * spec_tstride + base + c;
      a = sp[0];
      if ((N & 1) == 0) b = sp[(long)(N / 2) * jstride];
    }
    x0[vc] = a;
    xn[vc] = b;
  }
  __syncthreads();
  int tid, half;
  dft_task_of_thread<JS>(tid, half);
  const int g = tid / NC, c = tid - g * NC;
  const int cty = c / NCt, ccol = c - cty * NCt;
  const bool task = g < TK && c < NC && ccol < ncp && cty < mix.nt;
  int k[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK < Kh) ? g + q * TK : 0;
  if (task) {
    Cx<T> Xk[KQ], Xnk[KQ];
    dft_pair_outputs_js<T, -1, KQ, JS>(N, k, NC, ab + c, x0[c], xn[c], tw, half, Xk, Xnk);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK;
      if (kq < Kh) {
        if (JS == 1 || half == 0) S[kq * NC + c] = Xk[q];
        if ((JS == 1 || half == 1) && kq != 0 && 2 * kq != N) S[(N - kq) * NC + c] = Xnk[q];
      }
    }
  }
  __syncthreads();
  double e = 0.0;
  // one frequency of one physical column: combine the types (s: their spectra at it, in place -> psi), return the energy term
  auto combine = [&](Cx<T>* s, const T* G) {
    Cx<T> y[NTP];
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty) y[ty] = Cx<T>{T(0), T(0)};
    double en = 0.0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (ch >= mix.nch) break;
      Cx<T> Sp{T(0), T(0)};
#pragma unroll
      for (int ty = 0; ty < NTP; ++ty) { const T cc = (T)mix.c[ch][ty]; Sp.re += cc * s[ty].re; Sp.im += cc * s[ty].im; }
      en += (double)G[ch] * ((double)Sp.re * Sp.re + (double)Sp.im * Sp.im);
#pragma unroll
      for (int ty = 0; ty < NTP; ++ty) { const T cg = (T)mix.c[ch][ty] * G[ch]; y[ty].re += cg * Sp.re; y[ty].im += cg * Sp.im; }
    }
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty) s[ty] = y[ty];
    return en;
  };
#pragma unroll
  for (int u = 0; u < KQ; ++u) {
    const int t = threadIdx.x + u * kDftBlock;
    if (t >= H * NCt) continue;
    const int jj = t / NCt, cc = t - jj * NCt;
    Cx<T> s1[NTP], s2[NTP];
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty) s1[ty] = s2[ty] = Cx<T>{T(0), T(0)};
    if (cc < ncp) {
      const int k1 = 1 + jj, k2 = N - 1 - jj, kz = col0 + cc;
#pragma unroll
      for (int ty = 0; ty < NTP; ++ty) { s1[ty] = S[k1 * NC + ty * NCt + cc]; s2[ty] = S[k2 * NC + ty * NCt + cc]; }
      const double w = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
      const T G1[3] = {Gp[u][0][0], Gp[u][1][0], Gp[u][2][0]}, G2[3] = {Gp[u][0][1], Gp[u][1][1], Gp[u][2][1]};
      e += w * (combine(s1, G1) + combine(s2, G2));
    }
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty)
      ab[jj * NC + ty * NCt + cc] = PairCx<T>{s1[ty].re + s2[ty].re, s1[ty].im + s2[ty].im, s1[ty].re - s2[ty].re, s1[ty].im - s2[ty].im};
  }
  if ((int)threadIdx.x < NCt) {
    const int cc = threadIdx.x;
    Cx<T> a[NTP], b[NTP];
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty) a[ty] = b[ty] = Cx<T>{T(0), T(0)};
    if (cc < ncp) {
      const int kz = col0 + cc;
      const double w = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
#pragma unroll
      for (int ty = 0; ty < NTP; ++ty) a[ty] = S[ty * NCt + cc];
      e += w * combine(a, G0);
      if ((N & 1) == 0) {
#pragma unroll
        for (int ty = 0; ty < NTP; ++ty) b[ty] = S[(N / 2) * NC + ty * NCt + cc];
        e += w * combine(b, Gn);
      }
    }
#pragma unroll
    for (int ty = 0; ty < NTP; ++ty) { x0[ty * NCt + cc] = a[ty]; xn[ty * NCt + cc] = b[ty]; }
  }
  __syncthreads();
  if (task) {
    Cx<T> Xk[KQ], Xnk[KQ];
    dft_pair_outputs_js<T, +1, KQ, JS>(N, k, NC, ab + c, x0[c], xn[c], tw, half, Xk, Xnk);
    Cx<T>* sp = spec + (long)cty * spec_tstride + base + ccol;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK;
      if (kq < Kh) {
        if (JS == 1 || half == 0) sp[(long)kq * jstride] = Xk[q];
        if ((JS == 1 || half == 1) && kq != 0 && 2 * kq != N) sp[(long)(N - kq) * jstride] = Xnk[q];
      }
    }
  }
  e = block_reduce_sum<kDftBlock>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[slot], e);
}

// ---- z and y lines of one x plane in ONE workgroup (round 3) ------------------------------------------------------------
// A pass of the chain above is 11-14 us of kernel for 1-2 us of f64 arithmetic at 97^3: ~4 us between the kernel's start and
// its first block / its last block and its end, ~3 us of loads, ~1 us of stores, a 48-step dependent loop over < 2 waves per
// SIMD (tools/ubench/dftm_phases.hip).  The z and the y lines of one x plane touch only that plane: 97 x 97 reals and its
// 97 x 49 half spectrum fit the 160 KB of LDS together, so one workgroup of 1024 threads per plane does both transforms --
// one launch, one load of the plane, one store of the spectrum, and 16 waves per CU to hide the loop's LDS latency.  Measured
// at 97^3 f64 (rocprof / HIP events): z 11.2 + y 13.5 us as two passes; one workgroup per plane 32 us (97 of 256 CUs, bound by
// their f64 FMAs); two workgroups per plane (below) 24.5 us forward, 23.2 us inverse: the step 0.234 -> 0.219 ms.
// Phases of the forward kernel (clock64 stamps, one workgroup): load + pair sums 3.0 us, z lines 6.6 us (1261 tasks: a second
// round for 237 of them), pairing along y 0.5 us, y lines + stores 6.8 us.  Three outputs per z task (873 tasks, one round)
// measured 2 % slower than two; two lanes per z task, each summing half of the pair positions (three rounds of 24 steps
// instead of two of 48, idle lanes running along for the shuffles): 17 % slower.  The z results are paired in place along y (row j <- x_j + x_{N-j}, row N-j <-
// x_j - x_{N-j}: dft_pair_outputs_rows) between the two stages.
constexpr int kZyBlock = 1024;

template <class T>
struct ZyLayout {          // LDS carve-up of the plane kernels (bytes); the same for both directions
  size_t tw3, tw2, a, x0, b, total;
  __host__ __device__ ZyLayout(int N2, int N3) {
    const size_t H3 = (size_t)(N3 - 1) / 2, Kh = (size_t)N3 / 2 + 1;
    tw3 = 0;
    tw2 = tw3 + sizeof(Cx<T>) * N3;
    a = tw2 + sizeof(Cx<T>) * N2;                          // forward: z pair sums [H3][N2]; inverse: y results [N2][Kh]
    const size_t fa = sizeof(Cx<T>) * H3 * N2, ia = sizeof(Cx<T>) * N2 * Kh;
    x0 = a + (fa > ia ? fa : ia);                          // forward: x_0 and x_{N/2} of the z lines [2][N2]
    b = x0 + sizeof(Cx<T>) * N2;                           // the plane's half spectrum [N2][Kh]
    total = b + sizeof(Cx<T>) * N2 * Kh;
  }
};

// gridDim.z = 2 splits a plane between two workgroups (K1 planes alone leave most of the 256 CUs idle and the kernel is then
// bound by the f64 FMAs of its CUs): forward by kz halves -- a z line's outputs are independent, so each workgroup computes
// its half of the outputs of every z line and transforms its own columns along y; inverse by halves of the y output pairs
// (y, N-y) -- each workgroup transforms every column along y for its pairs only and then the z lines of those y.  Both
// load the whole plane.
// SPREAD (round 4, small systems in double precision): the plane is not read from a mesh -- the workgroup builds it in LDS from
// the sites (PlaneSpread: the scan spread of recip_kernels.hip, one x plane instead of one brick): it scans the stencil bases
// of all atoms, keeps those whose six x planes include this one (~6 na / K0 of them) and adds their 6 x 6 (y, z) patches with
// LDS atomics, straight into the slots of the z pair sums (element (l, j) is the real part of slot (j - 1, l) for j <= H3, the
// imaginary part of slot (N3 - 1 - j, l) above, x0[l] for j = 0): one in-place pass then turns (a, b) into (a + b, a - b).
// The spread kernel (17 us at 3072 atoms: 375 workgroups each scanning every atom), its dispatch and the mesh round trip through
// L2 go away; both workgroups of a plane build the whole plane (they both need it).  Weights: 4 threads per kept atom stage
// the three axes' splines and the folded multipoles (spread_entry_list's arithmetic), then one task per stencil point.
constexpr int kZyW = 42;                       // staged words per kept atom: 6 folded coefficients, 18 y and 18 z weights
constexpr int kZyScan = 3;                     // atoms a thread tests per scan round (their records are fetched together)
constexpr int kZySpreadMaxAtoms = 8192;        // a hit is 16 bits: atom (13) | x offset (3)
template <class T>
__host__ __device__ inline int zy_spread_sub(int N2, int N3) {      // kept atoms whose weights fit in the spectrum region (free
  const int Kh = N3 / 2 + 1;                                         // until the z lines write it) next to the hit list
  const long spare = (long)sizeof(Cx<T>) * N2 * Kh - (long)sizeof(unsigned short) * kZyScan * kZyBlock;
  const long n = spare / (long)(kZyW * sizeof(T) + 2 * sizeof(int));
  return n > kZyBlock / 3 ? kZyBlock / 3 : (int)n;                   // (three staging threads per kept atom)
}
template <class T>
__device__ __forceinline__ void zy_plane_spread(const PlaneSpread<T>& sp, int px, int N2, int N3, Cx<T>* p, T* x0, T* xn,
                                                unsigned char* scratch, int SUB) {
  __shared__ int s_nhit;
  unsigned short* hits = reinterpret_cast<unsigned short*>(scratch);             // [kZyScan * kZyBlock]
  T* wts = reinterpret_cast<T*>(hits + kZyScan * kZyBlock);                      // [SUB][kZyW]
  int* ebase = reinterpret_cast<int*>(wts + (size_t)SUB * kZyW);                 // [SUB][2]: stencil bases along y and z
  const int H3 = (N3 - 1) / 2;
  const Site<T>* sites = sp.sites + (size_t)blockIdx.y * sp.na;
  for (int t = threadIdx.x; t < H3 * N2; t += kZyBlock) p[t] = Cx<T>{T(0), T(0)};
  for (int t = threadIdx.x; t < 2 * N2; t += kZyBlock) x0[t] = T(0);                 // (x0 and xn are one array)
  const RecipGeom<T>& g = sp.g;
  for (int c0 = 0; c0 < sp.na; c0 += kZyScan * kZyBlock) {
    if (threadIdx.x == 0) s_nhit = 0;
    __syncthreads();                                                                   // (also: the zeroes above are in place)
    int bx[kZyScan];
#pragma unroll
    for (int r = 0; r < kZyScan; ++r) {                                               // independent loads, all in flight
      const int i = c0 + (int)threadIdx.x + r * kZyBlock;
      bx[r] = -1000;
      if (i < sp.na) {
        if (sp.bases) bx[r] = sp.bases[i].x;
        else {
          const T rr[3] = {sites[i].r[0], sites[i].r[1], sites[i].r[2]};
          (void)grid_ref(g, rr, 0, bx[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kZyScan; ++r) {
      int a = px - bx[r];
      if (a < 0) a += g.K[0];
      if (bx[r] >= 0 && a < 6)
        hits[atomicAdd(&s_nhit, 1)] = (unsigned short)((c0 + (int)threadIdx.x + r * kZyBlock) | (a << 13));
    }
    __syncthreads();
    const int nhit = s_nhit;
    const int nsub = (nhit + SUB - 1) / SUB, per = nsub > 0 ? (nhit + nsub - 1) / nsub : 0;      // balanced sub-rounds
    for (int sub = 0; sub < nhit; sub += per) {
      const int cnt = min(per, nhit - sub);
      if ((int)threadIdx.x < 3 * cnt) {
        const int e = threadIdx.x / 3, part = threadIdx.x - 3 * e;
        const int h = hits[sub + e];
        const Site<T>& site = sites[h & 0x1fff];
        T* w = wts + e * kZyW;
        const T r[3] = {site.r[0], site.r[1], site.r[2]};
        int base;
        T M[6], D1[6], D2[6], D3[6];
        const T f = grid_ref(g, r, part, base);
        bspline6(f, M, D1, D2, D3);
        if (part == 0) {
          // x factors of this plane folded into the multipole coefficients: c1 = (x, y, z), c2 = (xx, yy, zz, xy, xz, yz)
          const int a = h >> 13;
          T m0 = T(0), d0 = T(0), e0 = T(0);
#pragma unroll
          for (int k = 0; k < 6; ++k)
            if (k == a) { m0 = M[k]; d0 = D1[k]; e0 = D2[k]; }
          T Q[9], c1[3], c2[6];
#pragma unroll
          for (int k = 0; k < 9; ++k) Q[k] = site.Q[k];
          if (sp.lpol) { Q[1] += site.U[0]; Q[2] += site.U[1]; Q[3] += site.U[2]; }       // Q_global_tot, admp/pme.py:236
          fold_multipole(g, Q, c1, c2);
          w[0] = Q[0] * m0 + c1[0] * d0 + c2[0] * e0;      // P0 = w0 My + w1 My' + w2 My''   (times Mz)
          w[1] = c1[1] * m0 + c2[3] * d0;
          w[2] = c2[1] * m0;
          w[3] = c1[2] * m0 + c2[4] * d0;                  // P1 = w3 My + w4 My'             (times Mz')
          w[4] = c2[5] * m0;
          w[5] = c2[2] * m0;                               // P2 = w5 My                      (times Mz'')
        } else {
          const int o = part == 1 ? 6 : 24;
#pragma unroll
          for (int k = 0; k < 6; ++k) { w[o + k] = M[k]; w[o + 6 + k] = D1[k]; w[o + 12 + k] = D2[k]; }
          ebase[2 * e + part - 1] = base;
        }
      }
      __syncthreads();
      for (int task = threadIdx.x; task < cnt * 6; task += kZyBlock) {      // (kept atom, y point): six z adds
        const int e = task / 6, b = task - e * 6;
        const T* w = wts + e * kZyW;
        const T m1 = w[6 + b], d1 = w[12 + b], e1 = w[18 + b];
        const T P0 = w[0] * m1 + w[1] * d1 + w[2] * e1;
        const T P1 = w[3] * m1 + w[4] * d1;
        const T P2 = w[5] * m1;
        const int jb = wrap_add(ebase[2 * e], b, N2), bz = ebase[2 * e + 1];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const T v = P0 * w[24 + c] + P1 * w[30 + c] + P2 * w[36 + c];
          const int jc = wrap_add(bz, c, N3);
          T* dst;
          if (jc == 0) dst = &x0[jb];
          else if (jc <= H3) dst = &p[(jc - 1) * N2 + jb].re;
          else if (2 * jc == N3) dst = &xn[jb];
          else dst = &p[(N3 - 1 - jc) * N2 + jb].im;
          atomicAdd(dst, v);
        }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < H3 * N2; t += kZyBlock) {
    const Cx<T> ab = p[t];
    p[t] = Cx<T>{ab.re + ab.im, ab.re - ab.im};
  }
}

template <class T, int KQ, bool SPREAD>
__global__ __launch_bounds__(kZyBlock) void k_dft_zy_fwd(int N2, int N3, const T* __restrict__ mesh, Cx<T>* __restrict__ spec,
                                                         const Cx<T>* __restrict__ tw2g, const Cx<T>* __restrict__ tw3g,
                                                         long mesh_stride, long spec_stride, PlaneSpread<T> sp) {
  const ZyLayout<T> L(N2, N3);
  const int H3 = (N3 - 1) / 2, Kh = N3 / 2 + 1, H2 = (N2 - 1) / 2;
  const int kz0 = (int)(((long)Kh * blockIdx.z) / gridDim.z), kz1 = (int)(((long)Kh * (blockIdx.z + 1)) / gridDim.z);
  const int Kl = kz1 - kz0;                                     // this workgroup's columns
  Cx<T>* tw3 = reinterpret_cast<Cx<T>*>(dft_smem + L.tw3);
  Cx<T>* tw2 = reinterpret_cast<Cx<T>*>(dft_smem + L.tw2);
  Cx<T>* p = reinterpret_cast<Cx<T>*>(dft_smem + L.a);          // [H3][N2]
  T* x0 = reinterpret_cast<T*>(dft_smem + L.x0);                // [N2], then xn [N2]
  T* xn = x0 + N2;
  Cx<T>* Z = reinterpret_cast<Cx<T>*>(dft_smem + L.b);          // [N2][Kl]
  const T* plane = mesh + blockIdx.y * mesh_stride + (long)blockIdx.x * N2 * N3;
  Cx<T>* out = spec + blockIdx.y * spec_stride + (long)blockIdx.x * N2 * Kh;
  for (int t = threadIdx.x; t < N3; t += kZyBlock) tw3[t] = tw3g[t];
  for (int t = threadIdx.x; t < N2; t += kZyBlock) tw2[t] = tw2g[t];
  if (SPREAD) {
    zy_plane_spread<T>(sp, (int)blockIdx.x, N2, N3, p, x0, xn, dft_smem + L.b, zy_spread_sub<T>(N2, N3));
  } else {
    for (int t = threadIdx.x; t < H3 * N2; t += kZyBlock) {
      const int l = t / H3, jj = t - l * H3;
      const T* x = plane + (long)l * N3;
      const T a = x[1 + jj], b = x[N3 - 1 - jj];
      p[jj * N2 + l] = Cx<T>{a + b, a - b};
    }
    for (int l = threadIdx.x; l < N2; l += kZyBlock) {
      x0[l] = plane[(long)l * N3];
      xn[l] = (N3 & 1) ? T(0) : plane[(long)l * N3 + N3 / 2];
    }
  }
  __syncthreads();
  // z lines: task = (line l, group g of KQ of this workgroup's outputs)
  const int TK3 = (Kl + KQ - 1) / KQ;
  for (int task = threadIdx.x; task < N2 * TK3; task += kZyBlock) {
    const int l = task / TK3, g = task - l * TK3;
    int k[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK3 < Kl) ? kz0 + g + q * TK3 : 0;
    Cx<T> X[KQ];
    rdft_outputs<T, KQ>(N3, k, N2, p + l, x0[l], xn[l], tw3, X);
#pragma unroll
    for (int q = 0; q < KQ; ++q)
      if (g + q * TK3 < Kl) Z[l * Kl + g + q * TK3] = X[q];
  }
  __syncthreads();
  // pair the rows along y in place
  for (int t = threadIdx.x; t < H2 * Kl; t += kZyBlock) {
    const int jj = t / Kl, c = t - jj * Kl;
    const Cx<T> a = Z[(1 + jj) * Kl + c], b = Z[(N2 - 1 - jj) * Kl + c];
    Z[(1 + jj) * Kl + c] = Cx<T>{a.re + b.re, a.im + b.im};
    Z[(N2 - 1 - jj) * Kl + c] = Cx<T>{a.re - b.re, a.im - b.im};
  }
  __syncthreads();
  // y lines: task = (group g, column c): neighbouring threads store neighbouring kz
  const int Kh2 = N2 / 2 + 1, TK2 = (Kh2 + KQ - 1) / KQ;
  for (int task = threadIdx.x; task < TK2 * Kl; task += kZyBlock) {
    const int g = task / Kl, c = task - g * Kl;
    int k[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK2 < Kh2) ? g + q * TK2 : 0;
    Cx<T> Xk[KQ], Xnk[KQ];
    dft_pair_outputs_rows<T, -1, KQ>(N2, k, Kl, Z + c, tw2, Xk, Xnk);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK2;
      if (kq < Kh2) {
        out[(long)kq * Kh + kz0 + c] = Xk[q];
        if (kq != 0 && 2 * kq != N2) out[(long)(N2 - kq) * Kh + kz0 + c] = Xnk[q];
      }
    }
  }
}

template <class T, int KQ>
__global__ __launch_bounds__(kZyBlock) void k_dft_yz_inv(int N2, int N3, const Cx<T>* __restrict__ spec, T* __restrict__ mesh,
                                                         const Cx<T>* __restrict__ tw2g, const Cx<T>* __restrict__ tw3g,
                                                         long mesh_stride, long spec_stride, T* __restrict__ accum) {
  __shared__ int s_lines[2 * 160 + 2];
  __shared__ int s_nlines;
  const ZyLayout<T> L(N2, N3);
  const int Kh = N3 / 2 + 1, H2 = (N2 - 1) / 2;
  Cx<T>* tw3 = reinterpret_cast<Cx<T>*>(dft_smem + L.tw3);
  Cx<T>* tw2 = reinterpret_cast<Cx<T>*>(dft_smem + L.tw2);
  Cx<T>* V = reinterpret_cast<Cx<T>*>(dft_smem + L.a);          // y results [N2][Kh] (this workgroup's rows filled)
  Cx<T>* Z = reinterpret_cast<Cx<T>*>(dft_smem + L.b);          // the plane's spectrum [N2][Kh]
  const Cx<T>* in = spec + blockIdx.y * spec_stride + (long)blockIdx.x * N2 * Kh;
  T* plane = mesh + blockIdx.y * mesh_stride + (long)blockIdx.x * N2 * N3;
  T* acc = accum ? accum + (long)blockIdx.x * N2 * N3 : nullptr;      // SCF increment: phi += this mesh in the same pass
  const int Kh2 = N2 / 2 + 1, TK2 = (Kh2 + KQ - 1) / KQ;
  const int g0 = (int)(((long)TK2 * blockIdx.z) / gridDim.z), g1 = (int)(((long)TK2 * (blockIdx.z + 1)) / gridDim.z);
  for (int t = threadIdx.x; t < N3; t += kZyBlock) tw3[t] = tw3g[t];
  for (int t = threadIdx.x; t < N2; t += kZyBlock) tw2[t] = tw2g[t];
  if (threadIdx.x == 0) {            // the y lines this workgroup produces: both members of its output pairs
    int n = 0;
    for (int g = g0; g < g1; ++g)
      for (int q = 0; q < KQ; ++q) {
        const int kq = g + q * TK2;
        if (kq >= Kh2) continue;
        s_lines[n++] = kq;
        if (kq != 0 && 2 * kq != N2) s_lines[n++] = N2 - kq;
      }
    s_nlines = n;
  }
  // rows 0 and N/2 as they are, the others paired on the way in (row j and row N-j by the same thread)
  for (int t = threadIdx.x; t < (H2 + 1) * Kh; t += kZyBlock) {
    const int jj = t / Kh, c = t - jj * Kh;
    if (jj == H2) {                                   // rows 0 and (N2 even) N2/2
      Z[c] = in[c];
      if ((N2 & 1) == 0) Z[(N2 / 2) * Kh + c] = in[(long)(N2 / 2) * Kh + c];
      continue;
    }
    const Cx<T> a = in[(long)(1 + jj) * Kh + c], b = in[(long)(N2 - 1 - jj) * Kh + c];
    Z[(1 + jj) * Kh + c] = Cx<T>{a.re + b.re, a.im + b.im};
    Z[(N2 - 1 - jj) * Kh + c] = Cx<T>{a.re - b.re, a.im - b.im};
  }
  __syncthreads();
  for (int task = threadIdx.x; task < (g1 - g0) * Kh; task += kZyBlock) {
    const int g = g0 + task / Kh, c = task % Kh;
    int k[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) k[q] = (g + q * TK2 < Kh2) ? g + q * TK2 : 0;
    Cx<T> Xk[KQ], Xnk[KQ];
    dft_pair_outputs_rows<T, +1, KQ>(N2, k, Kh, Z + c, tw2, Xk, Xnk);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK2;
      if (kq < Kh2) {
        V[kq * Kh + c] = Xk[q];
        if (kq != 0 && 2 * kq != N2) V[(N2 - kq) * Kh + c] = Xnk[q];
      }
    }
  }
  __syncthreads();
  // z lines back: line l reads its half spectrum V[l][0 .. Kh)
  const int TK3 = (Kh + KQ - 1) / KQ, nl = s_nlines;
  for (int task = threadIdx.x; task < nl * TK3; task += kZyBlock) {
    const int l = s_lines[task / TK3], g = task % TK3;
    int j[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) j[q] = (g + q * TK3 < Kh) ? g + q * TK3 : 0;
    const Cx<T>* v = V + l * Kh;
    T xj[KQ], xnj[KQ];
    irdft_pair_outputs<T, KQ>(N3, j, 1, v + 1, v[0].re, (N3 & 1) ? T(0) : v[N3 / 2].re, tw3, xj, xnj);
    T* x = plane + (long)l * N3;
    T* ac = acc ? acc + (long)l * N3 : nullptr;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int jq = g + q * TK3;
      if (jq < Kh) {
        x[jq] = xj[q];
        if (ac) ac[jq] += xj[q];
        if (jq != 0 && 2 * jq != N3) {
          x[N3 - jq] = xnj[q];
          if (ac) ac[N3 - jq] += xnj[q];
        }
      }
    }
  }
}

// ---- launchers.  K = mesh dimensions, tw = (cos, sin) tables of K[0], K[1], K[2] back to back.
#define KQ_SWITCH(CALL) { constexpr int KQ = 2; constexpr int JS = 1; CALL; }

template <class T>
bool launch_dft_z(hipStream_t st, const int K[3], const T* tw, T* mesh, T* spec, int inverse, int nb, long mesh_stride,
                  long spec_stride, T* accum) {
  const int N = K[2], nlines = K[0] * K[1], H = (N - 1) / 2, TK = dft_tasks(N, dft_kq());
  const int NL = dft_cols(N, dft_kq(), sizeof(Cx<T>) * (size_t)H + 2 * sizeof(T), sizeof(Cx<T>) * (size_t)N);
  const size_t sh = sizeof(Cx<T>) * (size_t)(N + H * NL) + sizeof(T) * 2 * (size_t)NL;
  const Cx<T>* t2 = reinterpret_cast<const Cx<T>*>(tw) + K[0] + K[1];
  const dim3 grid((nlines + NL - 1) / NL, nb);
  const long ss = spec_stride / 2;       // strides are given in reals; the kernels index complex numbers
  if (inverse) {
    KQ_SWITCH((k_dft_z_c2r<T, KQ, JS><<<grid, kDftBlock, sh, st>>>(N, nlines, NL, TK, reinterpret_cast<const Cx<T>*>(spec), mesh, t2, mesh_stride, ss,
                                                              nb == 1 ? accum : nullptr)))
    return accum != nullptr && nb == 1;
  }
  KQ_SWITCH((k_dft_z_r2c<T, KQ, JS><<<grid, kDftBlock, sh, st>>>(N, nlines, NL, TK, mesh, reinterpret_cast<Cx<T>*>(spec), t2, mesh_stride, ss)))
  return false;
}
template <class T>
void launch_dft_y(hipStream_t st, const int K[3], const T* tw, T* spec, int inverse, int nb, long spec_stride) {
  const int N = K[1], Kh = K[2] / 2 + 1, H = (N - 1) / 2, TK = dft_tasks(N, dft_kq());
  const int NC = dft_cols(N, dft_kq(), sizeof(PairCx<T>) * (size_t)H + 2 * sizeof(Cx<T>), sizeof(Cx<T>) * (size_t)N);
  const size_t sh = sizeof(PairCx<T>) * (size_t)(H * NC) + sizeof(Cx<T>) * (size_t)(N + 2 * NC);
  const Cx<T>* t1 = reinterpret_cast<const Cx<T>*>(tw) + K[0];
  const dim3 grid((Kh + NC - 1) / NC, K[0], nb);
  Cx<T>* sp = reinterpret_cast<Cx<T>*>(spec);
  const long ss = spec_stride / 2;
  if (inverse) {
    KQ_SWITCH((k_dft_strided<T, +1, KQ, JS><<<grid, kDftBlock, sh, st>>>(N, Kh, NC, TK, (long)Kh, (long)K[1] * Kh, sp, t1, ss)))
  } else {
    KQ_SWITCH((k_dft_strided<T, -1, KQ, JS><<<grid, kDftBlock, sh, st>>>(N, Kh, NC, TK, (long)Kh, (long)K[1] * Kh, sp, t1, ss)))
  }
}
template <class T>
void launch_dft_x_conv(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, double* energies,
                       int slot, int nb, long spec_stride) {
  const int N = K[0], Kh = K[2] / 2 + 1, H = (N - 1) / 2, TK = dft_tasks(N, dft_kq());
  const int NC = dft_cols(N, dft_kq(), sizeof(PairCx<T>) * (size_t)H + sizeof(Cx<T>) * (size_t)(2 + N), sizeof(Cx<T>) * (size_t)N);
  const size_t sh = sizeof(PairCx<T>) * (size_t)(H * NC) + sizeof(Cx<T>) * (size_t)(N + 2 * NC + N * NC);
  const dim3 grid((Kh + NC - 1) / NC, K[1], nb);
  const XConvArgs<T> xa{N, Kh, NC, TK, (long)K[1] * Kh, (long)Kh, K[2], reinterpret_cast<Cx<T>*>(spec), tabs,
                        reinterpret_cast<const Cx<T>*>(tw), energies, slot, spec_stride / 2};
  KQ_SWITCH((k_dft_x_conv<T, KQ, JS><<<grid, kDftBlock, sh, st>>>(xa)))
}
// x pass of the typed dispersion meshes: spec holds mix.nt type spectra, spec_stride (reals) apart
template <class T>
void launch_dft_x_mix(hipStream_t st, const int K[3], const T* tw, T* spec, const DftTabs<T>& tabs, const MixTab& mix,
                      long spec_stride, double* energies, int slot) {
  const int N = K[0], Kh = K[2] / 2 + 1, H = (N - 1) / 2, TK = dft_tasks(N, dft_kq());
  const int ntp = mix.nt <= 1 ? 1 : (mix.nt == 2 ? 2 : 4);
  int NC = dft_cols(N, dft_kq(), sizeof(PairCx<T>) * (size_t)H + sizeof(Cx<T>) * (size_t)(2 + N), sizeof(Cx<T>) * (size_t)N);
  NC -= NC % ntp;
  if (NC < ntp) NC = ntp;
  const int NCt = NC / ntp;
  const size_t sh = sizeof(PairCx<T>) * (size_t)(H * NC) + sizeof(Cx<T>) * (size_t)(N + 2 * NC + N * NC);
  const dim3 grid((Kh + NCt - 1) / NCt, K[1], 1);
#define XMIX(NTP)                                                                                                          \
  k_dft_x_mix<T, 2, 1, NTP><<<grid, kDftBlock, sh, st>>>(N, Kh, NC, TK, (long)K[1] * Kh, (long)Kh, K[2],                    \
                                                         reinterpret_cast<Cx<T>*>(spec), tabs, mix,                        \
                                                         reinterpret_cast<const Cx<T>*>(tw), energies, slot, spec_stride / 2)
  if (ntp == 1) XMIX(1); else if (ntp == 2) XMIX(2); else XMIX(4);
#undef XMIX
}
// the two plane kernels in place of launch_dft_z + launch_dft_y (forward) / launch_dft_y + launch_dft_z (inverse)
template <class T>
bool dft_zy_fits(const int K[3]) {
  static const bool off = [] { const char* e = getenv("ADMP_DFT_PLANES"); return e && atoi(e) == 0; }();
  // two workgroups per plane must fit the chip in one round (one workgroup per CU): with more planes the separate passes win
  return !off && 2 * K[0] <= 256 && ZyLayout<T>(K[1], K[2]).total + 2048 <= 160 * 1024;      // (+ the kernels' static LDS)
}
// can the forward plane kernel build its planes from the sites (zy_plane_spread)?  double precision only: LDS float atomics
// run at a twentieth of the f64 rate on this chip (tools/ubench/lds_atomics.hip)
template <class T>
bool dft_zy_spread_fits(const int K[3], int na) {
  static const int mx = [] { const char* e = getenv("ADMP_FUSE_SPREAD_MAX"); return e ? atoi(e) : 8192; }();
  return sizeof(T) == 8 && na > 0 && na <= mx && na <= kZySpreadMaxAtoms && dft_zy_fits<T>(K) && zy_spread_sub<T>(K[1], K[2]) >= 16;
}
template <class T>
bool launch_dft_zy(hipStream_t st, const int K[3], const T* tw, T* mesh, T* spec, int inverse, int nb, long mesh_stride,
                   long spec_stride, T* accum, const PlaneSpread<T>* sp) {
  const size_t sh = ZyLayout<T>(K[1], K[2]).total;
  const Cx<T>* t1 = reinterpret_cast<const Cx<T>*>(tw) + K[0];
  const Cx<T>* t2 = t1 + K[1];
  const dim3 grid(K[0], nb, 2);                  // two workgroups per plane
  static size_t attr_set[2] = {0, 0};            // more than 64 KB of dynamic LDS has to be asked for (per kernel and size)
  if (inverse) {
    auto kern = k_dft_yz_inv<T, 2>;
    if (attr_set[1] < sh) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr_set[1] = sh; }
    kern<<<grid, kZyBlock, sh, st>>>(K[1], K[2], reinterpret_cast<const Cx<T>*>(spec), mesh, t1, t2, mesh_stride, spec_stride / 2,
                                     nb == 1 ? accum : nullptr);
    return accum != nullptr && nb == 1;
  }
  if (sp) {
    static size_t attr_sp = 0;
    auto kern = k_dft_zy_fwd<T, 2, true>;
    if (attr_sp < sh) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr_sp = sh; }
    kern<<<grid, kZyBlock, sh, st>>>(K[1], K[2], mesh, reinterpret_cast<Cx<T>*>(spec), t1, t2, mesh_stride, spec_stride / 2, *sp);
    return false;
  }
  auto kern = k_dft_zy_fwd<T, 2, false>;
  if (attr_set[0] < sh) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr_set[0] = sh; }
  kern<<<grid, kZyBlock, sh, st>>>(K[1], K[2], mesh, reinterpret_cast<Cx<T>*>(spec), t1, t2, mesh_stride, spec_stride / 2,
                                   PlaneSpread<T>());
  return false;
}
#undef KQ_SWITCH
#define INST(T)                                                                                   \
  template bool dft_zy_fits<T>(const int*);                                                       \
  template bool dft_zy_spread_fits<T>(const int*, int);                                           \
  template bool launch_dft_zy<T>(hipStream_t, const int*, const T*, T*, T*, int, int, long, long, T*, const PlaneSpread<T>*); \
  template bool launch_dft_z<T>(hipStream_t, const int*, const T*, T*, T*, int, int, long, long, T*); \
  template void launch_dft_y<T>(hipStream_t, const int*, const T*, T*, int, int, long);           \
  template void launch_dft_x_conv<T>(hipStream_t, const int*, const T*, T*, const DftTabs<T>&, double*, int, int, long); \
  template void launch_dft_x_mix<T>(hipStream_t, const int*, const T*, T*, const DftTabs<T>&, const MixTab&, long, double*, int);
INST(float)
INST(double)
#undef INST

}  // namespace admp
