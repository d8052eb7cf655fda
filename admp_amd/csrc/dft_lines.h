// Pieces of the direct-DFT line kernels shared between dft_kernels.hip and the kernels that run an x pass next to other
// work (pair_kernels.hip k_xconv_pair): block size, task helpers, the pair-sum loader and the body of the x pass.
#pragma once
#include "dft_math.h"
#include "launch.h"
#include "reduce.h"

namespace admp {

#ifndef ADMP_DFT_BLOCK
#define ADMP_DFT_BLOCK 256
#endif
constexpr int kDftBlock = ADMP_DFT_BLOCK;
constexpr size_t kDftLdsBudget = 60 * 1024;

// Two output pairs per thread (they share one read of the pair sums), one thread per output set.  Round 2 measured the
// alternatives on the 97^3 f64 mesh and dropped them (variants in the history): 1 or 4 output pairs per thread, and two
// lanes per output set (each summing half of the pair positions: x pass 36.8 -> 44.4 us) -- more, thinner threads do not
// help these latency-bound passes.  The kernels keep the two template parameters; one instantiation is compiled.
static inline int dft_kq() { return 2; }
static inline int dft_js() { return 1; }
// thread-tasks per line and lines (columns) per block
static inline int dft_tasks(int N, int KQ) { return (N / 2 + 1 + KQ - 1) / KQ; }
static inline int dft_cols(int N, int KQ, size_t bytes_per_col, size_t fixed_bytes) {
  int nc = (kDftBlock / dft_js()) / dft_tasks(N, KQ);
  if (nc < 1) nc = 1;
  while (nc > 1 && fixed_bytes + bytes_per_col * nc > kDftLdsBudget) --nc;
  return nc;
}

extern __shared__ __align__(32) unsigned char dft_smem[];

// task index of a thread and which half of the pair positions it sums (JS = 2: lanes 2i, 2i+1 share task i)
template <int JS>
__device__ __forceinline__ void dft_task_of_thread(int& tid, int& half) {
  tid = JS == 2 ? (int)(threadIdx.x >> 1) : (int)threadIdx.x;
  half = JS == 2 ? (int)(threadIdx.x & 1) : 0;
}
template <class T>
__device__ __forceinline__ T pair_lane_sum(T v) { return v + __shfl_xor(v, 1, 64); }

// dft_pair_core / real_pair_sums over this thread's share of the positions, shares combined: every lane gets the full outputs
template <class T, int SIGN, int KQ, int JS>
__device__ __forceinline__ void dft_pair_outputs_js(int N, const int* k, int stride, const PairCx<T>* ab, Cx<T> x0, Cx<T> xn,
                                                    const Cx<T>* tw, int half, Cx<T>* Xk, Cx<T>* Xnk) {
  T Are[KQ], Aim[KQ], Bre[KQ], Bim[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) Are[q] = Aim[q] = Bre[q] = Bim[q] = T(0);
  const int H = (N - 1) / 2, mid = JS == 2 ? dft_split(N) : H;
  dft_pair_partial<T, KQ>(N, k, [=](int j) { return ab[j * stride]; }, tw, half ? mid : 0, half ? H : mid, Are, Aim, Bre, Bim);
  if (JS == 2) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      Are[q] = pair_lane_sum(Are[q]); Aim[q] = pair_lane_sum(Aim[q]);
      Bre[q] = pair_lane_sum(Bre[q]); Bim[q] = pair_lane_sum(Bim[q]);
    }
  }
  dft_pair_finish<T, SIGN, KQ>(N, k, x0, xn, Are, Aim, Bre, Bim, Xk, Xnk);
}
template <class T, int KQ, int JS>
__device__ __forceinline__ void real_pair_sums_js(int N, const int* k, int stride, const Cx<T>* p, const Cx<T>* tw, int half,
                                                  T* P, T* R) {
#pragma unroll
  for (int q = 0; q < KQ; ++q) P[q] = R[q] = T(0);
  const int H = (N - 1) / 2, mid = JS == 2 ? dft_split(N) : H;
  real_pair_partial<T, KQ>(N, k, stride, p, tw, half ? mid : 0, half ? H : mid, P, R);
  if (JS == 2) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) { P[q] = pair_lane_sum(P[q]); R[q] = pair_lane_sum(R[q]); }
  }
}

// stage a tile of NC strided complex lines as pair sums: element (j, c) at spec[base + j * jstride + c]
template <class T>
__device__ __forceinline__ void load_pairs(int N, int NC, int nca, const Cx<T>* __restrict__ spec, long base, long jstride,
                                           PairCx<T>* ab, Cx<T>* x0, Cx<T>* xn) {
  const int H = (N - 1) / 2;
  for (int t = threadIdx.x; t < H * NC; t += kDftBlock) {
    const int jj = t / NC, c = t - jj * NC;
    PairCx<T> v{T(0), T(0), T(0), T(0)};
    if (c < nca) {
      const Cx<T> a = spec[base + (long)(1 + jj) * jstride + c], b = spec[base + (long)(N - 1 - jj) * jstride + c];
      v = PairCx<T>{a.re + b.re, a.im + b.im, a.re - b.re, a.im - b.im};
    }
    ab[t] = v;
  }
  if (threadIdx.x < NC) {
    const int c = threadIdx.x;
    Cx<T> a{T(0), T(0)}, b{T(0), T(0)};
    if (c < nca) {
      a = spec[base + c];
      if ((N & 1) == 0) b = spec[base + (long)(N / 2) * jstride + c];
    }
    x0[c] = a;
    xn[c] = b;
  }
}

// ---- x lines: forward, multiply by G (accumulating sum w G |S|^2, recip.py:400-414 / pme.py:240), inverse; in place
// (the body of k_dft_x_conv as a device function: bx, by, bz = the block's tile, y row and batch channel -- also run by the
// workgroups of k_xconv_pair, pair_kernels.hip, next to the workgroups of a real-space field kernel)
template <class T>
struct XConvArgs {
  int N, ncols, NC, TK;
  long jstride, fixstride;
  int K3;
  Cx<T>* spec;
  DftTabs<T> tabs;
  const Cx<T>* twg;
  double* energies;
  int slot;
  long spec_stride;
};
template <class T, int KQ, int JS>
__device__ __forceinline__ void dft_x_conv_body(const XConvArgs<T>& xa, int bx, int by, int bz) {
  const int N = xa.N, ncols = xa.ncols, NC = xa.NC, TK = xa.TK, K3 = xa.K3, slot = xa.slot;
  const long jstride = xa.jstride, fixstride = xa.fixstride;
  const Cx<T>* __restrict__ twg = xa.twg;
  double* energies = xa.energies;
  Cx<T>* __restrict__ spec = xa.spec + bz * xa.spec_stride;
  const T* __restrict__ gtab = xa.tabs.p[bz];
  const int H = (N - 1) / 2, Kh = N / 2 + 1;
  PairCx<T>* ab = reinterpret_cast<PairCx<T>*>(dft_smem);   // [H][NC]
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(ab + H * NC);          // [N]
  Cx<T>* x0 = tw + N;                                         // [NC]
  Cx<T>* xn = x0 + NC;                                        // [NC]
  Cx<T>* S = xn + NC;                                         // [N][NC]
  const int col0 = bx * NC;
  const int nca = min(NC, ncols - col0);
  const long base = (long)by * fixstride + col0;
  for (int t = threadIdx.x; t < N; t += kDftBlock) tw[t] = twg[t];
  // the G values this thread multiplies with after the forward transform: fetched now, so that their latency hides
  // behind the transform instead of sitting between two barriers (H * NC < KQ * kDftBlock by construction of NC)
  T Gp[KQ][2], G0 = T(0), Gn = T(0);
#pragma unroll
  for (int u = 0; u < KQ; ++u) {
    Gp[u][0] = Gp[u][1] = T(0);
    const int t = threadIdx.x + u * kDftBlock;
    if (t < H * NC) {
      const int jj = t / NC, cc = t - jj * NC;
      if (cc < nca) {
        Gp[u][0] = gtab[base + (long)(1 + jj) * jstride + cc];
        Gp[u][1] = gtab[base + (long)(N - 1 - jj) * jstride + cc];
      }
    }
  }
  if (threadIdx.x < nca) {
    G0 = gtab[base + threadIdx.x];
    if ((N & 1) == 0) Gn = gtab[base + (long)(N / 2) * jstride + threadIdx.x];
  }
  load_pairs<T>(N, NC, nca, spec, base, jstride, ab, x0, xn);
  __syncthreads();
  int tid, half;
  dft_task_of_thread<JS>(tid, half);
  const int g = tid / NC, c = tid - g * NC;
  const bool task = g < TK && c < nca;
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
#pragma unroll
  for (int u = 0; u < KQ; ++u) {
    const int t = threadIdx.x + u * kDftBlock;
    if (t >= H * NC) continue;
    const int jj = t / NC, cc = t - jj * NC;
    PairCx<T> v{T(0), T(0), T(0), T(0)};
    if (cc < nca) {
      const int k1 = 1 + jj, k2 = N - 1 - jj, kz = col0 + cc;
      const T G1 = Gp[u][0], G2 = Gp[u][1];
      const Cx<T> s1 = S[k1 * NC + cc], s2 = S[k2 * NC + cc];
      const double w = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
      e += w * ((double)G1 * ((double)s1.re * s1.re + (double)s1.im * s1.im) +
                (double)G2 * ((double)s2.re * s2.re + (double)s2.im * s2.im));
      const T ar = G1 * s1.re, ai = G1 * s1.im, br = G2 * s2.re, bi = G2 * s2.im;
      v = PairCx<T>{ar + br, ai + bi, ar - br, ai - bi};
    }
    ab[t] = v;
  }
  if (threadIdx.x < NC) {
    const int cc = threadIdx.x;
    Cx<T> a{T(0), T(0)}, b{T(0), T(0)};
    if (cc < nca) {
      const int kz = col0 + cc;
      const double w = (kz == 0 || ((K3 & 1) == 0 && kz == K3 / 2)) ? 0.5 : 1.0;
      const Cx<T> s0 = S[cc];
      e += w * (double)G0 * ((double)s0.re * s0.re + (double)s0.im * s0.im);
      a = Cx<T>{G0 * s0.re, G0 * s0.im};
      if ((N & 1) == 0) {
        const Cx<T> sn = S[(N / 2) * NC + cc];
        e += w * (double)Gn * ((double)sn.re * sn.re + (double)sn.im * sn.im);
        b = Cx<T>{Gn * sn.re, Gn * sn.im};
      }
    }
    x0[cc] = a;
    xn[cc] = b;
  }
  __syncthreads();
  if (task) {
    Cx<T> Xk[KQ], Xnk[KQ];
    dft_pair_outputs_js<T, +1, KQ, JS>(N, k, NC, ab + c, x0[c], xn[c], tw, half, Xk, Xnk);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int kq = g + q * TK;
      if (kq < Kh) {
        if (JS == 1 || half == 0) spec[base + (long)kq * jstride + c] = Xk[q];
        if ((JS == 1 || half == 1) && kq != 0 && 2 * kq != N) spec[base + (long)(N - kq) * jstride + c] = Xnk[q];
      }
    }
  }
  e = block_reduce_sum<kDftBlock>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[slot], e);
}

}  // namespace admp
