// Reciprocal-space kernels (reference admp/recip.py:21-431): B-spline spread of the multipoles onto
// the K1 x K2 x K3 mesh, the k-space multiply with the cached table G_k = 2 D C_k / theta_k^2, and the
// gather that produces dE/dQ and dE/dr from phi = c2r(G S) (the adjoint the reference leaves to jax.grad).
// The 3-D transforms themselves are rocFFT r2c / c2r plans driven from engine.hip.
#include "disp_math.h"
#include "launch.h"
#include "reduce.h"

namespace admp {

constexpr int kRecipBlock = 128;

template <class T>
__device__ __forceinline__ void site_qtot(const Site<T>& s, int lpol, T r[3], T Q[9]) {
  r[0] = s.r[0]; r[1] = s.r[1]; r[2] = s.r[2];
#pragma unroll
  for (int k = 0; k < 9; ++k) Q[k] = s.Q[k];
  if (lpol) { Q[1] += s.U[0]; Q[2] += s.U[1]; Q[3] += s.U[2]; }   // Q_global_tot, admp/pme.py:236
}

// Spread, first version: one thread per atom, 216 hardware float atomics each (Q_mesh.at[].add, recip.py:324-328).
template <class T>
__global__ __launch_bounds__(kRecipBlock) void k_spread(int na, const Site<T>* __restrict__ sites, int lpol,
                                                        RecipGeom<T> g, T* __restrict__ mesh) {
  int i = blockIdx.x * kRecipBlock + threadIdx.x;
  if (i >= na) return;
  T r[3], Q[9];
  site_qtot(sites[i], lpol, r, Q);
  spread_atom(g, r, Q, [&](long idx, T v) { atomicAdd(&mesh[idx], v); });
}

template <class T>
__global__ __launch_bounds__(kRecipBlock) void k_spread_scalar(int na, const T* __restrict__ pos,
                                                               const T* __restrict__ vals, int stride, int chan,
                                                               RecipGeom<T> g, T* __restrict__ mesh) {
  int i = blockIdx.x * kRecipBlock + threadIdx.x;
  if (i >= na) return;
  T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  T Q[9] = {vals[(long)stride * i + chan], 0, 0, 0, 0, 0, 0, 0, 0};
  spread_atom(g, r, Q, [&](long idx, T v) { atomicAdd(&mesh[idx], v); });
}

// G table over the r2c half spectrum [K1][K2][K3/2+1].  Frequencies are assigned axis by axis
// (k_d <- mesh axis d); the reference's meshgrid(kz, kx, ky) (recip.py:339-340) instead puts the
// frequencies of mesh axes (1,0,2) into k-columns (0,1,2), which is the same thing whenever
// K1 = K2 and |a| = |b| (orthorhombic) -- the only regime in which the reference is self-consistent.
template <class T>
__global__ void k_gtab(int K0, int K1, int K2, const double* __restrict__ binv, double volume, double kappa,
                       int which, T* __restrict__ gtab) {
  const int nh = K2 / 2 + 1;
  const long n = (long)K0 * K1 * nh;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const int i2 = (int)(t % nh);
    const int i1 = (int)((t / nh) % K1);
    const int i0 = (int)(t / ((long)nh * K1));
    const int m0 = signed_freq(i0, K0), m1 = signed_freq(i1, K1), m2 = signed_freq(i2, K2);
    const double tp = 6.283185307179586;
    double kx = tp * (m0 * binv[0] + m1 * binv[3] + m2 * binv[6]);
    double ky = tp * (m0 * binv[1] + m1 * binv[4] + m2 * binv[7]);
    double kz = tp * (m0 * binv[2] + m1 * binv[5] + m2 * binv[8]);
    double ksq = kx * kx + ky * ky + kz * kz;
    double th = theta_k_1d(m0, K0) * theta_k_1d(m1, K1) * theta_k_1d(m2, K2);
    double G;
    if (which == 1) {   // Ck_1 (recip.py:434-435), gamma point excluded (recip.py:413-415), x DIELECTRIC (:424)
      G = (t == 0) ? 0.0
                   : 2.0 * kDielectric * (tp / volume / ksq) * exp(-ksq / (4.0 * kappa * kappa)) / (th * th);
    } else {            // dispersion: gamma point included, no DIELECTRIC (recip.py:416-426)
      G = 2.0 * disp_ck(which, ksq, kappa, volume) / (th * th);
    }
    gtab[t] = (T)G;
  }
}

// spec <- G * spec, E += sum over the FULL spectrum of (G/2)|S|^2 (interior half-spectrum planes count twice)
template <class T>
__global__ __launch_bounds__(256) void k_kspace(int K0, int K1, int K2, const T* __restrict__ gtab,
                                                T* __restrict__ spec, double* energies, int slot) {
  const int nh = K2 / 2 + 1;
  const long n = (long)K0 * K1 * nh;
  double e = 0.0;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
    const int i2 = (int)(t % nh);
    const T G = gtab[t];
    T re = spec[2 * t], im = spec[2 * t + 1];
    const double w = (i2 == 0 || ((K2 & 1) == 0 && i2 == K2 / 2)) ? 0.5 : 1.0;
    e += w * (double)G * ((double)re * re + (double)im * im);
    spec[2 * t] = re * G;
    spec[2 * t + 1] = im * G;
  }
  e = block_reduce_sum<256>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[slot], e);
}

template <class T>
__global__ __launch_bounds__(kRecipBlock) void k_gather(int na, const Site<T>* __restrict__ sites, int lpol,
                                                        RecipGeom<T> g, const T* __restrict__ phi,
                                                        T* __restrict__ pot, T* __restrict__ grad) {
  int i = blockIdx.x * kRecipBlock + threadIdx.x;
  if (i >= na) return;
  T r[3], Q[9], F[NF];
  site_qtot(sites[i], lpol, r, Q);
  gather_atom(g, r, [&](long idx) { return phi[idx]; }, F);
  T P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gr[3] = {0, 0, 0};
  unfold_potential(g, Q, F, P, gr);
#pragma unroll
  for (int k = 0; k < 9; ++k) pot[9 * i + k] += P[k];
  if (grad) {
    grad[3 * i] += gr[0]; grad[3 * i + 1] += gr[1]; grad[3 * i + 2] += gr[2];
  }
}

template <class T>
__global__ __launch_bounds__(kRecipBlock) void k_gather_field(int na, const Site<T>* __restrict__ sites,
                                                              RecipGeom<T> g, const T* __restrict__ phi,
                                                              T* __restrict__ fld) {
  int i = blockIdx.x * kRecipBlock + threadIdx.x;
  if (i >= na) return;
  T r[3] = {sites[i].r[0], sites[i].r[1], sites[i].r[2]}, gx[3];
  gather_atom_field(g, r, [&](long idx) { return phi[idx]; }, gx);
  fld[3 * i] = gx[0]; fld[3 * i + 1] = gx[1]; fld[3 * i + 2] = gx[2];
}

template <class T>
__global__ __launch_bounds__(kRecipBlock) void k_gather_scalar(int na, const T* __restrict__ pos,
                                                               const T* __restrict__ vals, int stride, int chan,
                                                               RecipGeom<T> g, const T* __restrict__ phi,
                                                               T* __restrict__ grad) {
  int i = blockIdx.x * kRecipBlock + threadIdx.x;
  if (i >= na) return;
  T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}, gx[3];
  // g.Aop holds the true Jacobian here (see launch_gather_scalar): gx = Jac . F1, dE/dx = q gx
  gather_atom_field(g, r, [&](long idx) { return phi[idx]; }, gx);
  const T q = vals[(long)stride * i + chan];
  grad[3 * i] += q * gx[0]; grad[3 * i + 1] += q * gx[1]; grad[3 * i + 2] += q * gx[2];
}

static inline int nblk(int n, int b) { return (n + b - 1) / b; }

template <class T>
void launch_spread(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, T* mesh) {
  k_spread<T><<<nblk(na, kRecipBlock), kRecipBlock, 0, st>>>(na, sites, lpol, g, mesh);
}
template <class T>
void launch_spread_scalar(hipStream_t st, int na, const T* pos, const T* vals, int stride, int chan,
                          const RecipGeom<T>& g, T* mesh) {
  k_spread_scalar<T><<<nblk(na, kRecipBlock), kRecipBlock, 0, st>>>(na, pos, vals, stride, chan, g, mesh);
}
template <class T>
void launch_gtab(hipStream_t st, const int K[3], const double* box_inv, double volume, double kappa, int which, T* gtab) {
  const long n = (long)K[0] * K[1] * (K[2] / 2 + 1);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  k_gtab<T><<<blocks, 256, 0, st>>>(K[0], K[1], K[2], box_inv, volume, kappa, which, gtab);
}
template <class T>
void launch_kspace(hipStream_t st, const int K[3], const T* gtab, T* spec, double* energies, int slot) {
  const long n = (long)K[0] * K[1] * (K[2] / 2 + 1);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  k_kspace<T><<<blocks, 256, 0, st>>>(K[0], K[1], K[2], gtab, spec, energies, slot);
}
template <class T>
void launch_gather(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, const T* phi, T* pot,
                   T* grad) {
  k_gather<T><<<nblk(na, kRecipBlock), kRecipBlock, 0, st>>>(na, sites, lpol, g, phi, pot, grad);
}
template <class T>
void launch_gather_field(hipStream_t st, int na, const Site<T>* sites, const RecipGeom<T>& g, const T* phi, T* fld) {
  k_gather_field<T><<<nblk(na, kRecipBlock), kRecipBlock, 0, st>>>(na, sites, g, phi, fld);
}
template <class T>
void launch_gather_scalar(hipStream_t st, int na, const T* pos, const T* vals, int stride, int chan,
                          const RecipGeom<T>& g, const T* phi, T* grad) {
  // a scalar site's position gradient is q * Jac . F1: reuse the first-derivative gather with Aop := Jac
  RecipGeom<T> gj = g;
  for (int k = 0; k < 9; ++k) gj.Aop[k] = g.Jac[k];
  k_gather_scalar<T><<<nblk(na, kRecipBlock), kRecipBlock, 0, st>>>(na, pos, vals, stride, chan, gj, phi, grad);
}

#define INST(T)                                                                                                       \
  template void launch_spread<T>(hipStream_t, int, const Site<T>*, int, const RecipGeom<T>&, T*);                     \
  template void launch_spread_scalar<T>(hipStream_t, int, const T*, const T*, int, int, const RecipGeom<T>&, T*);     \
  template void launch_gtab<T>(hipStream_t, const int*, const double*, double, double, int, T*);                      \
  template void launch_kspace<T>(hipStream_t, const int*, const T*, T*, double*, int);                                \
  template void launch_gather<T>(hipStream_t, int, const Site<T>*, int, const RecipGeom<T>&, const T*, T*, T*);       \
  template void launch_gather_field<T>(hipStream_t, int, const Site<T>*, const RecipGeom<T>&, const T*, T*);          \
  template void launch_gather_scalar<T>(hipStream_t, int, const T*, const T*, int, int, const RecipGeom<T>&, const T*, T*);
INST(float)
INST(double)
#undef INST

}  // namespace admp
