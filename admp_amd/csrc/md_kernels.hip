// MD-driver kernels (the reference has no integrator): the harmonic bonded terms of the drivers' force
// field (examples/*/mpidwater.xml:16-21, OpenMM's HarmonicBondForce / HarmonicAngleForce: E = k/2 (r - r0)^2, k/2 (theta -
// theta0)^2) as ONE kernel over explicit bond / angle lists, and the two half steps of velocity Verlet as one elementwise
// kernel each.  Round 3's driver did this with ~60 torch launches per step (autograd through acos / norm, elementwise updates).
#include <hip/hip_runtime.h>

#include "launch.h"
#include "reduce.h"

namespace admp {

// items 0 .. nb-1: bonds (i, j; k, r0); items nb .. nb+na-1: angles (i, centre j, k; k_theta, theta0).  grad is ADDED to
// (hardware float atomics: two or three atoms per item); E[0] += bond energy, E[1] += angle energy.
template <class T>
__global__ __launch_bounds__(256) void k_md_bonded(int nb, const int* __restrict__ bidx, const T* __restrict__ bpar, int na,
                                                   const int* __restrict__ aidx, const T* __restrict__ apar,
                                                   const T* __restrict__ pos, Box<T> box, T* __restrict__ grad, double* E) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  double eb = 0.0, ea = 0.0;
  if (t < nb) {
    const int i = bidx[2 * t], j = bidx[2 * t + 1];
    T d[3] = {pos[3 * j] - pos[3 * i], pos[3 * j + 1] - pos[3 * i + 1], pos[3 * j + 2] - pos[3 * i + 2]};
    min_image(box, d);
    const T r = m_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const T k = bpar[2 * t], dr = r - bpar[2 * t + 1];
    eb = 0.5 * (double)k * (double)dr * (double)dr;
    const T s = k * dr / r;                      // dE/dr / r
#pragma unroll
    for (int c = 0; c < 3; ++c) { atomicAdd(&grad[3 * j + c], s * d[c]); atomicAdd(&grad[3 * i + c], -s * d[c]); }
  } else if (t < nb + na) {
    const int a = t - nb;
    const int i = aidx[3 * a], j = aidx[3 * a + 1], k3 = aidx[3 * a + 2];
    T u[3] = {pos[3 * i] - pos[3 * j], pos[3 * i + 1] - pos[3 * j + 1], pos[3 * i + 2] - pos[3 * j + 2]};
    T v[3] = {pos[3 * k3] - pos[3 * j], pos[3 * k3 + 1] - pos[3 * j + 1], pos[3 * k3 + 2] - pos[3 * j + 2]};
    min_image(box, u);
    min_image(box, v);
    const T ru = m_sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), rv = m_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    T c = (u[0] * v[0] + u[1] * v[1] + u[2] * v[2]) / (ru * rv);
    c = c > T(1) ? T(1) : (c < T(-1) ? T(-1) : c);
    const T th = (T)acos((double)c), kt = apar[2 * a], dth = th - apar[2 * a + 1];
    ea = 0.5 * (double)kt * (double)dth * (double)dth;
    // d theta / d u = -(v / (ru rv) - c u / ru^2) / sin theta
    T sn = m_sqrt(T(1) - c * c);
    sn = sn < T(1e-8) ? T(1e-8) : sn;
    const T f = -kt * dth / sn;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const T gu = f * (v[q] / (ru * rv) - c * u[q] / (ru * ru)), gv = f * (u[q] / (ru * rv) - c * v[q] / (rv * rv));
      atomicAdd(&grad[3 * i + q], gu);
      atomicAdd(&grad[3 * k3 + q], gv);
      atomicAdd(&grad[3 * j + q], -(gu + gv));
    }
  }
  eb = block_reduce_sum<256>(eb);
  ea = block_reduce_sum<256>(ea);
  if (threadIdx.x == 0) {
    if (eb != 0.0) atomicAdd(&E[0], eb);
    if (ea != 0.0) atomicAdd(&E[1], ea);
  }
}

// v -= half_dt_acc grad / m (grad = +dE/dr); then, if dt != 0, r += dt v; ekin (optional) += sum m v^2 / 2 AFTER the kick
template <class T>
__global__ __launch_bounds__(256) void k_md_kick_drift(int n, T* __restrict__ pos, T* __restrict__ vel, const T* __restrict__ grad,
                                                       const T* __restrict__ inv_mass, T half_dt_acc, T dt, double* ekin) {
  double ek = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const T im = inv_mass[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T v = vel[3 * i + c] - half_dt_acc * grad[3 * i + c] * im;
      vel[3 * i + c] = v;
      if (dt != T(0)) pos[3 * i + c] += dt * v;
      ek += 0.5 * (double)v * (double)v / (double)im;
    }
  }
  if (ekin) {
    ek = block_reduce_sum<256>(ek);
    if (threadIdx.x == 0) atomicAdd(ekin, ek);
  }
}

template <class T>
void launch_md_bonded(hipStream_t st, int nb, const int* bidx, const T* bpar, int na, const int* aidx, const T* apar, const T* pos,
                      const Box<T>& box, T* grad, double* E) {
  const int n = nb + na;
  if (n > 0) k_md_bonded<T><<<(n + 255) / 256, 256, 0, st>>>(nb, bidx, bpar, na, aidx, apar, pos, box, grad, E);
}
template <class T>
void launch_md_kick_drift(hipStream_t st, int n, T* pos, T* vel, const T* grad, const T* inv_mass, double half_dt_acc, double dt,
                          double* ekin) {
  if (n <= 0) return;
  int blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;      // (<= 1024 atomics on the kinetic-energy word)
  k_md_kick_drift<T><<<blocks, 256, 0, st>>>(n, pos, vel, grad, inv_mass, (T)half_dt_acc, (T)dt, ekin);
}
#define INST(T)                                                                                                              \
  template void launch_md_bonded<T>(hipStream_t, int, const int*, const T*, int, const int*, const T*, const T*, const Box<T>&, \
                                    T*, double*);                                                                            \
  template void launch_md_kick_drift<T>(hipStream_t, int, T*, T*, const T*, const T*, double, double, double*);
INST(float)
INST(double)
#undef INST

}  // namespace admp
