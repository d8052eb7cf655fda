// Wave64 / workgroup reductions (gfx950: 64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>

namespace admp {

template <class T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_reduce_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

// order-preserving bit pattern of a non-negative double (for atomicMax on an unsigned word)
__device__ __forceinline__ unsigned long long nonneg_bits(double v) { return (unsigned long long)__double_as_longlong(v); }

// result valid in thread 0
template <int BLOCK>
__device__ __forceinline__ double block_reduce_sum(double v) {
  __shared__ double part[BLOCK / 64];
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();   // protects `part` when called twice in a row
  if (lane == 0) part[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < BLOCK / 64; ++w) r += part[w];
  return r;
}
template <int BLOCK>
__device__ __forceinline__ double block_reduce_max(double v) {
  __shared__ double partm[BLOCK / 64];
  v = wave_reduce_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) partm[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < BLOCK / 64; ++w) r = fmax(r, partm[w]);
  return r;
}

}  // namespace admp
