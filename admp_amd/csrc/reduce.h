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

// XCD-aware block order (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, each with
// its own L2; remapping block b -> (b % 8) * ceil(n/8) + b / 8 gives every XCD one contiguous range of rows, so the
// partner rows gathered by neighbouring workgroups are shared in that XCD's L2 instead of being fetched by all eight.
__device__ __forceinline__ long xcd_block(unsigned b, unsigned n) {
  const unsigned per = (n + 7u) / 8u;
  if ((b >> 3) >= per) return -1;          // grid sized for more rows than there are (device-side row count)
  const unsigned m = (b & 7u) * per + (b >> 3);
  return m < n ? (long)m : -1;
}
static inline unsigned xcd_grid(unsigned n) { return ((n + 7u) / 8u) * 8u; }

// t / d for 0 <= t < 2^22 with inv = 1.0f / d: an integer division by a run-time divisor costs ~40 VALU instructions (a
// 64-bit one ~300), which adds up in per-element sweeps
__device__ __forceinline__ int fast_div(int t, int d, float inv) {
  int q = (int)((float)t * inv);
  if (q * d > t) --q;
  else if ((q + 1) * d <= t) ++q;
  return q;
}

}  // namespace admp
