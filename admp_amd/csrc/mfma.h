// Matrix-core wrappers shared by the direct-DFT kernels (dft_mfma.hip, pfa_kernels.hip).
// Operand maps (cdna_hip_programming.md "Fragment layout"): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15];
// C/D column = lane & 15, row = (lane >> 4) + 4 r for f64, 4 (lane >> 4) + r for f32, r = 0..3.
#pragma once
#include <hip/hip_runtime.h>

namespace admp {

template <class T>
struct Mfma;
template <>
struct Mfma<double> {
  typedef double Acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
  typedef float Acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// twiddle index of a lane for the cosine / sine matrices of a pair-symmetric direct DFT of length N:
// output index i (0-based: k - 1), position kk (0-based: jj) -> (1 + i)(1 + kk) mod N; advanced by 4 positions per step
struct TwIdx {
  int m, inc, N;
  // i < (N - 1) / 2 and kk0 < 4: both products stay below 2 N -- reduced by subtraction (a 64-bit modulo is ~300 VALU
  // instructions, and this constructor runs once per work item of the matrix-core stages)
  __device__ __forceinline__ TwIdx(int i, int kk0, int N_) : N(N_) {
    m = (1 + i) * (1 + kk0);
    while (m >= N_) m -= N_;
    inc = 4 * (1 + i);
    while (inc >= N_) inc -= N_;
  }
  __device__ __forceinline__ void step() { m += inc; if (m >= N) m -= N; }
};

}  // namespace admp
