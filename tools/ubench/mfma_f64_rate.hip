// Micro-benchmark: issue rate and dependent latency of v_mfma_f64_16x16x4_f64 (and the f32 16x16x4 form) on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k64(double* out, int iters, double a, double b) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == -1.0) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(double* out, int iters, float a, float b) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == -1.0f) out[0] = s;
}
template <class F>
void run(const char* name, F launch, int nacc, int waves_per_simd) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  launch(10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch(iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  const double per = ms * 1e6 / ((double)iters * nacc * waves_per_simd);     // ns per MFMA per SIMD
  printf("%-28s %d accumulators, %d waves/SIMD: %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", name, nacc, waves_per_simd, per, per * 2.4);
}
int main() {
  double* out; hipMalloc(&out, 64);
  // one wave per SIMD: 256 CUs x 4 waves = 256 blocks of 256 threads
  run("f64 16x16x4", [&](int it) { k64<1><<<256, 256>>>(out, it, 1.0, 2.0); }, 1, 1);
  run("f64 16x16x4", [&](int it) { k64<2><<<256, 256>>>(out, it, 1.0, 2.0); }, 2, 1);
  run("f64 16x16x4", [&](int it) { k64<4><<<256, 256>>>(out, it, 1.0, 2.0); }, 4, 1);
  run("f64 16x16x4", [&](int it) { k64<4><<<512, 256>>>(out, it, 1.0, 2.0); }, 4, 2);
  run("f32 16x16x4", [&](int it) { k32<1><<<256, 256>>>(out, it, 1.0f, 2.0f); }, 1, 1);
  run("f32 16x16x4", [&](int it) { k32<4><<<256, 256>>>(out, it, 1.0f, 2.0f); }, 4, 1);
  return 0;
}
