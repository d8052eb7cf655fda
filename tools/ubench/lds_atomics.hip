// Micro-benchmark: LDS float-atomic throughput on gfx950 (ds_add_f32 / ds_add_f64 / ds_add_u32 vs plain RMW).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomics.hip -o /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <class T, int MODE>   // MODE 0: atomicAdd, 1: plain read-add-write (lane-private cells)
__global__ __launch_bounds__(256) void k(const int* idx, T* out, int iters) {
  __shared__ T tile[4096];
  for (int t = threadIdx.x; t < 4096; t += 256) tile[t] = T(0);
  __syncthreads();
  int base = idx[threadIdx.x + 256 * (blockIdx.x & 7)];
  T v = T(1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int a = (base + u * 37 + it * 101) & 4095;
      if (MODE == 0) atomicAdd(&tile[a], v);
      else { int b = (threadIdx.x + 256 * ((u + it) & 15)) & 4095; tile[b] += v; }
    }
  }
  __syncthreads();
  T s = 0;
  for (int t = threadIdx.x; t < 4096; t += 256) s += tile[t];
  if (s == T(-1)) out[0] = s;
}

template <class T, int MODE>
void run(const char* name, const int* didx, int pattern) {
  T* out; hipMalloc(&out, 64);
  const int blocks = 256 * 8, iters = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<T, MODE><<<blocks, 256>>>(didx, out, 4);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<T, MODE><<<blocks, 256>>>(didx, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 256 * iters * 16;
  printf("%-28s pattern %d: %8.3f ms  %7.2f Gop/s  = %.3f lane-ops/clk/CU @2.4GHz\n", name, pattern, ms, ops / ms * 1e-6,
         ops / (ms * 1e-3) / 2.4e9 / 256);
  hipFree(out);
}

int main() {
  for (int pattern = 0; pattern < 2; ++pattern) {
    std::vector<int> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = pattern == 0 ? i : (int)((i * 2654435761u) >> 7) & 4095;   // 0: consecutive, 1: scattered
    int* d; hipMalloc(&d, 2048 * 4); hipMemcpy(d, h.data(), 2048 * 4, hipMemcpyHostToDevice);
    run<float, 0>("ds_add_f32 (atomicAdd)", d, pattern);
    run<double, 0>("ds_add_f64 (atomicAdd)", d, pattern);
    run<unsigned, 0>("ds_add_u32 (atomicAdd)", d, pattern);
    run<float, 1>("plain RMW f32", d, pattern);
    run<double, 1>("plain RMW f64", d, pattern);
    hipFree(d);
  }
  return 0;
}
