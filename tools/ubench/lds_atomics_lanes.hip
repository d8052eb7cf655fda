// Micro-benchmark: does the cost of a wave-level LDS f64 atomic add depend on the number of ACTIVE lanes and on the address
// pattern?  (decides whether packing more useful lanes per ds_add_f64 pays in the PME spread)
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomics_lanes.hip -o /tmp/lds_lanes
#include <hip/hip_runtime.h>
#include <cstdio>

// pattern 0: lane l -> word l (+ moving offset): conflict free; 1: scattered (hash); 2: stride 17 words;
// 3: 36 active lanes laid out as 6 rows of 6 consecutive words, row pitch `pitch`
// 4 / 5 / 6: scattered, but the words of each contiguous group of 16 / 32 / 64 lanes are distinct modulo 16 / 32 / 64
// (which grouping and modulus makes a scattered 64-bit atomic conflict free?)
template <int PATTERN, class V = double>
__global__ __launch_bounds__(256) void k(double* out, int iters, int active, int pitch) {
  __shared__ V tile[8192];
  for (int t = threadIdx.x; t < 8192; t += 256) tile[t] = V(0);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int base;
  if (PATTERN == 0) base = lane;
  else if (PATTERN == 1) base = (int)(((threadIdx.x + 1) * 2654435761u) >> 9) & 4095;
  else if (PATTERN == 2) base = lane * 17;
  else if (PATTERN == 4) base = (((int)(((threadIdx.x + 1) * 2654435761u) >> 9) & 4095) & ~15) | (lane & 15);
  else if (PATTERN == 5) base = (((int)(((threadIdx.x + 1) * 2654435761u) >> 9) & 4095) & ~31) | (lane & 31);
  else if (PATTERN == 6) base = (((int)(((threadIdx.x + 1) * 2654435761u) >> 9) & 4095) & ~63) | (lane & 63);
  else base = (lane / 6) * pitch + (lane % 6);
  base += (threadIdx.x >> 6) * 64;
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) atomicAdd(&tile[(base + (PATTERN >= 4 ? u * 128 + it * 64 : u * 97 + it * 13)) & 8191], V(1));
    }
  }
  __syncthreads();
  double s = 0;
  for (int t = threadIdx.x; t < 8192; t += 256) s += (double)tile[t];
  if (s == -1.0) out[0] = s;
}

template <int PATTERN, class V = double>
void run(const char* name, int active, int pitch) {
  double* out; hipMalloc(&out, 64);
  const int blocks = 256 * 8, iters = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<PATTERN, V><<<blocks, 256>>>(out, 4, active, pitch);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<PATTERN, V><<<blocks, 256>>>(out, iters, active, pitch);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)blocks * 4 * iters * 16;          // wave-level instructions
  const double cyc = ms * 1e-3 * 2.4e9 * 256 / winstr;            // CU-cycles per wave instruction (all 4 waves of a WG share a CU)
  printf("%-34s active %2d pitch %2d: %7.3f ms  %6.2f CU-cycles / wave-instr  %6.2f lane-adds/clk/CU\n", name, active, pitch, ms,
         cyc, active / cyc);
  hipFree(out);
}

int main() {
  for (int act : {64, 48, 36, 32, 16, 8}) run<0>("consecutive words", act, 0);
  for (int act : {64, 36, 16}) run<1>("scattered", act, 0);
  for (int act : {64, 36}) run<2>("stride 17 words", act, 0);
  for (int pitch : {16, 17, 20, 21, 22, 24}) run<3>("6 rows x 6 words", 36, pitch);
  // integer atomics (fixed-point accumulation): 64-bit and 32-bit
  run<0, unsigned long long>("u64 consecutive", 64, 0);
  run<1, unsigned long long>("u64 scattered", 64, 0);
  run<4, unsigned long long>("u64 scattered, distinct mod 16 per 16 lanes", 64, 0);
  run<5, unsigned long long>("u64 scattered, distinct mod 32 per 32 lanes", 64, 0);
  run<6, unsigned long long>("u64 scattered, distinct mod 64 per 64 lanes", 64, 0);
  run<0, unsigned>("u32 consecutive", 64, 0);
  run<1, unsigned>("u32 scattered", 64, 0);
  run<0, float>("f32 consecutive", 64, 0);
  run<1, float>("f32 scattered", 64, 0);
  return 0;
}
