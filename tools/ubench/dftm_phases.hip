// Where does a pass of the matrix-core DFT (admp_amd/csrc/dft_mfma.hip) spend its time?  Launches the y pass of a 97^3 f64
// spectrum with phase probes (ADMP_DFTM_TRACE) and prints, per phase, the mean over blocks and the spread of block start times.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -DADMP_DFTM_TRACE \
//     -Iadmp_amd/csrc -Iinclude tools/ubench/dftm_phases.hip -o /tmp/dftm_phases && /tmp/dftm_phases
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../admp_amd/csrc/dft_mfma.hip"

using namespace admp;

int main() {
  const int K[3] = {97, 97, 97};
  const int N = 97, Kh = 49;
  const size_t nspec = (size_t)97 * 97 * Kh;
  std::vector<double> h(2 * nspec);
  for (size_t i = 0; i < h.size(); ++i) h[i] = std::sin(0.37 * (double)i);
  std::vector<double> tw(2 * 3 * N);
  for (int d = 0; d < 3; ++d)
    for (int m = 0; m < N; ++m) { tw[2 * (d * N + m)] = std::cos(2 * M_PI * m / N); tw[2 * (d * N + m) + 1] = std::sin(2 * M_PI * m / N); }
  double *spec, *twd;
  long long* tr;
  hipMalloc(&spec, h.size() * 8); hipMalloc(&twd, tw.size() * 8);
  const int nblocks = (97 * Kh + 7) / 8;
  hipMalloc(&tr, sizeof(long long) * 8 * nblocks);
  hipMemcpy(spec, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(twd, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(g_dftm_trace), &tr, sizeof(tr));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 5; ++rep) {
    hipMemset(tr, 0, sizeof(long long) * 8 * nblocks);
    hipEventRecord(a);
    launch_dftm_y<double>(0, K, twd, spec, rep & 1, 1, 0);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> t(8 * nblocks);
    hipMemcpy(t.data(), tr, sizeof(long long) * t.size(), hipMemcpyDeviceToHost);
    long long t0 = t[0], t1 = 0;
    double ph[4] = {0, 0, 0, 0};
    for (int blk = 0; blk < nblocks; ++blk) {
      if (t[8 * blk] < t0) t0 = t[8 * blk];
      if (t[8 * blk + 4] > t1) t1 = t[8 * blk + 4];
      for (int p = 0; p < 4; ++p) ph[p] += (double)(t[8 * blk + p + 1] - t[8 * blk + p]) * 10.0 / nblocks;   // 100 MHz -> ns
    }
    long long smax = 0;
    for (int blk = 0; blk < nblocks; ++blk) if (t[8 * blk] - t0 > smax) smax = t[8 * blk] - t0;
    printf("rep %d: event %.1f us | first start -> last end %.1f us | latest block start +%.1f us | mean ns: load %.0f  mma %.0f  store %.0f  special %.0f\n",
           rep, ms * 1e3, (t1 - t0) * 0.01, smax * 0.01, ph[0], ph[1], ph[2], ph[3]);
  }
  return 0;
}
