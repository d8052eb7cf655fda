// Which output layout makes rocFFT's batched 2-D r2c / c2r of the y-z planes fastest?  (engine.hip plan2_f / plan2_b: 256^3 f32,
// one plane per x; the fused x pass that follows only needs a fixed stride between x planes.)
//   A: default -- spectrum [x][y][kz], kz fastest
//   C: transposed -- spectrum [x][kz][y], y fastest (out strides {K1, 1})
//   P: default order with the kz rows padded to a multiple of 16 complex (128-byte aligned rows)
// hipcc -O2 tools/ubench/rocfft_yz_layouts.cpp -lrocfft -o tools/ubench/rocfft_yz_layouts.bin
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e = (x); if (e != 0) { printf("error %d at line %d\n", (int)e, __LINE__); exit(1); } } while (0)

static float time_plan(rocfft_plan p, void* in, void* out, void* work, size_t wbytes, int reps) {
  rocfft_execution_info info;
  CK(rocfft_execution_info_create(&info));
  if (wbytes) CK(rocfft_execution_info_set_work_buffer(info, work, wbytes));
  void* i[1] = {in};
  void* o[1] = {out};
  for (int k = 0; k < 3; ++k) CK(rocfft_execute(p, i, o, info));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  for (int k = 0; k < reps; ++k) CK(rocfft_execute(p, i, o, info));
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  rocfft_execution_info_destroy(info);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const size_t K = argc > 1 ? atoi(argv[1]) : 256, Kh = K / 2 + 1;
  CK(rocfft_setup());
  float *mesh, *spec;
  void* work;
  const size_t wcap = 512u << 20;
  CK(hipMalloc(&mesh, K * K * K * sizeof(float)));
  CK(hipMalloc(&spec, K * K * ((Kh + 15) / 16 * 16) * 2 * sizeof(float)));
  CK(hipMalloc(&work, wcap));
  CK(hipMemset(mesh, 0, K * K * K * sizeof(float)));
  CK(hipMemset(spec, 0, K * K * ((Kh + 15) / 16 * 16) * 2 * sizeof(float)));
  const size_t len2[2] = {K, K};
  const size_t Kp = (Kh + 15) / 16 * 16;
  for (int variant = 0; variant < 3; ++variant) {
    rocfft_plan pf, pb;
    rocfft_plan_description df = nullptr, db = nullptr;
    if (variant >= 1) {
      const size_t rs[2] = {1, K};
      const size_t cs[2] = {variant == 1 ? K : 1, variant == 1 ? 1 : Kp};
      const size_t cdist = variant == 1 ? K * Kh : K * Kp;
      CK(rocfft_plan_description_create(&df));
      CK(rocfft_plan_description_set_data_layout(df, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr,
                                                 2, rs, K * K, 2, cs, cdist));
      CK(rocfft_plan_description_create(&db));
      CK(rocfft_plan_description_set_data_layout(db, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr,
                                                 2, cs, cdist, 2, rs, K * K));
    }
    CK(rocfft_plan_create(&pf, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_single, 2, len2, K, df));
    CK(rocfft_plan_create(&pb, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_single, 2, len2, K, db));
    size_t wf = 0, wb = 0;
    CK(rocfft_plan_get_work_buffer_size(pf, &wf));
    CK(rocfft_plan_get_work_buffer_size(pb, &wb));
    if (wf > wcap || wb > wcap) { printf("variant %d: work buffer too large\n", variant); continue; }
    const float tf = time_plan(pf, mesh, spec, work, wf, 20), tb = time_plan(pb, spec, mesh, work, wb, 20);
    printf("K=%zu variant %s: r2c %.1f us (work %zu MB), c2r %.1f us (work %zu MB)\n", K, variant == 0 ? "A [x][y][kz]" : (variant == 1 ? "C [x][kz][y]" : "P [x][y][kz pad 16]"), tf,
           wf >> 20, tb, wb >> 20);
    rocfft_plan_destroy(pf); rocfft_plan_destroy(pb);
  }
  return 0;
}
