// libadmp_hip: handle, device buffers, rocFFT plans, and the orchestration of one
// get_energy / get_forces evaluation behind the C ABI of include/admp_hip.h.
//
// Flow of admp_pme_energy_grad (reference call stack: admp/pme.py:58-86 get_energy ->
// :111-143 optimize_Uind -> :176-254 energy_pme, gradient by jax.value_and_grad :108):
//   prepare_sites   local frames, Q_local -> Q_global, packed site rows           [atom_kernels.hip]
//   (polarizable)   Jacobi SCF: pair_field + spread/r2c/kspace/c2r/gather_field + field_finish,
//                   one host read of max|field| per cycle (the reference syncs there too, pme.py:136)
//   pair_full       real-space energy, dE/dr, dE/dQ                                 [pair_kernels.hip]
//   spread, r2c, kspace (energy + G multiply), c2r, gather                          [recip_kernels.hip + rocFFT]
//   finish          self + penalty, frame adjoint, dE/dQ_local                      [atom_kernels.hip]
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/admp_hip.h"
#include "launch.h"

using namespace admp;

namespace {

struct Err {
  int code;
  std::string msg;
};

#define HIP_TRY(x)                                                                                       \
  do {                                                                                                   \
    hipError_t e_ = (x);                                                                                 \
    if (e_ != hipSuccess) throw Err{ADMP_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)};          \
  } while (0)
#define FFT_TRY(x)                                                                                       \
  do {                                                                                                   \
    rocfft_status s_ = (x);                                                                              \
    if (s_ != rocfft_status_success) throw Err{ADMP_E_FFT, std::string(#x) + ": rocfft status " + std::to_string((int)s_)}; \
  } while (0)
#define ARG_CHECK(c, m) \
  do { if (!(c)) throw Err{ADMP_E_ARG, m}; } while (0)

std::once_flag g_fft_once;

// growable device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  void need(size_t n) {
    if (n <= bytes) return;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    bytes = 0;
    HIP_TRY(hipMalloc(&p, n));
    bytes = n;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class U> U* as() const { return reinterpret_cast<U*>(p); }
};

// per-label launch timing with HIP events on the engine's stream
struct Profiler {
  bool on = false;
  struct Rec { int label; hipEvent_t a, b; };
  std::vector<std::string> labels;
  std::map<std::string, int> index;
  std::vector<double> total_ms;
  std::vector<int64_t> count;
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    return e;
  }
  int id(const char* name) {
    auto it = index.find(name);
    if (it != index.end()) return it->second;
    int k = (int)labels.size();
    labels.push_back(name); index[name] = k; total_ms.push_back(0.0); count.push_back(0);
    return k;
  }
  void begin(const char* name, hipStream_t st, Rec& r) {
    r.label = id(name); r.a = get(); r.b = get();
    HIP_TRY(hipEventRecord(r.a, st));
  }
  void end(hipStream_t st, Rec& r) {
    HIP_TRY(hipEventRecord(r.b, st));
    pending.push_back(r);
  }
  void collect(hipStream_t st) {
    if (pending.empty()) return;
    HIP_TRY(hipStreamSynchronize(st));
    for (auto& r : pending) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
      total_ms[r.label] += ms; count[r.label] += 1;
      pool.push_back(r.a); pool.push_back(r.b);
    }
    pending.clear();
  }
  void reset(hipStream_t st) {
    collect(st);
    for (auto& v : total_ms) v = 0.0;
    for (auto& v : count) v = 0;
  }
  void destroy() {
    for (auto& r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : pool) (void)hipEventDestroy(e);
    pending.clear(); pool.clear();
  }
};

struct Scoped {
  Profiler& p; hipStream_t st; Profiler::Rec r; bool active;
  Scoped(Profiler& p_, const char* name, hipStream_t st_) : p(p_), st(st_), active(p_.on) { if (active) p.begin(name, st, r); }
  ~Scoped() { if (active) p.end(st, r); }
};
#define TIMED(name) Scoped scoped_timer_(prof, name, stream)

void invert3(const double* h, double* inv, double* det) {
  double d = h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) + h[2] * (h[3] * h[7] - h[4] * h[6]);
  inv[0] = (h[4] * h[8] - h[5] * h[7]) / d; inv[1] = (h[2] * h[7] - h[1] * h[8]) / d; inv[2] = (h[1] * h[5] - h[2] * h[4]) / d;
  inv[3] = (h[5] * h[6] - h[3] * h[8]) / d; inv[4] = (h[0] * h[8] - h[2] * h[6]) / d; inv[5] = (h[2] * h[3] - h[0] * h[5]) / d;
  inv[6] = (h[3] * h[7] - h[4] * h[6]) / d; inv[7] = (h[1] * h[6] - h[0] * h[7]) / d; inv[8] = (h[0] * h[4] - h[1] * h[3]) / d;
  *det = d;
}

struct EngineBase {
  virtual ~EngineBase() {}
  int device = 0;
  int prec = 8;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Profiler prof;
  Topology top;
  NbrTable nbr;
  bool have_top = false, have_ewald = false, have_pairs = false;
  double kappa = 0;
  int K[3] = {0, 0, 0};
  int lmax = 2, lpol = 0;
  DevBuf scan_scratch;
  size_t scan_bytes = 0;

  virtual void set_ewald(double kappa_, int K1, int K2, int K3, int lmax_, int lpol_) = 0;
  virtual void pme(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                   const double* mS, const double* pS, void* U, int max_cycle, double thresh, double* E, void* dpos,
                   void* dQl, int* ncyc, int* conv, int on_device) = 0;
  virtual void disp(const void* pos, const double* box, const void* clist, int pmax, int ns, const double* mS, double* E,
                    void* dpos, int on_device) = 0;
  virtual void tt(const void* pos, const double* box, const void* abqc, int ns, const double* mS, double* E, void* dpos,
                  int on_device) = 0;

  void free_topology() {
    if (top.axis_type) (void)hipFree(top.axis_type);
    if (top.axis_idx) (void)hipFree(top.axis_idx);
    if (top.excl_ptr) (void)hipFree(top.excl_ptr);
    if (top.excl_col) (void)hipFree(top.excl_col);
    if (top.excl_nb) (void)hipFree(top.excl_nb);
    top = Topology();
    if (nbr.rowptr) (void)hipFree(nbr.rowptr);
    if (nbr.col) (void)hipFree(nbr.col);
    nbr = NbrTable();
    have_top = have_pairs = false;
  }

  void set_topology(int na, const int32_t* atype, const int32_t* aidx, const int32_t* eptr, const int32_t* ecol,
                    const int32_t* enb) {
    ARG_CHECK(na > 0 && na <= kColMask, "n_atoms out of range");
    free_topology();
    top.na = na;
    std::vector<int32_t> t5(na, NoAxisType), idx(3 * (size_t)na, -1);
    const int32_t* at = atype ? atype : t5.data();
    const int32_t* ai = aidx ? aidx : idx.data();
    for (int i = 0; i < na; ++i) {
      ARG_CHECK(at[i] >= 0 && at[i] <= 5, "axis_type outside 0..5");
      for (int k = 0; k < 3; ++k) ARG_CHECK(ai[3 * i + k] >= -1 && ai[3 * i + k] < na, "axis index out of range");
    }
    HIP_TRY(hipMalloc(&top.axis_type, sizeof(int) * na));
    HIP_TRY(hipMalloc(&top.axis_idx, sizeof(int) * 3 * na));
    HIP_TRY(hipMemcpy(top.axis_type, at, sizeof(int) * na, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(top.axis_idx, ai, sizeof(int) * 3 * na, hipMemcpyHostToDevice));
    if (eptr) {
      int nnz = eptr[na];
      ARG_CHECK(nnz >= 0, "bad exclusion rowptr");
      for (int k = 0; k < nnz; ++k) ARG_CHECK(ecol[k] >= 0 && ecol[k] < na && enb[k] >= 0 && enb[k] <= 15, "bad exclusion entry");
      HIP_TRY(hipMalloc(&top.excl_ptr, sizeof(int) * (na + 1)));
      HIP_TRY(hipMalloc(&top.excl_col, sizeof(int) * (nnz > 0 ? nnz : 1)));
      HIP_TRY(hipMalloc(&top.excl_nb, sizeof(int) * (nnz > 0 ? nnz : 1)));
      HIP_TRY(hipMemcpy(top.excl_ptr, eptr, sizeof(int) * (na + 1), hipMemcpyHostToDevice));
      if (nnz > 0) {
        HIP_TRY(hipMemcpy(top.excl_col, ecol, sizeof(int) * nnz, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(top.excl_nb, enb, sizeof(int) * nnz, hipMemcpyHostToDevice));
      }
    }
    have_top = true;
  }

  void set_pairs(int64_t n_rows, const int32_t* pairs, int on_device) {
    ARG_CHECK(have_top, "admp_set_topology must precede admp_set_pairs");
    ARG_CHECK(n_rows >= 0, "negative pair count");
    DevBuf staged;
    const int* dev = pairs;
    if (!on_device && n_rows > 0) {
      staged.need(sizeof(int) * 2 * (size_t)n_rows);
      HIP_TRY(hipMemcpyAsync(staged.p, pairs, sizeof(int) * 2 * (size_t)n_rows, hipMemcpyHostToDevice, stream));
      dev = staged.as<int>();
    }
    {
      TIMED("nbr_build");
      int rc = build_neighbour_table(stream, top, n_rows, dev, nbr, &scan_scratch.p, &scan_bytes);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("build_neighbour_table: ") + hipGetErrorString((hipError_t)rc)};
    }
    HIP_TRY(hipStreamSynchronize(stream));
    staged.release();
    have_pairs = true;
  }
};

template <class T>
struct Engine : EngineBase {
  // per-atom
  DevBuf sites, grad, pot, fld_pair, fld_recip, field, energies_d, fmax_d;
  // staging for host-pointer calls
  DevBuf s_pos, s_Q, s_pol, s_thole, s_U, s_out, s_dQ, s_par;
  // mesh
  DevBuf mesh, spec, gtab, fft_work, binv_d, bin_cells, bin_sorted, bin_scan;
  BinScratch bins;
  rocfft_plan plan_f = nullptr, plan_b = nullptr;
  rocfft_execution_info info_f = nullptr, info_b = nullptr;
  int planK[3] = {0, 0, 0};
  // validity of the cached G table
  double tab_box[9] = {0}, tab_kappa = -1;
  int tab_which = 0, tabK[3] = {0, 0, 0};

  ~Engine() override {
    destroy_plans();
    for (DevBuf* b : {&sites, &grad, &pot, &fld_pair, &fld_recip, &field, &energies_d, &fmax_d, &s_pos, &s_Q, &s_pol,
                      &s_thole, &s_U, &s_out, &s_dQ, &s_par, &mesh, &spec, &gtab, &fft_work, &binv_d, &scan_scratch, &bin_cells,
                      &bin_sorted, &bin_scan})
      b->release();
    free_topology();
    prof.destroy();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }

  void destroy_plans() {
    if (plan_f) rocfft_plan_destroy(plan_f);
    if (plan_b) rocfft_plan_destroy(plan_b);
    if (info_f) rocfft_execution_info_destroy(info_f);
    if (info_b) rocfft_execution_info_destroy(info_b);
    plan_f = plan_b = nullptr;
    info_f = info_b = nullptr;
    planK[0] = planK[1] = planK[2] = 0;
  }

  void set_ewald(double kappa_, int K1, int K2, int K3, int lmax_, int lpol_) override {
    ARG_CHECK(kappa_ > 0, "kappa must be positive");
    ARG_CHECK(K1 >= 6 && K2 >= 6 && K3 >= 6, "PME mesh must be at least 6 points per dimension (order-6 splines)");
    ARG_CHECK(lmax_ >= 0 && lmax_ <= 2, "l > 2 (beyond quadrupole) not supported");   // admp/recip.py:275
    kappa = kappa_; K[0] = K1; K[1] = K2; K[2] = K3; lmax = lmax_; lpol = lpol_ ? 1 : 0;
    have_ewald = true;
  }

  void ensure_mesh() {
    const size_t nreal = (size_t)K[0] * K[1] * K[2];
    const size_t nspec = (size_t)K[0] * K[1] * (K[2] / 2 + 1);
    mesh.need(nreal * sizeof(T));
    spec.need(nspec * 2 * sizeof(T));
    gtab.need(nspec * sizeof(T));
    binv_d.need(9 * sizeof(double));
    if (planK[0] == K[0] && planK[1] == K[1] && planK[2] == K[2] && plan_f) return;
    destroy_plans();
    std::call_once(g_fft_once, [] { rocfft_setup(); });
    const size_t len[3] = {(size_t)K[2], (size_t)K[1], (size_t)K[0]};   // rocFFT: fastest dimension first
    const rocfft_precision pr = sizeof(T) == 4 ? rocfft_precision_single : rocfft_precision_double;
    FFT_TRY(rocfft_plan_create(&plan_f, rocfft_placement_notinplace, rocfft_transform_type_real_forward, pr, 3, len, 1, nullptr));
    FFT_TRY(rocfft_plan_create(&plan_b, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, pr, 3, len, 1, nullptr));
    size_t wf = 0, wb = 0;
    FFT_TRY(rocfft_plan_get_work_buffer_size(plan_f, &wf));
    FFT_TRY(rocfft_plan_get_work_buffer_size(plan_b, &wb));
    fft_work.need((wf > wb ? wf : wb) + 16);
    FFT_TRY(rocfft_execution_info_create(&info_f));
    FFT_TRY(rocfft_execution_info_create(&info_b));
    if (wf) FFT_TRY(rocfft_execution_info_set_work_buffer(info_f, fft_work.p, wf));
    if (wb) FFT_TRY(rocfft_execution_info_set_work_buffer(info_b, fft_work.p, wb));
    planK[0] = K[0]; planK[1] = K[1]; planK[2] = K[2];
    tab_kappa = -1;   // mesh changed: table stale
  }

  void fft_forward() {
    TIMED("rocfft_r2c");
    FFT_TRY(rocfft_execution_info_set_stream(info_f, stream));
    void* in[1] = {mesh.p};
    void* out[1] = {spec.p};
    FFT_TRY(rocfft_execute(plan_f, in, out, info_f));
  }
  void fft_inverse() {
    TIMED("rocfft_c2r");
    FFT_TRY(rocfft_execution_info_set_stream(info_b, stream));
    void* in[1] = {spec.p};
    void* out[1] = {mesh.p};
    FFT_TRY(rocfft_execute(plan_b, in, out, info_b));
  }

  Box<T> make_box(const double* h, double* inv, double* vol) {
    Box<T> b;
    invert3(h, inv, vol);
    ARG_CHECK(std::fabs(*vol) > 1e-12, "singular box");
    for (int k = 0; k < 9; ++k) { b.h[k] = (T)h[k]; b.hinv[k] = (T)inv[k]; }
    return b;
  }

  RecipGeom<T> make_geom(const double* inv) {
    RecipGeom<T> g;
    for (int d = 0; d < 3; ++d) g.K[d] = K[d];
    for (int k = 0; k < 9; ++k) g.hinv[k] = (T)inv[k];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        g.Aop[3 * i + j] = (T)(-(double)K[i] * inv[3 * j + i]);   // -Nj_Aji_star[i][j] (admp/recip.py:52,177)
        g.Jac[3 * i + j] = (T)(-(double)K[j] * inv[3 * i + j]);   // d u_j / d x_i     (admp/recip.py:75-77)
      }
    return g;
  }

  void ensure_gtab(const double* box, const double* inv, double vol, int which) {
    bool same = tab_kappa == kappa && tab_which == which && tabK[0] == K[0] && tabK[1] == K[1] && tabK[2] == K[2];
    for (int k = 0; k < 9 && same; ++k) same = tab_box[k] == box[k];
    if (same) return;
    HIP_TRY(hipMemcpyAsync(binv_d.p, inv, 9 * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));   // `inv` is a caller stack array
    {
      TIMED("gtab");
      launch_gtab<T>(stream, K, binv_d.as<double>(), std::fabs(vol), kappa, which, gtab.as<T>());
    }
    std::memcpy(tab_box, box, sizeof(tab_box));
    tab_kappa = kappa; tab_which = which; tabK[0] = K[0]; tabK[1] = K[1]; tabK[2] = K[2];
  }

  ScaleTab<T> make_tab(int ns, const double* mS, const double* pS) {
    ARG_CHECK(ns >= 1 && mS, "mScales missing");
    ScaleTab<T> t;
    for (int nb = 0; nb < 16; ++nb) {
      int idx = ((nb - 1) % ns + ns) % ns;   // python-style wrap of mScales[nbonds-1] (admp/pme.py:682-683)
      t.mm[nb] = (T)(mS[idx] - 1.0);
      double p = pS ? pS[idx] : 0.0;
      t.p[nb] = (T)p;
      t.w0[nb] = (T)(1.0 / (std::exp((p - 1e-3) / 1e-5) + 1.0));   // switch_val weight of y0 (admp/pme.py:345-346)
    }
    return t;
  }

  const T* stage_in(DevBuf& b, const void* p, size_t n, int on_device) {
    if (!p) return nullptr;
    if (on_device) return reinterpret_cast<const T*>(p);
    b.need(n * sizeof(T));
    HIP_TRY(hipMemcpyAsync(b.p, p, n * sizeof(T), hipMemcpyHostToDevice, stream));
    return b.as<T>();
  }

  void ensure_bins(int na) {
    const BrickGrid bg = make_bricks(K);
    bin_cells.need(sizeof(int) * 2 * (size_t)(bg.ncell + 1));
    bin_sorted.need(sizeof(int) * 8 * (size_t)na);
    bin_scan.need(spread_scan_bytes(bg.ncell));
    bins.cell_start = bin_cells.as<int>();
    bins.cursor = bin_cells.as<int>() + (bg.ncell + 1);
    bins.sorted = bin_sorted.as<int>();
    bins.scan_tmp = bin_scan.p;
    bins.scan_bytes = bin_scan.bytes;
  }

  void recip_pass(int na, const RecipGeom<T>& g, bool field_only) {
    ensure_bins(na);
    {
      TIMED("spread");
      int rc = launch_spread<T>(stream, na, sites.as<Site<T>>(), lpol, g, bins, mesh.as<T>());
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)};
    }
    fft_forward();
    if (field_only) HIP_TRY(hipMemsetAsync(energies_d.as<double>() + E_SCF_RECIP, 0, sizeof(double), stream));
    { TIMED("kspace"); launch_kspace<T>(stream, K, gtab.as<T>(), spec.as<T>(), energies_d.as<double>(), field_only ? E_SCF_RECIP : E_RECIP); }
    fft_inverse();
  }

  void pme(const void* pos_, const double* box, const void* Ql_, const void* pol_, const void* thole_, int ns,
           const double* mS, const double* pS, void* U_, int max_cycle, double thresh, double* E, void* dpos_,
           void* dQl_, int* ncyc, int* conv, int on_device) override {
    ARG_CHECK(have_top && have_ewald && have_pairs, "topology, ewald parameters and pairs must be set first");
    ARG_CHECK(pos_ && box && Ql_ && E, "null argument");
    if (lpol) ARG_CHECK(pol_ && thole_ && U_ && pS, "polarizable handle needs pol, tholes, pScales and U_inout");
    const int na = top.na;
    HIP_TRY(hipSetDevice(device));
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    RecipGeom<T> g = make_geom(inv);
    ScaleTab<T> tab = make_tab(ns, mS, pS);
    ensure_mesh();
    ensure_gtab(box, inv, vol, 1);

    const T* pos = stage_in(s_pos, pos_, 3 * (size_t)na, on_device);
    const T* Ql = stage_in(s_Q, Ql_, 9 * (size_t)na, on_device);
    const T* pol = lpol ? stage_in(s_pol, pol_, na, on_device) : nullptr;
    const T* thole = lpol ? stage_in(s_thole, thole_, na, on_device) : nullptr;
    T* U = nullptr;
    if (lpol) {
      if (on_device) U = reinterpret_cast<T*>(U_);
      else { s_U.need(3 * (size_t)na * sizeof(T)); HIP_TRY(hipMemcpyAsync(s_U.p, U_, 3 * (size_t)na * sizeof(T), hipMemcpyHostToDevice, stream)); U = s_U.as<T>(); }
    }
    T* dpos = nullptr;
    if (dpos_) {
      if (on_device) dpos = reinterpret_cast<T*>(dpos_);
      else { s_out.need(3 * (size_t)na * sizeof(T)); dpos = s_out.as<T>(); }
    }
    T* dQl = nullptr;
    if (dQl_) {
      ARG_CHECK(dpos_, "dE_dQlocal requires dE_dpos");
      if (on_device) dQl = reinterpret_cast<T*>(dQl_);
      else { s_dQ.need(9 * (size_t)na * sizeof(T)); dQl = s_dQ.as<T>(); }
    }
    // the gradient buffer is needed internally even for energy-only calls
    grad.need(3 * (size_t)na * sizeof(T));
    T* gbuf = dpos ? dpos : grad.as<T>();

    sites.need(sizeof(Site<T>) * (size_t)na);
    pot.need(9 * (size_t)na * sizeof(T));
    energies_d.need(E_SLOTS * sizeof(double));
    fmax_d.need(sizeof(unsigned long long));
    Site<T>* S = sites.as<Site<T>>();
    double* Ed = energies_d.as<double>();

    HIP_TRY(hipMemsetAsync(Ed, 0, E_SLOTS * sizeof(double), stream));
    { TIMED("prepare_sites"); launch_prepare_sites<T>(stream, top, pos, Ql, U, pol, thole, bx, S); }

    // phi_valid: the mesh holds phi = c2r(G S) of the CURRENT dipoles (last SCF field evaluation, no update since):
    // the closing gather can then reuse it instead of spreading and transforming again.
    bool phi_valid = false;
    int cyc = 0, flag = 1;
    if (lpol) {
      fld_pair.need(3 * (size_t)na * sizeof(T));
      fld_recip.need(3 * (size_t)na * sizeof(T));
      field.need(3 * (size_t)na * sizeof(T));
      ARG_CHECK(max_cycle >= 1, "max_cycle must be >= 1");
      int i = 0;
      for (i = 0; i < max_cycle; ++i) {     // admp/pme.py:132-138
        HIP_TRY(hipMemsetAsync(fmax_d.p, 0, sizeof(unsigned long long), stream));
        { TIMED("pair_field"); launch_pair_field<T>(stream, na, nbr, S, bx, tab, (T)kappa, fld_pair.as<T>()); }
        recip_pass(na, g, true);
        { TIMED("gather_field"); launch_gather_field<T>(stream, na, S, g, mesh.as<T>(), fld_recip.as<T>()); }
        { TIMED("field_finish");
          launch_field_finish<T>(stream, na, S, pol, U, fld_pair.as<T>(), fld_recip.as<T>(), (T)kappa, field.as<T>(),
                                 fmax_d.as<unsigned long long>()); }
        unsigned long long bits = 0;
        HIP_TRY(hipMemcpyAsync(&bits, fmax_d.p, sizeof(bits), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        double fmax;
        std::memcpy(&fmax, &bits, sizeof(fmax));
        if (fmax < thresh) { phi_valid = true; break; }
        { TIMED("jacobi_update"); launch_jacobi_update<T>(stream, na, pol, field.as<T>(), U, S); }
      }
      if (i == max_cycle) i = max_cycle - 1;   // python's loop variable after exhaustion
      cyc = i;
      flag = (i != max_cycle - 1);             // admp/pme.py:139-143
    }

    { TIMED("pair_full"); launch_pair_full<T>(stream, na, nbr, S, bx, tab, (T)kappa, lpol, gbuf, pot.as<T>(), Ed); }
    if (!phi_valid) recip_pass(na, g, false);
    { TIMED("gather"); launch_gather<T>(stream, na, S, lpol, g, mesh.as<T>(), pot.as<T>(), gbuf); }
    { TIMED("finish");
      launch_finish<T>(stream, top, pos, bx, S, pol, U, lpol, (T)kappa, pot.as<T>(), dpos ? gbuf : nullptr, dQl, Ed); }

    double Eh[E_SLOTS];
    HIP_TRY(hipMemcpyAsync(Eh, Ed, sizeof(Eh), hipMemcpyDeviceToHost, stream));
    if (!on_device) {
      if (dpos_) HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
      if (dQl_) HIP_TRY(hipMemcpyAsync(dQl_, dQl, 9 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
      if (lpol) HIP_TRY(hipMemcpyAsync(U_, U, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    E[0] = Eh[E_REAL]; E[1] = phi_valid ? Eh[E_SCF_RECIP] : Eh[E_RECIP]; E[2] = Eh[E_SELF]; E[3] = Eh[E_PEN];
    if (ncyc) *ncyc = cyc;
    if (conv) *conv = flag;
  }

  // dispersion PME (admp/disp_pme.py:80-123): real-space pairs + one scalar reciprocal pass per power
  void disp(const void* pos_, const double* box, const void* clist_, int pmax, int ns, const double* mS, double* E,
            void* dpos_, int on_device) override {
    ARG_CHECK(have_top && have_ewald && have_pairs, "topology, ewald parameters and pairs must be set first");
    ARG_CHECK(pos_ && box && clist_ && E, "null argument");
    ARG_CHECK(pmax == 6 || pmax == 8 || pmax == 10, "pmax must be 6, 8 or 10");
    const int na = top.na;
    HIP_TRY(hipSetDevice(device));
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    RecipGeom<T> g = make_geom(inv);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    ensure_mesh();
    const T* pos = stage_in(s_pos, pos_, 3 * (size_t)na, on_device);
    const T* cl = stage_in(s_par, clist_, 3 * (size_t)na, on_device);
    T* dpos = nullptr;
    grad.need(3 * (size_t)na * sizeof(T));
    if (dpos_ && on_device) dpos = reinterpret_cast<T*>(dpos_);
    else dpos = grad.as<T>();
    energies_d.need(E_SLOTS * sizeof(double));
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_SLOTS * sizeof(double), stream));
    { TIMED("disp_pair"); launch_disp_pair<T>(stream, na, nbr, pos, cl, bx, tab, (T)kappa, pmax, dpos, Ed); }
    const size_t nreal = (size_t)K[0] * K[1] * K[2];
    double eself = 0.0;
    // self term (disp_pme.py:254-279) needs sum c_p^2: folded into the spread pass via a tiny host reduction
    std::vector<T> ch(3 * (size_t)na);
    HIP_TRY(hipMemcpyAsync(ch.data(), cl, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const double kp[3] = {std::pow(kappa, 6) / 12.0, std::pow(kappa, 8) / 48.0, std::pow(kappa, 10) / 240.0};
    for (int c = 0; c < (pmax - 4) / 2; ++c) {
      double s2 = 0.0;
      for (int i = 0; i < na; ++i) s2 += (double)ch[3 * (size_t)i + c] * (double)ch[3 * (size_t)i + c];
      eself -= kp[c] * s2;
      ensure_gtab(box, inv, vol, 6 + 2 * c);
      HIP_TRY(hipMemsetAsync(mesh.p, 0, nreal * sizeof(T), stream));
      { TIMED("spread_scalar"); launch_spread_scalar<T>(stream, na, pos, cl, 3, c, g, mesh.as<T>()); }
      fft_forward();
      { TIMED("kspace"); launch_kspace<T>(stream, K, gtab.as<T>(), spec.as<T>(), Ed, E_RECIP); }
      fft_inverse();
      { TIMED("gather_scalar"); launch_gather_scalar<T>(stream, na, pos, cl, 3, c, g, mesh.as<T>(), dpos); }
    }
    double Eh[E_SLOTS];
    HIP_TRY(hipMemcpyAsync(Eh, Ed, sizeof(Eh), hipMemcpyDeviceToHost, stream));
    if (dpos_ && !on_device) HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    E[0] = Eh[E_REAL]; E[1] = Eh[E_RECIP]; E[2] = eself;
  }

  void tt(const void* pos_, const double* box, const void* abqc_, int ns, const double* mS, double* E, void* dpos_,
          int on_device) override {
    ARG_CHECK(have_top && have_pairs, "topology and pairs must be set first");
    ARG_CHECK(pos_ && box && abqc_ && E, "null argument");
    const int na = top.na;
    HIP_TRY(hipSetDevice(device));
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    const T* pos = stage_in(s_pos, pos_, 3 * (size_t)na, on_device);
    const T* par = stage_in(s_par, abqc_, 4 * (size_t)na, on_device);
    grad.need(3 * (size_t)na * sizeof(T));
    T* dpos = (dpos_ && on_device) ? reinterpret_cast<T*>(dpos_) : grad.as<T>();
    energies_d.need(E_SLOTS * sizeof(double));
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_SLOTS * sizeof(double), stream));
    { TIMED("tt_pair"); launch_tt_pair<T>(stream, na, nbr, pos, par, bx, tab, dpos, Ed); }
    double Eh[E_SLOTS];
    HIP_TRY(hipMemcpyAsync(Eh, Ed, sizeof(Eh), hipMemcpyDeviceToHost, stream));
    if (dpos_ && !on_device) HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    E[0] = Eh[E_REAL];
  }
};

}  // namespace

struct admp_handle {
  std::unique_ptr<EngineBase> eng;
  std::string err;
};

static std::string g_create_err;

template <class F>
static int guarded(admp_handle* h, F&& f) {
  if (!h || !h->eng) return ADMP_E_ARG;
  try {
    (void)hipSetDevice(h->eng->device);
    f(*h->eng);
    return ADMP_OK;
  } catch (const Err& e) {
    h->err = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    h->err = e.what();
    return ADMP_E_ARG;
  }
}

extern "C" {

const char* admp_version(void) { return "admp_hip 0.1 (gfx950)"; }

int admp_create(admp_handle** out, int device, int precision) {
  if (!out || (precision != 4 && precision != 8)) return ADMP_E_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ADMP_E_NOGPU;
  try {
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<EngineBase> e;
    if (precision == 4) e.reset(new Engine<float>());
    else e.reset(new Engine<double>());
    e->device = device;
    e->prec = precision;
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
    admp_handle* h = new admp_handle();
    h->eng = std::move(e);
    *out = h;
    return ADMP_OK;
  } catch (const Err& e) {
    g_create_err = e.msg;
    return e.code;
  }
}

int admp_destroy(admp_handle* h) {
  if (!h) return ADMP_E_ARG;
  if (h->eng) { (void)hipSetDevice(h->eng->device); (void)hipStreamSynchronize(h->eng->stream); }
  delete h;
  return ADMP_OK;
}

const char* admp_last_error(const admp_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int admp_set_stream(admp_handle* h, void* hip_stream) {
  return guarded(h, [&](EngineBase& e) {
    HIP_TRY(hipStreamSynchronize(e.stream));
    if (e.own_stream && e.stream) HIP_TRY(hipStreamDestroy(e.stream));
    if (hip_stream) { e.stream = (hipStream_t)hip_stream; e.own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking)); e.own_stream = true; }
  });
}

int admp_synchronize(admp_handle* h) {
  return guarded(h, [&](EngineBase& e) { HIP_TRY(hipStreamSynchronize(e.stream)); });
}

int admp_set_topology(admp_handle* h, int n_atoms, const int32_t* axis_type, const int32_t* axis_idx,
                      const int32_t* excl_rowptr, const int32_t* excl_col, const int32_t* excl_nbonds) {
  return guarded(h, [&](EngineBase& e) { e.set_topology(n_atoms, axis_type, axis_idx, excl_rowptr, excl_col, excl_nbonds); });
}

int admp_set_ewald(admp_handle* h, double kappa, int K1, int K2, int K3, int lmax, int lpol) {
  return guarded(h, [&](EngineBase& e) { e.set_ewald(kappa, K1, K2, K3, lmax, lpol); });
}

int admp_set_pairs(admp_handle* h, int64_t n_rows, const int32_t* pairs, int on_device) {
  return guarded(h, [&](EngineBase& e) { e.set_pairs(n_rows, pairs, on_device); });
}

int64_t admp_num_pairs(const admp_handle* h) { return (h && h->eng) ? h->eng->nbr.n_half : -1; }

int admp_pme_energy_grad(admp_handle* h, const void* positions, const double* box, const void* Q_local,
                         const void* pol, const void* tholes, int n_scales, const double* mScales,
                         const double* pScales, const double* dScales, void* U_inout, int max_cycle, double thresh,
                         double* E_out, void* dE_dpos, void* dE_dQlocal, int* n_cycle, int* converged,
                         int on_device) {
  (void)dScales;   // accepted and ignored, as in the reference (uscales = 1, admp/pme.py:472)
  return guarded(h, [&](EngineBase& e) {
    e.pme(positions, box, Q_local, pol, tholes, n_scales, mScales, pScales, U_inout, max_cycle, thresh, E_out, dE_dpos,
          dE_dQlocal, n_cycle, converged, on_device);
  });
}

int admp_disp_energy_grad(admp_handle* h, const void* positions, const double* box, const void* c_list, int pmax,
                          int n_scales, const double* mScales, double* E_out, void* dE_dpos, int on_device) {
  return guarded(h, [&](EngineBase& e) { e.disp(positions, box, c_list, pmax, n_scales, mScales, E_out, dE_dpos, on_device); });
}

int admp_tt_energy_grad(admp_handle* h, const void* positions, const double* box, const void* abqc, int n_scales,
                        const double* mScales, double* E_out, void* dE_dpos, int on_device) {
  return guarded(h, [&](EngineBase& e) { e.tt(positions, box, abqc, n_scales, mScales, E_out, dE_dpos, on_device); });
}

int admp_profile_enable(admp_handle* h, int on) {
  return guarded(h, [&](EngineBase& e) { e.prof.collect(e.stream); e.prof.on = on != 0; });
}
int admp_profile_reset(admp_handle* h) {
  return guarded(h, [&](EngineBase& e) { e.prof.reset(e.stream); });
}
int admp_profile_count(admp_handle* h) {
  if (!h || !h->eng) return ADMP_E_ARG;
  try { h->eng->prof.collect(h->eng->stream); } catch (const Err& e) { h->err = e.msg; return e.code; }
  return (int)h->eng->prof.labels.size();
}
int admp_profile_entry(admp_handle* h, int idx, const char** label, double* total_ms, int64_t* launches) {
  return guarded(h, [&](EngineBase& e) {
    e.prof.collect(e.stream);
    ARG_CHECK(idx >= 0 && idx < (int)e.prof.labels.size(), "profile index out of range");
    if (label) *label = e.prof.labels[idx].c_str();
    if (total_ms) *total_ms = e.prof.total_ms[idx];
    if (launches) *launches = e.prof.count[idx];
  });
}

}  // extern "C"
