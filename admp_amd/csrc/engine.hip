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
#include <hip/hiprtc.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <functional>
#include <vector>

#include "../../include/admp_hip.h"
#include "dft_math.h"
#include "launch.h"
#include "rccl_comm.h"

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
  std::string only;   // when non-empty, only sections with exactly this label are bracketed
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
  Scoped(Profiler& p_, const char* name, hipStream_t st_) : p(p_), st(st_), active(p_.on && (p_.only.empty() || p_.only == name)) {
    if (active) p.begin(name, st, r);
  }
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
  int srank = 0, snranks = 1;   // x-slab decomposition (admp_slab_configure)
  admp_comm comm{};             // ... and the caller's communicator (admp_set_comm); unused with one rank
  bool have_comm = false;
  // collectives of a decomposed evaluation: every rank makes the same calls in the same order
  admp_rccl* rccl = nullptr;    // native communicator (admp_set_comm_rccl): the collectives go straight to RCCL on `stream`
  void comm_fail(const char* what, int rc) {
    if (rccl) throw Err{ADMP_E_COMM, std::string(what) + ": " + rccl_error()};
    throw Err{ADMP_E_COMM, std::string("communicator callback ") + what + " returned " + std::to_string(rc)};
  }
  void c_all_reduce(void* buf, int64_t n, int dtype, int op, int tag) {
    const int rc = rccl ? rccl_all_reduce(rccl, stream, buf, n, dtype, op, tag) : comm.all_reduce(comm.ctx, buf, n, dtype, op, tag);
    if (rc != 0) comm_fail("all_reduce", rc);
  }
  void c_all_to_all_v(const void* send, const int64_t* sc, void* recv, const int64_t* rc_, int dtype, int tag) {
    const int rc = rccl ? rccl_all_to_all_v(rccl, stream, send, sc, recv, rc_, dtype, tag)
                        : comm.all_to_all_v(comm.ctx, send, sc, recv, rc_, dtype, tag);
    if (rc != 0) comm_fail("all_to_all_v", rc);
  }
  void c_shift(const void* send, void* recv, int64_t n, int dtype, int to_next, int tag) {
    const int rc = rccl ? rccl_shift(rccl, stream, send, recv, n, dtype, to_next, tag)
                        : comm.shift(comm.ctx, send, recv, n, dtype, to_next, tag);
    if (rc != 0) comm_fail("shift", rc);
  }
  // how the polarizable calls of this handle were enqueued and how the guesses behind it turned out (admp_scf_stats):
  // [0] plain calls, [1] speculative calls, [2] ... whose first check failed (closing pass wasted), [3] chained calls,
  // [4] ... that needed more steps than enqueued (closing pass wasted), [5] ... that enqueued more steps than needed,
  // [6] increments that ran for nothing in those, [7] Jacobi steps in total
  int64_t scf_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const void* U_src = nullptr;  // admp_set_dipole_source: read-only initial dipoles of the NEXT polarizable evaluation
  const void* U_src_now = nullptr;   // ... taken over by that evaluation (Engine::pme), consumed by its site pass
  double cutoff = 0.0;          // admp_set_cutoff: listed pairs beyond it are skipped (0: every listed pair, as the reference)
  int ref_korder = 0;           // ADMP_OPT_REFERENCE_KPOINTS: the reference's k-point table (k_gtab)
  int keep_pol_sites = 0;       // ADMP_OPT_KEEP_POL_SITES: the caller vouches that the set {i : pol_i > 0} has not changed
  int side_stream_on = 1;       // ADMP_OPT_SIDE_STREAM: 0 keeps every kernel on the handle's stream (clean per-kernel event times)
  // a call that failed half way: whatever it left on a helper stream is waited for before the error is reported
  virtual void after_error() {}
  long nbr_gen = 0;             // bumped whenever the neighbour table is rebuilt
  // admp_prune_pairs: `nbr` is an INNER table (the entries of the table as built, nbr_full, below a shorter cutoff; own rowptr /
  // col, everything else shared with nbr_full) until the next prune, list build, class compile or admp_unprune.  Whatever frees
  // or rebuilds `nbr` calls unprune() first.
  NbrTable nbr_full;
  bool pruned = false;
  void unprune() {
    if (!pruned) return;
    nbr = nbr_full;
    nbr_full = NbrTable();
    pruned = false;
    ++nbr_gen;
  }
  // Neighbour table borrowed from another handle (admp_share_neighbors): the calculators of one system walk ONE compiled
  // table instead of compiling the same pair list once each.  `nbr` is then a copy of the lender's struct, refreshed at
  // the start of every call (adopt_shared), never freed or modified here.
  EngineBase* nbr_src = nullptr;
  long nbr_src_gen = -1;
  static std::set<EngineBase*>& live() { static std::set<EngineBase*> s; return s; }
  static std::mutex& live_mu() { static std::mutex m; return m; }
  static bool is_live(EngineBase* e) { std::lock_guard<std::mutex> g(live_mu()); return live().count(e) != 0; }
  void detach_shared() {
    if (!nbr_src) return;
    nbr = NbrTable();            // forget the lender's pointers
    nbr_src = nullptr; nbr_src_gen = -1;
    have_pairs = false;
    ++nbr_gen;
  }
  void adopt_shared() {
    if (!nbr_src) return;
    if (!is_live(nbr_src)) { nbr = NbrTable(); nbr_src = nullptr; have_pairs = false; throw Err{ADMP_E_ARG, "the handle the neighbour table was borrowed from has been destroyed"}; }
    if (!nbr_src->have_pairs || nbr_src->top.na != top.na) { nbr = NbrTable(); have_pairs = false; return; }
    if (nbr_src_gen != nbr_src->nbr_gen || nbr.col != nbr_src->nbr.col || nbr.order != nbr_src->nbr.order ||
        nbr.order_plain != nbr_src->nbr.order_plain) {
      if (nbr_src->stream != stream) HIP_TRY(hipStreamSynchronize(nbr_src->stream));   // built / ordered on the lender's stream
      nbr = nbr_src->nbr;
      nbr_src_gen = nbr_src->nbr_gen;
      ++nbr_gen;
    }
    have_pairs = true;
  }
  void share_neighbors(EngineBase* src) {
    ARG_CHECK(have_top, "admp_set_topology must precede admp_share_neighbors");
    if (!src) { detach_shared(); return; }
    ARG_CHECK(src != this && !src->nbr_src, "the lender must own its neighbour table");
    ARG_CHECK(src->have_top && src->top.na == top.na && src->device == device, "handles of different systems / devices");
    unprune();
    if (!nbr_src) nbr.free_all();   // drop the table this handle owns
    nbr = NbrTable();
    nbr_src = src; nbr_src_gen = -1;
    have_pairs = false;
    adopt_shared();
  }
  DevBuf scan_scratch;
  size_t scan_bytes = 0;

  virtual void set_ewald(double kappa_, int K1, int K2, int K3, int lmax_, int lpol_) = 0;
  virtual void pme(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                   const double* mS, const double* pS, void* U, int max_cycle, double thresh, double* E, void* dpos,
                   void* dQl, int* ncyc, int* conv, int on_device) = 0;
  virtual void pme_at_U(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                        const double* mS, const double* pS, const void* U, double* E, void* dpos, void* dU, void* dQl) = 0;
  virtual void local_frames(const void* pos, const double* box, void* out) = 0;
  virtual void pair_program_eval(int id, const void* pos, const double* box, const void* par, int ns, const double* mS,
                                 double* E, void* dpos, int on_device) = 0;

  // ---- user-defined pair kernels (admp_amd/xp.py traces a Python kernel into HIP source; compiled here with hiprtc) ----
  struct PairProgram { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; int n_params = 0; };
  std::vector<PairProgram> programs;
  int pair_program_build(const char* source, int n_params) {
    ARG_CHECK(source && n_params >= 0 && n_params <= 16, "bad pair program");
    hiprtcProgram prog = nullptr;
    auto rtc = [&](hiprtcResult r, const char* what) {
      if (r != HIPRTC_SUCCESS) throw Err{ADMP_E_HIP, std::string(what) + ": " + hiprtcGetErrorString(r)};
    };
    rtc(hiprtcCreateProgram(&prog, source, "admp_pair_custom.hip", 0, nullptr, nullptr), "hiprtcCreateProgram");
    const char* opts[] = {"--offload-arch=gfx950", "-O3", prec == 4 ? "-DREAL_T=float" : "-DREAL_T=double"};
    const hiprtcResult cr = hiprtcCompileProgram(prog, 3, opts);
    if (cr != HIPRTC_SUCCESS) {
      size_t n = 0;
      std::string log;
      if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) { log.resize(n); (void)hiprtcGetProgramLog(prog, &log[0]); }
      (void)hiprtcDestroyProgram(&prog);
      throw Err{ADMP_E_HIP, std::string("hiprtc could not compile the traced pair kernel: ") + log.substr(0, 1500)};
    }
    size_t sz = 0;
    rtc(hiprtcGetCodeSize(prog, &sz), "hiprtcGetCodeSize");
    std::vector<char> code(sz);
    rtc(hiprtcGetCode(prog, code.data()), "hiprtcGetCode");
    (void)hiprtcDestroyProgram(&prog);
    PairProgram p;
    p.n_params = n_params;
    HIP_TRY(hipModuleLoadData(&p.mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&p.fn, p.mod, "admp_pair_custom"));
    programs.push_back(p);
    return (int)programs.size() - 1;
  }
  void free_programs() {
    for (auto& p : programs) if (p.mod) (void)hipModuleUnload(p.mod);
    programs.clear();
  }
  virtual void pme_box_grad(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                            const double* mS, const double* pS, const void* U, double* E, double* dbox) = 0;
  virtual void disp_box_grad(const void* pos, const double* box, const void* clist, int pmax, int ns, const double* mS,
                             double* E, double* dbox) = 0;
  virtual void tt_box_grad(const void* pos, const double* box, const void* abqc, int ns, const double* mS, double* E,
                           double* dbox) = 0;
  virtual void disp_set_types(int nt, const void* types, const double* ctab) = 0;
  virtual void disp(const void* pos, const double* box, const void* clist, int pmax, int ns, const double* mS, double* E,
                    void* dpos, int on_device) = 0;
  virtual void tt(const void* pos, const double* box, const void* abqc, int ns, const double* mS, double* E, void* dpos,
                  int on_device) = 0;
  virtual void disp_param_grad(const void* pos, const double* box, const void* clist, int pmax, int ns, const double* mS,
                               void* out) = 0;
  virtual void tt_param_grad(const void* pos, const double* box, const void* abqc, int ns, const double* mS, void* out) = 0;
  virtual void thole_sums(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                          const double* mS, const double* pS, const void* U, void* sumX, void* sumXw) = 0;
  virtual void pscale_grad(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                           const double* mS, const double* pS, const void* U, double* out) = 0;
  virtual void mscale_grad(int kind, const void* pos, const double* box, const void* par, int pmax, int ns, double* out,
                           int on_device) = 0;
  virtual void md_bonded(const void* pos, const double* box, int nb, const int32_t* bidx, const void* bpar, int na,
                         const int32_t* aidx, const void* apar, double* E_dev, void* grad) = 0;
  virtual void md_kick_drift(int n, void* pos, void* vel, const void* grad, const void* inv_mass, double half_dt_acc, double dt,
                             double* ekin_dev) = 0;
  virtual void nbr_count(int na, const void* pos, const double* box, double rc, int64_t* n_pairs) = 0;
  virtual void nbr_fill(int32_t* pairs) = 0;
  virtual void nbr_table(const void* pos, const double* box, double rc) = 0;
  virtual void prune_pairs(const void* pos, const double* box, double rc) = 0;
  virtual void slab_info(int64_t* out) = 0;
  virtual void slab_home(int32_t* out, int* n_home, int* n_import) = 0;

  void free_topology() {
    if (top.axis_type) (void)hipFree(top.axis_type);
    if (top.axis_idx) (void)hipFree(top.axis_idx);
    if (top.excl_ptr) (void)hipFree(top.excl_ptr);
    if (top.excl_col) (void)hipFree(top.excl_col);
    if (top.excl_nb) (void)hipFree(top.excl_nb);
    if (top.inv_ptr) (void)hipFree(top.inv_ptr);
    if (top.inv_idx) (void)hipFree(top.inv_idx);
    if (top.grp_ptr) (void)hipFree(top.grp_ptr);
    if (top.rows_blk) (void)hipFree(top.rows_blk);
    if (top.grp_of) (void)hipFree(top.grp_of);
    if (top.gath_blk) (void)hipFree(top.gath_blk);
    top = Topology();
    unprune();
    if (!nbr_src) nbr.free_all();
    nbr_src = nullptr; nbr_src_gen = -1;
    nbr = NbrTable();
    cls_pending = false; cls_quiet = 1 << 20;
    slab_sites_na = -1;
    have_top = have_pairs = false;
    ++nbr_gen;
  }

  // row order of the freshly built table (see launch_row_order); ADMP_PAIR_SORT=0 keeps the natural order
  void order_rows() {
    static const bool off = [] { const char* e = getenv("ADMP_PAIR_SORT"); return e && atoi(e) == 0; }();
    if (off) {
      if (nbr.order) { (void)hipFree(nbr.order); nbr.order = nullptr; }
      if (nbr.order_plain) { (void)hipFree(nbr.order_plain); nbr.order_plain = nullptr; }
      return;
    }
    if (!nbr.order) HIP_TRY(hipMalloc(&nbr.order, sizeof(int) * (size_t)top.na));
    launch_row_order(stream, top.na, nbr.rowptr, nbr.order, nbr.cls);
    if (nbr.cls) {
      if (!nbr.order_plain) HIP_TRY(hipMalloc(&nbr.order_plain, sizeof(int) * (size_t)top.na));
      launch_row_order(stream, top.na, nbr.rowptr, nbr.order_plain, nullptr);
    }
  }
  // Site classes of the neighbour table (NbrTable::cls).  k_prepare_sites compares them with the sites of every evaluation
  // and leaves CLS_STALE / CLS_BETTER next to E_NACT; read_energies hands the word to cls_seen, and the next evaluation
  // recompiles the table (part the rows, regroup the row order) before it starts.  A table that is merely not as good as
  // it could be (CLS_BETTER) is left alone for a while after a CLS_STALE, so that a caller alternating between parameter
  // sets does not pay a recompilation per call.
  bool cls_pending = false;
  int cls_quiet = 1 << 20;      // evaluations since the last CLS_STALE
  int slab_sites_na = -1;       // slab rank: every row of the site table (at slab_sites_ptr) has been prepared at least once
  const void* slab_sites_ptr = nullptr;   //     for a system of this many atoms (Engine::stage_begin)
  void cls_seen(int flags) {
    if (flags & CLS_STALE) { cls_pending = true; cls_quiet = 0; }
    else if ((flags & CLS_BETTER) && cls_quiet > 8) cls_pending = true;
  }
  void apply_classes() {        // after a table build: part the fresh rows by the classes already known
    if (!nbr.cls) return;
    int rc = launch_class_partition(stream, top.na, nbr);
    if (rc != 0) throw Err{ADMP_E_HIP, std::string("class partition: ") + hipGetErrorString((hipError_t)rc)};
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
      for (int k = 0; k < nnz; ++k) ARG_CHECK(ecol[k] >= 0 && ecol[k] < na && enb[k] >= 0 && enb[k] <= kNbMask,
                                                "bad exclusion entry (atom index out of range, or nbonds outside 0..7)");
      HIP_TRY(hipMalloc(&top.excl_ptr, sizeof(int) * (na + 1)));
      HIP_TRY(hipMalloc(&top.excl_col, sizeof(int) * (nnz > 0 ? nnz : 1)));
      HIP_TRY(hipMalloc(&top.excl_nb, sizeof(int) * (nnz > 0 ? nnz : 1)));
      HIP_TRY(hipMemcpy(top.excl_ptr, eptr, sizeof(int) * (na + 1), hipMemcpyHostToDevice));
      if (nnz > 0) {
        HIP_TRY(hipMemcpy(top.excl_col, ecol, sizeof(int) * nnz, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(top.excl_nb, enb, sizeof(int) * nnz, hipMemcpyHostToDevice));
      }
    }
    {   // inverse frame map for the atomics-free closing kernel
      std::vector<std::vector<int>> inv(na);
      for (int i = 0; i < na; ++i) {
        if (at[i] == NoAxisType || ai[3 * i] < 0) { inv[i].push_back(i); continue; }
        int mem[4] = {i, ai[3 * i], at[i] != Zonly ? ai[3 * i + 1] : -1,
                      (at[i] == ZBisect || at[i] == ThreeFold) ? ai[3 * i + 2] : -1};
        for (int m = 0; m < 4; ++m) {
          if (mem[m] < 0) continue;
          bool dup = false;
          for (int q = 0; q < m; ++q) dup = dup || mem[q] == mem[m];
          if (!dup) inv[mem[m]].push_back(i);
        }
      }
      std::vector<int> ptr(na + 1, 0), idx;
      for (int a = 0; a < na; ++a) { ptr[a + 1] = ptr[a] + (int)inv[a].size(); idx.insert(idx.end(), inv[a].begin(), inv[a].end()); }
      HIP_TRY(hipMalloc(&top.inv_ptr, sizeof(int) * (na + 1)));
      HIP_TRY(hipMalloc(&top.inv_idx, sizeof(int) * (idx.empty() ? 1 : idx.size())));
      HIP_TRY(hipMemcpy(top.inv_ptr, ptr.data(), sizeof(int) * (na + 1), hipMemcpyHostToDevice));
      if (!idx.empty()) HIP_TRY(hipMemcpy(top.inv_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    }
    {   // frame groups (see Topology): union-find over "site i uses atom m in its frame"
      std::vector<int> parent(na);
      for (int i = 0; i < na; ++i) parent[i] = i;
      auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
      for (int i = 0; i < na; ++i) {
        if (at[i] == NoAxisType || ai[3 * i] < 0) continue;
        const int mem[3] = {ai[3 * i], at[i] != Zonly ? ai[3 * i + 1] : -1,
                            (at[i] == ZBisect || at[i] == ThreeFold) ? ai[3 * i + 2] : -1};
        for (int m = 0; m < 3; ++m)
          if (mem[m] >= 0) { const int a = find(i), b = find(mem[m]); if (a != b) parent[std::max(a, b)] = std::min(a, b); }
      }
      // the root of a component is its lowest atom: components are runs of consecutive atoms iff root(i) is
      // non-decreasing in i and every run is short enough
      std::vector<int> gp;
      bool ok = true;
      int prev_root = -1;
      for (int i = 0; i < na && ok; ++i) {
        const int r = find(i);
        if (r != prev_root) {
          if (r != i) ok = false;                      // joins an earlier, already closed run
          gp.push_back(i);
          prev_root = r;
        } else if (i - r >= kMaxGroup) ok = false;
      }
      gp.push_back(na);
      static const bool off = [] { const char* e = getenv("ADMP_FINISH_GROUPS"); return e && atoi(e) == 0; }();
      if (ok && !off) {
        HIP_TRY(hipMalloc(&top.grp_ptr, sizeof(int) * gp.size()));
        HIP_TRY(hipMemcpy(top.grp_ptr, gp.data(), sizeof(int) * gp.size(), hipMemcpyHostToDevice));
        top.ngroups = (int)gp.size() - 1;
        // row form: workgroups of whole groups (greedy runs of at most kFinishBlock atoms), group record per atom
        std::vector<int> blk(1, 0), gof((size_t)na);
        for (int gi = 0; gi < top.ngroups; ++gi) {
          const int a0 = gp[gi], n = gp[gi + 1] - a0;
          if (gp[gi + 1] - blk.back() > kFinishBlock) blk.push_back(a0);
          for (int m = 0; m < n; ++m) gof[(size_t)a0 + m] = (a0 << 2) | (n - 1);
        }
        blk.push_back(na);
        {   // runs of at most kGatherRun atoms for the gather with the closing epilogue
          std::vector<int> gb(1, 0);
          for (int gi = 0; gi < top.ngroups; ++gi)
            if (gp[gi + 1] - gb.back() > kGatherRun) gb.push_back(gp[gi]);
          gb.push_back(na);
          HIP_TRY(hipMalloc(&top.gath_blk, sizeof(int) * gb.size()));
          HIP_TRY(hipMemcpy(top.gath_blk, gb.data(), sizeof(int) * gb.size(), hipMemcpyHostToDevice));
          top.ngathblk = (int)gb.size() - 1;
        }
        HIP_TRY(hipMalloc(&top.rows_blk, sizeof(int) * blk.size()));
        HIP_TRY(hipMemcpy(top.rows_blk, blk.data(), sizeof(int) * blk.size(), hipMemcpyHostToDevice));
        top.nrowblk = (int)blk.size() - 1;
        HIP_TRY(hipMalloc(&top.grp_of, sizeof(int) * (size_t)na));
        HIP_TRY(hipMemcpy(top.grp_of, gof.data(), sizeof(int) * (size_t)na, hipMemcpyHostToDevice));
      }
    }
    have_top = true;
  }

  void set_pairs(int64_t n_rows, const int32_t* pairs, int on_device) {
    unprune();
    ARG_CHECK(have_top, "admp_set_topology must precede admp_set_pairs");
    ARG_CHECK(n_rows >= 0, "negative pair count");
    detach_shared();             // a pair list of its own ends a borrowed table
    DevBuf staged;
    const int* dev = pairs;
    if (!on_device && n_rows > 0) {
      staged.need(sizeof(int) * 2 * (size_t)n_rows);
      HIP_TRY(hipMemcpyAsync(staged.p, pairs, sizeof(int) * 2 * (size_t)n_rows, hipMemcpyHostToDevice, stream));
      dev = staged.as<int>();
    }
    {
      TIMED("nbr_build");
      if (nbr.built) { (void)hipFree(nbr.built); nbr.built = nullptr; }      // (a table from an explicit list holds every row)
      int rc = build_neighbour_table(stream, top, n_rows, dev, nbr, &scan_scratch.p, &scan_bytes);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("build_neighbour_table: ") + hipGetErrorString((hipError_t)rc)};
      apply_classes();
      order_rows();
    }
    HIP_TRY(hipStreamSynchronize(stream));
    staged.release();
    have_pairs = true;
    ++nbr_gen;
  }
};

// decomposition state of one evaluation on a slab rank (snranks > 1; slab_kernels.hip)
struct SlabState {
  DevBuf owner, owner_prev, mig, bits, counts, totals, lists, imp, exp, mig_in, mig_out, sendb, recvb, ghost, pack, tbuf;
  int prev_na = -1;             // owner_prev holds the owners of this handle's previous decomposed evaluation of prev_na atoms
  int64_t min_cnt[kSlabMaxRanks], mout_cnt[kSlabMaxRanks];   // atoms taken over from / given away to every rank since then
  int n_min = 0, n_mout = 0;
  int64_t tr_send[kSlabMaxRanks], tr_recv[kSlabMaxRanks];
  int64_t imp_cnt[kSlabMaxRanks], exp_cnt[kSlabMaxRanks];     // atoms imported from / exported to every rank
  int n_home = 0, n_act = 0, n_imp = 0, n_exp = 0;
  const int* home = nullptr;    // home atoms, ascending
  const int* rows = nullptr;    // home rows in the pair kernels' order (the table's class-grouped order, filtered)
  const int* act = nullptr;     // polarizable home atoms, ascending
  void release() {
    for (DevBuf* b : {&owner, &owner_prev, &mig, &bits, &counts, &totals, &lists, &imp, &exp, &mig_in, &mig_out, &sendb, &recvb,
                      &ghost, &pack, &tbuf}) b->release();
  }
};

template <class T>
struct Engine : EngineBase {
  // per-atom
  DevBuf sites, grad, pot, fld_pair, fld_recip, field, energies_d;
  bool fmax_clean = false, slot_clean[E_SLOTS] = {false};
  // The energy words are double-buffered: the evaluation of step n accumulates into half (n & 1) while its first
  // kernel zeroes the other half for step n+1 -- no memset dispatch per step.  Eh is pinned host memory for the
  // one device-to-host copy of a step.
  int ehalf = 0;
  bool other_clean = false;
  void* energies_seen = nullptr;
  double* Eh = nullptr;
  double* Ed_cur() { return energies_d.as<double>() + (size_t)ehalf * E_WORDS; }
  // staging for host-pointer calls
  DevBuf s_pos, s_Q, s_pol, s_thole, s_U, s_out, s_dQ, s_par;
  // mesh
  DevBuf mesh, spec, gtabs[4], fft_work, binv_d, bin_cells, bin_sorted, bin_scan, bin_cells_ind, bin_sorted_ind;   // gtabs: Ck_1, Ck_6, Ck_8, Ck_10
  T* gtab_cur = nullptr;
  BinScratch bins, bins_ind;   // brick lists of the site rows / of the compact rows of the SCF increments
  rocfft_plan plan_f = nullptr, plan_b = nullptr, plan_xf = nullptr, plan_xb = nullptr;
  // power-of-two x dimension on one rank: batched 2-D plans of the y-z planes around the fused x pass (fftx_kernels.hip)
  rocfft_plan plan2_f = nullptr, plan2_b = nullptr;
  int fx_khp = 0;              // row pitch (complex numbers) of the spectrum those plans write: K2/2+1 padded to whole lines
  bool use_fx = false;
  DevBuf fx_tw;
  rocfft_execution_info info_f = nullptr;
  int planK[3] = {0, 0, 0}, planR = 0, planRank = 0;
  DevBuf home_list;
  DevBuf act_tmp;
  DevBuf onehot_d, tcount_d;     // typed dispersion of small systems: (Na, n_types) one-hot weights; scratch of the type counts
  void* onehot_for = nullptr;    // ... built in this allocation
  DevBuf act_d, isites, mesh2;   // incremental SCF: polarizable-site list, their compact delta rows, the increment's mesh
  IndTable ind;                  // ... and the polarizable-polarizable part of the neighbour table
  long act_gen = 0, act_top_na = -1;   // act_gen: bumped when the list is rebuilt; the list belongs to a topology of act_top_na atoms
  int act_n = -1;                // its length once the host has seen it (-1: not yet)
  bool act_fresh = false;        // this evaluation rebuilt the list (count still on the device)
  long ind_nbr_gen = -1, ind_act_gen = -1;
  long eval_seq = 0, ind_bins_eval = -1, ind_bins_gen = -1;   // evaluation counter; the evaluation / active set bins_ind was built for
  int ind_bins_n = -1;
  const int* ind_bins_at = nullptr;
  DevBuf dft_tw;          // twiddle tables of the direct-DFT path
  DevBuf bases_d;         // int4 per atom: lowest stencil index on each mesh axis
  bool use_dft = false;   // mesh convolution through dft_kernels.hip instead of rocFFT (single rank, a Bluestein dimension)
  bool use_pfa = false;   // ... through pfa_kernels.hip: a Bluestein dimension too long for plain lines, split N = N1 * N2;
  PfaPlan pfa;            //     spectrum and G tables then live in slot order with pfa.Khp z columns
  DevBuf pfa_tw, pfa_fmap, pfa_ptab, gtab_nat;
  // validity of the cached G table
  struct TabKey { double box[9] = {0}, kappa = -1; int K[3] = {0, 0, 0}, Y0 = 0, ref = 0; bool pfa = false; } tabkey[4];
  static int tab_slot(int which) { return which == 1 ? 0 : (which == 6 ? 1 : (which == 8 ? 2 : 3)); }
  bool warm_regime = false;   // previous polarizable call converged at its first SCF check
  // Residual history of consecutive polarizable calls (MD: every call starts from the previous call's dipoles): the residual
  // of a call's first check is the previous call's last residual plus what one step of motion adds.  scf_growth = that
  // increase as last observed after a call of the same kind, scf_last = the residual the previous call ended with (< 0: no history).  The first cycle is
  // evaluated speculatively with the full kernels only when scf_last + scf_growth predicts that its check will pass.
  double scf_last = -1.0;
  double scf_growth[2][2] = {{0.0, 0.0}, {0.0, 0.0}};    // the last two observed increases after a call without / with a Jacobi
  int scf_nobs[2] = {0, 0};                              // step (they differ: the residual is a maximum norm, not additive);
  int scf_state = 0;                                     // a prediction needs two observations of the current kind
  double scf_contract = -1.0;   // factor by which one Jacobi step shrank the residual in the last call that took steps
                                // ((last / first residual)^(1 / steps); < 0: never observed)
  bool mono_ok = false;       // this evaluation may use the charge-only pair forms (no dE/dQ_local requested)

  ~Engine() override {
    destroy_plans();
    for (DevBuf* b : {&sites, &grad, &pot, &fld_pair, &fld_recip, &field, &energies_d, &s_pos, &s_Q, &s_pol,
                      &s_thole, &s_U, &s_out, &s_dQ, &s_par, &mesh, &spec, &gtabs[0], &gtabs[1], &gtabs[2], &gtabs[3], &fft_work, &binv_d, &scan_scratch, &bin_cells,
                      &bin_sorted, &bin_scan, &home_list, &dft_tw, &bases_d, &vir_d, &act_d, &isites, &mesh2, &act_tmp,
                      &rq_d, &pfa_tw, &pfa_fmap, &pfa_ptab, &gtab_nat, &fx_tw, &bin_cells_ind, &bin_sorted_ind, &srow_d, &onehot_d, &tcount_d, &prune_rowptr, &prune_cnt, &prune_col})
      b->release();
    free_topology();
    if (ind.end) (void)hipFree(ind.end);
    if (ind.col) (void)hipFree(ind.col);
    cells.release();
    sl.release();
    free_programs();
    if (Eh) (void)hipHostFree(Eh);
    prof.destroy();
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }

  void destroy_plans() {
    if (plan_f) rocfft_plan_destroy(plan_f);
    if (plan_b) rocfft_plan_destroy(plan_b);
    if (plan_xf) rocfft_plan_destroy(plan_xf);
    if (plan_xb) rocfft_plan_destroy(plan_xb);
    if (plan2_f) rocfft_plan_destroy(plan2_f);
    if (plan2_b) rocfft_plan_destroy(plan2_b);
    for (int k = 0; k < 4; ++k) {
      if (plan2n_f[k]) rocfft_plan_destroy(plan2n_f[k]);
      if (plan2n_b[k]) rocfft_plan_destroy(plan2n_b[k]);
      plan2n_f[k] = plan2n_b[k] = nullptr;
    }
    if (info_f) rocfft_execution_info_destroy(info_f);
    plan_f = plan_b = plan_xf = plan_xb = plan2_f = plan2_b = nullptr;
    use_fx = false;
    info_f = nullptr;
    planK[0] = planK[1] = planK[2] = 0;
  }

  void set_ewald(double kappa_, int K1, int K2, int K3, int lmax_, int lpol_) override {
    ARG_CHECK(kappa_ > 0, "kappa must be positive");
    ARG_CHECK(K1 >= 6 && K2 >= 6 && K3 >= 6, "PME mesh must be at least 6 points per dimension (order-6 splines)");
    ARG_CHECK(lmax_ >= 0 && lmax_ <= 2, "l > 2 (beyond quadrupole) not supported");   // admp/recip.py:275
    kappa = kappa_; K[0] = K1; K[1] = K2; K[2] = K3; lmax = lmax_; lpol = lpol_ ? 1 : 0;
    have_ewald = true;
  }

  // ---- slab decomposition state (nranks == 1: the whole mesh is local) ---------------------------------
  // rank s owns mesh planes [X0, X1) along x and, in the transposed (k-space) layout, rows [Y0, Y1) along y;
  // the local real mesh additionally carries kGhost planes above X1 (stencil overhang / phi halo).
  static constexpr int kGhost = 5;
  int X0 = 0, X1 = 0, Y0 = 0, Y1 = 0;
  void update_slab() {
    X0 = (int)((long)srank * K[0] / snranks); X1 = (int)((long)(srank + 1) * K[0] / snranks);
    Y0 = (int)((long)srank * K[1] / snranks); Y1 = (int)((long)(srank + 1) * K[1] / snranks);
    if (snranks > 1) {
      ARG_CHECK(have_comm, "slab-decomposed handle without a communicator (admp_set_comm)");
      const int64_t nh = K[2] / 2 + 1;
      for (int s = 0; s < snranks; ++s) {
        int w = (int)((long)(s + 1) * K[0] / snranks) - (int)((long)s * K[0] / snranks);
        ARG_CHECK(w >= 6, "slab decomposition needs at least 6 mesh planes per rank along x");
        int wy = (int)((long)(s + 1) * K[1] / snranks) - (int)((long)s * K[1] / snranks);
        ARG_CHECK(wy >= 1, "slab decomposition needs at least 1 mesh row per rank along y");
        // transposes: block (x in my slab) x (y in slab s) travels to rank s, block (x in slab s) x (my y rows) comes back
        sl.tr_send[s] = (int64_t)(X1 - X0) * wy * nh * 2;
        sl.tr_recv[s] = (int64_t)w * (Y1 - Y0) * nh * 2;
      }
    }
  }
  // ---- decomposition state of one evaluation (snranks > 1; slab_kernels.hip) ------------------------------------------
  SlabState sl;
  static constexpr int real_dtype() { return sizeof(T) == 4 ? ADMP_T_F32 : ADMP_T_F64; }
  // Who owns what in this evaluation, derived by every rank from the replicated inputs (no communication): home atoms, home
  // rows in pair-kernel order, polarizable home atoms, imports per owner, exports per reader.  One host read (the counts).
  void decompose() { decompose(K[0], X0, X1, nbr.order, lpol != 0); }
  // K0v / X0v / X1v: the planes the ownership rule works on (the PME mesh's; a handle without a mesh takes a virtual one);
  // order: the row order of the table to filter (nullptr: natural order); with_dipoles: the evaluation carries induced
  // dipoles from call to call (atoms that change hands take theirs along)
  void decompose(int K0v, int X0v, int X1v, const int* order, bool with_dipoles = false) {
    const int na = top.na, N = snranks, me = srank;
    ARG_CHECK(N <= kSlabMaxRanks, "at most 28 slab ranks");
    sl.owner.need(sizeof(int) * (size_t)na);
    sl.bits.need(sizeof(int) * (size_t)na);
    // Atoms that changed hands since the previous evaluation (the home rule follows the positions): their dipoles are valid on
    // the PREVIOUS owner only when the caller keeps home rows (outputs of a decomposed call), so they travel with the atom.
    const bool track = with_dipoles && sl.prev_na == na && sl.owner_prev.p;
    sl.owner_prev.need(sizeof(int) * (size_t)na);
    if (track) sl.mig.need(sizeof(int) * (size_t)na);
    // Columns of the ordered compaction.  Only `rows` (the home rows in the table's order: a different sequence) takes a pass
    // of its own; home / polarizable home / the per-peer import, export and migrant columns are bins of ONE pass over the atoms
    // (slab_kernels.hip k_slab_bins; ADMP_SLAB_BINS=0: a pass per column as in round 3, for A/B and tests).
    static const bool bins_on = [] { const char* e = getenv("ADMP_SLAB_BINS"); return !(e && atoi(e) == 0); }();
    SlabCols cs;
    SlabBins sb;
    sb.N = bins_on ? N : 0; sb.me = me;
    auto col = [&](const int* seq, int len, int mask, int want, const int* src = nullptr, bool binned = false) {
      cs.seq[cs.ncols] = seq; cs.len[cs.ncols] = len; cs.mask[cs.ncols] = mask; cs.want[cs.ncols] = want; cs.src[cs.ncols] = src;
      cs.binned[cs.ncols] = binned && bins_on ? 1 : 0;
      return cs.ncols++;
    };
    const int c_home = col(nullptr, na, kSlabHome, kSlabHome, nullptr, true);
    const int c_rows = col(order, na, kSlabHome, kSlabHome);
    const int c_act = col(nullptr, na, kSlabHome | kSlabPolar, kSlabHome | kSlabPolar, nullptr, true);
    sb.c_home = c_home; sb.c_act = c_act;
    int c_imp[kSlabMaxRanks], c_exp[kSlabMaxRanks], c_min[kSlabMaxRanks], c_mout[kSlabMaxRanks];
    for (int t = 0; t < kSlabMaxRanks; ++t) sb.c_imp[t] = sb.c_exp[t] = sb.c_min[t] = sb.c_mout[t] = -1;
    for (int t = 0; t < N; ++t) {
      c_imp[t] = c_exp[t] = c_min[t] = c_mout[t] = -1;
      if (t == me) continue;
      c_imp[t] = col(nullptr, na, kSlabHome | (1 << t), 1 << t, nullptr, true);
      c_exp[t] = col(nullptr, na, kSlabHome | (1 << t), kSlabHome | (1 << t), nullptr, true);
      if (track) {
        c_min[t] = col(nullptr, na, kSlabHome | (1 << t), 1 << t, sl.mig.as<int>(), true);
        c_mout[t] = col(nullptr, na, kSlabHome | (1 << t), kSlabHome | (1 << t), sl.mig.as<int>(), true);
      }
      sb.c_imp[t] = c_imp[t]; sb.c_exp[t] = c_exp[t]; sb.c_min[t] = c_min[t]; sb.c_mout[t] = c_mout[t];
    }
    sl.counts.need(sizeof(int) * (size_t)cs.ncols * (size_t)std::max(1, slab_compact_blocks(na)));
    sl.totals.need(sizeof(int) * (kSlabMaxCols + 1));
    sl.lists.need(sizeof(int) * (size_t)cs.ncols * (size_t)na);
    {
      TIMED("slab_decompose");
      int rc = launch_slab_decompose(stream, na, nbr, top, ev.bases, ev.pol, (int)sizeof(T), X1v - X0v, K0v, X0v, N, me,
                                     sl.owner.as<int>(), sl.bits.as<int>(), cs, sb, sl.counts.as<int>(), sl.totals.as<int>(),
                                     sl.lists.as<int>(), track ? sl.owner_prev.as<int>() : nullptr, track ? sl.mig.as<int>() : nullptr);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("slab decomposition: ") + hipGetErrorString((hipError_t)rc)};
    }
    int tot[kSlabMaxCols + 1];
    HIP_TRY(hipMemcpyAsync(tot, sl.totals.p, sizeof(int) * (cs.ncols + 1), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (tot[cs.ncols])
      throw Err{ADMP_E_STATE, "an atom entered this rank's slab whose row the neighbour table does not hold: the table of a slab "
                              "rank covers its slab plus a margin of half the list cutoff (admp_set_pairs_from_positions) -- "
                              "rebuild the pair list"};
    const int* L = sl.lists.as<int>();
    sl.n_home = tot[c_home]; sl.home = L + (size_t)c_home * na;
    sl.rows = L + (size_t)c_rows * na;
    sl.n_act = tot[c_act]; sl.act = L + (size_t)c_act * na;
    if (tot[c_rows] != sl.n_home) throw Err{ADMP_E_STATE, "slab decomposition: row order and home list disagree"};
    SlabSegs si, se;
    si.off[0] = se.off[0] = 0;
    for (int t = 0; t < N; ++t) {
      sl.imp_cnt[t] = t == me ? 0 : tot[c_imp[t]];
      sl.exp_cnt[t] = t == me ? 0 : tot[c_exp[t]];
      if (t == me) continue;
      si.col[si.n] = c_imp[t]; si.off[si.n + 1] = si.off[si.n] + tot[c_imp[t]]; ++si.n;
      se.col[se.n] = c_exp[t]; se.off[se.n + 1] = se.off[se.n] + tot[c_exp[t]]; ++se.n;
    }
    sl.n_imp = si.off[si.n]; sl.n_exp = se.off[se.n];
    sl.imp.need(sizeof(int) * (size_t)std::max(1, sl.n_imp));
    sl.exp.need(sizeof(int) * (size_t)std::max(1, sl.n_exp));
    launch_slab_concat(stream, si, L, (long)na, sl.imp.as<int>());
    launch_slab_concat(stream, se, L, (long)na, sl.exp.as<int>());
    sl.n_min = sl.n_mout = 0;
    if (track) {
      SlabSegs mi, mo;
      mi.off[0] = mo.off[0] = 0;
      for (int t = 0; t < N; ++t) {
        sl.min_cnt[t] = t == me ? 0 : tot[c_min[t]];
        sl.mout_cnt[t] = t == me ? 0 : tot[c_mout[t]];
        if (t == me) continue;
        mi.col[mi.n] = c_min[t]; mi.off[mi.n + 1] = mi.off[mi.n] + tot[c_min[t]]; ++mi.n;
        mo.col[mo.n] = c_mout[t]; mo.off[mo.n + 1] = mo.off[mo.n] + tot[c_mout[t]]; ++mo.n;
      }
      sl.n_min = mi.off[mi.n]; sl.n_mout = mo.off[mo.n];
      sl.mig_in.need(sizeof(int) * (size_t)std::max(1, sl.n_min));
      sl.mig_out.need(sizeof(int) * (size_t)std::max(1, sl.n_mout));
      launch_slab_concat(stream, mi, L, (long)na, sl.mig_in.as<int>());
      launch_slab_concat(stream, mo, L, (long)na, sl.mig_out.as<int>());
    }
    // the owners of this evaluation are the "previous owners" of the next one
    HIP_TRY(hipMemcpyAsync(sl.owner_prev.p, sl.owner.p, sizeof(int) * (size_t)na, hipMemcpyDeviceToDevice, stream));
    sl.prev_na = with_dipoles ? na : -1;
    const size_t w = 9 * sizeof(T) * (size_t)std::max(1, std::max(std::max(sl.n_imp, sl.n_exp), std::max(sl.n_min, sl.n_mout)));
    sl.sendb.need(w); sl.recvb.need(w);
    ev.n_home = sl.n_home;
    ev.home = sl.home;
  }
  // the dipoles of the atoms that changed hands: previous owner -> new owner
  void exchange_migrants() {
    TIMED("comm_halo_dipoles");                            // (every rank takes part: the caller's condition is rank-uniform)
    int64_t sc[kSlabMaxRanks], rc[kSlabMaxRanks];
    for (int t = 0; t < snranks; ++t) { sc[t] = 3 * sl.mout_cnt[t]; rc[t] = 3 * sl.min_cnt[t]; }
    launch_halo_u_pack<T>(stream, sl.n_mout, 0, sl.mig_out.as<int>(), ev.U, sites.as<Site<T>>(), sl.sendb.as<T>());
    c_all_to_all_v(sl.sendb.p, sc, sl.recvb.p, rc, real_dtype(), ADMP_TAG_HALO_DIPOLES);
    launch_halo_u_unpack<T>(stream, sl.n_min, 0, sl.mig_in.as<int>(), sl.recvb.as<T>(), ev.U, sites.as<Site<T>>());
  }
  // rows of width w of the halo atoms: owners -> readers (to_readers: exports out, imports in) or back (gradient contributions)
  void halo_counts(int w, bool to_readers, int64_t* sc, int64_t* rc) const {
    for (int t = 0; t < snranks; ++t) {
      sc[t] = (to_readers ? sl.exp_cnt[t] : sl.imp_cnt[t]) * w;
      rc[t] = (to_readers ? sl.imp_cnt[t] : sl.exp_cnt[t]) * w;
    }
  }
  // what 0: the Cartesian dipoles of the atoms this rank reads <- their owners' values (start of an evaluation: the result
  // must not depend on rows of U the rank does not own); what 1: the last Jacobi step's change of those dipoles
  void exchange_U(int what) {
    TIMED("comm_halo_dipoles");
    int64_t sc[kSlabMaxRanks], rc[kSlabMaxRanks];
    halo_counts(3, true, sc, rc);
    launch_halo_u_pack<T>(stream, sl.n_exp, what, sl.exp.as<int>(), ev.U, sites.as<Site<T>>(), sl.sendb.as<T>());
    c_all_to_all_v(sl.sendb.p, sc, sl.recvb.p, rc, real_dtype(), ADMP_TAG_HALO_DIPOLES);
    launch_halo_u_unpack<T>(stream, sl.n_imp, what, sl.imp.as<int>(), sl.recvb.as<T>(), ev.U, sites.as<Site<T>>());
  }
  // what the closing kernel added to atoms of other ranks (frame adjoint of molecules across a slab face) goes to the owners
  void exchange_grad(T* grad_p) {
    TIMED("comm_halo_gradient");
    int64_t sc[kSlabMaxRanks], rc[kSlabMaxRanks];
    halo_counts(3, false, sc, rc);
    launch_rows_gather<T>(stream, sl.n_imp, 3, sl.imp.as<int>(), grad_p, sl.sendb.as<T>());
    c_all_to_all_v(sl.sendb.p, sc, sl.recvb.p, rc, real_dtype(), ADMP_TAG_HALO_GRADIENT);
    launch_rows_scatter<T>(stream, 1, sl.n_exp, 3, sl.exp.as<int>(), sl.recvb.as<T>(), grad_p);
  }
  void slab_home(int32_t* out, int* n_home, int* n_import) override {
    ARG_CHECK(snranks > 1 && sl.home, "no decomposed evaluation has run on this handle");
    if (out) HIP_TRY(hipMemcpyAsync(out, sl.home, sizeof(int) * (size_t)sl.n_home, hipMemcpyDeviceToDevice, stream));
    if (n_home) *n_home = sl.n_home;
    if (n_import) *n_import = sl.n_imp;
  }
  // the residual word of an SCF check: maximum over the ranks (bit patterns of non-negative doubles are doubles)
  void reduce_check_word(unsigned long long* word) {
    if (snranks > 1) { TIMED("comm_scf_max"); c_all_reduce(word, 1, ADMP_T_F64, ADMP_OP_MAX, ADMP_TAG_SCF_MAX); }
  }
  int nloc0() const { return snranks > 1 ? (X1 - X0) + kGhost : K[0]; }
  int nxown() const { return snranks > 1 ? (X1 - X0) : K[0]; }
  int nyown() const { return snranks > 1 ? (Y1 - Y0) : K[1]; }

  void ensure_mesh() {
    update_slab();
    const size_t K2h = (size_t)(K[2] / 2 + 1);
    const size_t nreal = (size_t)nloc0() * K[1] * K[2];
    size_t nspec = (size_t)nxown() * K[1] * K2h;
    const size_t nspec_t = (size_t)K[0] * nyown() * K2h;
    if (nspec_t > nspec) nspec = nspec_t;
    // fused-x path: the kz rows of the spectrum are padded to whole 128-byte lines.  With K2/2+1 = 129 complex numbers per row
    // every row of a tile straddles two lines; padded to 144, rocFFT's batched 2-D r2c / c2r of the 256^3 f32 mesh take
    // 61 / 64 us instead of 88 / 91 (tools/ubench/rocfft_yz_layouts.cpp).
    static const bool fx_off = [] { const char* e = getenv("ADMP_FUSED_X"); return e && atoi(e) == 0; }();
    const bool want_fx = snranks == 1 && !fx_off && fftx_usable(K[0]);
    const size_t per_line = 128 / (2 * sizeof(T));
    // (a slab rank pads the rows of its batched 2-D transforms the same way; the transposes carry the unpadded rows)
    const int khp = (want_fx || snranks > 1) ? (int)((K2h + per_line - 1) / per_line * per_line) : (int)K2h;
    if ((size_t)nxown() * K[1] * khp > nspec) nspec = (size_t)nxown() * K[1] * khp;
    mesh.need(nreal * sizeof(T));
    spec.need(nspec * 2 * sizeof(T));
    binv_d.need(9 * sizeof(double));
    if (planK[0] == K[0] && planK[1] == K[1] && planK[2] == K[2] && planR == snranks && planRank == srank && plan_f) return;
    fx_khp = khp;
    destroy_plans();
    std::call_once(g_fft_once, [] { rocfft_setup(); });
    const rocfft_precision pr = sizeof(T) == 4 ? rocfft_precision_single : rocfft_precision_double;
    size_t wmax = 0, w = 0;
    if (snranks == 1) {
      const size_t len[3] = {(size_t)K[2], (size_t)K[1], (size_t)K[0]};   // rocFFT: fastest dimension first
      FFT_TRY(rocfft_plan_create(&plan_f, rocfft_placement_notinplace, rocfft_transform_type_real_forward, pr, 3, len, 1, nullptr));
      FFT_TRY(rocfft_plan_create(&plan_b, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, pr, 3, len, 1, nullptr));
      if (want_fx) {
        const size_t len2[2] = {(size_t)K[2], (size_t)K[1]};
        const size_t rs[2] = {1, (size_t)K[2]}, cs[2] = {1, (size_t)khp};
        const size_t rdist = (size_t)K[1] * K[2], cdist = (size_t)K[1] * khp;
        rocfft_plan_description df = nullptr, db = nullptr;
        FFT_TRY(rocfft_plan_description_create(&df));
        FFT_TRY(rocfft_plan_description_set_data_layout(df, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr,
                                                        nullptr, 2, rs, rdist, 2, cs, cdist));
        FFT_TRY(rocfft_plan_description_create(&db));
        FFT_TRY(rocfft_plan_description_set_data_layout(db, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr,
                                                        nullptr, 2, cs, cdist, 2, rs, rdist));
        FFT_TRY(rocfft_plan_create(&plan2_f, rocfft_placement_notinplace, rocfft_transform_type_real_forward, pr, 2, len2, (size_t)K[0], df));
        FFT_TRY(rocfft_plan_create(&plan2_b, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, pr, 2, len2, (size_t)K[0], db));
        rocfft_plan_description_destroy(df);
        rocfft_plan_description_destroy(db);
        FFT_TRY(rocfft_plan_get_work_buffer_size(plan2_f, &w)); if (w > wmax) wmax = w;
        FFT_TRY(rocfft_plan_get_work_buffer_size(plan2_b, &w)); if (w > wmax) wmax = w;
        std::vector<T> tw((size_t)K[0]);
        for (int m = 0; m < K[0] / 2; ++m) {
          const double th = 2.0 * M_PI * (double)m / (double)K[0];
          tw[2 * m] = (T)std::cos(th);
          tw[2 * m + 1] = (T)std::sin(th);
        }
        fx_tw.need(tw.size() * sizeof(T));
        HIP_TRY(hipMemcpy(fx_tw.p, tw.data(), tw.size() * sizeof(T), hipMemcpyHostToDevice));
        use_fx = true;
      }
    } else {
      // distributed transform = batched 2-D r2c over the owned planes, all-to-all transpose (done by the caller over
      // RCCL), batched strided 1-D c2c along x
      const size_t len2[2] = {(size_t)K[2], (size_t)K[1]};
      {
        const size_t rs[2] = {1, (size_t)K[2]}, cs[2] = {1, (size_t)khp};
        const size_t rdist = (size_t)K[1] * K[2], cdist = (size_t)K[1] * khp;
        rocfft_plan_description df = nullptr, db = nullptr;
        FFT_TRY(rocfft_plan_description_create(&df));
        FFT_TRY(rocfft_plan_description_set_data_layout(df, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr,
                                                        nullptr, 2, rs, rdist, 2, cs, cdist));
        FFT_TRY(rocfft_plan_description_create(&db));
        FFT_TRY(rocfft_plan_description_set_data_layout(db, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr,
                                                        nullptr, 2, cs, cdist, 2, rs, rdist));
        FFT_TRY(rocfft_plan_create(&plan_f, rocfft_placement_notinplace, rocfft_transform_type_real_forward, pr, 2, len2, (size_t)nxown(), df));
        FFT_TRY(rocfft_plan_create(&plan_b, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, pr, 2, len2, (size_t)nxown(), db));
        rocfft_plan_description_destroy(df);
        rocfft_plan_description_destroy(db);
      }
      const size_t len1[1] = {(size_t)K[0]};
      const size_t stride[1] = {(size_t)nyown() * K2h};
      const size_t batch = (size_t)nyown() * K2h;
      for (int dir = 0; dir < 2; ++dir) {
        rocfft_plan_description desc = nullptr;
        FFT_TRY(rocfft_plan_description_create(&desc));
        FFT_TRY(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved,
                                                        rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, stride, 1,
                                                        1, stride, 1));
        FFT_TRY(rocfft_plan_create(dir == 0 ? &plan_xf : &plan_xb, rocfft_placement_inplace,
                                   dir == 0 ? rocfft_transform_type_complex_forward : rocfft_transform_type_complex_inverse,
                                   pr, 1, len1, batch, desc));
        FFT_TRY(rocfft_plan_description_destroy(desc));
      }
      FFT_TRY(rocfft_plan_get_work_buffer_size(plan_xf, &w)); if (w > wmax) wmax = w;
      FFT_TRY(rocfft_plan_get_work_buffer_size(plan_xb, &w)); if (w > wmax) wmax = w;
      if (!fx_off && fftx_usable(K[0])) {   // the fused x pass works on the transposed layout [K0][ny][K2/2+1] as well
        std::vector<T> tw((size_t)K[0]);
        for (int m = 0; m < K[0] / 2; ++m) {
          const double th = 2.0 * M_PI * (double)m / (double)K[0];
          tw[2 * m] = (T)std::cos(th);
          tw[2 * m + 1] = (T)std::sin(th);
        }
        fx_tw.need(tw.size() * sizeof(T));
        HIP_TRY(hipMemcpy(fx_tw.p, tw.data(), tw.size() * sizeof(T), hipMemcpyHostToDevice));
        use_fx = true;
      }
    }
    FFT_TRY(rocfft_plan_get_work_buffer_size(plan_f, &w)); if (w > wmax) wmax = w;
    FFT_TRY(rocfft_plan_get_work_buffer_size(plan_b, &w)); if (w > wmax) wmax = w;
    fft_work.need(wmax + 16);
    FFT_TRY(rocfft_execution_info_create(&info_f));
    if (wmax) FFT_TRY(rocfft_execution_info_set_work_buffer(info_f, fft_work.p, wmax));
    planK[0] = K[0]; planK[1] = K[1]; planK[2] = K[2]; planR = snranks; planRank = srank;
    for (auto& k : tabkey) k.kappa = -1;   // mesh changed: tables stale
    setup_dft();
  }

  // Direct-DFT convolution (dft_kernels.hip) when rocFFT would need Bluestein for some dimension (largest prime factor
  // above 13) and the mesh is small enough for O(N^2) lines.  ADMP_DFT=0 disables it, ADMP_DFT=1 forces it for any mesh
  // with all dimensions <= 160 (tests).
  void setup_dft() {
    use_dft = use_pfa = false;
    if (snranks != 1) return;
    const char* e = getenv("ADMP_DFT");
    const int mode = e ? atoi(e) : -1;
    if (mode == 0) return;
    bool small = true, hard = false;
    for (int d = 0; d < 3; ++d) {
      small = small && K[d] >= 2 && K[d] <= 160;
      hard = hard || largest_prime_factor(K[d]) > 13;
    }
    if (mode == 2 && setup_pfa()) return;                 // tests: the split form for any mesh it can take
    if (!small && hard && setup_pfa()) return;
    if (!small || !(hard || mode == 1 || mode == 2)) return;
    std::vector<T> tw(2 * (size_t)(K[0] + K[1] + K[2]));
    size_t o = 0;
    for (int d = 0; d < 3; ++d)
      for (int m = 0; m < K[d]; ++m) {
        const double th = 2.0 * M_PI * (double)m / (double)K[d];
        tw[o++] = (T)std::cos(th);
        tw[o++] = (T)std::sin(th);
      }
    dft_tw.need(tw.size() * sizeof(T));
    HIP_TRY(hipMemcpy(dft_tw.p, tw.data(), tw.size() * sizeof(T), hipMemcpyHostToDevice));
    use_dft = true;
  }
  // Two-level direct DFT (pfa_kernels.hip): every dimension either fits the plain lines (<= 160) or splits into a smooth
  // cofactor N1 <= 32 and a prime power N2 <= 160.  ADMP_DFT=2 forces it for any mesh it can split (tests); ADMP_PFA=0 off.
  bool setup_pfa() {
    static const bool off = [] { const char* e = getenv("ADMP_PFA"); return e && atoi(e) == 0; }();
    if (off) return false;
    for (int d = 0; d < 3; ++d)
      if (!pfa_split(K[d], &pfa.ax[d])) return false;
    pfa.Khp = pfa.ax[2].N1 * (pfa.ax[2].N2 / 2 + 1);
    std::vector<T> tw;
    for (int d = 0; d < 3; ++d) {
      pfa.tw_off[d] = (int)(tw.size() / 2);
      for (int n : {pfa.ax[d].N2, pfa.ax[d].N1})
        for (int m = 0; m < n; ++m) {
          const double th = 2.0 * M_PI * (double)m / (double)n;
          tw.push_back((T)std::cos(th));
          tw.push_back((T)std::sin(th));
        }
    }
    pfa_tw.need(tw.size() * sizeof(T));
    HIP_TRY(hipMemcpy(pfa_tw.p, tw.data(), tw.size() * sizeof(T), hipMemcpyHostToDevice));
    std::vector<int> fm((size_t)K[0] + K[1] + pfa.Khp);
    pfa_freq_of_slot(pfa.ax[0], fm.data());
    pfa_freq_of_slot(pfa.ax[1], fm.data() + K[0]);
    pfa_freq_of_zcolumn(pfa.ax[2], fm.data() + K[0] + K[1]);
    pfa_fmap.need(fm.size() * sizeof(int));
    HIP_TRY(hipMemcpy(pfa_fmap.p, fm.data(), fm.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<int> pt((size_t)K[0] + K[1] + K[2]);
    pfa_index_table(pfa.ax[0], pt.data());
    pfa_index_table(pfa.ax[1], pt.data() + K[0]);
    pfa_index_table(pfa.ax[2], pt.data() + K[0] + K[1]);
    pfa_ptab.need(pt.size() * sizeof(int));
    HIP_TRY(hipMemcpy(pfa_ptab.p, pt.data(), pt.size() * sizeof(int), hipMemcpyHostToDevice));
    pfa.ptab[0] = pfa_ptab.as<int>(); pfa.ptab[1] = pfa.ptab[0] + K[0]; pfa.ptab[2] = pfa.ptab[1] + K[1];
    spec.need((size_t)K[0] * K[1] * pfa.Khp * 2 * sizeof(T));
    use_pfa = true;
    return true;
  }
  // mesh <- IFFT( G * FFT(mesh) ), energies[slot] += sum w G |S|^2 : the whole k-space leg of one reciprocal pass
  // accum (optional): a second mesh the result is to be ADDED to; returns true when the last pass did that itself (the direct
  // DFT writes every word once anyway), false when the caller still has to add
  // out (optional, rocFFT paths): phi is written there instead of over mesh_p
  // sp: the forward transform builds its planes from these sites (no spread kernel ran, mesh_p holds nothing yet): small
  // double-precision systems on the direct-DFT plane kernels (spread_fused())
  bool spread_fused(int n) const {
    return use_dft && !use_pfa && snranks == 1 && !ev.home && dft_zy_spread_fits<T>(K, n);
  }
  PlaneSpread<T> plane_spread(int n, const Site<T>* rows, int lp, const int4* bases) const {
    PlaneSpread<T> sp;
    sp.na = n; sp.lpol = lp; sp.sites = rows; sp.bases = bases; sp.g = ev.g;
    return sp;
  }
  // rider: an SCF field kernel whose workgroups run inside the x pass (pair_kernels.hip k_xconv_pair; direct-DFT meshes only)
  bool rider_ok() const {
    static const bool on = [] { const char* e = getenv("ADMP_FIELD_RIDER"); return !(e && atoi(e) == 0); }();
    return on && use_dft && !use_pfa && snranks == 1 && !overlap_ok();
  }
  bool convolve(T* mesh_p, T* spec_p, const T* gtab, int slot, T* accum = nullptr, T* out = nullptr,
                const PlaneSpread<T>* sp = nullptr, const FieldRider<T>* rider = nullptr) {
    double* Ed = Ed_cur();
    T* mesh_o = out ? out : mesh_p;
    ARG_CHECK(!sp || (use_dft && !use_pfa && snranks == 1), "internal: plane spread on a transform path without it");
    ARG_CHECK(!(rider && rider->kind) || (use_dft && !use_pfa && snranks == 1), "internal: field rider on a transform path without it");
    if (snranks > 1) {
      // x-slab ranks: the stencils of the home atoms overhang into kGhost planes of the next rank (added there before the
      // transform), the 3-D transform is batched 2-D r2c on the owned planes -> transpose (all-to-all over the ranks: every
      // GPU exchanges a block with each of the others) -> x lines forward * G * inverse on the y rows this rank owns ->
      // transpose back -> 2-D c2r, and the gather needs the next rank's first planes of phi as its ghost planes
      const size_t plane = (size_t)K[1] * K[2];
      const int nx = nxown(), ny = nyown(), nh = K[2] / 2 + 1;
      sl.ghost.need(kGhost * plane * sizeof(T));
      sl.pack.need((size_t)nx * K[1] * nh * 2 * sizeof(T));
      sl.tbuf.need((size_t)K[0] * ny * nh * 2 * sizeof(T));
      { TIMED("comm_ghost"); c_shift(mesh_p + (size_t)nx * plane, sl.ghost.p, (int64_t)(kGhost * plane), real_dtype(), 1, ADMP_TAG_GHOST); }
      { TIMED("mesh_add"); launch_mesh_add<T>(stream, (long)(kGhost * plane), mesh_p, sl.ghost.as<T>()); }
      fft_forward(mesh_p, spec_p);
      { TIMED("transpose_pack"); launch_transpose_pack<T>(stream, nx, K[1], nh, fx_khp, snranks, 0, spec_p, sl.pack.as<T>()); }
      { TIMED("comm_transpose"); c_all_to_all_v(sl.pack.p, sl.tr_send, sl.tbuf.p, sl.tr_recv, real_dtype(), ADMP_TAG_TRANSPOSE); }
      if (use_fx) {
        TIMED("fftx_kspace");
        launch_fftx_conv<T>(stream, K, fx_tw.as<T>(), sl.tbuf.as<T>(), gtab, Ed, slot, nh, ny);
      } else {
        fft_x(sl.tbuf.as<T>(), 0);
        { TIMED("kspace"); launch_kspace<T>(stream, K, ny, gtab, sl.tbuf.as<T>(), Ed, slot); }
        fft_x(sl.tbuf.as<T>(), 1);
      }
      { TIMED("comm_transpose"); c_all_to_all_v(sl.tbuf.p, sl.tr_recv, sl.pack.p, sl.tr_send, real_dtype(), ADMP_TAG_TRANSPOSE); }
      { TIMED("transpose_pack"); launch_transpose_pack<T>(stream, nx, K[1], nh, fx_khp, snranks, 1, spec_p, sl.pack.as<T>()); }
      fft_inverse(spec_p, mesh_o);
      { TIMED("comm_ghost"); c_shift(mesh_o, mesh_o + (size_t)nx * plane, (int64_t)(kGhost * plane), real_dtype(), 0, ADMP_TAG_GHOST); }
      (void)accum;
      return false;
    }
    if (use_pfa) {
      const T* tw = pfa_tw.as<T>();
      { TIMED("dft_z_r2c"); launch_pfa_z<T>(stream, pfa, tw, mesh_p, spec_p, 0); }
      { TIMED("dft_y_fwd"); launch_pfa_y<T>(stream, pfa, tw, spec_p, 0); }
      DftTabs<T> tabs;
      tabs.p[0] = gtab;
      { TIMED("dft_x_kspace"); launch_pfa_x_conv<T>(stream, pfa, tw, spec_p, tabs, Ed, slot); }
      { TIMED("dft_y_inv"); launch_pfa_y<T>(stream, pfa, tw, spec_p, 1); }
      { TIMED("dft_z_c2r"); launch_pfa_z<T>(stream, pfa, tw, mesh_p, spec_p, 1); }
      return false;
    }
    if (use_dft) {
      const T* tw = dft_tw.as<T>();
      const bool planes = dft_zy_fits<T>(K);         // z and y lines of a plane in one workgroup (dft_kernels.hip)
      ARG_CHECK(!sp || planes, "internal: plane spread without the plane kernels");
      if (planes && sp) { TIMED("dft_spread_zy_fwd"); launch_dft_zy<T>(stream, K, tw, mesh_p, spec_p, 0, 1, 0, 0, nullptr, sp); }
      else if (planes) { TIMED("dft_zy_fwd"); launch_dft_zy<T>(stream, K, tw, mesh_p, spec_p, 0); }
      else {
        { TIMED("dft_z_r2c"); launch_dft_z<T>(stream, K, tw, mesh_p, spec_p, 0); }
        { TIMED("dft_y_fwd"); launch_dft_y<T>(stream, K, tw, spec_p, 0); }
      }
      DftTabs<T> tabs;
      tabs.p[0] = gtab;
      if (rider && rider->kind) { TIMED("dft_x_kspace"); launch_dft_x_conv_rider<T>(stream, K, tw, spec_p, tabs, Ed, slot, *rider); }
      else { TIMED("dft_x_kspace"); launch_dft_x_conv<T>(stream, K, tw, spec_p, tabs, Ed, slot); }
      bool added;
      if (planes) { TIMED("dft_yz_inv"); added = launch_dft_zy<T>(stream, K, tw, mesh_p, spec_p, 1, 1, 0, 0, accum); }
      else {
        { TIMED("dft_y_inv"); launch_dft_y<T>(stream, K, tw, spec_p, 1); }
        { TIMED("dft_z_c2r"); added = launch_dft_z<T>(stream, K, tw, mesh_p, spec_p, 1, 1, 0, 0, accum); }
      }
      return added;
    }
    if (use_fx) {      // rocFFT for the y-z planes, one fused kernel for x forward * G * x inverse
      run_plan("rocfft_r2c_yz", plan2_f, mesh_p, spec_p);
      { TIMED("fftx_kspace"); launch_fftx_conv<T>(stream, K, fx_tw.as<T>(), spec_p, gtab, Ed, slot, fx_khp); }
      run_plan("rocfft_c2r_yz", plan2_b, spec_p, mesh_o);
      return false;
    }
    fft_forward(mesh_p, spec_p);
    { TIMED("kspace"); launch_kspace<T>(stream, K, nyown(), gtab, spec_p, Ed, slot); }
    fft_inverse(spec_p, mesh_o);
    return false;
  }

  // Batched y-z plans of the fused-x path for nb meshes at once (dispersion PME: C6 / C8 / C10): the same 2-D r2c / c2r
  // plans with batch nb * K0 over meshes / spectra stored back to back -- two rocFFT executions per call instead of 2 nb.
  rocfft_plan plan2n_f[4] = {nullptr, nullptr, nullptr, nullptr}, plan2n_b[4] = {nullptr, nullptr, nullptr, nullptr};
  void ensure_batched_plans(int nb) {
    ARG_CHECK(nb >= 2 && nb <= 3 && use_fx && snranks == 1, "internal: batched y-z plans");
    if (plan2n_f[nb]) return;
    const rocfft_precision pr = sizeof(T) == 4 ? rocfft_precision_single : rocfft_precision_double;
    const size_t len2[2] = {(size_t)K[2], (size_t)K[1]};
    const size_t rs[2] = {1, (size_t)K[2]}, cs[2] = {1, (size_t)fx_khp};
    const size_t rdist = (size_t)K[1] * K[2], cdist = (size_t)K[1] * fx_khp;
    rocfft_plan_description df = nullptr, db = nullptr;
    FFT_TRY(rocfft_plan_description_create(&df));
    FFT_TRY(rocfft_plan_description_set_data_layout(df, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr,
                                                    nullptr, 2, rs, rdist, 2, cs, cdist));
    FFT_TRY(rocfft_plan_description_create(&db));
    FFT_TRY(rocfft_plan_description_set_data_layout(db, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr,
                                                    nullptr, 2, cs, cdist, 2, rs, rdist));
    FFT_TRY(rocfft_plan_create(&plan2n_f[nb], rocfft_placement_notinplace, rocfft_transform_type_real_forward, pr, 2, len2,
                               (size_t)nb * K[0], df));
    FFT_TRY(rocfft_plan_create(&plan2n_b[nb], rocfft_placement_notinplace, rocfft_transform_type_real_inverse, pr, 2, len2,
                               (size_t)nb * K[0], db));
    rocfft_plan_description_destroy(df);
    rocfft_plan_description_destroy(db);
    size_t w = 0, wmax = 0;
    FFT_TRY(rocfft_plan_get_work_buffer_size(plan2n_f[nb], &w)); if (w > wmax) wmax = w;
    FFT_TRY(rocfft_plan_get_work_buffer_size(plan2n_b[nb], &w)); if (w > wmax) wmax = w;
    if (wmax + 16 > fft_work.bytes) {
      HIP_TRY(hipStreamSynchronize(stream));
      fft_work.need(wmax + 16);
    }
    if (fft_work.bytes > 16) FFT_TRY(rocfft_execution_info_set_work_buffer(info_f, fft_work.p, fft_work.bytes - 16));
  }
  // nb meshes at mesh_p + b * nreal -> phi_b in place; energies of all channels summed into energies[slot]
  void convolve_batch(T* mesh_p, T* spec_p, const DftTabs<T>& tabs, int nb, size_t nreal, size_t nspec, int slot) {
    ensure_batched_plans(nb);
    run_plan("rocfft_r2c_yz", plan2n_f[nb], mesh_p, spec_p);
    { TIMED("fftx_kspace");
      launch_fftx_conv_batch<T>(stream, K, fx_tw.as<T>(), spec_p, tabs, nb, (long)nspec, Ed_cur(), slot, fx_khp); }
    run_plan("rocfft_c2r_yz", plan2n_b[nb], spec_p, mesh_p);
    (void)nreal;
  }

  void run_plan(const char* label, rocfft_plan plan, void* in, void* out) {
    TIMED(label);
    FFT_TRY(rocfft_execution_info_set_stream(info_f, stream));
    void* i[1] = {in};
    void* o[1] = {out};
    FFT_TRY(rocfft_execute(plan, i, o, info_f));
  }
  // real mesh (owned planes) <-> half spectrum in `spec`: full 3-D transform (1 rank) or its y-z part (slabs)
  void fft_forward(T* mesh_p, T* spec_p) { run_plan("rocfft_r2c", plan_f, mesh_p, spec_p); }
  void fft_inverse(T* spec_p, T* mesh_p) { run_plan("rocfft_c2r", plan_b, spec_p, mesh_p); }
  void fft_x(T* buf, int inverse) { run_plan(inverse ? "rocfft_x_inv" : "rocfft_x_fwd", inverse ? plan_xb : plan_xf, buf, buf); }

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
    g.whole_mesh();
    if (snranks > 1) { g.xoff = X0; g.nloc0 = nloc0(); g.wrap0 = 1 << 30; }
    return g;
  }

  void ensure_gtab(const double* box, const double* inv, double vol, int which) {
    const int slot = tab_slot(which);
    TabKey& key = tabkey[slot];
    DevBuf& buf = gtabs[slot];
    const size_t nspec_t = (size_t)K[0] * nyown() * (use_pfa ? pfa.Khp : K[2] / 2 + 1);
    buf.need(nspec_t * sizeof(T));
    gtab_cur = buf.template as<T>();
    bool same = key.kappa == kappa && key.K[0] == K[0] && key.K[1] == K[1] && key.K[2] == K[2] &&
                key.Y0 == (snranks > 1 ? Y0 : 0) && key.ref == ref_korder && key.pfa == use_pfa;
    for (int k = 0; k < 9 && same; ++k) same = key.box[k] == box[k];
    if (same) return;
    HIP_TRY(hipMemcpyAsync(binv_d.p, inv, 9 * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));   // `inv` is a caller stack array
    {
      TIMED("gtab");
      launch_gtab<T>(stream, K, snranks > 1 ? Y0 : 0, nyown(), binv_d.as<double>(), std::fabs(vol), kappa, which, gtab_cur, ref_korder,
                     use_pfa ? pfa_fmap.as<int>() : nullptr, use_pfa ? pfa.Khp : 0);
    }
    std::memcpy(key.box, box, sizeof(key.box));
    key.kappa = kappa; key.K[0] = K[0]; key.K[1] = K[1]; key.K[2] = K[2]; key.Y0 = snranks > 1 ? Y0 : 0; key.ref = ref_korder;
    key.pfa = use_pfa;
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

  void ensure_bins(int na) { ensure_bins(na, bins, bin_cells, bin_sorted); }
  void ensure_bins(int na, BinScratch& b, DevBuf& cells, DevBuf& sorted) {
    const int dims[3] = {nloc0(), K[1], K[2]};
    const BrickGrid bg = make_bricks(dims);
    cells.need(sizeof(int) * 3 * (size_t)(bg.ncell + 1));
    sorted.need(sizeof(int) * 8 * (size_t)na);
    bin_scan.need(spread_scan_bytes(bg.ncell));
    if (b.cell_start != cells.as<int>() || b.cursor != cells.as<int>() + (bg.ncell + 1))
      b.counters_zero = false;                     // fresh or re-laid-out storage
    b.cell_start = cells.as<int>();
    b.cursor = cells.as<int>() + (bg.ncell + 1);
    b.fillcur = cells.as<int>() + 2 * (size_t)(bg.ncell + 1);
    b.sorted = sorted.as<int>();
    b.scan_tmp = bin_scan.p;
    b.scan_bytes = bin_scan.bytes;
  }

  // ---- one evaluation, step by step --------------------------------------------------------------------------
  // pme() runs the steps back to back.  On a slab-decomposed handle the same steps work on the rank's home atoms and the
  // collectives of the caller's communicator (admp_set_comm) sit between them: same kernels, same SCF forms.
  struct Eval {
    const T* pos = nullptr; const T* Ql = nullptr; const T* pol = nullptr; const T* thole = nullptr;
    T* U = nullptr;
    Box<T> bx; RecipGeom<T> g; ScaleTab<T> tab;
    int n_home = 0; const int* home = nullptr;   // this rank's atoms (nullptr = all, in order)
    const int4* bases = nullptr;                 // stencil base indices per atom (written by prepare_sites)
    bool active = false;
  } ev;
  const int* pair_rows() const { return snranks > 1 ? sl.rows : nbr.order; }      // row order of the pair kernels

  int stage_begin(const void* pos_, const double* box, const void* Ql_, const void* pol_, const void* thole_, int ns,
                  const double* mS, const double* pS, void* U_) {
    ARG_CHECK(have_top && have_ewald && have_pairs, "topology, ewald parameters and pairs must be set first");
    ARG_CHECK(pos_ && box && Ql_, "null argument");
    if (lpol) ARG_CHECK(pol_ && thole_ && U_ && pS, "polarizable handle needs pol, tholes, pScales and U_inout");
    const int na = top.na;
    double inv[9], vol;
    ++eval_seq;
    ff_done = nullptr;
    ensure_mesh();
    ev.bx = make_box(box, inv, &vol);
    ev.g = make_geom(inv);
    ev.tab = make_tab(ns, mS, pS);
    ensure_gtab(box, inv, vol, 1);
    ev.pos = reinterpret_cast<const T*>(pos_);
    ev.Ql = reinterpret_cast<const T*>(Ql_);
    ev.pol = lpol ? reinterpret_cast<const T*>(pol_) : nullptr;
    ev.thole = lpol ? reinterpret_cast<const T*>(thole_) : nullptr;
    ev.U = lpol ? reinterpret_cast<T*>(U_) : nullptr;
    // admp_set_dipole_source (one shot): the initial dipoles are read from there, ev.U starts as their copy (k_prepare_sites)
    const T* U_first = lpol && U_src_now ? reinterpret_cast<const T*>(U_src_now) : ev.U;
    U_src_now = nullptr;
    sites.need(sizeof(Site<T>) * (size_t)na);
    pot.need(9 * (size_t)na * sizeof(T));
    energies_d.need(2 * E_WORDS * sizeof(double));
    if (energies_d.p != energies_seen) { energies_seen = energies_d.p; other_clean = false; }
    if (lpol) {
      fld_pair.need(3 * (size_t)na * sizeof(T));
      fld_recip.need(3 * (size_t)na * sizeof(T));
      field.need(3 * (size_t)na * sizeof(T));
    }
    ensure_bins(na);
    bases_d.need(sizeof(int4) * (size_t)na);
    if (other_clean) {
      ehalf ^= 1;                      // zeroed by the previous evaluation's first kernel
    } else {
      ehalf = 0;
      HIP_TRY(hipMemsetAsync(energies_d.p, 0, E_WORDS * sizeof(double), stream));   // energies and the max|field| word
    }
    other_clean = false;
    fmax_clean = true;
    for (bool& c : slot_clean) c = true;
    // Slab rank, steady state: only the rows this rank reads are prepared -- its home atoms and its imports (round 3 prepared
    // the rows of ALL atoms on every rank: 0.056 ms at 1M atoms whatever the rank count).  The ownership rule needs the stencil
    // base planes of all atoms first: a 28-byte-per-atom pass (k_atom_bases).  The first evaluation of a handle / topology /
    // buffer, and evaluations that recompile the table's site classes, take the all-atom pass below, so that every row of
    // `sites` has held valid data at least once (the class compilation reads all of them).  ADMP_SLAB_SUBSET=0: always all.
    static const bool subset_on = [] { const char* e = getenv("ADMP_SLAB_SUBSET"); return !(e && atoi(e) == 0); }();
    if (snranks > 1 && subset_on && slab_sites_na == na && slab_sites_ptr == sites.p && !cls_pending && have_pairs) {
      { TIMED("atom_bases"); launch_atom_bases<T>(stream, na, ev.pos, ev.g, bases_d.as<int4>()); }
      ev.bases = bases_d.as<int4>();
      cls_sites_na = na;
      ++cls_quiet;
      const bool tracked = lpol && sl.prev_na == na;
      decompose();
      {
        TIMED("prepare_sites");
        const int* lists[2] = {sl.home, sl.imp.as<int>()};
        const int counts[2] = {sl.n_home, sl.n_imp};
        for (int k = 0; k < 2; ++k)
          launch_prepare_sites<T>(stream, top, ev.pos, ev.Ql, U_first, ev.pol, ev.thole, ev.bx, sites.as<Site<T>>(),
                                  k == 0 ? energies_d.as<double>() + (size_t)(ehalf ^ 1) * E_WORDS : nullptr, ev.g, nullptr,
                                  nullptr, nullptr, nbr.cls, cls_flags_dev(), rq_p(), U_first != ev.U ? ev.U : nullptr, lists[k],
                                  counts[k]);
      }
      other_clean = true;
      if (tracked) exchange_migrants();
      if (lpol) exchange_U(0);
      ev.active = true;
      return ev.n_home;
    }
    {
      TIMED("prepare_sites");
      // list of the polarizable sites for the incremental SCF: rebuilt by this kernel unless the caller vouches that the
      // set is the one of the previous call (ADMP_OPT_KEEP_POL_SITES; the Python wrapper sets it while `pol` is unchanged).
      // A slab rank takes its list (the polarizable HOME atoms of this evaluation) from the decomposition instead.
      const bool have_list = act_n >= 0 && act_top_na == na && act_d.p;
      const bool want_act = lpol && snranks == 1 && !(keep_pol_sites && have_list);
      act_fresh = want_act;
      if (want_act) { act_d.need(sizeof(int) * (size_t)na); act_n = -1; act_top_na = na; ++act_gen; }
      if (cls_pending && have_pairs && cls_sites_na == na && !nbr_src) {   // `sites` still holds the last evaluation's
        unprune();                                                        // (classes are compiled into the table as built)
        if (!nbr.cls) HIP_TRY(hipMalloc(&nbr.cls, sizeof(int) * (size_t)na));
        launch_site_classes<T>(stream, na, sites.as<Site<T>>(), nbr.cls);
        apply_classes();
        order_rows();
      }
      cls_pending = false;
      cls_sites_na = na;
      ++cls_quiet;
      launch_prepare_sites<T>(stream, top, ev.pos, ev.Ql, U_first, ev.pol, ev.thole, ev.bx, sites.as<Site<T>>(),
                              energies_d.as<double>() + (size_t)(ehalf ^ 1) * E_WORDS, ev.g, bases_d.as<int4>(),
                              want_act ? act_d.as<int>() : nullptr, want_act ? nact_dev() : nullptr, nbr.cls,
                              cls_flags_dev(), rq_p(), U_first != ev.U ? ev.U : nullptr);
      ev.bases = bases_d.as<int4>();
      // First evaluation on a table compiled without classes: look at the flags right away (one extra host read, once) so
      // that this call already walks the parted rows -- a caller who evaluates once gets the reduced forms too.
      if (!nbr.cls && have_pairs && !nbr_src && !cls_first_done) {
        cls_first_done = true;
        int flags = 0;
        HIP_TRY(hipMemcpyAsync(&flags, cls_flags_dev(), sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (flags & CLS_BETTER) {
          unprune();
          HIP_TRY(hipMalloc(&nbr.cls, sizeof(int) * (size_t)na));
          launch_site_classes<T>(stream, na, sites.as<Site<T>>(), nbr.cls);
          apply_classes();
          order_rows();
          HIP_TRY(hipMemsetAsync(cls_flags_dev(), 0, sizeof(int), stream));   // the table now agrees with these sites
        }
      }
    }
    other_clean = true;
    if (snranks > 1) {
      slab_sites_na = na; slab_sites_ptr = sites.p;        // every row of `sites` is valid from here on
      const bool tracked = lpol && sl.prev_na == na;       // (decompose() then lists the atoms that changed hands)
      decompose();
      if (tracked) exchange_migrants();
      if (lpol) exchange_U(0);
    } else {
      ev.n_home = na;
      ev.home = nullptr;
    }
    ev.active = true;
    return ev.n_home;
  }

  void need_eval() { ARG_CHECK(ev.active, "internal: no evaluation in progress"); }

  void stage_spread(T* mesh_p) {
    need_eval();
    TIMED("spread");
    int rc = launch_spread<T>(stream, ev.n_home, sites.as<Site<T>>(), lpol, ev.g, bins, mesh_p, ev.home, ev.bases);
    if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)};
  }
  // max|field| lives in the last word of the energies buffer (bit pattern of a non-negative double), so that one
  // device->host copy can fetch it together with the energies
  unsigned long long* fmax_word() { return reinterpret_cast<unsigned long long*>(Ed_cur() + E_FMAX); }
  void launch_field_finish_only() {      // total dE/dU of every home atom and its maximum (over all ranks) into fmax_word()
    need_eval();
    if (!fmax_clean) HIP_TRY(hipMemsetAsync(fmax_word(), 0, sizeof(unsigned long long), stream));
    fmax_clean = false;
    { TIMED("field_finish");
      launch_field_finish<T>(stream, ev.n_home, sites.as<Site<T>>(), ev.pol, ev.U, fld_pair.as<T>(), fld_recip.as<T>(),
                             (T)kappa, field.as<T>(), fmax_word(), ev.home); }
    reduce_check_word(fmax_word());
  }
  // The charge-only pair forms read the compact rq rows and the class marks of the table: asking for them without those is
  // a bug of the caller, not something to paper over (round 2: a slab handle passed the flag word but no rq rows, and the
  // kernels' prefetch `rq[0]` read address 0 -- DESIGN.md section 7a).
  void check_mono_inputs(bool use_mono) {
    if (use_mono && !rq_d.p) throw Err{ADMP_E_STATE, "charge-only pair forms requested without the compact site rows"};
  }
  // ---- a second stream for the real-space kernels of small systems (round 3) ---------------------------------------------
  // At a few thousand atoms a kernel fills a fraction of the chip and a step is a chain of ~20 short dispatches; the pair
  // kernels (sites -> gradient / potential / field rows) do not depend on the mesh chain (spread -> convolution) that runs
  // next to them, so they go to a side stream between a fork (side waits for what main has enqueued so far) and a join (main
  // waits for the side stream before the first kernel that reads or adds to the pair kernel's rows: the gathers).
  // ADMP_OVERLAP_MAX: atom count up to which this is done (0 = never; default 200 000: 98k atoms 0.316 -> 0.306 ms per step,
  // 1M atoms 1.72 -> 1.99 -- there every kernel fills the chip and the two streams only get in each other's way).
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_busy = false;
  // ... and from which it is done (ADMP_OVERLAP_MIN, default 4096): the fork + join cost ~12 us of the main chain, what they hide
  // is the pair kernels -- 15 us at 3072 atoms (round 4, after the chain lost its spread and closing kernels: 0.1893 ms per
  // step on one stream against 0.1929 with the side stream, and a third of the run-to-run spread), 30 us at 6144 atoms
  // (0.241 against 0.261 ms), more above.
  bool overlap_ok() const {
    static const int mx = [] { const char* e = getenv("ADMP_OVERLAP_MAX"); return e ? atoi(e) : 200000; }();
    static const int mn = [] { const char* e = getenv("ADMP_OVERLAP_MIN"); return e ? atoi(e) : 4096; }();
    return side_stream_on && snranks == 1 && top.na <= mx && top.na >= mn;
  }
  // An exception between on_side() and join_side() (a failed launch, a refused argument further down the call) would leave
  // kernels of this call running on the side stream with nobody waiting for them -- the next call would race them on the
  // gradient rows (round-3 verdict, weak #10).  guarded() calls this before it reports the error.
  void after_error() override {
    if (side) (void)hipStreamSynchronize(side);
    side_busy = false;
    ev.active = false;
  }
  // Order of submission around a fork: dispatch-bound systems (<= 16384 atoms) submit the side work AFTER the first kernel of
  // the main chain that follows (see recip_pass); larger ones first -- there the pair kernel is long and wants the early start.
  // ADMP_SIDE_FIRST = 0 / 1 forces one order (A/B).
  bool side_first() const {
    static const int mode = [] { const char* e = getenv("ADMP_SIDE_FIRST"); return e ? atoi(e) : -1; }();
    return mode >= 0 ? mode != 0 : top.na > 16384;
  }
  template <class F>
  void on_side(F&& f) {
    if (!overlap_ok()) { f(); return; }
    if (!side) {
      HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(ev_fork, stream));
    HIP_TRY(hipStreamWaitEvent(side, ev_fork, 0));
    hipStream_t main_stream = stream;
    stream = side;
    try { f(); } catch (...) { stream = main_stream; throw; }      // (guarded() -> after_error() waits for the side stream)
    stream = main_stream;
    HIP_TRY(hipEventRecord(ev_join, side));
    side_busy = true;
  }
  void join_side() {
    if (!side_busy) return;
    HIP_TRY(hipStreamWaitEvent(stream, ev_join, 0));
    side_busy = false;
  }

  void stage_pair_full(T* grad_p, T* fld_out = nullptr) {
    need_eval();
    if (snranks > 1)   // the closing kernel ADDS frame-adjoint contributions to atoms of other ranks: those rows start at zero
      launch_rows_scatter<T>(stream, 2, sl.n_imp, 3, sl.imp.as<int>(), nullptr, grad_p);
    if (!slot_clean[E_REAL]) {
      HIP_TRY(hipMemsetAsync(Ed_cur() + E_REAL, 0, sizeof(double), stream));
      HIP_TRY(hipMemsetAsync(Ed_cur() + E_RPARTS, 0, E_PARTS * sizeof(double), stream));
    }
    slot_clean[E_REAL] = false;
    check_mono_inputs(mono_ok);
    TIMED("pair_full");
    launch_pair_full<T>(stream, ev.n_home, nbr, sites.as<Site<T>>(), ev.bx, ev.tab, (T)kappa, lpol, grad_p, pot.as<T>(),
                        Ed_cur(), pair_rows(), fld_out, mono_ok ? 1 : 0, cls_flags_dev(), rq_d.as<RQ4<T>>(), ev.thole);
  }
  // with_field_finish: the gather also forms the total dE/dU and its maximum (launch_field_finish's work, fused)
  // Small single-rank systems with frame groups: the closing kernel's work rides in the gather's epilogue (recip_kernels.hip,
  // k_gather_staged<.., FIN>); the launch_finish_only that follows is then a no-op.  ADMP_FUSE_FIN_MAX = 0 turns it off (A/B).
  bool fin_fused = false;
  bool fuse_fin_ok() const {
    const char* e = getenv("ADMP_FUSE_FIN_MAX");       // (read per call: the parity tests run both forms in one process)
    const int fin_max = e ? atoi(e) : 8192;
    return snranks == 1 && !ev.home && top.gath_blk && top.na <= fin_max;
  }
  FinishArgs<T> finish_args(bool want_grad, T* dQl) {
    FinishArgs<T> f;
    f.pol = ev.pol; f.kappa = (T)kappa; f.dQlocal = dQl; f.energies = Ed_cur(); f.want_grad = want_grad ? 1 : 0;
    return f;
  }
  // ff_given: a field epilogue on a word of the caller's choice (field_epilogue(word); the chained SCF's last residual)
  void stage_gather(const T* mesh_p, T* grad_p, T* fld_out = nullptr, bool with_field_finish = false,
                    double* e_recip = nullptr, const FinishArgs<T>* fin = nullptr, const FieldFin<T>* ff_given = nullptr) {
    need_eval();
    FieldFin<T> ff;
    if (ff_given) ff = *ff_given;
    else if (with_field_finish) {
      if (!fmax_clean) HIP_TRY(hipMemsetAsync(fmax_word(), 0, sizeof(unsigned long long), stream));
      fmax_clean = false;
      ff.pol = ev.pol; ff.Ucart = ev.U; ff.fld_pair = fld_pair.as<T>(); ff.kappa = (T)kappa;
      ff.field = field.as<T>(); ff.fmax_bits = fmax_word();
    }
    join_side();                       // the gather adds to the rows the pair kernel has set
    TIMED("gather");
    if (fin && fuse_fin_ok()) {
      launch_gather<T>(stream, ev.n_home, sites.as<Site<T>>(), lpol, ev.g, mesh_p, pot.as<T>(), grad_p, ev.home, fld_out, ff,
                       e_recip, &top, &ev.bx, *fin);
      fin_fused = true;
      return;
    }
    launch_gather<T>(stream, ev.n_home, sites.as<Site<T>>(), lpol, ev.g, mesh_p, pot.as<T>(), grad_p, ev.home, fld_out, ff,
                     e_recip);
  }
  // closes the evaluation: E_out = (real, recip[slot], self, penalty) of THIS rank's share
  // with_field_finish (single rank, pull kernel): the SCF residual and its maximum are formed by this kernel too
  void launch_finish_only(T* grad_p, T* dQl, bool with_field_finish = false) {
    need_eval();
    if (fin_fused) { fin_fused = false; return; }      // (the gather just before did this work)
    FieldFin<T> ff;
    if (with_field_finish) {
      if (!fmax_clean) HIP_TRY(hipMemsetAsync(fmax_word(), 0, sizeof(unsigned long long), stream));
      fmax_clean = false;
      ff.pol = ev.pol; ff.Ucart = ev.U; ff.fld_pair = fld_pair.as<T>(); ff.fld_recip = fld_recip.as<T>();
      ff.kappa = (T)kappa; ff.field = field.as<T>(); ff.fmax_bits = fmax_word();
    }
    { TIMED("finish");
      launch_finish<T>(stream, top, ev.pos, ev.bx, sites.as<Site<T>>(), ev.pol, ev.U, lpol, (T)kappa, pot.as<T>(), grad_p,
                       dQl, Ed_cur(), ev.home, ev.n_home, ff, snranks > 1 ? sl.bits.as<int>() : nullptr); }
    if (snranks > 1 && grad_p) exchange_grad(grad_p);
  }
  // ---- incremental SCF ------------------------------------------------------------------------------------------
  // The field dE/dU is LINEAR in the induced dipoles and only sites with pol > 0 ever change theirs.  So after the first
  // (full) field evaluation of a call, every further SCF cycle evaluates only the field of the Jacobi step's dipole CHANGE
  // dU: real space over polarizable-polarizable pairs (k_pair_field_ind), reciprocal space by spreading the dU of the
  // polarizable sites alone, and adds it to the stored field; phi is kept current by adding the increment's mesh.  For
  // water that is 1/9 of the pairs and 1/3 of the spread / gather work per cycle; the arithmetic is the reference's
  // (admp/pme.py:130-138) regrouped, same dipoles and cycle count to round-off.  On a slab rank the rows are the polarizable
  // HOME atoms and the dU of the imported atoms arrive by one all-to-all-v per Jacobi step (exchange_U).
  enum { E_PARTS_SUM = -1 };   // read_energies: the reciprocal energy is the sum of the E_PARTS partial words
  int* nact_dev() { return reinterpret_cast<int*>(Ed_cur() + E_NACT); }
  DevBuf rq_d;                                      // compact (position, charge) rows: what the pair kernels read of a
  RQ4<T>* rq_p() {                                  // charge-only partner
    rq_d.need(sizeof(RQ4<T>) * (size_t)top.na);
    return rq_d.as<RQ4<T>>();
  }
  int* cls_flags_dev() { return nact_dev() + 1; }   // the other half of that word: CLS_* of this evaluation (k_prepare_sites)
  int cls_sites_na = -1;                            // `sites` holds an evaluation of this many atoms
  bool cls_first_done = false;                      // the flags of the first evaluation have been looked at
  // the polarizable rows of this evaluation: all polarizable atoms (one rank), or the rank's polarizable home atoms
  const int* act_list() const { return snranks > 1 ? sl.act : act_d.as<int>(); }
  // device-side count for the kernels of the first cycle when the list is fresh (nullptr: the host knows it)
  const int* nact_arg() { return snranks == 1 && act_fresh ? nact_dev() : nullptr; }
  int nact_rows() const { return snranks > 1 ? sl.n_act : (act_fresh ? top.na : act_n); }      // grid bound of those kernels
  int nact_known() const { return snranks > 1 ? sl.n_act : act_n; }
  void nact_seen() {                                                 // after a read_energies of this evaluation
    if (snranks > 1 || !act_fresh) return;
    int n = 0;
    std::memcpy(&n, &Eh[E_NACT], sizeof(n));
    act_n = n;
    act_fresh = false;
    // The workgroups of k_prepare_sites append their chunks in completion order; in atom order the rows that neighbouring
    // workgroups (and XCDs) of the pair kernels work on are neighbours in space again (measured HBM traffic of the field
    // pass over the list at 1M atoms: 1.42 GB unsorted -- more than the full pair kernel's 0.44 GB).
    act_tmp.need(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int rc = sort_ints(stream, act_d.as<int>(), act_tmp.as<int>(), n, &scan_scratch.p, &scan_bytes);
    if (rc != 0) throw Err{ADMP_E_HIP, std::string("sort_ints: ") + hipGetErrorString((hipError_t)rc)};
  }
  void first_pair_field() {          // real-space dE/dU of the polarizable rows, all partners
    check_mono_inputs(true);         // (the flag word handed over below switches the charge-only form on)
    TIMED("pair_field");
    launch_pair_field<T>(stream, nact_rows(), nbr, sites.as<Site<T>>(), ev.bx, ev.tab, (T)kappa, fld_pair.as<T>(),
                         act_list(), nact_arg(), cls_flags_dev(), rq_d.as<RQ4<T>>(), ev.thole);
  }
  void first_gather_field(const FieldFin<T>& ff) {   // reciprocal dE/dU of the polarizable rows from phi
    join_side();                     // (its epilogue reads the pair field)
    TIMED("gather_field");
    launch_gather_field<T>(stream, nact_rows(), sites.as<Site<T>>(), ev.g, mesh.as<T>(), fld_recip.as<T>(), act_list(), 1,
                           nact_arg(), nullptr, ff);
  }
  double scf_check(int* n_act) {          // total field + its maximum over the polarizable sites; one host read
    if (ff_done == fmax_word()) ff_done = nullptr;            // formed by the field gather before this check
    else launch_field_check(next_check_word());
    double dummy[4];
    const double fmax = read_energies(E_SCF_RECIP, dummy, false);
    nact_seen();
    *n_act = nact_known();
    return fmax;
  }
  void scf_jacobi(int n_act, const unsigned long long* gate = nullptr, double gate_min = 0.0) {
    if (n_act > 0) {
      isites.need(sizeof(Site<T>) * (size_t)n_act);
      TIMED("jacobi_update");
      launch_jacobi_delta<T>(stream, n_act, act_list(), ev.pol, field.as<T>(), ev.U, sites.as<Site<T>>(),
                             isites.as<Site<T>>(), gate, gate_min);
    }
    if (snranks > 1) exchange_U(1);       // every rank takes part, whatever its own count
  }
  // Small systems are dispatch-bound: the SCF residual rides in the epilogue of the field gather that precedes a check
  // (field_epilogue(word) describes it; the check's own kernel is then skipped).  word: a zero word of the energy block.
  const unsigned long long* ff_done = nullptr;
  bool fuse_ok() const {
    static const int fuse_max = [] { const char* e = getenv("ADMP_FUSE_FF_MAX"); return e ? atoi(e) : 16384; }();
    return top.na <= fuse_max && snranks == 1 && !ev.home;
  }
  FieldFin<T> field_epilogue(unsigned long long* word) {
    FieldFin<T> ff;
    if (!word || !fuse_ok()) return ff;
    ff.pol = ev.pol; ff.Ucart = ev.U; ff.fld_pair = fld_pair.as<T>(); ff.kappa = (T)kappa;
    ff.field = field.as<T>(); ff.fmax_bits = word; ff.sites = sites.as<Site<T>>();
    ff_done = word;
    return ff;
  }
  // the word the next plain check writes: fmax_word(), cleared if this evaluation has used it
  unsigned long long* next_check_word() {
    if (!fmax_clean) HIP_TRY(hipMemsetAsync(fmax_word(), 0, sizeof(unsigned long long), stream));
    fmax_clean = false;
    return fmax_word();
  }
  // total field of the polarizable rows and its maximum (over all ranks) into `word` (a zero word of this evaluation's
  // energy block); no host read
  void launch_field_check(unsigned long long* word) {
    if (ff_done == word) { ff_done = nullptr; return; }     // the field gather before it has done this (one rank only)
    ff_done = nullptr;
    if (nact_rows() > 0) {
      TIMED("field_finish");
      launch_field_finish<T>(stream, nact_rows(), sites.as<Site<T>>(), ev.pol, ev.U, fld_pair.as<T>(), fld_recip.as<T>(),
                             (T)kappa, field.as<T>(), word, act_list(), nact_arg());
    }
    reduce_check_word(word);
  }
  size_t nreal_local() const { return (size_t)nloc0() * K[1] * K[2]; }
  // check_word: the zero word the check after this increment writes (its residual then rides in the field gather)
  // extra_side: more work for the side stream of this increment (the closing pair kernel of a chained call, whose dipoles are final)
  // field_from_total_phi: do not gather the increment's field -- the caller's next kernel is a full gather of the accumulated
  // phi with a field epilogue, which yields the same reciprocal field (and the residual) for every atom
  void scf_increment(int n_act, unsigned long long* check_word = nullptr,   // fld_pair / fld_recip / phi <- their values for the dipoles after scf_jacobi
                     const std::function<void()>& extra_side = nullptr, bool field_from_total_phi = false) {
    if (n_act <= 0 && snranks == 1) {
      if (extra_side) on_side(extra_side);
      return;
    }
    const bool first = side_first();
    FieldRider<T> fr;      // (filled below, once the sub-table is current: the field kernel then rides in the x pass)
    bool ride = false;
    auto side_work = [&] {
      if (!ride)
        on_side([&] {
          TIMED("pair_field_ind");
          launch_pair_field_ind<T>(stream, n_act, ind, sites.as<Site<T>>(), ev.bx, ev.tab, (T)kappa, fld_pair.as<T>(),
                                   act_list());
        });
      if (extra_side) on_side(extra_side);
    };
    if (ind_nbr_gen != nbr_gen || ind_act_gen != act_gen) {      // neighbour table or polarizable set changed
      TIMED("ind_table");
      int rc = build_ind_table<T>(stream, top.na, nbr, sites.as<Site<T>>(), ind);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("build_ind_table: ") + hipGetErrorString((hipError_t)rc)};
      ind_nbr_gen = nbr_gen; ind_act_gen = act_gen;
    }
    ride = rider_ok() && field_rider_ind<T>(fr, n_act, ind, sites.as<Site<T>>(), ev.bx, ev.tab, (T)kappa, fld_pair.as<T>(),
                                            act_list());
    if (first) side_work();
    const size_t nreal = nreal_local();
    mesh2.need(nreal * sizeof(T));
    const bool fused = spread_fused(n_act);
    if (fused) { if (!first) side_work(); }
    else { TIMED("spread_ind");
      // the compact rows keep their positions and their order through the SCF cycles of one evaluation: the brick lists of
      // its first increment serve the later ones (two binning passes and a scan less per cycle)
      ensure_bins(std::max(n_act, 1), bins_ind, bin_cells_ind, bin_sorted_ind);
      const bool reuse = ind_bins_eval == eval_seq && ind_bins_n == n_act && ind_bins_gen == act_gen &&
                         ind_bins_at == bins_ind.cell_start;
      int rc = launch_spread<T>(stream, n_act, isites.as<Site<T>>(), 1, ev.g, bins_ind, mesh2.as<T>(), nullptr, nullptr, 1,
                                reuse ? 1 : 0, 0);
      ind_bins_eval = eval_seq; ind_bins_n = n_act; ind_bins_gen = act_gen; ind_bins_at = bins_ind.cell_start;
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)}; }
    if (!first && !fused) side_work();
    const PlaneSpread<T> sp = plane_spread(n_act, isites.as<Site<T>>(), 1, nullptr);
    const bool added = convolve(mesh2.as<T>(), spec.as<T>(), gtab_cur, E_SCRATCH, mesh.as<T>(), nullptr, fused ? &sp : nullptr,
                                ride ? &fr : nullptr);
    join_side();
    if (!field_from_total_phi) {
      TIMED("gather_field_ind");
      launch_gather_field<T>(stream, n_act, isites.as<Site<T>>(), ev.g, mesh2.as<T>(), fld_recip.as<T>(), nullptr, 1, nullptr,
                             act_list(), check_word ? field_epilogue(check_word) : FieldFin<T>()); }
    if (!added) { TIMED("mesh_add"); launch_mesh_add<T>(stream, (long)nreal, mesh.as<T>(), mesh2.as<T>()); }
  }

  // one device->host copy + sync: energies (and the max|field| word, returned).  On a slab rank the four parts are first
  // summed over the ranks (want_sum; the SCF checks only need the residual word, which is already the global maximum).
  double read_energies(int recip_slot, double* E, bool want_sum = true) {
    if (!Eh) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&Eh), E_WORDS * sizeof(double), hipHostMallocDefault));
    const bool summed = snranks > 1 && want_sum;
    if (summed) {
      TIMED("comm_energies");
      launch_energy_pack(stream, Ed_cur(), recip_slot, Ed_cur() + E_RED);
      c_all_reduce(Ed_cur() + E_RED, 4, ADMP_T_F64, ADMP_OP_SUM, ADMP_TAG_ENERGIES);
    }
    HIP_TRY(hipMemcpyAsync(Eh, Ed_cur(), E_WORDS * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    {
      int w[2];
      std::memcpy(w, &Eh[E_NACT], sizeof(w));
      cls_seen(w[1]);
    }
    if (summed) {
      for (int k = 0; k < 4; ++k) E[k] = Eh[E_RED + k];
      return Eh[E_FMAX];
    }
    E[0] = Eh[E_REAL]; E[1] = recip_slot >= 0 ? Eh[recip_slot] : 0.0; E[2] = Eh[E_SELF]; E[3] = Eh[E_PEN];
    for (int k = 0; k < E_PARTS; ++k) E[0] += Eh[E_RPARTS + k];      // k_pair_full's partial words
    if (recip_slot == E_PARTS_SUM) {         // atom-side reciprocal energy: the partial words of k_gather<.., true>
      double e = 0.0;
      for (int k = 0; k < E_PARTS; ++k) e += Eh[E_SLOTS + k];
      E[1] = e;
    }
    return Eh[E_FMAX];   // same bits as the device word
  }
  void stage_finish(T* grad_p, T* dQl, int recip_slot, double* E) {
    launch_finish_only(grad_p, dQl);
    read_energies(recip_slot, E);
    ev.active = false;
  }

  // full reciprocal pass on one rank: spread -> r2c -> G multiply (+energy) -> c2r ; mesh then holds phi
  // side_work: real-space kernels that run NEXT to the convolution (on_side).  They are submitted after the spread: the host
  // needs ~10 us for the event record / stream wait / launch of a fork, and submitted first (round 3) that time stood between
  // the site pass and the spread on the main stream as well (kernel trace of the S1 loop: 10-12 us idle there, every
  // evaluation); behind the spread it is hidden by a running kernel and the side kernels still have the three transform
  // passes to hide behind (side_first()).
  template <class F>
  void recip_pass(int slot, F&& side_work, const FieldRider<T>* rider = nullptr) {
    need_eval();
    const bool fused = spread_fused(ev.n_home);      // the forward transform spreads (dft_kernels.hip, zy_plane_spread)
    const bool first = side_first() || fused;
    if (first) side_work();
    if (!fused) stage_spread(mesh.as<T>());
    if (!first) side_work();
    if (!slot_clean[slot]) HIP_TRY(hipMemsetAsync(Ed_cur() + slot, 0, sizeof(double), stream));
    slot_clean[slot] = false;
    const PlaneSpread<T> sp = plane_spread(ev.n_home, sites.as<Site<T>>(), lpol, ev.bases);
    convolve(mesh.as<T>(), spec.as<T>(), gtab_cur, slot, nullptr, nullptr, fused ? &sp : nullptr, rider);
  }
  void recip_pass(int slot) { recip_pass(slot, [] {}); }
  // first field evaluation of a call: the mesh chain and the real-space field of all partners -- on the side stream, or (small
  // systems, one stream, direct-DFT mesh) with the field kernel's workgroups inside the x pass
  void recip_pass_first_field(int slot) {
    FieldRider<T> fr;
    if (rider_ok()) {
      check_mono_inputs(true);
      if (field_rider_full<T>(fr, nact_rows(), nbr, sites.as<Site<T>>(), ev.bx, ev.tab, (T)kappa, fld_pair.as<T>(), act_list(),
                              nact_arg(), cls_flags_dev(), rq_d.as<RQ4<T>>(), ev.thole)) {
        recip_pass(slot, [] {}, &fr);
        return;
      }
    }
    recip_pass(slot, [&] { on_side([&] { first_pair_field(); }); });
  }

  void pme(const void* pos_, const double* box, const void* Ql_, const void* pol_, const void* thole_, int ns,
           const double* mS, const double* pS, void* U_, int max_cycle, double thresh, double* E, void* dpos_,
           void* dQl_, int* ncyc, int* conv, int on_device) override {
    // (the one-shot dipole source is taken off the handle before anything can fail: a call that throws must not leave a
    // pointer behind for the next one)
    U_src_now = on_device ? U_src : nullptr;
    U_src = nullptr;
    ARG_CHECK(snranks == 1 || on_device, "a slab-decomposed handle takes device pointers");
    ARG_CHECK(pos_ && box && Ql_ && E, "null argument");
    const int na = top.na;
    HIP_TRY(hipSetDevice(device));
    const T* pos = stage_in(s_pos, pos_, 3 * (size_t)na, on_device);
    const T* Ql = stage_in(s_Q, Ql_, 9 * (size_t)na, on_device);
    const T* pol = lpol ? stage_in(s_pol, pol_, na, on_device) : nullptr;
    const T* thole = lpol ? stage_in(s_thole, thole_, na, on_device) : nullptr;
    T* U = nullptr;
    if (lpol) {
      ARG_CHECK(U_, "polarizable handle needs U_inout");
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
    grad.need(3 * (size_t)na * sizeof(T));   // the gradient buffer is needed internally even for energy-only calls
    T* gbuf = dpos ? dpos : grad.as<T>();
    mono_ok = (dQl == nullptr);

    stage_begin(pos, box, Ql, pol, thole, ns, mS, pS, U);

    // phi_valid: the mesh holds phi = c2r(G S) of the CURRENT dipoles (last SCF field evaluation, no update since):
    // the closing gather can then reuse it instead of spreading and transforming again.
    // phi_accum: that phi was assembled from increments, so no single k-space pass saw its energy -- the closing gather
    // sums it over the atoms instead.
    bool phi_valid = false, done = false, finished = false, phi_accum = false;
    int cyc = 0, flag = 1;
    if (lpol) {
      ARG_CHECK(max_cycle >= 1, "max_cycle must be >= 1");
      int i = 0, n_act = 0;
      bool have_base = false;    // fld_pair / fld_recip / phi belong to the dipoles before the last Jacobi step
      // which form of the first cycle: ADMP_SPECULATE=0 / 1 forces the plain / the speculative one (A/B, tests)
      static const int spec_mode = [] { const char* e = getenv("ADMP_SPECULATE"); return e ? atoi(e) : -1; }();
      // a failed speculation wastes the full pair kernel, the gather and the closing kernel; a successful one saves the field
      // kernels and one synchronisation: at 3072 atoms that is 30 against 45 us, at 98k atoms about even, at 1M atoms 0.55
      // against 0.18 ms -- very large systems speculate only on a clear prediction
      const double spec_infl = top.na <= 200000 ? 1.0 : 1.5;    // weight of the observed growth (0 on a static geometry)
      const bool have_pred = scf_last >= 0.0 && scf_nobs[scf_state] >= 2;
      const double g_hi = std::max(scf_growth[scf_state][0], scf_growth[scf_state][1]);
      const double g_lo = std::min(scf_growth[scf_state][0], scf_growth[scf_state][1]);
      const bool speculate = spec_mode >= 0 ? spec_mode != 0
                                            : (have_pred ? scf_last + spec_infl * g_hi < thresh : (scf_last < 0.0 && warm_regime));
      double f_first = -1.0, f_final = -1.0;     // residuals of the first and of the last check of this call
      // Chained form (up to 200k atoms; at 3072 atoms a host synchronisation costs as much as three kernels): when the history says the
      // first check will fail and n Jacobi steps will do, the whole call is enqueued at once -- first field evaluation and its
      // check, then n times (Jacobi step GATED on the previous check's residual on the device: a zero step once a check has
      // passed, after which every later residual repeats the passing one; increment; check), the closing pass -- and read back
      // with one synchronisation.  The host then replays the reference's decisions on the n + 1 residuals: same dipoles,
      // cycle count and flag as the plain loop; a wrong guess costs the kernels that ran for nothing (closing pass when more
      // cycles are needed, increments after the check that passed).
      static const int chain_max = [] { const char* e = getenv("ADMP_SCF_CHAIN_MAX"); return e ? atoi(e) : 200000; }();
      const double pred = have_pred ? scf_last + 0.5 * (g_hi + g_lo) : -1.0;
      // number of Jacobi steps the history predicts: the residual contracts by scf_contract per step
      int nhat = 0;
      if (have_pred && scf_last + g_lo >= 1.1 * thresh && scf_contract > 0.0 && scf_contract < 0.95 && thresh > 0.0) {
        double r = pred;
        while (nhat <= E_CHAIN && r >= thresh) { r *= scf_contract; ++nhat; }
      }
      // (every input of these decisions is the same on every rank of a decomposed handle: the residuals are global maxima)
      const bool chain = spec_mode < 0 && !speculate && top.na <= chain_max && nhat >= 1 && nhat <= E_CHAIN &&
                         nhat + 2 <= max_cycle && (snranks > 1 || (!act_fresh && act_n > 0));
      static const bool scf_trace = getenv("ADMP_SCF_TRACE") != nullptr;     // one line per call: which form ran
      if (scf_trace) fprintf(stderr, "[admp scf] %s (predicted residual %.4g, threshold %.4g, %d steps)\n",
                             chain ? "chained" : (speculate ? "speculative" : "plain"), pred, thresh, chain ? nhat : 0);
      ++scf_stats[chain ? 3 : (speculate ? 1 : 0)];
      if (chain) {
        n_act = nact_known();
        recip_pass_first_field(E_SCF_RECIP);
        auto word = [&](int k) {      // residual of check k: E_FMAX, then the (still zero) chain words of this evaluation
          return k == 0 ? fmax_word() : reinterpret_cast<unsigned long long*>(Ed_cur() + E_FMAX1 + (k - 1));
        };
        next_check_word();
        first_gather_field(field_epilogue(word(0)));
        launch_field_check(word(0));
        const bool early_full = overlap_ok();       // the closing pair kernel next to the last increment's mesh chain
        // the last residual rides in the closing gather (it reads the accumulated phi anyway): the field gather of the last
        // increment is not run (small single-rank systems; ADMP_CHAIN_LAST_FIELD=1 keeps it: A/B, tests)
        static const bool keep_last = [] { const char* e = getenv("ADMP_CHAIN_LAST_FIELD"); return e && atoi(e) != 0; }();
        const bool last_in_gather = fuse_ok() && !keep_last && n_act > 0;
        for (int c = 0; c < nhat; ++c) {
          scf_jacobi(n_act, word(c), thresh);        // a zero step once a check has passed: later residuals repeat it
          const bool last = c == nhat - 1;
          if (early_full && last)      // (the dipoles are final now)
            scf_increment(n_act, word(c + 1), [&] { stage_pair_full(gbuf); }, last_in_gather);
          else
            scf_increment(n_act, word(c + 1), nullptr, last && last_in_gather);
          if (!(last && last_in_gather)) launch_field_check(word(c + 1));
        }
        if (!early_full) stage_pair_full(gbuf);
        const FinishArgs<T> fin_c = finish_args(dpos != nullptr, dQl);
        if (last_in_gather) {
          const FieldFin<T> ffl = field_epilogue(word(nhat));
          stage_gather(mesh.as<T>(), gbuf, fld_recip.as<T>(), false, Ed_cur() + E_SLOTS, &fin_c, &ffl);
          launch_field_check(word(nhat));            // (done by the epilogue: clears the mark)
        } else {
          stage_gather(mesh.as<T>(), gbuf, nullptr, false, Ed_cur() + E_SLOTS, &fin_c);
        }
        launch_finish_only(dpos ? gbuf : nullptr, dQl);
        read_energies(E_PARTS_SUM, E);
        nact_seen();
        f_first = Eh[E_FMAX];
        have_base = true;
        phi_accum = true;
        int hit = -1;
        for (int k = 0; k <= nhat && hit < 0; ++k)
          if ((k == 0 ? Eh[E_FMAX] : Eh[E_FMAX1 + k - 1]) < thresh) hit = k;
        if (hit >= 0) {
          i = hit;
          f_final = hit == 0 ? Eh[E_FMAX] : Eh[E_FMAX1 + hit - 1];
          phi_valid = done = finished = true;
          ev.active = false;
          if (hit < nhat) { ++scf_stats[5]; scf_stats[6] += nhat - hit; }
        } else {
          ++scf_stats[4];   // more cycles are needed: undo the energy sums of the closing pass, go on like the plain loop
          HIP_TRY(hipMemsetAsync(Ed_cur() + E_SELF, 0, 2 * sizeof(double), stream));
          HIP_TRY(hipMemsetAsync(Ed_cur() + E_SLOTS, 0, E_PARTS * sizeof(double), stream));
          f_final = Eh[E_FMAX1 + nhat - 1];
          scf_jacobi(n_act);
          i = nhat + 1;
        }
      }
      if (speculate) {
        // Steady-state MD regime (the previous call converged at its first check): evaluate the FIRST SCF cycle
        // with the full kernels -- they produce dE/dU alongside the gradient -- so that, when the check passes
        // again, the step is already finished (no separate field kernels, no second pass).  Same arithmetic and
        // same (U, flag, i) as the plain loop; a failed check only costs the difference between the kernels.
        // The closing kernel is enqueued speculatively as well, so the step has ONE host synchronisation.
        recip_pass(E_SCF_RECIP, [&] { on_side([&] { stage_pair_full(gbuf, fld_pair.as<T>()); }); });
        // small systems are dispatch-bound: the field finish rides in the gather's epilogue; larger ones keep the two
        // kernels (98k atoms: gather 26 -> 47 us fused against 9 us saved; 1M atoms: 0.40 vs 0.31 + 0.056 ms)
        static const int fuse_max = [] { const char* e = getenv("ADMP_FUSE_FF_MAX"); return e ? atoi(e) : 16384; }();
        const bool fuse_ff = top.na <= fuse_max;
        const bool fuse_g = fuse_ff && snranks == 1;
        const FinishArgs<T> fin_s = finish_args(dpos != nullptr, dQl);
        stage_gather(mesh.as<T>(), gbuf, fld_recip.as<T>(), fuse_g, nullptr, fuse_g ? &fin_s : nullptr);
        const bool pull = !ev.home && top.inv_ptr;          // the atomics-free closing kernel can carry the field finish
        if (!fuse_g && !pull) launch_field_finish_only();
        launch_finish_only(dpos ? gbuf : nullptr, dQl, !fuse_g && pull);
        const double fmax = read_energies(E_SCF_RECIP, E);
        f_first = f_final = fmax;
        nact_seen();
        n_act = nact_known();
        if (fmax < thresh) {
          phi_valid = done = finished = true;
          ev.active = false;
        } else {   // undo the speculative energy sums; gradient / dQ are rewritten by the regular closing pass
          ++scf_stats[2];
          HIP_TRY(hipMemsetAsync(Ed_cur() + E_SELF, 0, 2 * sizeof(double), stream));
          scf_jacobi(n_act);
          i = 1;
          have_base = true;      // the full kernels left the field of the old dipoles behind: continue by increments
        }
      }
      for (; !done && i < max_cycle; ++i) {     // admp/pme.py:132-138
        if (!have_base) {        // first field evaluation of the call: everything, at the polarizable sites
          recip_pass_first_field(E_SCF_RECIP);
          unsigned long long* w = fuse_ok() ? next_check_word() : nullptr;
          first_gather_field(field_epilogue(w));
          have_base = true;
        } else {
          scf_increment(n_act, fuse_ok() ? next_check_word() : nullptr);
          phi_accum = true;
        }
        const double fmax = scf_check(&n_act);
        if (f_first < 0.0) f_first = fmax;
        f_final = fmax;
        if (fmax < thresh) { phi_valid = true; break; }
        scf_jacobi(n_act);
      }
      if (i == max_cycle) i = max_cycle - 1;   // python's loop variable after exhaustion
      cyc = i;
      scf_stats[7] += cyc;
      flag = (i != max_cycle - 1);             // admp/pme.py:139-143
      warm_regime = (cyc == 0);
      if (f_first >= 0.0) {
        if (scf_last >= 0.0) {
          scf_growth[scf_state][1] = scf_growth[scf_state][0];
          scf_growth[scf_state][0] = f_first - scf_last;
          ++scf_nobs[scf_state];
        }
        scf_last = f_final;
        scf_state = cyc > 0 ? 1 : 0;
        if (cyc >= 1 && f_first > 0.0 && f_final > 0.0 && f_final < f_first) scf_contract = std::pow(f_final / f_first, 1.0 / cyc);
      }
    }

    const bool atoms_energy = phi_valid && phi_accum;
    if (!done) {
      if (!phi_valid) recip_pass(E_RECIP, [&] { on_side([&] { stage_pair_full(gbuf); }); });
      else stage_pair_full(gbuf);
      // (the partial words are zero: nothing else of this evaluation writes them)
      const FinishArgs<T> fin_t = finish_args(dpos != nullptr, dQl);
      stage_gather(mesh.as<T>(), gbuf, nullptr, false, atoms_energy ? Ed_cur() + E_SLOTS : nullptr, finished ? nullptr : &fin_t);
    }
    if (!finished)
      stage_finish(dpos ? gbuf : nullptr, dQl, atoms_energy ? (int)E_PARTS_SUM : (phi_valid ? (int)E_SCF_RECIP : (int)E_RECIP), E);

    join_side();
    if (!on_device) {
      if (dpos_) HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
      if (dQl_) HIP_TRY(hipMemcpyAsync(dQl_, dQl, 9 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
      if (lpol) HIP_TRY(hipMemcpyAsync(U_, U, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
    if (ncyc) *ncyc = cyc;
    if (conv) *conv = flag;
  }

  // energy_fn / grad_U_fn / grad_pos_fn of the reference (admp/pme.py:69-78): energy and its derivatives at dipoles the
  // CALLER supplies -- no SCF.  Device pointers only.  dU = dE/dUind_global (Cartesian, incl. the self and penalty terms).
  void pme_at_U(const void* pos_, const double* box, const void* Ql_, const void* pol_, const void* thole_, int ns,
                const double* mS, const double* pS, const void* U_, double* E, void* dpos_, void* dU_, void* dQl_) override {
    ARG_CHECK(lpol, "polarizable handle required (the non-polarizable energy is admp_pme_energy_grad)");
    ARG_CHECK(pos_ && box && Ql_ && U_ && E, "null argument");
    ARG_CHECK(!dQl_ || dpos_, "dE_dQlocal requires dE_dpos");
    const int na = top.na;
    grad.need(3 * (size_t)na * sizeof(T));
    T* gbuf = dpos_ ? reinterpret_cast<T*>(dpos_) : grad.as<T>();
    mono_ok = (dQl_ == nullptr);
    if (snranks > 1) {      // slab rank: the rows of the atoms it reads are filled in from their owners -- in a copy, the caller's
      s_U.need(3 * (size_t)na * sizeof(T));                            // array is an input here (its home rows must be valid)
      HIP_TRY(hipMemcpyAsync(s_U.p, U_, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToDevice, stream));
      U_ = s_U.p;
    }
    stage_begin(pos_, box, Ql_, pol_, thole_, ns, mS, pS, const_cast<void*>(U_));
    const bool wantU = dU_ != nullptr;
    stage_pair_full(gbuf, wantU ? fld_pair.as<T>() : nullptr);
    recip_pass(E_RECIP);
    stage_gather(mesh.as<T>(), gbuf, wantU ? fld_recip.as<T>() : nullptr);
    if (wantU) {
      launch_field_finish_only();
      HIP_TRY(hipMemcpyAsync(dU_, field.p, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToDevice, stream));
    }
    stage_finish(dpos_ ? gbuf : nullptr, reinterpret_cast<T*>(dQl_), E_RECIP, E);
    warm_regime = false;
    scf_last = -1.0;
    scf_nobs[0] = scf_nobs[1] = 0;
  }

  void local_frames(const void* pos, const double* box, void* out) override {
    ARG_CHECK(have_top && pos && box && out, "topology must be set; null argument");
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    launch_local_frames<T>(stream, top, reinterpret_cast<const T*>(pos), bx, reinterpret_cast<T*>(out));
  }

  // ---- box gradient: dE/dbox at fixed Cartesian positions = jax.grad(get_energy, argnums=1) of the
  // reference (admp/pme.py:108 with argnums; README.md:7 "force and virial").  The device kernels accumulate
  //   xw[9]  sum shift (x) dE/dr over boundary-crossing pairs and frame vectors + sum_atoms x (x) dE_recip/dx
  //   yy[9]  sum_atoms dE_recip/dAop (the multipole operators of the spread)
  //   tk[6]  sum_k w dG/dk^2 |S|^2 k (x) k
  // and the host folds them with box^-1 (A = inv, rows c, columns j; Aop[i][j] = -K_i A[j][i]):
  //   dE/dbox = -A^T xw  -  A^T (dE/dA)_op A^T  -  2 tk A^T  -  E_recip A^T ,   (dE/dA)_op[j][i] = -K_i yy[i][j]
  DevBuf vir_d;
  enum { V_XW = 0, V_Y = 9, V_TK = 18, V_WORDS = 24 };
  double* vir_begin() {
    vir_d.need(V_WORDS * sizeof(double));
    HIP_TRY(hipMemsetAsync(vir_d.p, 0, V_WORDS * sizeof(double), stream));
    return vir_d.as<double>();
  }
  void vir_assemble(const double* inv, double Erec, double* out) {
    double h[V_WORDS];
    HIP_TRY(hipMemcpyAsync(h, vir_d.p, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    auto A = [&](int c, int j) { return inv[3 * c + j]; };
    const double* xw = h + V_XW;
    const double* yy = h + V_Y;
    const double tk[9] = {h[V_TK + 0], h[V_TK + 3], h[V_TK + 4], h[V_TK + 3], h[V_TK + 1], h[V_TK + 5],
                          h[V_TK + 4], h[V_TK + 5], h[V_TK + 2]};
    double dA[9];   // (dE/dA)_op[j][i]
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) dA[3 * j + i] = -(double)K[i] * yy[3 * i + j];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double v = 0.0;
        for (int c = 0; c < 3; ++c) v -= A(c, a) * xw[3 * c + b];
        for (int j = 0; j < 3; ++j)
          for (int i = 0; i < 3; ++i) v -= A(j, a) * dA[3 * j + i] * A(b, i);
        for (int j = 0; j < 3; ++j) v -= 2.0 * tk[3 * a + j] * A(b, j);
        v -= Erec * A(b, a);
        out[3 * a + b] = v;
      }
  }
  // spread -> r2c -> k-tensor sums -> G multiply (+energy) -> c2r, always through rocFFT (the fused direct-DFT x pass
  // never holds the bare spectrum)
  void recip_pass_virial(int slot, int which, double vol, double* acc, int reuse_bins = 0, int lpol_sites = -1) {
    need_eval_or_disp();
    {
      TIMED("spread");
      int rc = launch_spread<T>(stream, vs_n, vs_sites, lpol_sites < 0 ? lpol : lpol_sites, vs_g, bins, mesh.as<T>(), nullptr,
                                nullptr, 1, reuse_bins);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)};
    }
    fft_forward(mesh.as<T>(), spec.as<T>());
    launch_kspace_virial<T>(stream, K, binv_d.as<double>(), std::fabs(vol), kappa, which, ref_korder, spec.as<T>(), acc + V_TK);
    const T* gt = gtab_cur;
    if (use_pfa) {   // this pass goes through rocFFT: it needs the table in the natural layout, not the slot-ordered one
      gtab_nat.need((size_t)K[0] * K[1] * (K[2] / 2 + 1) * sizeof(T));
      launch_gtab<T>(stream, K, 0, K[1], binv_d.as<double>(), std::fabs(vol), kappa, which, gtab_nat.as<T>(), ref_korder);
      gt = gtab_nat.as<T>();
    }
    { TIMED("kspace"); launch_kspace<T>(stream, K, nyown(), gt, spec.as<T>(), Ed_cur(), slot); }
    fft_inverse(spec.as<T>(), mesh.as<T>());
  }
  int vs_n = 0; const Site<T>* vs_sites = nullptr; RecipGeom<T> vs_g;
  void need_eval_or_disp() { ARG_CHECK(vs_sites != nullptr, "internal: virial pass without sites"); }
  void upload_binv(const double* inv) {
    HIP_TRY(hipMemcpyAsync(binv_d.p, inv, 9 * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));   // `inv` is a caller stack array
  }

  // The same on a slab rank (round 4): the pair / gather / frame sums run over the rank's home rows and atoms, the k-tensor
  // sums over the y rows it holds of the transposed spectrum (between the x transforms of the distributed convolution, which
  // therefore takes the unfused x passes here), and the 24 device words are summed over the ranks before the host folds
  // them -- every rank returns the full dE/dbox.
  void convolve_slab_virial(T* mesh_p, T* spec_p, const T* gtab, int slot, int which, double vol, double* acc) {
    double* Ed = Ed_cur();
    const size_t plane = (size_t)K[1] * K[2];
    const int nx = nxown(), ny = nyown(), nh = K[2] / 2 + 1;
    sl.ghost.need(kGhost * plane * sizeof(T));
    sl.pack.need((size_t)nx * K[1] * nh * 2 * sizeof(T));
    sl.tbuf.need((size_t)K[0] * ny * nh * 2 * sizeof(T));
    { TIMED("comm_ghost"); c_shift(mesh_p + (size_t)nx * plane, sl.ghost.p, (int64_t)(kGhost * plane), real_dtype(), 1, ADMP_TAG_GHOST); }
    { TIMED("mesh_add"); launch_mesh_add<T>(stream, (long)(kGhost * plane), mesh_p, sl.ghost.as<T>()); }
    fft_forward(mesh_p, spec_p);
    { TIMED("transpose_pack"); launch_transpose_pack<T>(stream, nx, K[1], nh, fx_khp, snranks, 0, spec_p, sl.pack.as<T>()); }
    { TIMED("comm_transpose"); c_all_to_all_v(sl.pack.p, sl.tr_send, sl.tbuf.p, sl.tr_recv, real_dtype(), ADMP_TAG_TRANSPOSE); }
    fft_x(sl.tbuf.as<T>(), 0);
    launch_kspace_virial<T>(stream, K, binv_d.as<double>(), std::fabs(vol), kappa, which, ref_korder, sl.tbuf.as<T>(), acc + V_TK,
                            Y0, ny);
    { TIMED("kspace"); launch_kspace<T>(stream, K, ny, gtab, sl.tbuf.as<T>(), Ed, slot); }
    fft_x(sl.tbuf.as<T>(), 1);
    { TIMED("comm_transpose"); c_all_to_all_v(sl.tbuf.p, sl.tr_recv, sl.pack.p, sl.tr_send, real_dtype(), ADMP_TAG_TRANSPOSE); }
    { TIMED("transpose_pack"); launch_transpose_pack<T>(stream, nx, K[1], nh, fx_khp, snranks, 1, spec_p, sl.pack.as<T>()); }
    fft_inverse(spec_p, mesh_p);
    { TIMED("comm_ghost"); c_shift(mesh_p, mesh_p + (size_t)nx * plane, (int64_t)(kGhost * plane), real_dtype(), 0, ADMP_TAG_GHOST); }
  }
  void pme_box_grad_slab(const void* pos_, const double* box, const void* Ql_, const void* pol_, const void* thole_, int ns,
                         const double* mS, const double* pS, const void* U_, double* E, double* dbox) {
    const int na = top.na;
    grad.need(3 * (size_t)na * sizeof(T));
    T* gbuf = grad.as<T>();
    mono_ok = true;
    stage_begin(pos_, box, Ql_, pol_, thole_, ns, mS, pS, const_cast<void*>(U_));
    double inv[9], vol;
    make_box(box, inv, &vol);
    upload_binv(inv);
    double* acc = vir_begin();
    stage_pair_full(gbuf);
    launch_pair_virial<T>(stream, na, nbr, sites.as<Site<T>>(), ev.bx, ev.tab, (T)kappa, lpol, acc + V_XW, pair_rows(), ev.n_home);
    stage_spread(mesh.as<T>());
    if (!slot_clean[E_RECIP]) HIP_TRY(hipMemsetAsync(Ed_cur() + E_RECIP, 0, sizeof(double), stream));
    slot_clean[E_RECIP] = false;
    convolve_slab_virial(mesh.as<T>(), spec.as<T>(), gtab_cur, E_RECIP, 1, vol, acc);
    stage_gather(mesh.as<T>(), gbuf);
    launch_gather_virial<T>(stream, ev.n_home, sites.as<Site<T>>(), lpol, ev.g, mesh.as<T>(), acc + V_XW, acc + V_Y, ev.home);
    if (lmax > 0)
      launch_frame_virial<T>(stream, top, ev.pos, ev.bx, sites.as<Site<T>>(), lpol, (T)kappa, pot.as<T>(), acc + V_XW, ev.home,
                             ev.n_home);
    // (the frame sums need the TOTAL potential of the home sites: pot holds pair + reciprocal space, the self term is added
    // by the kernel itself, as on one rank)
    stage_finish(nullptr, nullptr, E_RECIP, E);
    { TIMED("comm_energies"); c_all_reduce(acc, V_WORDS, ADMP_T_F64, ADMP_OP_SUM, ADMP_TAG_ENERGIES); }
    vir_assemble(inv, E[1], dbox);
    warm_regime = false;
    scf_last = -1.0;
    scf_nobs[0] = scf_nobs[1] = 0;
  }

  void pme_box_grad(const void* pos_, const double* box, const void* Ql_, const void* pol_, const void* thole_, int ns,
                    const double* mS, const double* pS, const void* U_, double* E, double* dbox) override {
    ARG_CHECK(pos_ && box && Ql_ && E && dbox, "null argument");
    if (snranks > 1) {
      if (lpol) ARG_CHECK(U_, "polarizable handle needs the induced dipoles");
      pme_box_grad_slab(pos_, box, Ql_, pol_, thole_, ns, mS, pS, U_, E, dbox);
      return;
    }
    if (lpol) ARG_CHECK(U_, "polarizable handle needs the induced dipoles");
    const int na = top.na;
    grad.need(3 * (size_t)na * sizeof(T));
    T* gbuf = grad.as<T>();
    mono_ok = true;
    stage_begin(pos_, box, Ql_, pol_, thole_, ns, mS, pS, const_cast<void*>(U_));
    double inv[9], vol;
    make_box(box, inv, &vol);
    upload_binv(inv);
    double* acc = vir_begin();
    stage_pair_full(gbuf);
    launch_pair_virial<T>(stream, na, nbr, sites.as<Site<T>>(), ev.bx, ev.tab, (T)kappa, lpol, acc + V_XW);
    vs_n = na; vs_sites = sites.as<Site<T>>(); vs_g = ev.g;
    if (!slot_clean[E_RECIP]) HIP_TRY(hipMemsetAsync(Ed_cur() + E_RECIP, 0, sizeof(double), stream));
    slot_clean[E_RECIP] = false;
    recip_pass_virial(E_RECIP, 1, vol, acc);
    stage_gather(mesh.as<T>(), gbuf);
    launch_gather_virial<T>(stream, na, sites.as<Site<T>>(), lpol, ev.g, mesh.as<T>(), acc + V_XW, acc + V_Y);
    if (lmax > 0)
      launch_frame_virial<T>(stream, top, ev.pos, ev.bx, sites.as<Site<T>>(), lpol, (T)kappa, pot.as<T>(), acc + V_XW);
    stage_finish(nullptr, nullptr, E_RECIP, E);
    vs_sites = nullptr;
    vir_assemble(inv, E[1], dbox);
    warm_regime = false;
    scf_last = -1.0;
    scf_nobs[0] = scf_nobs[1] = 0;
  }

  // dispersion PME box gradient on a slab rank (round 4): the pair sums over its home rows, the channels spread / gathered on
  // its home atoms through the distributed convolution with the k-tensor sums between the x transforms, the self term over
  // its home atoms; the 24 device sums are added over the ranks
  void disp_box_grad_slab(const T* pos, const double* box, const T* cl, int pmax, int ns, const double* mS, double* E,
                          double* dbox) {
    const int na = top.na;
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    ensure_mesh();
    RecipGeom<T> g = make_geom(inv);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    ARG_CHECK(spread_uses_bricks(1 << 30, g), "slab-decomposed dispersion PME needs at least 17 local mesh planes and K2, K3 >= 17");
    grad.need(3 * (size_t)na * sizeof(T));
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, 2 * E_WORDS * sizeof(double), stream));
    double* acc = vir_begin();
    cls_sites_na = slab_sites_na = -1;
    const ScalarRows sr = scalar_rows(pos, g, K[0], X0, X1, true);
    launch_disp_pair<T>(stream, na, nbr, pack_srows(pos, cl, 3), bx, tab, (T)kappa, pmax, grad.as<T>(), Ed, sr.rows, sr.n, cutoff);
    launch_scalar_pair_virial<T>(stream, 0, na, nbr, pos, cl, bx, tab, (T)kappa, pmax, acc + V_XW, cutoff, sr.rows, sr.n);
    const double kp[3] = {-std::pow(kappa, 6) / 12.0, -std::pow(kappa, 8) / 48.0, -std::pow(kappa, 10) / 240.0};
    const int nch = (pmax - 4) / 2;
    const size_t nreal = nreal_local();
    mesh.need(nch * nreal * sizeof(T));
    sites.need(sizeof(Site<T>) * (size_t)na);
    ensure_bins(std::max(sr.n, 1));
    {
      int rc = launch_bin_bricks<T>(stream, sr.n, (const Site<T>*)nullptr, g, bins, sr.home, bases_d.as<int4>());
      if (rc == 0) rc = launch_spread_scalar<T>(stream, nch, pos, cl, 3, g, bins, mesh.as<T>(), (long)nreal);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("dispersion spread: ") + hipGetErrorString((hipError_t)rc)};
      bins.counters_zero = true;
    }
    double* scratchE = Ed + E_WORDS;              // (the all-atom self sums of launch_scalar_sites are not this rank's: discarded)
    for (int c = 0; c < nch; ++c) {
      ensure_gtab(box, inv, vol, 6 + 2 * c);
      upload_binv(inv);
      convolve_slab_virial(mesh.as<T>() + c * nreal, spec.as<T>(), gtab_cur, E_RECIP, 6 + 2 * c, vol, acc);
      launch_scalar_sites<T>(stream, na, pos, cl, 3, c, 0.0, sites.as<Site<T>>(), scratchE);
      launch_gather_virial<T>(stream, sr.n, sites.as<Site<T>>(), 0, g, mesh.as<T>() + c * nreal, acc + V_XW, acc + V_Y, sr.home);
    }
    launch_scalar_self<T>(stream, nch, sr.n, cl, 3, sr.home, kp, Ed);
    read_scalar_energies(Ed, E, 3);
    { TIMED("comm_energies"); c_all_reduce(acc, V_WORDS, ADMP_T_F64, ADMP_OP_SUM, ADMP_TAG_ENERGIES); }
    vir_assemble(inv, E[1], dbox);
  }

  void disp_box_grad(const void* pos_, const double* box, const void* clist_, int pmax, int ns, const double* mS, double* E,
                     double* dbox) override {
    ARG_CHECK(have_top && have_ewald && have_pairs, "topology, ewald parameters and pairs must be set first");
    ARG_CHECK(pos_ && box && clist_ && E && dbox, "null argument");
    ARG_CHECK(pmax == 6 || pmax == 8 || pmax == 10, "pmax must be 6, 8 or 10");
    if (snranks > 1) {
      disp_box_grad_slab(reinterpret_cast<const T*>(pos_), box, reinterpret_cast<const T*>(clist_), pmax, ns, mS, E, dbox);
      return;
    }
    const int na = top.na;
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    RecipGeom<T> g = make_geom(inv);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    ensure_mesh();
    const T* pos = reinterpret_cast<const T*>(pos_);
    const T* cl = reinterpret_cast<const T*>(clist_);
    grad.need(3 * (size_t)na * sizeof(T));
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_SLOTS * sizeof(double), stream));
    double* acc = vir_begin();
    launch_disp_pair<T>(stream, na, nbr, pack_srows(pos, cl, 3), bx, tab, (T)kappa, pmax, grad.as<T>(), Ed, nullptr, 0, cutoff);
    launch_scalar_pair_virial<T>(stream, 0, na, nbr, pos, cl, bx, tab, (T)kappa, pmax, acc + V_XW, cutoff);
    cls_sites_na = slab_sites_na = -1;   // other rows than an electrostatics evaluation's
    sites.need(sizeof(Site<T>) * (size_t)na);
    ensure_bins(na);
    const double kp[3] = {-std::pow(kappa, 6) / 12.0, -std::pow(kappa, 8) / 48.0, -std::pow(kappa, 10) / 240.0};
    const int nch = (pmax - 4) / 2;
    vs_n = na; vs_sites = sites.as<Site<T>>(); vs_g = g;
    for (int c = 0; c < nch; ++c) {
      ensure_gtab(box, inv, vol, 6 + 2 * c);
      upload_binv(inv);
      launch_scalar_sites<T>(stream, na, pos, cl, 3, c, kp[c], sites.as<Site<T>>(), Ed);
      recip_pass_virial(E_RECIP, 6 + 2 * c, vol, acc, c > 0, 0);
      launch_gather_virial<T>(stream, na, sites.as<Site<T>>(), 0, g, mesh.as<T>(), acc + V_XW, acc + V_Y);
    }
    vs_sites = nullptr;
    double Eh2[E_SLOTS];
    HIP_TRY(hipMemcpyAsync(Eh2, Ed, sizeof(Eh2), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    E[0] = Eh2[E_REAL]; E[1] = Eh2[E_RECIP]; E[2] = Eh2[E_SELF];
    vir_assemble(inv, E[1], dbox);
  }

  void tt_box_grad(const void* pos_, const double* box, const void* abqc_, int ns, const double* mS, double* E,
                   double* dbox) override {
    ARG_CHECK(have_top && have_pairs, "topology and pairs must be set first");
    ARG_CHECK(pos_ && box && abqc_ && E && dbox, "null argument");
    const int na = top.na;
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    const T* pos = reinterpret_cast<const T*>(pos_);
    const T* par = reinterpret_cast<const T*>(abqc_);
    grad.need(3 * (size_t)na * sizeof(T));
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_WORDS * sizeof(double), stream));
    double* acc = vir_begin();
    ScalarRows sr{nullptr, na, nullptr};
    if (snranks > 1) {     // ownership by x-slabs of a virtual mesh, as in tt(); the sums are added over the ranks
      ARG_CHECK(have_comm, "slab-decomposed handle without a communicator (admp_set_comm)");
      RecipGeom<T> g;
      const int Kv = 64 * snranks;
      g.K[0] = Kv; g.K[1] = g.K[2] = 32;
      for (int k = 0; k < 9; ++k) { g.hinv[k] = (T)inv[k]; g.Aop[k] = g.Jac[k] = T(0); }
      g.xoff = 64 * srank; g.nloc0 = 64 + kGhost; g.wrap0 = 1 << 30;
      sr = scalar_rows(pos, g, Kv, 64 * srank, 64 * (srank + 1), true);
    }
    launch_tt_pair<T>(stream, na, nbr, pack_srows(pos, par, 4), bx, tab, grad.as<T>(), Ed, sr.rows, sr.n, cutoff);
    launch_scalar_pair_virial<T>(stream, 1, na, nbr, pos, par, bx, tab, T(0), 0, acc + V_XW, cutoff, sr.rows, sr.n);
    read_scalar_energies(Ed, E, 1);
    if (snranks > 1) { TIMED("comm_energies"); c_all_reduce(acc, V_WORDS, ADMP_T_F64, ADMP_OP_SUM, ADMP_TAG_ENERGIES); }
    vir_assemble(inv, 0.0, dbox);      // only the xw sums are non-zero here
  }

  // ---- MD-driver helpers (md_kernels.hip): device pointers, nothing read back -----------------------------
  void md_bonded(const void* pos, const double* box, int nb, const int32_t* bidx, const void* bpar, int na, const int32_t* aidx,
                 const void* apar, double* E_dev, void* grad_) override {
    ARG_CHECK(pos && box && E_dev && grad_ && nb >= 0 && na >= 0, "bad argument");
    ARG_CHECK((nb == 0 || (bidx && bpar)) && (na == 0 || (aidx && apar)), "bond / angle lists missing");
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    TIMED("md_bonded");
    launch_md_bonded<T>(stream, nb, bidx, reinterpret_cast<const T*>(bpar), na, aidx, reinterpret_cast<const T*>(apar),
                        reinterpret_cast<const T*>(pos), bx, reinterpret_cast<T*>(grad_), E_dev);
  }
  void md_kick_drift(int n, void* pos, void* vel, const void* grad_, const void* inv_mass, double half_dt_acc, double dt,
                     double* ekin_dev) override {
    ARG_CHECK(n >= 0 && vel && grad_ && inv_mass && (dt == 0.0 || pos), "bad argument");
    TIMED("md_kick_drift");
    launch_md_kick_drift<T>(stream, n, reinterpret_cast<T*>(pos), reinterpret_cast<T*>(vel), reinterpret_cast<const T*>(grad_),
                            reinterpret_cast<const T*>(inv_mass), half_dt_acc, dt, ekin_dev);
  }

  // ---- neighbour search (cell list) ------------------------------------------------------------------
  CellScratch cells;
  int nb_na = 0; const T* nb_pos = nullptr; Box<T> nb_box; double nb_rc = 0;
  void nbr_count(int na, const void* pos, const double* box, double rc, int64_t* n_pairs) override {
    ARG_CHECK(na > 0 && pos && box && rc > 0 && n_pairs, "bad argument");
    double inv[9], vol;
    nb_box = make_box(box, inv, &vol);
    double heights[3];
    for (int d = 0; d < 3; ++d)   // perpendicular height along lattice direction d = 1 / |column d of box^-1|
      heights[d] = 1.0 / std::sqrt(inv[0 + d] * inv[0 + d] + inv[3 + d] * inv[3 + d] + inv[6 + d] * inv[6 + d]);
    for (int d = 0; d < 3; ++d) ARG_CHECK(rc <= 0.5 * heights[d] * (1 + 1e-12), "rc exceeds half the box height (minimum image)");
    nb_na = na; nb_pos = reinterpret_cast<const T*>(pos); nb_rc = rc;
    long long n = 0;
    TIMED("neighbor_count");
    int r = cell_count_pairs<T>(stream, na, nb_pos, nb_box, heights, rc, cells, &n);
    if (r != 0) throw Err{ADMP_E_HIP, std::string("cell_count_pairs: ") + hipGetErrorString((hipError_t)r)};
    *n_pairs = (int64_t)n;
  }
  void nbr_fill(int32_t* pairs) override {
    ARG_CHECK(nb_pos && pairs, "admp_neighbor_count must precede admp_neighbor_fill");
    TIMED("neighbor_fill");
    int r = cell_fill_pairs<T>(stream, nb_na, nb_pos, nb_box, nb_rc, cells, pairs);
    if (r != 0) throw Err{ADMP_E_HIP, std::string("cell_fill_pairs: ") + hipGetErrorString((hipError_t)r)};
    nb_pos = nullptr;
  }

  // admp_prune_pairs: from now on the calculators walk the entries of the current table that lie below rc (see EngineBase::pruned)
  DevBuf prune_rowptr, prune_cnt, prune_col;
  void prune_pairs(const void* pos, const double* box, double rc) override {
    ARG_CHECK(have_top && have_pairs, "a pair list must be set before it can be pruned");
    ARG_CHECK(!nbr_src, "prune the lender's table: a borrowed one follows it");
    ARG_CHECK(snranks == 1, "admp_prune_pairs: single-rank handles only");
    ARG_CHECK(pos && box, "null argument");
    if (rc <= 0) { unprune(); return; }
    unprune();                                   // (always from the table as built)
    const int na = top.na;
    double inv[9], vol;
    const Box<T> b = make_box(box, inv, &vol);
    const size_t entries = 2 * (size_t)nbr.n_half;
    prune_rowptr.need(sizeof(int) * ((size_t)na + 1));
    prune_cnt.need(sizeof(int) * ((size_t)na + 1));
    prune_col.need(sizeof(int) * (entries + 1));
    int64_t total = 0;
    TIMED("prune_pairs");
    const int r = prune_table<T>(stream, na, nbr, reinterpret_cast<const T*>(pos), b, rc, prune_rowptr.as<int>(),
                                 prune_cnt.as<int>(), prune_col.as<int>(), &scan_scratch.p, &scan_bytes, &total);
    if (r != 0) throw Err{ADMP_E_HIP, std::string("prune_table: ") + hipGetErrorString((hipError_t)r)};
    nbr_full = nbr;
    nbr.rowptr = prune_rowptr.as<int>();
    nbr.col = prune_col.as<int>();
    nbr.n_half = total / 2;
    nbr.cap = (int64_t)entries + 1;
    nbr.col_alt = nullptr; nbr.cap_alt = 0;      // (build scratch of the full table: not this one's)
    nbr.deg = nullptr; nbr.deg_na = 0;
    pruned = true;
    ++nbr_gen;
  }
  void nbr_table(const void* pos, const double* box, double rc) override {
    ARG_CHECK(have_top, "admp_set_topology must precede admp_set_pairs_from_positions");
    ARG_CHECK(pos && box && rc > 0, "bad argument");
    unprune();
    detach_shared();
    double inv[9], vol;
    Box<T> b = make_box(box, inv, &vol);
    double heights[3];
    for (int d = 0; d < 3; ++d)
      heights[d] = 1.0 / std::sqrt(inv[0 + d] * inv[0 + d] + inv[3 + d] * inv[3 + d] + inv[6 + d] * inv[6 + d]);
    for (int d = 0; d < 3; ++d) ARG_CHECK(rc <= 0.5 * heights[d] * (1 + 1e-12), "rc exceeds half the box height (minimum image)");
    // Slab rank with a mesh (round 4): only the rows of the atoms whose stencil base plane lies within the rank's slab plus a
    // margin -- half the list cutoff (far more than half a skin) plus 4 planes for the pair potentials' slab rule -- are built:
    // the search then costs the rank its share of the box, not the box.  ADMP_SLAB_ROWS=0: every row on every rank.
    RowFilter rf;
    static const bool rows_on = [] { const char* e = getenv("ADMP_SLAB_ROWS"); return !(e && atoi(e) == 0); }();
    if (rows_on && snranks > 1 && have_ewald && have_comm) {
      update_slab();
      const int mp = (int)std::ceil(0.5 * rc / (heights[0] / K[0])) + 4;
      const int width = (X1 - X0) + 2 * mp;
      if (width < K[0]) { rf.on = 1; rf.K0 = K[0]; rf.lo = ((X0 - mp) % K[0] + K[0]) % K[0]; rf.width = width; }
    }
    TIMED("neighbor_table");
    int r = cell_build_table<T>(stream, top, reinterpret_cast<const T*>(pos), b, heights, rc, cells, nbr, rf);
    if (r != 0) throw Err{ADMP_E_HIP, std::string("cell_build_table: ") + hipGetErrorString((hipError_t)r)};
    apply_classes();
    order_rows();
    have_pairs = true;
    ++nbr_gen;
  }

  void slab_info(int64_t* o) override {
    ARG_CHECK(have_ewald, "admp_set_ewald first");
    update_slab();
    o[0] = X0; o[1] = X1; o[2] = Y0; o[3] = Y1; o[4] = nloc0(); o[5] = kGhost; o[6] = K[0]; o[7] = K[1]; o[8] = K[2] / 2 + 1;
    o[9] = srank; o[10] = snranks;
  }
  DevBuf srow_d;      // packed (position, parameters) rows of the scalar pair kernels, rebuilt per call
  const SRow<T>* pack_srows(const T* pos, const T* par, int np) {
    srow_d.need(sizeof(SRow<T>) * (size_t)top.na);
    launch_pack_scalar_rows<T>(stream, top.na, np, pos, par, srow_d.as<SRow<T>>());
    return srow_d.as<SRow<T>>();
  }
  // energies of a dispersion / pair-potential call: (real, recip, self) of all ranks
  void read_scalar_energies(double* Ed, double* E, int n, bool types_checked = false) {
    double Eh2[E_WORDS];
    if (snranks > 1) {
      TIMED("comm_energies");
      launch_energy_pack(stream, Ed, E_RECIP, Ed + E_RED);
      c_all_reduce(Ed + E_RED, 4, ADMP_T_F64, ADMP_OP_SUM, ADMP_TAG_ENERGIES);
    }
    HIP_TRY(hipMemcpyAsync(Eh2, Ed, sizeof(Eh2), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const double* src = snranks > 1 ? Eh2 + E_RED : Eh2;      // (E_REAL, E_RECIP, E_SELF are words 0, 1, 2 either way)
    if (types_checked && Eh2[E_FMAX] != 0.0)      // (k_types_check counted rows that differ from their type's coefficients)
      throw Err{ADMP_E_ARG, "admp_disp_set_types: c_list rows differ from the coefficients of their types"};
    for (int k = 0; k < n; ++k) E[k] = src[k];
  }
  // rows of a scalar pair / dispersion call: all atoms in the table's length-sorted order, or -- on a slab rank -- the
  // home rows of this evaluation (ownership by the stencil base plane of mesh geometry g)
  struct ScalarRows { const int* rows; int n; const int* home; };
  ScalarRows scalar_rows(const T* pos, const RecipGeom<T>& g, int K0v, int X0v, int X1v, bool need_bases) {
    const int na = top.na;
    const int* order = nbr.order_plain ? nbr.order_plain : nbr.order;
    if (snranks > 1 || need_bases) {
      bases_d.need(sizeof(int4) * (size_t)na);
      TIMED("atom_bases");
      launch_atom_bases<T>(stream, na, pos, g, bases_d.as<int4>());
    }
    if (snranks == 1) return {order, na, nullptr};
    ev.bases = bases_d.as<int4>(); ev.pol = nullptr; ev.U = nullptr;
    decompose(K0v, X0v, X1v, order);
    return {sl.rows, sl.n_home, sl.home};
  }

  // admp_disp_set_types: the atoms' coefficient rows take nt <= 3 distinct values; types (device int32[Na], the caller's) picks
  // the row ctab[type][3] of every atom.  Used by the single-rank fused-x path in single precision (typed meshes: one spread
  // and one gather per ATOM instead of per channel, nt transforms instead of one per channel); verified against c_list in
  // every call that uses it.  nt = 0: forget.
  int disp_nt = 0;
  const int* disp_types = nullptr;
  bool disp_counts_ok = false, disp_onehot_ok = false;      // per type table: atoms per type (host), one-hot weights (device)
  double disp_counts[4] = {0, 0, 0, 0};
  // self term of the typed form: sum_p kp[p] sum_t n_t c_p,t^2 (admp/disp_pme.py:254-279 summed by type)
  double typed_self_energy(int na, int nch, const double* kp) {
    if (!disp_counts_ok) {
      tcount_d.need(256);
      int* cd = reinterpret_cast<int*>(tcount_d.p);
      int ch[4];
      HIP_TRY(hipMemsetAsync(cd, 0, 4 * sizeof(int), stream));
      launch_type_counts(stream, na, disp_types, cd);
      HIP_TRY(hipMemcpyAsync(ch, cd, sizeof(ch), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      for (int t = 0; t < 4; ++t) disp_counts[t] = (double)ch[t];
      disp_counts_ok = true;
    }
    double e = 0.0;
    for (int p = 0; p < nch; ++p)
      for (int t = 0; t < disp_nt; ++t) e += kp[p] * disp_counts[t] * disp_ctab[t][p] * disp_ctab[t][p];
    return e;
  }
  double disp_ctab[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  void disp_set_types(int nt, const void* types, const double* ctab) override {
    if (nt <= 0 || !types || !ctab) { disp_nt = 0; disp_types = nullptr; return; }
    ARG_CHECK(nt <= 3, "admp_disp_set_types: at most 3 types");
    disp_nt = nt;
    disp_types = reinterpret_cast<const int*>(types);
    disp_counts_ok = disp_onehot_ok = false;
    for (int t = 0; t < nt; ++t)
      for (int c = 0; c < 3; ++c) disp_ctab[t][c] = ctab[3 * t + c];
  }
  // dispersion PME (admp/disp_pme.py:80-123): real-space pairs + one scalar reciprocal pass per power
  void disp(const void* pos_, const double* box, const void* clist_, int pmax, int ns, const double* mS, double* E,
            void* dpos_, int on_device) override {
    ARG_CHECK(have_top && have_ewald && have_pairs, "topology, ewald parameters and pairs must be set first");
    ARG_CHECK(snranks == 1 || on_device, "a slab-decomposed handle takes device pointers");
    ARG_CHECK(pos_ && box && clist_ && E, "null argument");
    ARG_CHECK(pmax == 6 || pmax == 8 || pmax == 10, "pmax must be 6, 8 or 10");
    const int na = top.na;
    HIP_TRY(hipSetDevice(device));
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    ensure_mesh();
    RecipGeom<T> g = make_geom(inv);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    const T* pos = stage_in(s_pos, pos_, 3 * (size_t)na, on_device);
    const T* cl = stage_in(s_par, clist_, 3 * (size_t)na, on_device);
    T* dpos = nullptr;
    grad.need(3 * (size_t)na * sizeof(T));
    if (dpos_ && on_device) dpos = reinterpret_cast<T*>(dpos_);
    else dpos = grad.as<T>();
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_WORDS * sizeof(double), stream));
    const double kp[3] = {-std::pow(kappa, 6) / 12.0, -std::pow(kappa, 8) / 48.0, -std::pow(kappa, 10) / 240.0};
    const int nch = (pmax - 4) / 2;
    cls_sites_na = slab_sites_na = -1;   // other rows than an electrostatics evaluation's
    // At scale (brick regime) and on every slab rank the channels go through ONE binning, ONE spread and ONE gather straight
    // from the caller's position / coefficient arrays (disp_kernels.hip); small single-GPU systems keep the batched
    // scan-spread / direct-DFT path below (dispatch-bound: nine launches for the three powers).
    const bool fused = !(use_dft || use_pfa) && (snranks > 1 || spread_uses_bricks(na, g));
    if (snranks > 1) ARG_CHECK(spread_uses_bricks(1 << 30, g), "slab-decomposed dispersion PME needs at least 17 local mesh planes and K2, K3 >= 17");
    const ScalarRows sr = scalar_rows(pos, g, K[0], X0, X1, fused);
    { TIMED("disp_pair"); launch_disp_pair<T>(stream, na, nbr, pack_srows(pos, cl, 3), bx, tab, (T)kappa, pmax, dpos, Ed, sr.rows, sr.n, cutoff); }
    if (fused) {
      const size_t nreal = nreal_local();
      mesh.need(nch * nreal * sizeof(T));
      ensure_bins(std::max(sr.n, 1));
      static const bool typed_on = [] { const char* e = getenv("ADMP_DISP_TYPES"); return !(e && atoi(e) == 0); }();
      static const bool batch_on_t = [] { const char* e = getenv("ADMP_DISP_BATCH"); return !(e && atoi(e) == 0); }();
      // (more types than powers would mean more transforms than the per-power form: pmax 6 with two types keeps its one mesh)
      const bool typed = typed_on && batch_on_t && snranks == 1 && use_fx && sizeof(T) == 4 && disp_nt >= 1 && disp_nt <= nch &&
                         disp_types;
      if (typed) {
        // Typed meshes (disp_kernels.hip): nt type meshes through the transforms, combined per k in the x pass
        const int nt = disp_nt;
        MixTab mix;
        mix.nch = nch; mix.nt = nt;
        for (int c = 0; c < 3; ++c)
          for (int t = 0; t < 4; ++t) mix.c[c][t] = (c < nch && t < nt) ? disp_ctab[t][c] : 0.0;
        launch_types_check<T>(stream, na, cl, 3, disp_types, mix, Ed + E_FMAX);
        const size_t nspec = 2 * (size_t)K[0] * K[1] * fx_khp;
        mesh.need((size_t)std::max(nt, nch) * nreal * sizeof(T));
        spec.need((size_t)nt * nspec * sizeof(T));
        { TIMED("spread");
          int rc = launch_bin_bricks<T>(stream, sr.n, (const Site<T>*)nullptr, g, bins, sr.home, bases_d.as<int4>());
          if (rc == 0) rc = launch_spread_typed<T>(stream, nt, pos, disp_types, g, bins, mesh.as<T>(), (long)nreal);
          if (rc != 0) throw Err{ADMP_E_HIP, std::string("dispersion spread (typed): ") + hipGetErrorString((hipError_t)rc)};
          bins.counters_zero = true; }
        DftTabs<T> tabs;
        for (int c = 0; c < nch; ++c) { ensure_gtab(box, inv, vol, 6 + 2 * c); tabs.p[c] = gtab_cur; }
        if (nt >= 2) ensure_batched_plans(nt);
        run_plan("rocfft_r2c_yz", nt >= 2 ? plan2n_f[nt] : plan2_f, mesh.p, spec.p);
        { TIMED("fftx_kspace");
          launch_fftx_mix<T>(stream, K, fx_tw.as<T>(), spec.as<T>(), tabs, mix, (long)nspec, Ed, E_RECIP, fx_khp); }
        run_plan("rocfft_c2r_yz", nt >= 2 ? plan2n_b[nt] : plan2_b, spec.p, mesh.p);
        { TIMED("gather_field");
          launch_gather_scalar<T>(stream, 1, sr.n, pos, cl, 3, g, mesh.as<T>(), (long)nreal, dpos, sr.home, 0, disp_types); }
        const double e_self = typed_self_energy(na, nch, kp);
        read_scalar_energies(Ed, E, 3, true);
        E[2] = e_self;
        if (dpos_ && !on_device) {
          HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
          HIP_TRY(hipStreamSynchronize(stream));
        }
        return;
      }
      { TIMED("spread");
        int rc = launch_bin_bricks<T>(stream, sr.n, (const Site<T>*)nullptr, g, bins, sr.home, bases_d.as<int4>());
        if (rc == 0) rc = launch_spread_scalar<T>(stream, nch, pos, cl, 3, g, bins, mesh.as<T>(), (long)nreal);
        if (rc != 0) throw Err{ADMP_E_HIP, std::string("dispersion spread: ") + hipGetErrorString((hipError_t)rc)};
        bins.counters_zero = true; }
      // every channel is gathered right after its transform.  The gather is bound by the cache-line rate of its 36 mesh loads
      // per lane, whatever it sums: 0.175 ms per channel at 1M atoms in every form tried (one pass over the three meshes
      // 0.52-0.61 ms; one pass per channel from its own or from a shared, cache-resident phi buffer 0.525; round 2's
      // k_gather_field_staged on site rows 0.525) -- what the fused path saves is the site rows and the scale-add pass.
      // Round 4, one rank on a fused-x mesh: the nch meshes go through ONE batched r2c, ONE x pass (a G table per channel)
      // and ONE batched c2r; then the channels of every mesh point are laid side by side (one streaming pass) and ONE gather
      // fetches them with a single load per stencil point -- the gather is bound by its load instructions (36 per lane and
      // mesh), not by the bytes they return.  ADMP_DISP_BATCH=0: the per-channel loop (A/B, tests).
      static const bool batch_on = [] { const char* e = getenv("ADMP_DISP_BATCH"); return !(e && atoi(e) == 0); }();
      if (batch_on && snranks == 1 && use_fx && nch >= 2) {
        const size_t nspec = 2 * (size_t)K[0] * K[1] * fx_khp;
        spec.need(nch * nspec * sizeof(T));
        mesh2.need(nch * nreal * sizeof(T));
        DftTabs<T> tabs;
        for (int c = 0; c < nch; ++c) { ensure_gtab(box, inv, vol, 6 + 2 * c); tabs.p[c] = gtab_cur; }
        convolve_batch(mesh.as<T>(), spec.as<T>(), tabs, nch, nreal, nspec, E_RECIP);
        { TIMED("interleave"); launch_interleave<T>(stream, nch, (long)nreal, mesh.as<T>(), (long)nreal, mesh2.as<T>()); }
        { TIMED("gather_field");
          launch_gather_scalar<T>(stream, nch, sr.n, pos, cl, 3, g, mesh2.as<T>(), (long)nreal, dpos, sr.home, 1); }
      } else
      for (int c = 0; c < nch; ++c) {
        ensure_gtab(box, inv, vol, 6 + 2 * c);
        convolve(mesh.as<T>() + c * nreal, spec.as<T>(), gtab_cur, E_RECIP);
        TIMED("gather_field");
        launch_gather_scalar<T>(stream, 1, sr.n, pos, cl + c, 3, g, mesh.as<T>() + c * nreal, (long)nreal, dpos, sr.home);
      }
      { TIMED("scalar_self"); launch_scalar_self<T>(stream, nch, sr.n, cl, 3, sr.home, kp, Ed); }
      read_scalar_energies(Ed, E, 3);
      if (dpos_ && !on_device) {
        HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
      }
      return;
    }
    // one scalar reciprocal pass per power through the same spread / gather kernels as the electrostatics:
    // the channel is packed into charge-only site rows (which also accumulates the self term, disp_pme.py:254-279)
    sites.need(sizeof(Site<T>) * (size_t)na);
    fld_recip.need(3 * (size_t)na * sizeof(T));
    ensure_bins(na);
    RecipGeom<T> gj = g;                       // scalar sites: dE/dr = c * Jac . F1 (gather_field applies g.Aop)
    for (int k = 0; k < 9; ++k) gj.Aop[k] = g.Jac[k];
    // Typed meshes on a direct-DFT mesh (small systems; see admp_disp_set_types): the types take the place of the powers in the
    // batch -- one-hot weights through the same site / spread / gather kernels, the x pass combines (dft_kernels.hip k_dft_x_mix)
    static const bool typed_small_on = [] { const char* e = getenv("ADMP_DISP_TYPES"); return !(e && atoi(e) == 0); }();
    if (typed_small_on && disp_nt >= 1 && disp_nt < nch && disp_types && use_dft && !use_pfa && snranks == 1 &&
        !spread_uses_bricks(na, g)) {
      const int nt = disp_nt;
      MixTab mix;
      mix.nch = nch; mix.nt = nt;
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 4; ++t) mix.c[c][t] = (c < nch && t < nt) ? disp_ctab[t][c] : 0.0;
      launch_types_check<T>(stream, na, cl, 3, disp_types, mix, Ed + E_FMAX);
      const size_t nreal = (size_t)K[0] * K[1] * K[2], nspec = 2 * (size_t)K[0] * K[1] * (K[2] / 2 + 1);
      mesh.need(nt * nreal * sizeof(T));
      spec.need(nt * nspec * sizeof(T));
      sites.need(sizeof(Site<T>) * (size_t)na * nt);
      fld_recip.need(3 * (size_t)na * sizeof(T) * nt);
      onehot_d.need((size_t)na * nt * sizeof(T));
      bases_d.need(sizeof(int4) * (size_t)na);
      DftTabs<T> tabs;
      for (int c = 0; c < nch; ++c) { ensure_gtab(box, inv, vol, 6 + 2 * c); tabs.p[c] = gtab_cur; }
      const double no_self[3] = {0.0, 0.0, 0.0};
      { TIMED("scalar_sites");
        if (!disp_onehot_ok || onehot_for != onehot_d.p) {
          launch_onehot<T>(stream, na, nt, disp_types, onehot_d.as<T>());
          disp_onehot_ok = true; onehot_for = onehot_d.p;
        }
        launch_scalar_sites_batch<T>(stream, na, pos, onehot_d.as<T>(), nt, nt, no_self, sites.as<Site<T>>(), Ed, &g,
                                     bases_d.as<int4>()); }
      { TIMED("spread");
        int rc = launch_spread<T>(stream, na, sites.as<Site<T>>(), 0, g, bins, mesh.as<T>(), nullptr, bases_d.as<int4>(), nt);
        if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)}; }
      const T* tw = dft_tw.as<T>();
      { TIMED("dft_z_r2c"); launch_dft_z<T>(stream, K, tw, mesh.as<T>(), spec.as<T>(), 0, nt, (long)nreal, (long)nspec); }
      { TIMED("dft_y_fwd"); launch_dft_y<T>(stream, K, tw, spec.as<T>(), 0, nt, (long)nspec); }
      { TIMED("dft_x_kspace"); launch_dft_x_mix<T>(stream, K, tw, spec.as<T>(), tabs, mix, (long)nspec, Ed, E_RECIP); }
      { TIMED("dft_y_inv"); launch_dft_y<T>(stream, K, tw, spec.as<T>(), 1, nt, (long)nspec); }
      { TIMED("dft_z_c2r"); launch_dft_z<T>(stream, K, tw, mesh.as<T>(), spec.as<T>(), 1, nt, (long)nreal, (long)nspec); }
      { TIMED("gather_field"); launch_gather_field<T>(stream, na, sites.as<Site<T>>(), gj, mesh.as<T>(), fld_recip.as<T>(), nullptr, nt); }
      { TIMED("scale_add"); launch_scale_add<T>(stream, na, onehot_d.as<T>(), nt, 0, fld_recip.as<T>(), dpos, nt); }
      const double e_self = typed_self_energy(na, nch, kp);
      read_scalar_energies(Ed, E, 3, true);
      E[2] = e_self;
      if (dpos_ && !on_device) {
        HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
      }
      return;
    }
    if ((use_dft || use_pfa) && nch > 1) {
      // direct-DFT meshes are small and dispatch bound: the powers are spread into separate meshes and transformed
      // as ONE batch (5 launches instead of 5 per power); gather per power from its own mesh
      const size_t nreal = (size_t)K[0] * K[1] * K[2], nspec = 2 * (size_t)K[0] * K[1] * (use_pfa ? pfa.Khp : K[2] / 2 + 1);
      mesh.need(nch * nreal * sizeof(T));
      spec.need(nch * nspec * sizeof(T));
      DftTabs<T> tabs;
      for (int c = 0; c < nch; ++c) {
        ensure_gtab(box, inv, vol, 6 + 2 * c);
        tabs.p[c] = gtab_cur;
      }
      sites.need(sizeof(Site<T>) * (size_t)na * nch);
      fld_recip.need(3 * (size_t)na * sizeof(T) * nch);
      const bool batch_spread = !spread_uses_bricks(na, g);        // the scan-spread regime takes the channels as a batch
      if (batch_spread) {
        // the stencil records ride along: every (brick, channel) workgroup of the scan spread reads them instead of
        // redoing three grid_ref per atom (3072 atoms x 1029 workgroups at 97^3)
        bases_d.need(sizeof(int4) * (size_t)na);
        { TIMED("scalar_sites"); launch_scalar_sites_batch<T>(stream, na, pos, cl, 3, nch, kp, sites.as<Site<T>>(), Ed, &g,
                                                              bases_d.as<int4>()); }
        TIMED("spread");
        int rc = launch_spread<T>(stream, na, sites.as<Site<T>>(), 0, g, bins, mesh.as<T>(), nullptr, bases_d.as<int4>(), nch);
        if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)};
      } else {
        for (int c = 0; c < nch; ++c) {
          { TIMED("scalar_sites"); launch_scalar_sites<T>(stream, na, pos, cl, 3, c, kp[c], sites.as<Site<T>>(), Ed); }
          TIMED("spread");
          int rc = launch_spread<T>(stream, na, sites.as<Site<T>>(), 0, g, bins, mesh.as<T>() + c * nreal, nullptr, nullptr, 1,
                                    c > 0);
          if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)};
        }
      }
      if (use_pfa) {
        const T* tw = pfa_tw.as<T>();
        { TIMED("dft_z_r2c"); launch_pfa_z<T>(stream, pfa, tw, mesh.as<T>(), spec.as<T>(), 0, nch, (long)nreal, (long)nspec); }
        { TIMED("dft_y_fwd"); launch_pfa_y<T>(stream, pfa, tw, spec.as<T>(), 0, nch, (long)nspec); }
        { TIMED("dft_x_kspace"); launch_pfa_x_conv<T>(stream, pfa, tw, spec.as<T>(), tabs, Ed, E_RECIP, nch, (long)nspec); }
        { TIMED("dft_y_inv"); launch_pfa_y<T>(stream, pfa, tw, spec.as<T>(), 1, nch, (long)nspec); }
        { TIMED("dft_z_c2r"); launch_pfa_z<T>(stream, pfa, tw, mesh.as<T>(), spec.as<T>(), 1, nch, (long)nreal, (long)nspec); }
      } else {
        const T* tw = dft_tw.as<T>();
        { TIMED("dft_z_r2c"); launch_dft_z<T>(stream, K, tw, mesh.as<T>(), spec.as<T>(), 0, nch, (long)nreal, (long)nspec); }
        { TIMED("dft_y_fwd"); launch_dft_y<T>(stream, K, tw, spec.as<T>(), 0, nch, (long)nspec); }
        { TIMED("dft_x_kspace"); launch_dft_x_conv<T>(stream, K, tw, spec.as<T>(), tabs, Ed, E_RECIP, nch, (long)nspec); }
        { TIMED("dft_y_inv"); launch_dft_y<T>(stream, K, tw, spec.as<T>(), 1, nch, (long)nspec); }
        { TIMED("dft_z_c2r"); launch_dft_z<T>(stream, K, tw, mesh.as<T>(), spec.as<T>(), 1, nch, (long)nreal, (long)nspec); }
      }
      // the site rows hold the positions (only the charge slot differs per power): one gather over the batch of meshes
      { TIMED("gather_field"); launch_gather_field<T>(stream, na, sites.as<Site<T>>(), gj, mesh.as<T>(), fld_recip.as<T>(), nullptr, nch); }
      { TIMED("scale_add"); launch_scale_add<T>(stream, na, cl, 3, 0, fld_recip.as<T>(), dpos, nch); }
    } else
    for (int c = 0; c < nch; ++c) {
      ensure_gtab(box, inv, vol, 6 + 2 * c);
      { TIMED("scalar_sites"); launch_scalar_sites<T>(stream, na, pos, cl, 3, c, kp[c], sites.as<Site<T>>(), Ed); }
      {
        TIMED("spread");
        int rc = launch_spread<T>(stream, na, sites.as<Site<T>>(), 0, g, bins, mesh.as<T>(), nullptr, nullptr, 1, c > 0);
        if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)};
      }
      convolve(mesh.as<T>(), spec.as<T>(), gtab_cur, E_RECIP);
      { TIMED("gather_field"); launch_gather_field<T>(stream, na, sites.as<Site<T>>(), gj, mesh.as<T>(), fld_recip.as<T>(), nullptr); }
      { TIMED("scale_add"); launch_scale_add<T>(stream, na, cl, 3, c, fld_recip.as<T>(), dpos); }
    }
    double Eh[E_SLOTS];
    HIP_TRY(hipMemcpyAsync(Eh, Ed, sizeof(Eh), hipMemcpyDeviceToHost, stream));
    if (dpos_ && !on_device) HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    E[0] = Eh[E_REAL]; E[1] = Eh[E_RECIP]; E[2] = Eh[E_SELF];
  }

  void tt(const void* pos_, const double* box, const void* abqc_, int ns, const double* mS, double* E, void* dpos_,
          int on_device) override {
    ARG_CHECK(have_top && have_pairs, "topology and pairs must be set first");
    ARG_CHECK(snranks == 1 || on_device, "a slab-decomposed handle takes device pointers");
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
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_WORDS * sizeof(double), stream));
    ScalarRows sr{nullptr, na, nullptr};
    if (snranks > 1) {     // no mesh on a pair potential: ownership by x-slabs of a virtual mesh of 64 planes per rank
      ARG_CHECK(have_comm, "slab-decomposed handle without a communicator (admp_set_comm)");
      RecipGeom<T> g;
      const int Kv = 64 * snranks;
      g.K[0] = Kv; g.K[1] = g.K[2] = 32;
      for (int k = 0; k < 9; ++k) { g.hinv[k] = (T)inv[k]; g.Aop[k] = g.Jac[k] = T(0); }
      g.xoff = 64 * srank; g.nloc0 = 64 + kGhost; g.wrap0 = 1 << 30;
      sr = scalar_rows(pos, g, Kv, 64 * srank, 64 * (srank + 1), true);
    }
    { TIMED("tt_pair"); launch_tt_pair<T>(stream, na, nbr, pack_srows(pos, par, 4), bx, tab, dpos, Ed, sr.rows, sr.n, cutoff); }
    read_scalar_energies(Ed, E, 1);
    if (dpos_ && !on_device) {
      HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
  }

  // dE/dc_list (Na,3) of the dispersion PME energy (real + reciprocal + self; admp/disp_pme.py:80-123): the pair part is a
  // per-atom sum of (m + g_p - 1) c_j / r^p, the reciprocal part the mesh potential of every channel at the atom (E_recip is
  // a quadratic form of the channel's coefficients), the self part 2 kp c.  Device pointers; on request only.
  void disp_param_grad(const void* pos_, const double* box, const void* clist_, int pmax, int ns, const double* mS,
                       void* out_) override {
    ARG_CHECK(have_top && have_ewald && have_pairs, "topology, ewald parameters and pairs must be set first");
    ARG_CHECK(pos_ && box && clist_ && out_, "null argument");
    ARG_CHECK(pmax == 6 || pmax == 8 || pmax == 10, "pmax must be 6, 8 or 10");
    const int na = top.na;
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    ensure_mesh();
    RecipGeom<T> g = make_geom(inv);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    const T* pos = reinterpret_cast<const T*>(pos_);
    const T* cl = reinterpret_cast<const T*>(clist_);
    T* out = reinterpret_cast<T*>(out_);
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_WORDS * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(out, 0, 3 * (size_t)na * sizeof(T), stream));
    if (snranks > 1) {      // slab rank (round 4): pair sums over its home rows, the channels' mesh potentials at its home atoms
      ARG_CHECK(spread_uses_bricks(1 << 30, g), "slab-decomposed dispersion PME needs at least 17 local mesh planes and K2, K3 >= 17");
      cls_sites_na = slab_sites_na = -1;
      const ScalarRows sr = scalar_rows(pos, g, K[0], X0, X1, true);
      launch_scalar_pair_pgrad<T>(stream, 0, na, nbr, pos, cl, bx, tab, (T)kappa, pmax, out, cutoff, sr.rows, sr.n);
      const double kq[3] = {-std::pow(kappa, 6) / 12.0, -std::pow(kappa, 8) / 48.0, -std::pow(kappa, 10) / 240.0};
      const int nc = (pmax - 4) / 2;
      const size_t nreal = nreal_local();
      mesh.need(nc * nreal * sizeof(T));
      ensure_bins(std::max(sr.n, 1));
      int rc = launch_bin_bricks<T>(stream, sr.n, (const Site<T>*)nullptr, g, bins, sr.home, bases_d.as<int4>());
      if (rc == 0) rc = launch_spread_scalar<T>(stream, nc, pos, cl, 3, g, bins, mesh.as<T>(), (long)nreal);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("dispersion spread: ") + hipGetErrorString((hipError_t)rc)};
      bins.counters_zero = true;
      for (int c = 0; c < nc; ++c) {
        ensure_gtab(box, inv, vol, 6 + 2 * c);
        convolve(mesh.as<T>() + c * nreal, spec.as<T>(), gtab_cur, E_RECIP);
        launch_gather_value<T>(stream, sr.n, pos, cl, 3, c, g, mesh.as<T>() + c * nreal, 2.0 * kq[c], out, sr.home);
      }
      HIP_TRY(hipStreamSynchronize(stream));
      return;
    }
    launch_scalar_pair_pgrad<T>(stream, 0, na, nbr, pos, cl, bx, tab, (T)kappa, pmax, out, cutoff);
    cls_sites_na = slab_sites_na = -1;
    sites.need(sizeof(Site<T>) * (size_t)na);
    ensure_bins(na);
    const double kp[3] = {-std::pow(kappa, 6) / 12.0, -std::pow(kappa, 8) / 48.0, -std::pow(kappa, 10) / 240.0};
    const int nch = (pmax - 4) / 2;
    for (int c = 0; c < nch; ++c) {
      ensure_gtab(box, inv, vol, 6 + 2 * c);
      launch_scalar_sites<T>(stream, na, pos, cl, 3, c, kp[c], sites.as<Site<T>>(), Ed);
      int rc = launch_spread<T>(stream, na, sites.as<Site<T>>(), 0, g, bins, mesh.as<T>(), nullptr, nullptr, 1, c > 0);
      if (rc != 0) throw Err{ADMP_E_HIP, std::string("launch_spread: ") + hipGetErrorString((hipError_t)rc)};
      convolve(mesh.as<T>(), spec.as<T>(), gtab_cur, E_RECIP);
      launch_gather_value<T>(stream, na, pos, cl, 3, c, g, mesh.as<T>(), 2.0 * kp[c], out);
    }
    HIP_TRY(hipStreamSynchronize(stream));
  }
  // dE/d(a, b, q, c6) (Na,4) of the Tang-Toennies pair term (admp/pairwise.py:94-113).  Device pointers; on request only.
  void tt_param_grad(const void* pos_, const double* box, const void* abqc_, int ns, const double* mS, void* out_) override {
    ARG_CHECK(have_top && have_pairs, "topology and pairs must be set first");
    ARG_CHECK(pos_ && box && abqc_ && out_, "null argument");
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    ScaleTab<T> tab = make_tab(ns, mS, nullptr);
    const T* pos = reinterpret_cast<const T*>(pos_);
    ScalarRows sr{nullptr, top.na, nullptr};
    if (snranks > 1) {      // the rows of the rank's home atoms (per-atom outputs: home rows are this rank's results)
      ARG_CHECK(have_comm, "slab-decomposed handle without a communicator (admp_set_comm)");
      RecipGeom<T> gv;
      const int Kv = 64 * snranks;
      gv.K[0] = Kv; gv.K[1] = gv.K[2] = 32;
      for (int k = 0; k < 9; ++k) { gv.hinv[k] = (T)inv[k]; gv.Aop[k] = gv.Jac[k] = T(0); }
      gv.xoff = 64 * srank; gv.nloc0 = 64 + kGhost; gv.wrap0 = 1 << 30;
      sr = scalar_rows(pos, gv, Kv, 64 * srank, 64 * (srank + 1), true);
    }
    launch_scalar_pair_pgrad<T>(stream, 1, top.na, nbr, pos, reinterpret_cast<const T*>(abqc_), bx,
                                tab, T(0), 0, reinterpret_cast<T*>(out_), cutoff, sr.rows, sr.n);
    HIP_TRY(hipStreamSynchronize(stream));
  }

  // a traced pair kernel (pair_program_build) on the current neighbour table: admp/pairwise.py:67-91 with any kernel
  DevBuf prog_consts;
  void pair_program_eval(int id, const void* pos_, const double* box, const void* par_, int ns, const double* mS, double* E,
                         void* dpos_, int on_device) override {
    ARG_CHECK(have_top && have_pairs, "topology and pairs must be set first");
    ARG_CHECK(id >= 0 && id < (int)programs.size(), "unknown pair program");
    ARG_CHECK(pos_ && box && E && ns >= 1 && mS, "null argument");
    const PairProgram& pg = programs[id];
    ARG_CHECK(pg.n_params == 0 || par_, "atomic parameters missing");
    const int na = top.na;
    double inv[9], vol;
    make_box(box, inv, &vol);
    const T* pos = stage_in(s_pos, pos_, 3 * (size_t)na, on_device);
    const T* par = pg.n_params ? stage_in(s_par, par_, (size_t)pg.n_params * na, on_device) : nullptr;
    grad.need(3 * (size_t)na * sizeof(T));
    T* dpos = (dpos_ && on_device) ? reinterpret_cast<T*>(dpos_) : grad.as<T>();
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* Ed = energies_d.as<double>();
    HIP_TRY(hipMemsetAsync(Ed, 0, E_SLOTS * sizeof(double), stream));
    T consts[34];                                   // box (9), box^-1 (9), mscale per covalent class (16)
    for (int k = 0; k < 9; ++k) { consts[k] = (T)box[k]; consts[9 + k] = (T)inv[k]; }
    for (int nb = 0; nb < 16; ++nb) consts[18 + nb] = (T)mS[((nb - 1) % ns + ns) % ns];      // admp/pme.py:681-683 wrap
    prog_consts.need(sizeof(consts));
    HIP_TRY(hipMemcpyAsync(prog_consts.p, consts, sizeof(consts), hipMemcpyHostToDevice, stream));
    const T* boxd = prog_consts.as<T>();
    const T* mtab = boxd + 18;
    int na_arg = na;
    const int* rowptr = nbr.rowptr; const int* colp = nbr.col; const int* order = nbr.order;
    T* gradp = dpos_ ? dpos : nullptr;
    double* eptr = Ed + E_REAL;
    void* args[] = {&na_arg, &rowptr, &colp, &order, &pos, &par, &boxd, &mtab, &gradp, &eptr};
    const unsigned grid = (unsigned)(((long)na * 8 + 255) / 256);
    {
      TIMED("pair_custom");
      HIP_TRY(hipModuleLaunchKernel(pg.fn, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr));
    }
    double Eh2[E_SLOTS];
    HIP_TRY(hipMemcpyAsync(Eh2, Ed, sizeof(Eh2), hipMemcpyDeviceToHost, stream));
    if (dpos_ && !on_device) HIP_TRY(hipMemcpyAsync(dpos_, dpos, 3 * (size_t)na * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    E[0] = Eh2[E_REAL];
  }

  // raw per-atom sums behind dE/dpol and dE/dtholes (device pointers only); the caller finishes them (admp_amd/pme.py)
  void thole_sums(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                  const double* mS, const double* pS, const void* U, void* sumX, void* sumXw) override {
    ARG_CHECK(lpol, "polarizable handle required");
    ARG_CHECK(sumX && sumXw && U, "null argument");
    HIP_TRY(hipSetDevice(device));
    stage_begin(pos, box, Ql, pol, thole, ns, mS, pS, const_cast<void*>(U));
    // (slab rank: the sums of its home rows -- per-atom outputs like the gradient: the home rows are this rank's results)
    { TIMED("thole_sums"); launch_thole_sums<T>(stream, top.na, nbr, sites.as<Site<T>>(), ev.bx, ev.tab, (T*)sumX, (T*)sumXw,
                                                snranks > 1 ? pair_rows() : nullptr, ev.n_home); }
    ev.active = false;
    HIP_TRY(hipStreamSynchronize(stream));
  }

  // dE/dpScales[k] at the dipoles given (device pointers): class sums of pair_pscale_deriv, folded like dE/dmScales
  void pscale_grad(const void* pos, const double* box, const void* Ql, const void* pol, const void* thole, int ns,
                   const double* mS, const double* pS, const void* U, double* out) override {
    ARG_CHECK(lpol, "polarizable handle required");
    ARG_CHECK(out && U && ns >= 1 && ns <= 16, "bad argument");
    stage_begin(pos, box, Ql, pol, thole, ns, mS, pS, const_cast<void*>(U));
    vir_d.need(V_WORDS * sizeof(double));
    double* cls = vir_d.as<double>();
    HIP_TRY(hipMemsetAsync(cls, 0, 16 * sizeof(double), stream));
    { TIMED("pscale_grad"); launch_pscale_sums<T>(stream, top.na, nbr, sites.as<Site<T>>(), ev.bx, ev.tab, cls,
                                                  snranks > 1 ? pair_rows() : nullptr, ev.n_home); }
    if (snranks > 1) { TIMED("comm_energies"); c_all_reduce(cls, 16, ADMP_T_F64, ADMP_OP_SUM, ADMP_TAG_ENERGIES); }
    ev.active = false;
    double h16[16];
    HIP_TRY(hipMemcpyAsync(h16, cls, sizeof(h16), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int k = 0; k < ns; ++k) out[k] = 0.0;
    for (int nb = 0; nb < 16; ++nb) out[((nb - 1) % ns + ns) % ns] += h16[nb];
  }

  // dE/dmScales[k] = sum over the covalent classes nb that read mScales[k] (index (nb - 1) mod ns, with the reference's
  // negative-index wrap for non-bonded pairs) of the class sums produced by launch_mscale_sums
  void mscale_grad(int kind, const void* pos_, const double* box, const void* par_, int pmax, int ns, double* out,
                   int on_device) override {
    ARG_CHECK(have_top && have_pairs, "topology and pairs must be set first");
    ARG_CHECK(pos_ && box && par_ && out && ns >= 1 && ns <= 16, "bad argument");
    ARG_CHECK(kind >= 0 && kind <= 2, "kind must be 0 (multipolar PME), 1 (dispersion) or 2 (Tang-Toennies)");
    ARG_CHECK(snranks == 1 || on_device, "a slab-decomposed handle takes device pointers");
    const int na = top.na;
    HIP_TRY(hipSetDevice(device));
    double inv[9], vol;
    Box<T> bx = make_box(box, inv, &vol);
    const T* pos = stage_in(s_pos, pos_, 3 * (size_t)na, on_device);
    const size_t npar = kind == 0 ? 9 : (kind == 1 ? 3 : 4);
    const T* par = stage_in(kind == 0 ? s_Q : s_par, par_, npar * (size_t)na, on_device);
    energies_d.need(2 * E_WORDS * sizeof(double));
    ehalf = 0; other_clean = false;
    double* cls = energies_d.as<double>();          // 16 class sums
    HIP_TRY(hipMemsetAsync(cls, 0, 16 * sizeof(double), stream));
    if (kind == 0) {
      ARG_CHECK(have_ewald, "ewald parameters must be set first");
      cls_sites_na = slab_sites_na = -1;   // other rows than an electrostatics evaluation's
      sites.need(sizeof(Site<T>) * (size_t)na);
      RecipGeom<T> g = make_geom(inv);
      launch_prepare_sites<T>(stream, top, pos, par, nullptr, nullptr, nullptr, bx, sites.as<Site<T>>(), nullptr, g, nullptr);
    }
    ScalarRows sr{nullptr, na, nullptr};
    if (snranks > 1) {      // the class sums of the rank's home rows, added over the ranks
      ARG_CHECK(have_comm, "slab-decomposed handle without a communicator (admp_set_comm)");
      RecipGeom<T> gv;
      if (have_ewald) { update_slab(); gv = make_geom(inv); sr = scalar_rows(pos, gv, K[0], X0, X1, true); }
      else {                // a pair potential has no mesh: x-slabs of a virtual one, as in tt()
        const int Kv = 64 * snranks;
        gv.K[0] = Kv; gv.K[1] = gv.K[2] = 32;
        for (int k = 0; k < 9; ++k) { gv.hinv[k] = (T)inv[k]; gv.Aop[k] = gv.Jac[k] = T(0); }
        gv.xoff = 64 * srank; gv.nloc0 = 64 + kGhost; gv.wrap0 = 1 << 30;
        sr = scalar_rows(pos, gv, Kv, 64 * srank, 64 * (srank + 1), true);
      }
    }
    { TIMED("mscale_grad"); launch_mscale_sums<T>(stream, kind, na, nbr, sites.as<Site<T>>(), pos, par, bx, pmax, cls, cutoff,
                                                  sr.rows, sr.n); }
    if (snranks > 1) { TIMED("comm_energies"); c_all_reduce(cls, 16, ADMP_T_F64, ADMP_OP_SUM, ADMP_TAG_ENERGIES); }
    double h16[16];
    HIP_TRY(hipMemcpyAsync(h16, cls, sizeof(h16), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int k = 0; k < ns; ++k) out[k] = 0.0;
    for (int nb = 0; nb < 16; ++nb) out[((nb - 1) % ns + ns) % ns] += h16[nb];
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
    (void)hipGetLastError();                 // drop anything stale from other users of this thread
    h->eng->adopt_shared();
    f(*h->eng);
    // a bad launch configuration (too much LDS, too many registers after a flag change, ...) is reported by neither the
    // launch statement nor hipStreamSynchronize: every <<<>>> of this call is covered by one check here
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) throw Err{ADMP_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(le)};
    return ADMP_OK;
  } catch (const Err& e) {
    h->err = e.msg;
    h->eng->U_src = h->eng->U_src_now = nullptr;      // a failed call leaves no one-shot state behind
    h->eng->after_error();
    return e.code;
  } catch (const std::exception& e) {
    h->err = e.what();
    h->eng->U_src = h->eng->U_src_now = nullptr;
    h->eng->after_error();
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
    { std::lock_guard<std::mutex> g(EngineBase::live_mu()); EngineBase::live().insert(h->eng.get()); }
    *out = h;
    return ADMP_OK;
  } catch (const Err& e) {
    g_create_err = e.msg;
    return e.code;
  }
}

int admp_destroy(admp_handle* h) {
  if (!h) return ADMP_E_ARG;
  if (h->eng) {
    (void)hipSetDevice(h->eng->device);
    (void)hipStreamSynchronize(h->eng->stream);
    { std::lock_guard<std::mutex> g(EngineBase::live_mu()); EngineBase::live().erase(h->eng.get()); }
  }
  delete h;
  return ADMP_OK;
}

const char* admp_last_error(const admp_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int admp_use_default_stream(admp_handle* h) {
  return guarded(h, [&](EngineBase& e) {
    HIP_TRY(hipStreamSynchronize(e.stream));
    if (e.own_stream && e.stream) HIP_TRY(hipStreamDestroy(e.stream));
    e.stream = nullptr;          // the legacy default stream
    e.own_stream = false;
  });
}

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

int admp_share_neighbors(admp_handle* h, admp_handle* lender) {
  return guarded(h, [&](EngineBase& e) {
    ARG_CHECK(!lender || lender->eng, "bad lender handle");
    e.share_neighbors(lender ? lender->eng.get() : nullptr);
  });
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

int admp_pme_energy_fixed_dipoles(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                         const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                         double* E_out, void* dE_dpos, void* dE_dU, void* dE_dQlocal) {
  return guarded(h, [&](EngineBase& e) {
    e.pme_at_U(positions, box, Q_local, pol, tholes, n_scales, mScales, pScales, U, E_out, dE_dpos, dE_dU, dE_dQlocal);
  });
}

int admp_pme_box_grad(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                      const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                      double* E_out, double* dE_dbox) {
  return guarded(h, [&](EngineBase& e) {
    e.pme_box_grad(positions, box, Q_local, pol, tholes, n_scales, mScales, pScales, U, E_out, dE_dbox);
  });
}
int admp_disp_box_grad(admp_handle* h, const void* positions, const double* box, const void* c_list, int pmax, int n_scales,
                       const double* mScales, double* E_out, double* dE_dbox) {
  return guarded(h, [&](EngineBase& e) { e.disp_box_grad(positions, box, c_list, pmax, n_scales, mScales, E_out, dE_dbox); });
}
int admp_tt_box_grad(admp_handle* h, const void* positions, const double* box, const void* abqc, int n_scales,
                     const double* mScales, double* E_out, double* dE_dbox) {
  return guarded(h, [&](EngineBase& e) { e.tt_box_grad(positions, box, abqc, n_scales, mScales, E_out, dE_dbox); });
}

int admp_pair_program_build(admp_handle* h, const char* hip_source, int n_params, int* program_id) {
  return guarded(h, [&](EngineBase& e) {
    ARG_CHECK(program_id, "null");
    *program_id = e.pair_program_build(hip_source, n_params);
  });
}
int admp_pair_program_energy_grad(admp_handle* h, int program_id, const void* positions, const double* box,
                                  const void* params, int n_scales, const double* mScales, double* E_out, void* dE_dpos,
                                  int on_device) {
  return guarded(h, [&](EngineBase& e) {
    e.pair_program_eval(program_id, positions, box, params, n_scales, mScales, E_out, dE_dpos, on_device);
  });
}

int admp_local_frames(admp_handle* h, const void* positions, const double* box, void* frames_out) {
  return guarded(h, [&](EngineBase& e) { e.local_frames(positions, box, frames_out); });
}

int admp_set_option(admp_handle* h, int option, int value) {
  return guarded(h, [&](EngineBase& e) {
    switch (option) {
      case ADMP_OPT_REFERENCE_KPOINTS: e.ref_korder = value ? 1 : 0; break;
      case ADMP_OPT_KEEP_POL_SITES: e.keep_pol_sites = value ? 1 : 0; break;
      case ADMP_OPT_SIDE_STREAM: e.side_stream_on = value ? 1 : 0; break;
      default: throw Err{ADMP_E_ARG, "unknown option"};
    }
  });
}

int admp_disp_energy_grad(admp_handle* h, const void* positions, const double* box, const void* c_list, int pmax,
                          int n_scales, const double* mScales, double* E_out, void* dE_dpos, int on_device) {
  return guarded(h, [&](EngineBase& e) { e.disp(positions, box, c_list, pmax, n_scales, mScales, E_out, dE_dpos, on_device); });
}

int admp_prune_pairs(admp_handle* h, const void* positions, const double* box, double rc) {
  return guarded(h, [&](EngineBase& e) { e.prune_pairs(positions, box, rc); });
}

int admp_disp_set_types(admp_handle* h, int n_types, const void* type_of_atom, const double* coefficients) {
  return guarded(h, [&](EngineBase& e) { e.disp_set_types(n_types, type_of_atom, coefficients); });
}

int admp_tt_energy_grad(admp_handle* h, const void* positions, const double* box, const void* abqc, int n_scales,
                        const double* mScales, double* E_out, void* dE_dpos, int on_device) {
  return guarded(h, [&](EngineBase& e) { e.tt(positions, box, abqc, n_scales, mScales, E_out, dE_dpos, on_device); });
}

int admp_mscale_grad(admp_handle* h, int kind, const void* positions, const double* box, const void* params, int pmax,
                     int n_scales, double* dE_dmScales, int on_device) {
  return guarded(h, [&](EngineBase& e) { e.mscale_grad(kind, positions, box, params, pmax, n_scales, dE_dmScales, on_device); });
}

int admp_disp_param_grad(admp_handle* h, const void* positions, const double* box, const void* c_list, int pmax, int n_scales,
                         const double* mScales, void* dE_dc) {
  return guarded(h, [&](EngineBase& e) { e.disp_param_grad(positions, box, c_list, pmax, n_scales, mScales, dE_dc); });
}
int admp_tt_param_grad(admp_handle* h, const void* positions, const double* box, const void* abqc, int n_scales,
                       const double* mScales, void* dE_dabqc) {
  return guarded(h, [&](EngineBase& e) { e.tt_param_grad(positions, box, abqc, n_scales, mScales, dE_dabqc); });
}

int admp_pscale_grad(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                     const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                     double* dE_dpScales) {
  return guarded(h, [&](EngineBase& e) {
    e.pscale_grad(positions, box, Q_local, pol, tholes, n_scales, mScales, pScales, U, dE_dpScales);
  });
}

int admp_thole_sums(admp_handle* h, const void* positions, const double* box, const void* Q_local, const void* pol,
                    const void* tholes, int n_scales, const double* mScales, const double* pScales, const void* U,
                    void* sumX, void* sumXw) {
  return guarded(h, [&](EngineBase& e) { e.thole_sums(positions, box, Q_local, pol, tholes, n_scales, mScales, pScales, U, sumX, sumXw); });
}

int admp_md_bonded(admp_handle* h, const void* positions, const double* box, int n_bonds, const int32_t* bond_idx,
                   const void* bond_par, int n_angles, const int32_t* angle_idx, const void* angle_par, double* E_dev,
                   void* grad_inout) {
  return guarded(h, [&](EngineBase& e) {
    e.md_bonded(positions, box, n_bonds, bond_idx, bond_par, n_angles, angle_idx, angle_par, E_dev, grad_inout);
  });
}
int admp_md_kick_drift(admp_handle* h, int n_atoms, void* positions, void* velocities, const void* grad, const void* inv_mass,
                       double half_dt_acc, double dt, double* ekin_dev) {
  return guarded(h, [&](EngineBase& e) { e.md_kick_drift(n_atoms, positions, velocities, grad, inv_mass, half_dt_acc, dt, ekin_dev); });
}

int admp_neighbor_count(admp_handle* h, int n_atoms, const void* positions, const double* box, double rc, int64_t* n_pairs) {
  return guarded(h, [&](EngineBase& e) { e.nbr_count(n_atoms, positions, box, rc, n_pairs); });
}
int admp_neighbor_fill(admp_handle* h, int32_t* pairs_out) {
  return guarded(h, [&](EngineBase& e) { e.nbr_fill(pairs_out); });
}

int admp_set_pairs_from_positions(admp_handle* h, const void* positions, const double* box, double rc) {
  return guarded(h, [&](EngineBase& e) { e.nbr_table(positions, box, rc); });
}

int admp_slab_configure(admp_handle* h, int rank, int nranks) {
  return guarded(h, [&](EngineBase& e) {
    ARG_CHECK(nranks >= 1 && nranks <= kSlabMaxRanks && rank >= 0 && rank < nranks, "bad rank / nranks (at most 28 ranks)");
    e.srank = rank; e.snranks = nranks;
  });
}
int admp_slab_info(admp_handle* h, int64_t* out11) {
  return guarded(h, [&](EngineBase& e) { ARG_CHECK(out11, "null"); e.slab_info(out11); });
}
int admp_set_comm(admp_handle* h, const admp_comm* comm) {
  return guarded(h, [&](EngineBase& e) {
    if (!comm) { e.have_comm = false; e.comm = admp_comm{}; return; }
    ARG_CHECK(comm->all_reduce && comm->all_to_all_v && comm->shift, "communicator with a missing callback");
    e.comm = *comm;
    e.have_comm = true;
  });
}
int admp_set_comm_rccl(admp_handle* h, admp_rccl* c) {
  return guarded(h, [&](EngineBase& e) {
    if (!c) { e.rccl = nullptr; if (!e.comm.all_reduce) e.have_comm = false; return; }
    ARG_CHECK(rccl_device(c) == e.device, "the communicator was created on another device");
    ARG_CHECK(rccl_nranks(c) <= kSlabMaxRanks, "at most 28 slab ranks");
    e.rccl = c;
    e.srank = rccl_rank(c); e.snranks = rccl_nranks(c);
    e.have_comm = true;
  });
}
int admp_set_dipole_source(admp_handle* h, const void* U_init) {
  return guarded(h, [&](EngineBase& e) { e.U_src = U_init; });
}
int admp_set_cutoff(admp_handle* h, double rc) {
  return guarded(h, [&](EngineBase& e) {
    ARG_CHECK(rc >= 0.0, "negative cutoff");
    e.cutoff = rc;
  });
}
int admp_scf_stats(admp_handle* h, int64_t* out8, int reset) {
  return guarded(h, [&](EngineBase& e) {
    ARG_CHECK(out8, "null");
    for (int k = 0; k < 8; ++k) { out8[k] = e.scf_stats[k]; if (reset) e.scf_stats[k] = 0; }
  });
}
int admp_slab_home(admp_handle* h, int32_t* home_out, int* n_home, int* n_import) {
  return guarded(h, [&](EngineBase& e) { e.slab_home(home_out, n_home, n_import); });
}

int admp_profile_enable(admp_handle* h, int on) {
  return guarded(h, [&](EngineBase& e) { e.prof.collect(e.stream); e.prof.on = on != 0; });
}
int admp_profile_filter(admp_handle* h, const char* label) {
  return guarded(h, [&](EngineBase& e) { e.prof.only = label ? label : ""; });
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
