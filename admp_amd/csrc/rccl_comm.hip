// Native communicator of a slab-decomposed handle: the three collectives of include/admp_hip.h's admp_comm issued straight to
// RCCL on the handle's stream -- no host language in the step (round 3 went C -> Python -> torch.distributed per collective:
// ten or more interpreter hops per polarizable call).
//
//   all_reduce    ncclAllReduce in place (SUM; MAX of the SCF residual word as ncclUint64: the bit patterns of non-negative
//                 doubles order like the numbers, and a NaN pattern wins the maximum instead of vanishing)
//   all_to_all_v  ONE group of ncclSend / ncclRecv to every peer with a non-empty segment (RCCL runs the N-1 exchanges of a
//                 GPU concurrently, one per xGMI link); the rank's own segment is a device copy on the same stream
//   shift         a grouped send to one ring neighbour + receive from the other
//
// RCCL is bound at run time (dlopen / dlsym): a single-GPU user never loads the 0.5 GB library, and inside a PyTorch process
// the copy PyTorch has already loaded (same soname, librccl.so.1) is the one used, so a process holds ONE RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/admp_hip.h"
#include "rccl_comm.h"

namespace {

struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

std::mutex g_mu;
Api g_api;
std::string g_err;

bool load_api() {
  if (g_api.lib) return true;
  // the copy already in the process first (PyTorch's), then the ROCm installation's
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { g_err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
  Api a;
  a.lib = h;
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(h, name);
    if (!p) g_err = std::string("librccl lacks ") + name;
    return p;
  };
#define BIND(field, name) \
  if (!(a.field = reinterpret_cast<decltype(a.field)>(sym(name)))) return false
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(CommAbort, "ncclCommAbort");
  BIND(AllReduce, "ncclAllReduce");
  BIND(Send, "ncclSend");
  BIND(Recv, "ncclRecv");
  BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd");
  BIND(GetErrorString, "ncclGetErrorString");
  BIND(GetVersion, "ncclGetVersion");
#undef BIND
  g_api = a;
  return true;
}

int fail(const char* what, ncclResult_t r) {
  g_err = std::string(what) + ": " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "RCCL error") + " (" + std::to_string((int)r) + ")";
  return 1;
}
#define NCCL_TRY(call, what)                        \
  do {                                              \
    ncclResult_t r_ = (call);                       \
    if (r_ != ncclSuccess) return fail(what, r_);   \
  } while (0)

size_t elem_bytes(int dtype) { return dtype == ADMP_T_F64 ? 8 : 4; }
ncclDataType_t nccl_type(int dtype) { return dtype == ADMP_T_F64 ? ncclDouble : (dtype == ADMP_T_F32 ? ncclFloat : ncclInt32); }

}  // namespace

struct admp_rccl {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, nranks = 1;
  bool self_sendrecv = false;      // tests on one GPU: the rank's own segment goes through ncclSend / ncclRecv as well
  int64_t bytes[ADMP_RCCL_NTAGS] = {0}, calls[ADMP_RCCL_NTAGS] = {0};
  void count(int tag, int64_t nbytes) {
    const int t = tag >= 0 && tag < ADMP_RCCL_NTAGS ? tag : 0;
    bytes[t] += nbytes; calls[t] += 1;
  }
};

namespace admp {

int rccl_all_reduce(admp_rccl* c, hipStream_t st, void* buf, int64_t count, int dtype, int op, int tag) {
  std::lock_guard<std::mutex> g(g_mu);
  c->count(tag, 2 * (int64_t)(c->nranks - 1) * count * (int64_t)elem_bytes(dtype) / c->nranks);      // ring: 2 (N-1)/N of the buffer
  ncclDataType_t t = nccl_type(dtype);
  ncclRedOp_t o = ncclSum;
  if (op == ADMP_OP_MAX) {
    o = ncclMax;
    if (dtype == ADMP_T_F64) t = ncclUint64;          // bit patterns of non-negative doubles (see the header comment)
  }
  NCCL_TRY(g_api.AllReduce(buf, buf, (size_t)count, t, o, c->comm, st), "ncclAllReduce");
  return 0;
}

int rccl_all_to_all_v(admp_rccl* c, hipStream_t st, const void* send, const int64_t* sc, void* recv, const int64_t* rc, int dtype,
                      int tag) {
  std::lock_guard<std::mutex> g(g_mu);
  const size_t w = elem_bytes(dtype);
  const ncclDataType_t t = nccl_type(dtype);
  const char* s = static_cast<const char*>(send);
  char* r = static_cast<char*>(recv);
  int64_t so = 0, ro = 0, sent = 0;
  bool grouped = false;
  for (int p = 0; p < c->nranks; ++p) {
    if (p == c->rank && !c->self_sendrecv) {
      if (sc[p] != rc[p]) { g_err = "all_to_all_v: a rank's own send and receive counts differ"; return 1; }
      if (sc[p] > 0 && hipMemcpyAsync(r + ro * w, s + so * w, (size_t)sc[p] * w, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        g_err = "all_to_all_v: device copy of the rank's own segment failed";
        return 1;
      }
    } else {
      if (!grouped && (sc[p] > 0 || rc[p] > 0)) { NCCL_TRY(g_api.GroupStart(), "ncclGroupStart"); grouped = true; }
      if (sc[p] > 0) { NCCL_TRY(g_api.Send(s + so * w, (size_t)sc[p], t, p, c->comm, st), "ncclSend"); sent += sc[p]; }
      if (rc[p] > 0) NCCL_TRY(g_api.Recv(r + ro * w, (size_t)rc[p], t, p, c->comm, st), "ncclRecv");
    }
    so += sc[p]; ro += rc[p];
  }
  if (grouped) NCCL_TRY(g_api.GroupEnd(), "ncclGroupEnd");
  c->count(tag, sent * (int64_t)w);
  return 0;
}

int rccl_shift(admp_rccl* c, hipStream_t st, const void* send, void* recv, int64_t count, int dtype, int to_next, int tag) {
  std::lock_guard<std::mutex> g(g_mu);
  const size_t w = elem_bytes(dtype);
  if (count <= 0) return 0;
  if (c->nranks == 1 && !c->self_sendrecv) {
    if (send != recv && hipMemcpyAsync(recv, send, (size_t)count * w, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      g_err = "shift: device copy failed";
      return 1;
    }
    return 0;
  }
  const int n = c->nranks;
  const int dst = (c->rank + (to_next ? 1 : n - 1)) % n, src = (c->rank + (to_next ? n - 1 : 1)) % n;
  const ncclDataType_t t = nccl_type(dtype);
  NCCL_TRY(g_api.GroupStart(), "ncclGroupStart");
  NCCL_TRY(g_api.Send(send, (size_t)count, t, dst, c->comm, st), "ncclSend");
  NCCL_TRY(g_api.Recv(recv, (size_t)count, t, src, c->comm, st), "ncclRecv");
  NCCL_TRY(g_api.GroupEnd(), "ncclGroupEnd");
  c->count(tag, count * (int64_t)w);
  return 0;
}

int rccl_rank(const admp_rccl* c) { return c->rank; }
int rccl_nranks(const admp_rccl* c) { return c->nranks; }
int rccl_device(const admp_rccl* c) { return c->device; }
std::string rccl_error() { std::lock_guard<std::mutex> g(g_mu); return g_err; }

}  // namespace admp

extern "C" {

const char* admp_rccl_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> g(g_mu);
  copy = g_err;
  return copy.c_str();
}

int admp_rccl_unique_id(void* out128) {
  std::lock_guard<std::mutex> g(g_mu);
  if (!out128) { g_err = "null"; return ADMP_E_ARG; }
  if (!load_api()) return ADMP_E_COMM;
  static_assert(sizeof(ncclUniqueId) == ADMP_RCCL_ID_BYTES, "unique id size");
  ncclUniqueId id;
  if (g_api.GetUniqueId(&id) != ncclSuccess) { g_err = "ncclGetUniqueId failed"; return ADMP_E_COMM; }
  std::memcpy(out128, &id, sizeof(id));
  return ADMP_OK;
}

int admp_rccl_create(admp_rccl** out, int device, const void* id128, int rank, int nranks) {
  std::lock_guard<std::mutex> g(g_mu);
  if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { g_err = "bad argument"; return ADMP_E_ARG; }
  if (!load_api()) return ADMP_E_COMM;
  if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return ADMP_E_HIP; }
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  admp_rccl* c = new admp_rccl();
  c->device = device; c->rank = rank; c->nranks = nranks;
  const ncclResult_t r = g_api.CommInitRank(&c->comm, nranks, id, rank);      // blocks until every rank has joined
  if (r != ncclSuccess) { fail("ncclCommInitRank", r); delete c; return ADMP_E_COMM; }
  const char* e = getenv("ADMP_RCCL_SELF_SENDRECV");
  c->self_sendrecv = e && atoi(e) != 0;
  *out = c;
  return ADMP_OK;
}

int admp_rccl_destroy(admp_rccl* c) {
  std::lock_guard<std::mutex> g(g_mu);
  if (!c) return ADMP_E_ARG;
  if (c->comm && g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
  delete c;
  return ADMP_OK;
}

int admp_rccl_abort(admp_rccl* c) {
  std::lock_guard<std::mutex> g(g_mu);
  if (!c) return ADMP_E_ARG;
  if (c->comm && g_api.CommAbort) (void)g_api.CommAbort(c->comm);
  c->comm = nullptr;
  return ADMP_OK;
}

int admp_rccl_stats(admp_rccl* c, int64_t* bytes_out, int64_t* calls_out, int reset) {
  std::lock_guard<std::mutex> g(g_mu);
  if (!c) return ADMP_E_ARG;
  for (int t = 0; t < ADMP_RCCL_NTAGS; ++t) {
    if (bytes_out) bytes_out[t] = c->bytes[t];
    if (calls_out) calls_out[t] = c->calls[t];
    if (reset) c->bytes[t] = c->calls[t] = 0;
  }
  return ADMP_OK;
}

int admp_rccl_all_reduce(admp_rccl* c, void* buf, int64_t count, int dtype, int op, void* hip_stream) {
  if (!c || !c->comm || !buf || count < 0) return ADMP_E_ARG;
  return admp::rccl_all_reduce(c, (hipStream_t)hip_stream, buf, count, dtype, op, 7) == 0 ? ADMP_OK : ADMP_E_COMM;
}

int admp_rccl_all_to_all_v(admp_rccl* c, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                           int dtype, void* hip_stream) {
  if (!c || !c->comm || !send_counts || !recv_counts) return ADMP_E_ARG;
  return admp::rccl_all_to_all_v(c, (hipStream_t)hip_stream, send, send_counts, recv, recv_counts, dtype, 0) == 0 ? ADMP_OK : ADMP_E_COMM;
}

int admp_rccl_shift(admp_rccl* c, const void* send, void* recv, int64_t count, int dtype, int to_next, void* hip_stream) {
  if (!c || !c->comm || count < 0) return ADMP_E_ARG;
  return admp::rccl_shift(c, (hipStream_t)hip_stream, send, recv, count, dtype, to_next, 0) == 0 ? ADMP_OK : ADMP_E_COMM;
}

int admp_rccl_version(int* version) {
  std::lock_guard<std::mutex> g(g_mu);
  if (!version) return ADMP_E_ARG;
  if (!load_api()) return ADMP_E_COMM;
  return g_api.GetVersion(version) == ncclSuccess ? ADMP_OK : ADMP_E_COMM;
}

}  // extern "C"
