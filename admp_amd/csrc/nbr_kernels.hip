// Neighbour-table compiler: turns the caller's (n_rows, 2) pair list into the i-grouped table the pair
// kernels walk.  Replaces the per-call `pairs[pairs[:,0] < pairs[:,1]]` filter, the covalent_map lookup
// `nbonds = covalent_map[pairs[:,0], pairs[:,1]]` and the per-pair gathers of the reference
// (admp/pme.py:671-693, admp/pairwise.py:68-75).
//   1. count: degree of each atom over valid rows (0 <= i < j < na)            -- int atomics
//   2. exclusive scan of the degrees (hipcub)                                   -- rowptr
//   3. fill: both directions of each pair, nbonds from the CSR covalent map packed in bits 28..31
//   4. sort each row by partner index (short rows: in-thread insertion sort)    -- deterministic order
#include <hipcub/hipcub.hpp>

#include "launch.h"

namespace admp {

__global__ void k_nbr_count(int64_t n_rows, const int* __restrict__ pairs, int na, int* __restrict__ deg,
                            unsigned long long* n_valid) {
  unsigned long long local = 0;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_rows; p += (int64_t)gridDim.x * blockDim.x) {
    int i = pairs[2 * p], j = pairs[2 * p + 1];
    if (i >= 0 && i < j && j < na) {
      atomicAdd(&deg[i], 1);
      atomicAdd(&deg[j], 1);
      ++local;
    }
  }
  if (local) atomicAdd(n_valid, local);
}

__device__ __forceinline__ int lookup_nbonds(const Topology& top, int i, int j) {
  if (!top.excl_ptr) return 0;
  for (int k = top.excl_ptr[i]; k < top.excl_ptr[i + 1]; ++k)
    if (top.excl_col[k] == j) return top.excl_nb[k] & 15;
  return 0;
}

__global__ void k_nbr_fill(int64_t n_rows, const int* __restrict__ pairs, Topology top, int* __restrict__ cursor,
                           int* __restrict__ col) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_rows; p += (int64_t)gridDim.x * blockDim.x) {
    int i = pairs[2 * p], j = pairs[2 * p + 1];
    if (i >= 0 && i < j && j < top.na) {
      int nb = lookup_nbonds(top, i, j) << 28;
      col[atomicAdd(&cursor[i], 1)] = j | nb;
      col[atomicAdd(&cursor[j], 1)] = i | nb;
    }
  }
}

__global__ void k_nbr_sort(int na, const int* __restrict__ rowptr, int* __restrict__ col) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= na) return;
  int b = rowptr[r], e = rowptr[r + 1];
  for (int a = b + 1; a < e; ++a) {
    int v = col[a], key = v & kColMask, k = a - 1;
    while (k >= b && (col[k] & kColMask) > key) { col[k + 1] = col[k]; --k; }
    col[k + 1] = v;
  }
}

// Class partition of the rows (NbrTable::cls): entries whose partner is charge-only get kColMono and move behind the
// others, both runs keeping their order.  8 lanes per row, two sweeps (count, then place through ballot prefix counts).
constexpr int kPartLanes = 8;
__global__ __launch_bounds__(256) void k_class_partition(int na, const int* __restrict__ rowptr, const int* __restrict__ cin,
                                                         int* __restrict__ cout, const int* __restrict__ cls) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(t / kPartLanes), sub = (int)(t % kPartLanes);
  const int shift = (threadIdx.x & 63) & ~(kPartLanes - 1);
  const unsigned below = (1u << sub) - 1u, gmask = (1u << kPartLanes) - 1u;
  const int b = row < na ? rowptr[row] : 0, e = row < na ? rowptr[row + 1] : 0;
  int nfull = 0;
  for (int k0 = b; k0 < e; k0 += kPartLanes) {
    const int k = k0 + sub;
    const bool full = k < e && cls[cin[k] & kColMask] == 0;
    nfull += __popc((unsigned)(__ballot(full) >> shift) & gmask);
  }
  int pf = b, pm = b + nfull;
  for (int k0 = b; k0 < e; k0 += kPartLanes) {
    const int k = k0 + sub;
    const int c = k < e ? cin[k] : 0;
    const bool mono = k < e && cls[c & kColMask] != 0, full = k < e && !mono;
    const unsigned mf = (unsigned)(__ballot(full) >> shift) & gmask, mm = (unsigned)(__ballot(mono) >> shift) & gmask;
    if (full) cout[pf + __popc(mf & below)] = c & ~kColMono;
    if (mono) cout[pm + __popc(mm & below)] = c | kColMono;
    pf += __popc(mf); pm += __popc(mm);
  }
}
int launch_class_partition(hipStream_t st, int na, NbrTable& nb) {
  if (!nb.cls || !nb.col || na <= 0 || nb.cap <= 0) return 0;
  int* out = nullptr;
  hipError_t e = hipMalloc(&out, sizeof(int) * (size_t)nb.cap);
  if (e != hipSuccess) return (int)e;
  k_class_partition<<<(unsigned)(((long)na * kPartLanes + 255) / 256), 256, 0, st>>>(na, nb.rowptr, nb.col, out, nb.cls);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipFree(out); return (int)e; }
  (void)hipFree(nb.col);
  nb.col = out;
  return 0;
}

#define NB_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {
struct DevTmp {   // freed on every return path
  void* p = nullptr;
  ~DevTmp() { if (p) (void)hipFree(p); }
};
}  // namespace

int build_neighbour_table(hipStream_t st, const Topology& top, int64_t n_rows, const int* pairs_dev, NbrTable& nb,
                          void** scratch, size_t* scratch_bytes) {
  const int na = top.na;
  if (!nb.rowptr) NB_CHECK(hipMalloc(&nb.rowptr, sizeof(int) * (na + 1)));
  // na + 1 degrees (reused as the fill cursor), padded to an even count, then one 64-bit pair counter
  const size_t ndeg = (size_t)((na + 2) & ~1);
  DevTmp degbuf;
  NB_CHECK(hipMalloc(&degbuf.p, sizeof(int) * ndeg + sizeof(unsigned long long)));
  int* deg = (int*)degbuf.p;
  unsigned long long* n_valid = (unsigned long long*)(deg + ndeg);
  NB_CHECK(hipMemsetAsync(deg, 0, sizeof(int) * ndeg + sizeof(unsigned long long), st));
  int blocks = (int)((n_rows + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (n_rows > 0) { k_nbr_count<<<blocks, 256, 0, st>>>(n_rows, pairs_dev, na, deg, n_valid); NB_CHECK(hipGetLastError()); }
  size_t need = 0;
  NB_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, deg, nb.rowptr, na + 1, st));
  if (need > *scratch_bytes) {
    if (*scratch) NB_CHECK(hipFree(*scratch));
    *scratch = nullptr; *scratch_bytes = 0;
    NB_CHECK(hipMalloc(scratch, need));
    *scratch_bytes = need;
  }
  NB_CHECK(hipcub::DeviceScan::ExclusiveSum(*scratch, need, deg, nb.rowptr, na + 1, st));
  unsigned long long nv = 0;
  NB_CHECK(hipMemcpyAsync(&nv, n_valid, sizeof(nv), hipMemcpyDeviceToHost, st));
  NB_CHECK(hipStreamSynchronize(st));
  nb.n_half = (int64_t)nv;
  if (2 * nb.n_half > nb.cap) {
    if (nb.col) NB_CHECK(hipFree(nb.col));
    nb.col = nullptr; nb.cap = 0;
    NB_CHECK(hipMalloc(&nb.col, sizeof(int) * (2 * nb.n_half + 1024)));
    nb.cap = 2 * nb.n_half + 1024;
  }
  NB_CHECK(hipMemcpyAsync(deg, nb.rowptr, sizeof(int) * (na + 1), hipMemcpyDeviceToDevice, st));
  if (n_rows > 0) {
    k_nbr_fill<<<blocks, 256, 0, st>>>(n_rows, pairs_dev, top, deg, nb.col);
    k_nbr_sort<<<(na + 127) / 128, 128, 0, st>>>(na, nb.rowptr, nb.col);
    NB_CHECK(hipGetLastError());
  }
  NB_CHECK(hipStreamSynchronize(st));
  return 0;
}

// sub-table of the polarizable-polarizable entries: mode 0 counts per atom row (cnt[i]; 0 for non-polarizable atoms),
// mode 1 copies the entries to icol at irow[i] (row order preserved).  8 lanes per row.
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_ind_table(int na, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const Site<T>* __restrict__ sites, int* __restrict__ cnt,
                                                          const int* __restrict__ irow, int* __restrict__ icol) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(t >> 3), sub = (int)(t & 7);
  const bool live = row < na && sites[row].p6 > T(0);
  const int beg = live ? rowptr[row] : 0, end = live ? rowptr[row + 1] : 0;
  int base = (MODE == 1 && live) ? irow[row] : 0, total = 0;
  for (int k0 = beg; k0 < end; k0 += 8) {      // the 8 lanes of a row share the trip count; the ballot is read per group
    const int k = k0 + sub;
    const int c = k < end ? col[k] : 0;
    const bool keep = k < end && sites[c & kColMask].p6 > T(0);
    // rank of this lane's entry among the kept ones of its 8-lane group
    const unsigned long long m = __ballot(keep);
    const int g0 = (threadIdx.x & 63) & ~7;
    const unsigned grp = (unsigned)((m >> g0) & 0xffull);
    if (MODE == 1 && keep) icol[base + __popc(grp & ((1u << sub) - 1u))] = c;
    base += __popc(grp);
    total += __popc(grp);
  }
  if (MODE == 0 && row < na && sub == 0) cnt[row] = total;
}

// it <- the polarizable-polarizable entries of nb (rows keyed by atom).  One host synchronisation (the entry count).
template <class T>
int build_ind_table(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, IndTable& it, void** scratch,
                    size_t* scratch_bytes) {
  if (!it.rowptr) NB_CHECK(hipMalloc(&it.rowptr, sizeof(int) * (na + 1)));
  DevTmp cntbuf;
  NB_CHECK(hipMalloc(&cntbuf.p, sizeof(int) * (na + 1)));
  int* cnt = (int*)cntbuf.p;
  NB_CHECK(hipMemsetAsync(cnt + na, 0, sizeof(int), st));
  const unsigned grid = (unsigned)(((long)na * 8 + 255) / 256);
  k_ind_table<T, 0><<<grid, 256, 0, st>>>(na, nb.rowptr, nb.col, sites, cnt, nullptr, nullptr);
  NB_CHECK(hipGetLastError());
  size_t need = 0;
  NB_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, cnt, it.rowptr, na + 1, st));
  if (need > *scratch_bytes) {
    if (*scratch) NB_CHECK(hipFree(*scratch));
    *scratch = nullptr; *scratch_bytes = 0;
    NB_CHECK(hipMalloc(scratch, need));
    *scratch_bytes = need;
  }
  NB_CHECK(hipcub::DeviceScan::ExclusiveSum(*scratch, need, cnt, it.rowptr, na + 1, st));
  int total = 0;
  NB_CHECK(hipMemcpyAsync(&total, it.rowptr + na, sizeof(int), hipMemcpyDeviceToHost, st));
  NB_CHECK(hipStreamSynchronize(st));
  it.n = total;
  if (total > it.cap) {
    if (it.col) NB_CHECK(hipFree(it.col));
    it.col = nullptr; it.cap = 0;
    NB_CHECK(hipMalloc(&it.col, sizeof(int) * ((size_t)total + 1024)));
    it.cap = (int64_t)total + 1024;
  }
  if (total > 0) {
    k_ind_table<T, 1><<<grid, 256, 0, st>>>(na, nb.rowptr, nb.col, sites, nullptr, it.rowptr, it.col);
    NB_CHECK(hipGetLastError());
  }
  return 0;
}
template int build_ind_table<float>(hipStream_t, int, const NbrTable&, const Site<float>*, IndTable&, void**, size_t*);
template int build_ind_table<double>(hipStream_t, int, const NbrTable&, const Site<double>*, IndTable&, void**, size_t*);

// ascending sort of n ints in place (keys_tmp: n ints of scratch); returns a hipError_t as int
int sort_ints(hipStream_t st, int* keys, int* keys_tmp, int n, void** scratch, size_t* scratch_bytes) {
  if (n <= 1) return 0;
  size_t need = 0;
  NB_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, keys, keys_tmp, n, 0, 32, st));
  if (need > *scratch_bytes) {
    if (*scratch) NB_CHECK(hipFree(*scratch));
    *scratch = nullptr; *scratch_bytes = 0;
    NB_CHECK(hipMalloc(scratch, need));
    *scratch_bytes = need;
  }
  NB_CHECK(hipcub::DeviceRadixSort::SortKeys(*scratch, need, keys, keys_tmp, n, 0, 32, st));
  NB_CHECK(hipMemcpyAsync(keys, keys_tmp, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
  return 0;
}

// slab owner of global mesh plane gx: the s with floor(s K / N) <= gx < floor((s + 1) K / N)  (engine.hip update_slab)
__device__ __forceinline__ int slab_owner(int gx, int K0, int N) {
  int s = (int)(((long)gx * N) / K0);
  while ((int)(((long)s * K0) / N) > gx) --s;
  while ((int)(((long)(s + 1) * K0) / N) <= gx) ++s;
  return s;
}
__global__ __launch_bounds__(256) void k_mark_imports(int n_home, const int* __restrict__ home, const int* __restrict__ rowptr,
                                                      const int* __restrict__ col, Topology top,
                                                      const int4* __restrict__ bases, int width, int K0, int X0, int N,
                                                      int* __restrict__ mark) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int slot = (int)(t >> 3), sub = (int)(t & 7);
  if (slot >= n_home) return;
  const int i = home[slot];
  auto visit = [&](int j) {
    const int b = bases[j].x;                      // local plane index of j's stencil base, relative to this rank's X0
    if (b < width) return;                         // home atom
    int gx = b + X0;
    if (gx >= K0) gx -= K0;
    mark[j] = 1 + slab_owner(gx, K0, N);           // same value from every writer
  };
  for (int k = rowptr[i] + sub; k < rowptr[i + 1]; k += 8) visit(col[k] & kColMask);
  if (sub == 0 && top.axis_type) {
    const int type = top.axis_type[i];
    if (type != NoAxisType) {
      const int iz = top.axis_idx[3 * i], ix = top.axis_idx[3 * i + 1], iy = top.axis_idx[3 * i + 2];
      if (iz >= 0) visit(iz);
      if (type != Zonly && ix >= 0) visit(ix);
      if ((type == ZBisect || type == ThreeFold) && iy >= 0) visit(iy);
    }
  }
}
void launch_mark_imports(hipStream_t st, int n_home, const int* home, const NbrTable& nb, const Topology& top,
                         const int4* bases, int width, int K0, int X0, int nranks, int* mark) {
  if (n_home <= 0) return;
  k_mark_imports<<<(unsigned)(((long)n_home * 8 + 255) / 256), 256, 0, st>>>(n_home, home, nb.rowptr, nb.col, top, bases, width,
                                                                         K0, X0, nranks, mark);
}

// one workgroup per kRowWindow rows = kRowWindow / W windows: bitonic network of (length << 10 | local index) in LDS,
// stopped at span W so that every aligned W-chunk is sorted on its own (ascending)
// With site classes (NbrTable::cls) the rows of the whole kRowWindow are first parted by class (full network on
// class << 10 | index), then every W-chunk is sorted by (class, length): a wavefront of the pair kernel then holds rows of
// one class, but for one chunk per window.
__global__ __launch_bounds__(256) void k_row_order(int na, const int* __restrict__ rowptr, int* __restrict__ order, int W,
                                                   const int* __restrict__ cls) {
  __shared__ unsigned key[kRowWindow];
  const int w0 = blockIdx.x * kRowWindow;
  if (cls) {
    for (int t = threadIdx.x; t < kRowWindow; t += 256) {
      const int i = w0 + t;
      key[t] = ((i < na ? (cls[i] != 0 ? 1u : 0u) : 2u) << 10) | (unsigned)t;
    }
    __syncthreads();
    for (int k = 2; k <= kRowWindow; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < kRowWindow; t += 256) {
          const int p = t ^ j;
          if (p > t) {
            const unsigned a = key[t], b = key[p];
            if ((a > b) == ((t & k) == 0)) { key[t] = b; key[p] = a; }
          }
        }
        __syncthreads();
      }
  }
  unsigned mine[kRowWindow / 256];
  for (int t = threadIdx.x, q = 0; t < kRowWindow; t += 256, ++q) {
    const int loc = cls ? (int)(key[t] & 1023u) : t, i = w0 + loc;
    unsigned len = 0xfffffu;                                    // padding sorts to the end
    if (i < na) { len = (unsigned)(rowptr[i + 1] - rowptr[i]); if (len > 0xffffeu) len = 0xffffeu; }
    const unsigned c = cls ? (key[t] >> 10 != 0 ? 1u : 0u) : 0u;
    mine[q] = (c << 31) | (len << 10) | (unsigned)loc;
  }
  __syncthreads();
  for (int t = threadIdx.x, q = 0; t < kRowWindow; t += 256, ++q) key[t] = mine[q];
  __syncthreads();
  for (int k = 2; k <= W; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < kRowWindow; t += 256) {
        const int p = t ^ j;
        if (p > t) {
          const unsigned a = key[t], b = key[p];
          const bool up = k == W || (t & k) == 0;
          if ((a > b) == up) { key[t] = b; key[p] = a; }
        }
      }
      __syncthreads();
    }
  for (int t = threadIdx.x; t < kRowWindow; t += 256)
    if (w0 + t < na) order[w0 + t] = w0 + (int)(key[t] & 1023u);
}
void launch_row_order(hipStream_t st, int na, const int* rowptr, int* order, const int* cls) {
  // window = the rows one workgroup of the pair kernel owns (256 lanes / lanes per row): measured at 1M atoms
  // 0.446 ms (window 128 = one workgroup) vs 0.458-0.474 (64, 256, 512, 1024) vs 0.495 unsorted
  static const int Wenv = [] { const char* e = getenv("ADMP_ROW_WINDOW"); return e ? atoi(e) : 0; }();
  int W = 256 / pair_lanes_per_row(na);
  if (W < 64) W = 64;
  if (Wenv >= 64 && Wenv <= kRowWindow) W = Wenv;
  int p2 = 64;
  while (p2 * 2 <= W) p2 *= 2;                 // power of two (the bitonic network's span)
  if (na > 0) k_row_order<<<(na + kRowWindow - 1) / kRowWindow, 256, 0, st>>>(na, rowptr, order, p2, cls);
}

}  // namespace admp
