// Neighbour-table compiler: turns the caller's (n_rows, 2) pair list into the i-grouped table the pair
// kernels walk.  Replaces the per-call `pairs[pairs[:,0] < pairs[:,1]]` filter, the covalent_map lookup
// `nbonds = covalent_map[pairs[:,0], pairs[:,1]]` and the per-pair gathers of the reference
// (admp/pme.py:671-693, admp/pairwise.py:68-75).
//   1. count: degree of each atom over valid rows (0 <= i < j < na)            -- int atomics
//   2. exclusive scan of the degrees (hipcub)                                   -- rowptr
//   3. fill: both directions of each pair, nbonds from the CSR covalent map packed in bits 28..31
//   4. sort each row by partner index (ranking in LDS, out of place)             -- deterministic order
#include <hipcub/hipcub.hpp>

#include <utility>

#include "launch.h"

namespace admp {

// Runs of equal row atoms inside a wavefront are combined into ONE counter update: pair lists arrive grouped by i (the
// reference's jax_md OrderedSparse lists, this package's cell builder), so the 64 lanes of a wave used to hammer a handful
// of words -- k_nbr_count / k_nbr_fill took 0.83 / 1.26 ms for the 13.8 M pairs of 1M atoms.  key < 0: lane idle.
// Gives the lane's rank inside its run, the run's length and the lane that leads it.
__device__ __forceinline__ void wave_runs(int key, int& rank, int& len, int& head) {
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(key, 1, 64);
  const bool is_head = lane == 0 || prev != key;
  const unsigned long long H = __ballot(is_head);
  const unsigned long long below = H & ((2ull << lane) - 1ull);          // heads at or below this lane (lane 63: all)
  head = 63 - __clzll((long long)below);
  const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1)) << (lane + 1);
  const int next = above ? __ffsll((long long)above) - 1 : 64;           // first head after this lane
  rank = lane - head;
  len = next - head;
}

__global__ void k_nbr_count(int64_t n_rows, const int* __restrict__ pairs, int na, int* __restrict__ deg,
                            unsigned long long* n_valid) {
  unsigned long long local = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (n_rows + stride - 1) / stride;                 // wave-uniform trip count (shuffles inside)
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t p = r * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int i = -1, j = -1;
    if (p < n_rows) { i = pairs[2 * p]; j = pairs[2 * p + 1]; }
    const bool ok = i >= 0 && i < j && j < na;
    int rank, len, head;
    wave_runs(ok ? i : -1, rank, len, head);
    if (ok) {
      if (rank == 0) atomicAdd(&deg[i], len);
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
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (n_rows + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t p = r * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int i = -1, j = -1;
    if (p < n_rows) { i = pairs[2 * p]; j = pairs[2 * p + 1]; }
    const bool ok = i >= 0 && i < j && j < top.na;
    int rank, len, head;
    wave_runs(ok ? i : -1, rank, len, head);
    int base = 0;
    if (ok && rank == 0) base = atomicAdd(&cursor[i], len);              // one update per run of equal row atoms
    base = __shfl(base, head, 64);
    if (ok) {
      const int nb = lookup_nbonds(top, i, j) << 28;
      col[base + rank] = j | nb;
      col[atomicAdd(&cursor[j], 1)] = i | nb;
    }
  }
}

// Every row sorted by partner index (fixed summation order of the pair kernels), OUT OF PLACE by ranking: the entries of the
// 32 rows of a workgroup are one contiguous segment of `cin`, loaded into LDS with coalesced reads; 8 lanes per row then
// count, for each of their entries, the entries of the row with a smaller key (ties by position: duplicates keep their
// order) and store the entry at its rank.  One read and one write of the table; the insertion sort this replaces (one thread
// per row, in place in global memory) moved 7.4 GB to order a 111 MB table at 1M atoms and took 0.96-1.8 ms.
constexpr int kSortRows = 32, kSortLanes = 8, kSortCap = 8192;
__global__ __launch_bounds__(kSortRows * kSortLanes) void k_nbr_rank_sort(int na, const int* __restrict__ rowptr,
                                                                          const int* __restrict__ cin,
                                                                          int* __restrict__ cout) {
  __shared__ int seg[kSortCap];
  const int r0 = blockIdx.x * kSortRows, r1 = min(na, r0 + kSortRows);
  const int sb = rowptr[r0], se = rowptr[r1];
  const bool lds = se - sb <= kSortCap;                                   // workgroup-uniform
  if (lds) {
    for (int t = threadIdx.x; t < se - sb; t += kSortRows * kSortLanes) seg[t] = cin[sb + t];
    __syncthreads();
  }
  const int row = r0 + (int)threadIdx.x / kSortLanes, sub = (int)threadIdx.x % kSortLanes;
  if (row >= r1) return;
  const int b = rowptr[row], e = rowptr[row + 1];
  // (no `seg - sb` base pointer: an LDS pointer below its array wraps in 32 bits, and the flat pointer made from it does not
  // wrap back when the index is added -- the address leaves the LDS aperture)
  auto entry = [&](int k) { return lds ? seg[k - sb] : cin[k]; };          // entry k of the table
  constexpr int kMine = 8;                                                // rows of up to 8 * kMine = 64 entries: the lane
  if (e - b <= kSortLanes * kMine) {                                      // keeps its entries in registers and walks the row
    int v[kMine], key[kMine], rank[kMine];                                // ONCE (one LDS read per row entry serves all of them)
#pragma unroll
    for (int q = 0; q < kMine; ++q) {
      const int k = b + sub + q * kSortLanes;
      v[q] = k < e ? entry(k) : 0;
      key[q] = k < e ? (v[q] & kColMask) : 0x7fffffff;
      rank[q] = 0;
    }
    for (int m = b; m < e; ++m) {
      const int u = entry(m) & kColMask;
#pragma unroll
      for (int q = 0; q < kMine; ++q) rank[q] += (u < key[q] || (u == key[q] && m < b + sub + q * kSortLanes)) ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < kMine; ++q)
      if (b + sub + q * kSortLanes < e) cout[b + rank[q]] = v[q];
    return;
  }
  for (int k = b + sub; k < e; k += kSortLanes) {
    const int v = entry(k), key = v & kColMask;
    int rank = 0;
    for (int m = b; m < e; ++m) {
      const int u = entry(m) & kColMask;
      rank += (u < key || (u == key && m < k)) ? 1 : 0;
    }
    cout[b + rank] = v;
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
// the second column buffer of a table (sort and class partition work out of place and swap the two): grown with `cap`
static int ensure_col_alt(NbrTable& nb) {
  if (nb.col_alt && nb.cap_alt >= nb.cap) return 0;
  if (nb.col_alt) { hipError_t e = hipFree(nb.col_alt); nb.col_alt = nullptr; nb.cap_alt = 0; if (e != hipSuccess) return (int)e; }
  hipError_t e = hipMalloc(&nb.col_alt, sizeof(int) * (size_t)nb.cap);
  if (e != hipSuccess) return (int)e;
  nb.cap_alt = nb.cap;
  return 0;
}
int launch_class_partition(hipStream_t st, int na, NbrTable& nb) {
  if (!nb.cls || !nb.col || na <= 0 || nb.cap <= 0) return 0;
  int rc = ensure_col_alt(nb);
  if (rc != 0) return rc;
  k_class_partition<<<(unsigned)(((long)na * kPartLanes + 255) / 256), 256, 0, st>>>(na, nb.rowptr, nb.col, nb.col_alt, nb.cls);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  std::swap(nb.col, nb.col_alt);                 // stream order keeps the readers of the old buffer ahead of its next writer
  std::swap(nb.cap, nb.cap_alt);
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
  // na + 1 degrees (reused as the fill cursor), padded to an even count, then one 64-bit pair counter: kept with the table
  const size_t ndeg = (size_t)((na + 2) & ~1);
  if (!nb.deg || nb.deg_na != na) {
    if (nb.deg) NB_CHECK(hipFree(nb.deg));
    nb.deg = nullptr;
    NB_CHECK(hipMalloc(&nb.deg, sizeof(int) * ndeg + sizeof(unsigned long long)));
    nb.deg_na = na;
  }
  int* deg = nb.deg;
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
  NB_CHECK(hipStreamSynchronize(st));             // the one host read of a build: the entry count sizes the buffers
  nb.n_half = (int64_t)nv;
  if (2 * nb.n_half > nb.cap) {
    if (nb.col) NB_CHECK(hipFree(nb.col));
    nb.col = nullptr; nb.cap = 0;
    NB_CHECK(hipMalloc(&nb.col, sizeof(int) * (2 * nb.n_half + 1024)));
    nb.cap = 2 * nb.n_half + 1024;
  }
  { int rc = ensure_col_alt(nb); if (rc != 0) return rc; }
  NB_CHECK(hipMemcpyAsync(deg, nb.rowptr, sizeof(int) * (na + 1), hipMemcpyDeviceToDevice, st));
  if (n_rows > 0) {
    k_nbr_fill<<<blocks, 256, 0, st>>>(n_rows, pairs_dev, top, deg, nb.col_alt);
    k_nbr_rank_sort<<<(na + kSortRows - 1) / kSortRows, kSortRows * kSortLanes, 0, st>>>(na, nb.rowptr, nb.col_alt, nb.col);
    NB_CHECK(hipGetLastError());
  }
  return 0;
}

// sub-table of the polarizable-polarizable entries: the kept entries of row i go to icol at the row's own offset rowptr[i]
// (row order preserved), iend[i] = one past the last (= rowptr[i] for non-polarizable atoms).  8 lanes per row.
template <class T>
__global__ __launch_bounds__(256) void k_ind_table(int na, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const Site<T>* __restrict__ sites, int* __restrict__ iend,
                                                   int* __restrict__ icol) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(t >> 3), sub = (int)(t & 7);
  const bool live = row < na && sites[row].p6 > T(0);
  const int beg = row < na ? rowptr[row] : 0, end = live ? rowptr[row + 1] : beg;
  int base = beg;
  for (int k0 = beg; k0 < end; k0 += 8) {      // the 8 lanes of a row share the trip count; the ballot is read per group
    const int k = k0 + sub;
    const int c = k < end ? col[k] : 0;
    const bool keep = k < end && sites[c & kColMask].p6 > T(0);
    // rank of this lane's entry among the kept ones of its 8-lane group
    const unsigned long long m = __ballot(keep);
    const int g0 = (threadIdx.x & 63) & ~7;
    const unsigned grp = (unsigned)((m >> g0) & 0xffull);
    if (keep) icol[base + __popc(grp & ((1u << sub) - 1u))] = c;
    base += __popc(grp);
  }
  if (row < na && sub == 0) iend[row] = base;
}

// it <- the polarizable-polarizable entries of nb (rows keyed by atom, row order preserved), one pass, nothing read back
template <class T>
int build_ind_table(hipStream_t st, int na, const NbrTable& nb, const Site<T>* sites, IndTable& it) {
  if (na > it.na_cap) {
    if (it.end) NB_CHECK(hipFree(it.end));
    it.end = nullptr; it.na_cap = 0;
    NB_CHECK(hipMalloc(&it.end, sizeof(int) * (size_t)na));
    it.na_cap = na;
  }
  const int64_t need = 2 * nb.n_half;
  if (need > it.cap) {
    if (it.col) NB_CHECK(hipFree(it.col));
    it.col = nullptr; it.cap = 0;
    const int64_t cap = need + need / 8 + 1024;                   // (a rebuilt list is a little longer or shorter: no regrowth)
    NB_CHECK(hipMalloc(&it.col, sizeof(int) * (size_t)cap));
    it.cap = cap;
  }
  it.beg = nb.rowptr;
  if (na > 0) {
    const unsigned grid = (unsigned)(((long)na * 8 + 255) / 256);
    k_ind_table<T><<<grid, 256, 0, st>>>(na, nb.rowptr, nb.col, sites, it.end, it.col);
    NB_CHECK(hipGetLastError());
  }
  return 0;
}
template int build_ind_table<float>(hipStream_t, int, const NbrTable&, const Site<float>*, IndTable&);
template int build_ind_table<double>(hipStream_t, int, const NbrTable&, const Site<double>*, IndTable&);

// ---- inner list of an MD loop (round 4, admp_prune_pairs) ------------------------------------------------------------------
// A Verlet list with a skin holds (rc + skin)^3 / rc^3 times the pairs inside rc (rc 4 + 1 A: 1.95 x) and the multipolar kernels
// evaluate every listed pair.  Between two rebuilds of that OUTER table the calculators can walk an INNER one: the entries of
// the outer table whose current distance is below rc + a small margin (what the atoms can move until the next prune), compacted
// row by row in the outer table's order (class runs and marks survive).  MODE 0 counts per row, MODE 1 copies; 8 lanes per row.
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_prune_rows(int na, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    const T* __restrict__ pos, Box<T> box, T rc2, int* __restrict__ cnt,
                                                    const int* __restrict__ irow, int* __restrict__ icol) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(t >> 3), sub = (int)(t & 7);
  const bool live = row < na;
  const int beg = live ? rowptr[row] : 0, end = live ? rowptr[row + 1] : 0;
  T ri[3] = {0, 0, 0};
  if (live) { ri[0] = pos[3 * row]; ri[1] = pos[3 * row + 1]; ri[2] = pos[3 * row + 2]; }
  int base = (MODE == 1 && live) ? irow[row] : 0, total = 0;
  for (int k0 = beg; k0 < end; k0 += 8) {
    const int k = k0 + sub;
    const int c = k < end ? col[k] : 0;
    bool keep = false;
    if (k < end) {
      const int j = c & kColMask;
      T d[3] = {ri[0] - pos[3 * j], ri[1] - pos[3 * j + 1], ri[2] - pos[3 * j + 2]};
      min_image(box, d);
      keep = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < rc2;
    }
    const unsigned long long m = __ballot(keep);
    const int g0 = (threadIdx.x & 63) & ~7;
    const unsigned grp = (unsigned)((m >> g0) & 0xffull);
    if (MODE == 1 && keep) icol[base + __popc(grp & ((1u << sub) - 1u))] = c;
    base += __popc(grp);
    total += __popc(grp);
  }
  if (MODE == 0 && live && sub == 0) cnt[row] = total;
}
// inner table of `full` for the pairs below rc (rowptr_out: na + 1, col_out: as many entries as full.col, cnt: na + 1 scratch);
// *total = directed entries kept.  One host synchronisation.  hipError_t as int.
template <class T>
int prune_table(hipStream_t st, int na, const NbrTable& full, const T* pos, const Box<T>& box, double rc, int* rowptr_out,
                int* cnt, int* col_out, void** scratch, size_t* scratch_bytes, int64_t* total) {
  const unsigned grid = (unsigned)(((long)na * 8 + 255) / 256);
  const T rc2 = (T)(rc * rc);
  NB_CHECK(hipMemsetAsync(cnt + na, 0, sizeof(int), st));
  k_prune_rows<T, 0><<<grid, 256, 0, st>>>(na, full.rowptr, full.col, pos, box, rc2, cnt, nullptr, nullptr);
  NB_CHECK(hipGetLastError());
  size_t need = 0;
  NB_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, cnt, rowptr_out, na + 1, st));
  if (need > *scratch_bytes) {
    if (*scratch) NB_CHECK(hipFree(*scratch));
    *scratch = nullptr; *scratch_bytes = 0;
    NB_CHECK(hipMalloc(scratch, need));
    *scratch_bytes = need;
  }
  NB_CHECK(hipcub::DeviceScan::ExclusiveSum(*scratch, need, cnt, rowptr_out, na + 1, st));
  k_prune_rows<T, 1><<<grid, 256, 0, st>>>(na, full.rowptr, full.col, pos, box, rc2, nullptr, rowptr_out, col_out);
  NB_CHECK(hipGetLastError());
  int tot = 0;
  NB_CHECK(hipMemcpyAsync(&tot, rowptr_out + na, sizeof(int), hipMemcpyDeviceToHost, st));
  NB_CHECK(hipStreamSynchronize(st));
  *total = tot;
  return 0;
}
template int prune_table<float>(hipStream_t, int, const NbrTable&, const float*, const Box<float>&, double, int*, int*, int*, void**, size_t*, int64_t*);
template int prune_table<double>(hipStream_t, int, const NbrTable&, const double*, const Box<double>&, double, int*, int*, int*, void**, size_t*, int64_t*);

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
