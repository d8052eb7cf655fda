// Cell-list neighbour search on the GPU: positions -> half pair list (i < j, minimum-image r < rc).
// Stands in for jax_md.partition.neighbor_list(..., format=OrderedSparse) of the reference's drivers
// (examples/water_1024/run_admp.py:109-112) -- the producer of the `pairs` array the PME path consumes.
//
//   cells in fractional coordinates, n_d = floor(height_d / rc) cells along lattice direction d (perpendicular
//   height, so a sphere of radius rc around any point stays within the 27 neighbouring cells; directions with
//   fewer than 3 cells visit every cell once);  counting sort of the atoms by cell (wave-aggregated atomics),
//   cells sorted by atom index;  one thread per atom counts, then writes, its partners j > i.
// The minimum-image distance uses the same arithmetic as the pair kernel (min_image, admp/spatial.py:13-32).
#include <hipcub/hipcub.hpp>

#include "launch.h"

namespace admp {

struct CellGrid {
  int n[3];
  int ncell;
};

template <class T>
__device__ __forceinline__ int cell_of(const Box<T>& b, const CellGrid& cg, const T* r, int c[3]) {
  T s[3];
  s[0] = r[0] * b.hinv[0] + r[1] * b.hinv[3] + r[2] * b.hinv[6];
  s[1] = r[0] * b.hinv[1] + r[1] * b.hinv[4] + r[2] * b.hinv[7];
  s[2] = r[0] * b.hinv[2] + r[1] * b.hinv[5] + r[2] * b.hinv[8];
  for (int d = 0; d < 3; ++d) {
    T f = s[d] - m_floor(s[d]);
    int k = (int)(f * T(cg.n[d]));
    c[d] = k >= cg.n[d] ? cg.n[d] - 1 : (k < 0 ? 0 : k);
  }
  return (c[0] * cg.n[1] + c[1]) * cg.n[2] + c[2];
}

__device__ __forceinline__ int wave_agg_add_i(int* __restrict__ counter, int key, bool pred) {
  const int lane = threadIdx.x & 63;
  int leader_of_me = lane, rank = 0, cnt = 0;
  unsigned long long remaining = __ballot(pred);
  while (remaining) {                       // one pass per distinct key: shuffles and ballots only
    const int leader = __ffsll((long long)remaining) - 1;
    const int k = __shfl(key, leader, 64);
    const unsigned long long same = __ballot(pred && key == k);
    if (pred && key == k) {
      leader_of_me = leader;
      rank = __popcll(same & ((1ull << lane) - 1ull));
      cnt = __popcll(same);
    }
    remaining &= ~same;
  }
  int base = 0;
  if (pred && lane == leader_of_me) base = atomicAdd(&counter[key], cnt);   // all leaders in ONE instruction:
  base = __shfl(base, leader_of_me, 64);                                      // a single atomic round trip per call
  return pred ? base + rank : -1;
}

template <class T, int MODE>
__global__ __launch_bounds__(256) void k_cell_bin(int na, const T* __restrict__ pos, Box<T> box, CellGrid cg,
                                                  int* __restrict__ counter, int* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int cid = 0;
  if (i < na) {
    int c[3];
    T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    cid = cell_of(box, cg, r, c);
  }
  const int slot = wave_agg_add_i(counter, cid, i < na);
  if (MODE == 1 && i < na) sorted[slot] = i;
}

// cell-sorted record of one atom: the row sweep reads ONE 16-B (f32) / 32-B (f64) word per candidate instead of an index
// and three scattered coordinates
template <class T>
struct alignas(4 * sizeof(T)) CellAtom {
  T x, y, z;
  int id;
};

template <class T>
__global__ void k_cell_sort(int ncell, const int* __restrict__ start, int* __restrict__ sorted, const T* __restrict__ pos,
                            CellAtom<T>* __restrict__ spos) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  int b = start[c], e = start[c + 1];
  for (int a = b + 1; a < e; ++a) {
    int v = sorted[a], k = a - 1;
    while (k >= b && sorted[k] > v) { sorted[k + 1] = sorted[k]; --k; }
    sorted[k + 1] = v;
  }
  if (spos)
    for (int a = b; a < e; ++a) {
      const int j = sorted[a];
      spos[a] = CellAtom<T>{pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], j};
    }
}

// MODE 0: count[i] = number of partners j > i within rc ; MODE 1: write them at offs[i]
template <class T, int MODE>
__global__ __launch_bounds__(128) void k_cell_pairs(int na, const T* __restrict__ pos, Box<T> box, CellGrid cg, T rc2,
                                                    const int* __restrict__ start, const int* __restrict__ sorted,
                                                    long long* __restrict__ count, const long long* __restrict__ offs,
                                                    int* __restrict__ pairs) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= na) return;
  T ri[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  int c[3];
  cell_of(box, cg, ri, c);
  long long n = 0, w = MODE ? offs[i] : 0;
  int lo[3], cnt[3];
  for (int d = 0; d < 3; ++d) {
    if (cg.n[d] >= 3) { lo[d] = c[d] - 1; cnt[d] = 3; } else { lo[d] = 0; cnt[d] = cg.n[d]; }
  }
  for (int a = 0; a < cnt[0]; ++a)
    for (int b = 0; b < cnt[1]; ++b)
      for (int e = 0; e < cnt[2]; ++e) {
        int cx = (lo[0] + a + cg.n[0]) % cg.n[0], cy = (lo[1] + b + cg.n[1]) % cg.n[1], cz = (lo[2] + e + cg.n[2]) % cg.n[2];
        int cid = (cx * cg.n[1] + cy) * cg.n[2] + cz;
        for (int k = start[cid]; k < start[cid + 1]; ++k) {
          int j = sorted[k];
          if (j <= i) continue;
          T d[3] = {ri[0] - pos[3 * j], ri[1] - pos[3 * j + 1], ri[2] - pos[3 * j + 2]};
          min_image(box, d);
          if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < rc2) {
            if (MODE) { pairs[2 * w] = i; pairs[2 * w + 1] = j; ++w; }
            else ++n;
          }
        }
      }
  if (!MODE) count[i] = n;
}

// Full (both directions) neighbour rows straight from the cell list: row i = every j != i within rc, in the
// deterministic cell-sweep order; entry = j | nbonds << 28 (covalent map looked up in CSR form).  MODE 0 counts.
__device__ __forceinline__ int cell_lookup_nbonds(const Topology& top, int i, int j) {
  if (!top.excl_ptr) return 0;
  for (int k = top.excl_ptr[i]; k < top.excl_ptr[i + 1]; ++k)
    if (top.excl_col[k] == j) return top.excl_nb[k] & 15;
  return 0;
}

// 4 lanes per row: lane l sweeps the neighbour cells l, l+4, ... (of <= 27) of the row atom's cell; MODE 0 stores the
// four partial counts (deg4) and the row length, MODE 1 writes each lane's segment at rowptr[i] + the partial counts
// before it (lane-major, candidates in cell order: a fixed, reproducible order).  Writes beyond `cap` are dropped (the
// host re-runs the fill after growing the buffer).
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_cell_rows(Topology top, const T* __restrict__ pos, Box<T> box, CellGrid cg, T rc2,
                                                   const int* __restrict__ start, const CellAtom<T>* __restrict__ spos,
                                                   int* __restrict__ deg, int* __restrict__ deg4,
                                                   const int* __restrict__ rowptr, int* __restrict__ col, long cap,
                                                   RowFilter rf, unsigned char* __restrict__ built) {
  // rows are visited in cell-sorted order: the lanes of a wavefront then sweep the same few cells (L1-resident)
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int slot = (int)(t >> 2), l = (int)(t & 3);
  int n = 0;
  long w = 0;
  bool live = slot < top.na;
  int i = 0;
  bool skip = false;
  if (live && rf.on) {      // slab rank: only the rows of the atoms near its slab (RowFilter); the others stay empty
    const CellAtom<T> me = spos[slot];
    T f = me.x * box.hinv[0] + me.y * box.hinv[3] + me.z * box.hinv[6];
    f -= m_floor(f);
    int b = (int)(f * (T)rf.K0) - 2 - rf.lo;
    b %= rf.K0;
    if (b < 0) b += rf.K0;
    skip = b >= rf.width;
    if (!MODE && l == 0) built[me.id] = skip ? 0 : 1;
    if (skip) {
      i = me.id;
      if (!MODE) deg4[4 * i + l] = 0;
      live = false;
    }
  }
  if (live) {
    const CellAtom<T> me = spos[slot];
    i = me.id;
    T ri[3] = {me.x, me.y, me.z};
    int c[3];
    cell_of(box, cg, ri, c);
    // the row atom's exclusions (covalent neighbours) in registers: the per-hit lookup is then compare-only
    constexpr int kEx = 6;
    int ex_col[kEx], ex_nb[kEx], nex = 0, ex_beg = 0, ex_end = 0;
    if (MODE && top.excl_ptr) {
      ex_beg = top.excl_ptr[i]; ex_end = top.excl_ptr[i + 1];
      nex = ex_end - ex_beg < kEx ? ex_end - ex_beg : kEx;
#pragma unroll
      for (int q = 0; q < kEx; ++q) {
        ex_col[q] = q < nex ? top.excl_col[ex_beg + q] : -1;
        ex_nb[q] = q < nex ? (top.excl_nb[ex_beg + q] & 15) : 0;
      }
    }
    int lo[3], cnt[3];
    for (int d = 0; d < 3; ++d) {
      if (cg.n[d] >= 3) { lo[d] = c[d] - 1; cnt[d] = 3; } else { lo[d] = 0; cnt[d] = cg.n[d]; }
    }
    if (MODE) {
      w = rowptr[i];
      for (int q = 0; q < l; ++q) w += deg4[4 * i + q];
    }
    const int ncell = cnt[0] * cnt[1] * cnt[2];
    for (int ci = l; ci < ncell; ci += 4) {
      const int e = ci % cnt[2], ab = ci / cnt[2], b = ab % cnt[1], a = ab / cnt[1];
      int cx = lo[0] + a, cy = lo[1] + b, cz = lo[2] + e;
      cx = cx < 0 ? cx + cg.n[0] : (cx >= cg.n[0] ? cx - cg.n[0] : cx);
      cy = cy < 0 ? cy + cg.n[1] : (cy >= cg.n[1] ? cy - cg.n[1] : cy);
      cz = cz < 0 ? cz + cg.n[2] : (cz >= cg.n[2] ? cz - cg.n[2] : cz);
      const int cid = (cx * cg.n[1] + cy) * cg.n[2] + cz;
      const int kend = start[cid + 1];
      for (int k = start[cid]; k < kend; ++k) {
        const CellAtom<T> p = spos[k];
        if (p.id == i) continue;
        T d[3] = {ri[0] - p.x, ri[1] - p.y, ri[2] - p.z};
        min_image(box, d);
        if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < rc2) {
          if (MODE) {
            int nb = 0;
#pragma unroll
            for (int q = 0; q < kEx; ++q) nb = (ex_col[q] == p.id) ? ex_nb[q] : nb;
            for (int q = ex_beg + kEx; q < ex_end; ++q)      // rows with more than kEx covalent neighbours
              if (top.excl_col[q] == p.id) nb = top.excl_nb[q] & 15;
            if (w < cap) col[w] = p.id | (nb << 28);
            ++w;
          } else {
            ++n;
          }
        }
      }
    }
  }
  if (!MODE) {
    if (live) deg4[4 * i + l] = n;
    n += __shfl_xor(n, 1, 64);
    n += __shfl_xor(n, 2, 64);
    if ((live || skip) && l == 0) deg[i] = n;
  }
}

// Tiny systems (<= kBruteMax atoms): no cells at all -- every row tests every atom, 16 lanes per row (lane l takes the
// atoms j = l, l + 16, ...), positions staged through LDS.  Two launches (count, fill) replace the six of the cell
// path, whose dispatch latency dominates the rebuild at this size (3072 atoms: 0.21 ms -> see DESIGN.md 6).
// Row order: lane-major, j ascending within a lane -- fixed and reproducible.
constexpr int kBruteMax = 4096;
constexpr int kBruteTile = 1024;

template <class T, int MODE>
__global__ __launch_bounds__(256) void k_brute_rows(Topology top, const T* __restrict__ pos, Box<T> box, T rc2,
                                                    int* __restrict__ deg, int* __restrict__ deg16,
                                                    const int* __restrict__ rowptr, int* __restrict__ col, long cap) {
  __shared__ T tile[kBruteTile][3];
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const bool live = i < top.na;
  T ri[3] = {0, 0, 0};
  if (live) { ri[0] = pos[3 * i]; ri[1] = pos[3 * i + 1]; ri[2] = pos[3 * i + 2]; }
  constexpr int kEx = 6;
  int ex_col[kEx], ex_nb[kEx], ex_beg = 0, ex_end = 0;
#pragma unroll
  for (int q = 0; q < kEx; ++q) { ex_col[q] = -1; ex_nb[q] = 0; }
  long w = 0;
  if (MODE && live) {
    if (top.excl_ptr) {
      ex_beg = top.excl_ptr[i]; ex_end = top.excl_ptr[i + 1];
#pragma unroll
      for (int q = 0; q < kEx; ++q)
        if (ex_beg + q < ex_end) { ex_col[q] = top.excl_col[ex_beg + q]; ex_nb[q] = top.excl_nb[ex_beg + q] & 15; }
    }
    w = rowptr[i];
    for (int q = 0; q < l; ++q) w += deg16[16 * i + q];
  }
  int n = 0;
  for (int j0 = 0; j0 < top.na; j0 += kBruteTile) {
    const int nt = min(kBruteTile, top.na - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * nt; t += 256) (&tile[0][0])[t] = pos[3 * (long)j0 + t];
    __syncthreads();
    if (live)
      for (int jj = l; jj < nt; jj += 16) {
        const int j = j0 + jj;
        if (j == i) continue;
        T d[3] = {ri[0] - tile[jj][0], ri[1] - tile[jj][1], ri[2] - tile[jj][2]};
        min_image(box, d);
        if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < rc2) {
          if (MODE) {
            int nb = 0;
#pragma unroll
            for (int q = 0; q < kEx; ++q) nb = (ex_col[q] == j) ? ex_nb[q] : nb;
            for (int q = ex_beg + kEx; q < ex_end; ++q)
              if (top.excl_col[q] == j) nb = top.excl_nb[q] & 15;
            if (w < cap) col[w] = j | (nb << 28);
            ++w;
          } else {
            ++n;
          }
        }
      }
  }
  if (!MODE) {
    if (live) deg16[16 * i + l] = n;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) n += __shfl_xor(n, off, 64);
    if (live && l == 0) deg[i] = n;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// Phase 1: bins the atoms and counts the pairs.  scratch layout is owned by the engine (see CellScratch).
template <class T>
int cell_count_pairs(hipStream_t st, int na, const T* pos, const Box<T>& box, const double* heights, double rc,
                     CellScratch& cs, long long* n_pairs) {
  CellGrid cg;
  for (int d = 0; d < 3; ++d) {
    int n = (int)(heights[d] / rc);
    cg.n[d] = n < 1 ? 1 : (n > 1024 ? 1024 : n);
  }
  // keep the cell table bounded for huge dilute boxes
  while ((long)cg.n[0] * cg.n[1] * cg.n[2] > 64L * 1024 * 1024) { for (int d = 0; d < 3; ++d) cg.n[d] = (cg.n[d] + 1) / 2; }
  cg.ncell = cg.n[0] * cg.n[1] * cg.n[2];
  cs.n[0] = cg.n[0]; cs.n[1] = cg.n[1]; cs.n[2] = cg.n[2];
  if (cs.ensure(na, cg.ncell) != 0) return (int)hipErrorOutOfMemory;
  const int blocks = (na + 255) / 256;
  CK(hipMemsetAsync(cs.cursor, 0, sizeof(int) * (cg.ncell + 1), st));
  k_cell_bin<T, 0><<<blocks, 256, 0, st>>>(na, pos, box, cg, cs.cursor, nullptr);
  size_t need = cs.scan_bytes;
  CK(hipcub::DeviceScan::ExclusiveSum(cs.scan_tmp, need, cs.cursor, cs.start, cg.ncell + 1, st));
  CK(hipMemcpyAsync(cs.cursor, cs.start, sizeof(int) * (cg.ncell + 1), hipMemcpyDeviceToDevice, st));
  k_cell_bin<T, 1><<<blocks, 256, 0, st>>>(na, pos, box, cg, cs.cursor, cs.sorted);
  k_cell_sort<T><<<(cg.ncell + 127) / 128, 128, 0, st>>>(cg.ncell, cs.start, cs.sorted, pos, nullptr);
  k_cell_pairs<T, 0><<<(na + 127) / 128, 128, 0, st>>>(na, pos, box, cg, (T)(rc * rc), cs.start, cs.sorted, cs.count, nullptr, nullptr);
  need = cs.scan_bytes;
  CK(hipcub::DeviceScan::ExclusiveSum(cs.scan_tmp, need, cs.count, cs.offs, na + 1, st));
  CK(hipMemcpyAsync(n_pairs, cs.offs + na, sizeof(long long), hipMemcpyDeviceToHost, st));
  CK(hipStreamSynchronize(st));
  return 0;
}

template <class T>
int cell_fill_pairs(hipStream_t st, int na, const T* pos, const Box<T>& box, double rc, CellScratch& cs, int* pairs) {
  CellGrid cg;
  cg.n[0] = cs.n[0]; cg.n[1] = cs.n[1]; cg.n[2] = cs.n[2];
  cg.ncell = cg.n[0] * cg.n[1] * cg.n[2];
  k_cell_pairs<T, 1><<<(na + 127) / 128, 128, 0, st>>>(na, pos, box, cg, (T)(rc * rc), cs.start, cs.sorted, nullptr, cs.offs, pairs);
  CK(hipStreamSynchronize(st));
  return 0;
}

// positions -> i-grouped neighbour table of the pair kernels in one go (no pair array, no atomics, no sort)
template <class T>
int cell_build_table(hipStream_t st, const Topology& top, const T* pos, const Box<T>& box, const double* heights,
                     double rc, CellScratch& cs, NbrTable& nb, const RowFilter& rf_in) {
  const int na = top.na;
  RowFilter rf = rf_in;
  if (na <= kBruteMax) rf.on = 0;                       // (tiny systems: every row)
  if (rf.on) { if (!nb.built) CK(hipMalloc(&nb.built, (size_t)na)); }
  else if (nb.built) { (void)hipFree(nb.built); nb.built = nullptr; }
  if (na <= kBruteMax) {
    if (cs.ensure(na, 1) != 0) return (int)hipErrorOutOfMemory;
    if (!nb.rowptr) CK(hipMalloc(&nb.rowptr, sizeof(int) * (na + 1)));
    int* deg = reinterpret_cast<int*>(cs.count);
    CK(hipMemsetAsync(deg + na, 0, sizeof(int), st));
    const int rblocks = (na + 15) / 16;
    k_brute_rows<T, 0><<<rblocks, 256, 0, st>>>(top, pos, box, (T)(rc * rc), deg, cs.deg4, nullptr, nullptr, 0);
    size_t need = cs.scan_bytes;
    CK(hipcub::DeviceScan::ExclusiveSum(cs.scan_tmp, need, deg, nb.rowptr, na + 1, st));
    int total = 0;
    if (nb.cap > 0)
      k_brute_rows<T, 1><<<rblocks, 256, 0, st>>>(top, pos, box, (T)(rc * rc), nullptr, cs.deg4, nb.rowptr, nb.col, (long)nb.cap);
    CK(hipMemcpyAsync(&total, nb.rowptr + na, sizeof(int), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (total > nb.cap) {
      if (nb.col) CK(hipFree(nb.col));
      nb.cap = (int64_t)total + total / 8 + 1024;
      CK(hipMalloc(&nb.col, sizeof(int) * nb.cap));
      k_brute_rows<T, 1><<<rblocks, 256, 0, st>>>(top, pos, box, (T)(rc * rc), nullptr, cs.deg4, nb.rowptr, nb.col, (long)nb.cap);
      CK(hipStreamSynchronize(st));
    }
    nb.n_half = total / 2;
    return 0;
  }
  CellGrid cg;
  for (int d = 0; d < 3; ++d) {
    int n = (int)(heights[d] / rc);
    cg.n[d] = n < 1 ? 1 : (n > 1024 ? 1024 : n);
  }
  while ((long)cg.n[0] * cg.n[1] * cg.n[2] > 64L * 1024 * 1024) { for (int d = 0; d < 3; ++d) cg.n[d] = (cg.n[d] + 1) / 2; }
  cg.ncell = cg.n[0] * cg.n[1] * cg.n[2];
  cs.n[0] = cg.n[0]; cs.n[1] = cg.n[1]; cs.n[2] = cg.n[2];
  if (cs.ensure(na, cg.ncell) != 0) return (int)hipErrorOutOfMemory;
  const int blocks = (na + 255) / 256;
  CK(hipMemsetAsync(cs.cursor, 0, sizeof(int) * (cg.ncell + 1), st));
  k_cell_bin<T, 0><<<blocks, 256, 0, st>>>(na, pos, box, cg, cs.cursor, nullptr);
  size_t need = cs.scan_bytes;
  CK(hipcub::DeviceScan::ExclusiveSum(cs.scan_tmp, need, cs.cursor, cs.start, cg.ncell + 1, st));
  CK(hipMemcpyAsync(cs.cursor, cs.start, sizeof(int) * (cg.ncell + 1), hipMemcpyDeviceToDevice, st));
  k_cell_bin<T, 1><<<blocks, 256, 0, st>>>(na, pos, box, cg, cs.cursor, cs.sorted);
  CellAtom<T>* spos = reinterpret_cast<CellAtom<T>*>(cs.spos);
  k_cell_sort<T><<<(cg.ncell + 127) / 128, 128, 0, st>>>(cg.ncell, cs.start, cs.sorted, pos, spos);
  if (!nb.rowptr) CK(hipMalloc(&nb.rowptr, sizeof(int) * (na + 1)));
  int* deg = reinterpret_cast<int*>(cs.count);        // na + 1 ints fit in the (na + 1) long long scratch
  CK(hipMemsetAsync(deg + na, 0, sizeof(int), st));   // the scan's last element (the total)
  const int rblocks = (int)(((long)na * 4 + 255) / 256);
  k_cell_rows<T, 0><<<rblocks, 256, 0, st>>>(top, pos, box, cg, (T)(rc * rc), cs.start, spos, deg, cs.deg4, nullptr, nullptr, 0,
                                             rf, nb.built);
  need = cs.scan_bytes;
  CK(hipcub::DeviceScan::ExclusiveSum(cs.scan_tmp, need, deg, nb.rowptr, na + 1, st));
  // The fill is enqueued optimistically with the buffer of the previous build (entries beyond it are dropped) and the
  // total is read afterwards: one host synchronisation per rebuild unless the list outgrew its buffer.
  int total = 0;
  if (nb.cap > 0)
    k_cell_rows<T, 1><<<rblocks, 256, 0, st>>>(top, pos, box, cg, (T)(rc * rc), cs.start, spos, nullptr, cs.deg4, nb.rowptr,
                                               nb.col, (long)nb.cap, rf, nb.built);
  CK(hipMemcpyAsync(&total, nb.rowptr + na, sizeof(int), hipMemcpyDeviceToHost, st));
  CK(hipStreamSynchronize(st));
  if (total > nb.cap) {
    if (nb.col) CK(hipFree(nb.col));
    nb.cap = (int64_t)total + total / 8 + 1024;
    CK(hipMalloc(&nb.col, sizeof(int) * nb.cap));
    k_cell_rows<T, 1><<<rblocks, 256, 0, st>>>(top, pos, box, cg, (T)(rc * rc), cs.start, spos, nullptr, cs.deg4, nb.rowptr,
                                               nb.col, (long)nb.cap, rf, nb.built);
    CK(hipStreamSynchronize(st));
  }
  nb.n_half = total / 2;
  return 0;
}

int CellScratch::ensure(int na, int ncell) {
  size_t scan = 0;
  long long* pl = nullptr;
  int* pi = nullptr;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, pl, pl, na + 1, (hipStream_t)0);
  size_t scan2 = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan2, pi, pi, ncell + 1, (hipStream_t)0);
  if (scan2 > scan) scan = scan2;
  scan += 256;
  if (na > cap_atoms || ncell > cap_cells || scan > scan_bytes) {
    release();
    cap_atoms = na; cap_cells = ncell; scan_bytes = scan;
    if (hipMalloc(&start, sizeof(int) * (ncell + 1)) != hipSuccess) return -1;
    if (hipMalloc(&cursor, sizeof(int) * (ncell + 1)) != hipSuccess) return -1;
    if (hipMalloc(&sorted, sizeof(int) * (size_t)na) != hipSuccess) return -1;
    if (hipMalloc(&count, sizeof(long long) * ((size_t)na + 1)) != hipSuccess) return -1;
    if (hipMalloc(&offs, sizeof(long long) * ((size_t)na + 1)) != hipSuccess) return -1;
    if (hipMalloc(&spos, 32 * (size_t)na + 32) != hipSuccess) return -1;
    // partial row lengths: 4 per row (cell sweep) or 16 per row (brute-force rows of tiny systems)
    if (hipMalloc(&deg4, sizeof(int) * (na <= 4096 ? 16 : 4) * ((size_t)na + 1)) != hipSuccess) return -1;
    if (hipMalloc(&scan_tmp, scan) != hipSuccess) return -1;
    (void)hipMemset(count, 0, sizeof(long long) * ((size_t)na + 1));   // count[na] stays 0: the scans' total slot
  }
  return 0;
}

void CellScratch::release() {
  for (void* p : {(void*)start, (void*)cursor, (void*)sorted, (void*)count, (void*)offs, spos, (void*)deg4, scan_tmp})
    if (p) (void)hipFree(p);
  start = cursor = sorted = nullptr;
  spos = nullptr; deg4 = nullptr;
  count = offs = nullptr;
  scan_tmp = nullptr;
  cap_atoms = cap_cells = 0;
  scan_bytes = 0;
}

#define INST(T)                                                                                                   \
  template int cell_count_pairs<T>(hipStream_t, int, const T*, const Box<T>&, const double*, double, CellScratch&, \
                                   long long*);                                                                   \
  template int cell_fill_pairs<T>(hipStream_t, int, const T*, const Box<T>&, double, CellScratch&, int*);                \
  template int cell_build_table<T>(hipStream_t, const Topology&, const T*, const Box<T>&, const double*, double,   \
                                   CellScratch&, NbrTable&, const RowFilter&);
INST(float)
INST(double)
#undef INST

}  // namespace admp
