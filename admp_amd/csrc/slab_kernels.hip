// Multi-GPU (x-slab) decomposition kernels: who owns which atom, what a rank imports / exports, the ordered lists the
// per-row kernels walk, and the pack / unpack passes around the RCCL exchanges.  No counterpart in the reference (single
// device); SURVEY.md 8e.
//
// Every rank is handed the same full input arrays and the same neighbour table, so every rank can derive the WHOLE
// decomposition by itself, deterministically and without communication: owner[i] of every atom from its stencil base plane,
// its own import set (atoms it reads without owning them) and -- by the symmetry of the i-grouped table, which holds both
// directions of every pair, and the inverse frame map -- its export set (its atoms that other ranks read).  One word per
// atom carries it all:
//     home atom      bits = kHome | (pol > 0 ? kPolar : 0) | export mask (bit t: rank t reads this atom)
//     other atoms    bits = 1 << owner if this rank reads the atom (partner of a home row, axis atom of a home frame), else 0
// and the index lists (home atoms, home rows in pair-kernel order, polarizable home atoms, imports per owner, exports per
// reader -- all ascending, so both ends of an exchange agree on the order) are ordered compactions of that array.
#include "dft_math.h"
#include "launch.h"
#include "reduce.h"

namespace admp {

// slab owner of global mesh plane gx: the s with floor(s K / N) <= gx < floor((s + 1) K / N)  (engine.hip update_slab)
__device__ __forceinline__ int slab_owner(int gx, int K0, int N) {
  int s = (int)(((long)gx * N) / K0);
  while ((int)(((long)s * K0) / N) > gx) --s;
  while ((int)(((long)(s + 1) * K0) / N) <= gx) ++s;
  return s;
}

// owner[i] for every atom; bits[i] = kHome (| kPolar) for the rank's own atoms, 0 elsewhere
template <class T>
__global__ __launch_bounds__(256) void k_slab_owner(int na, const int4* __restrict__ bases, const T* __restrict__ pol, int width,
                                                    int K0, int X0, int N, int me, int* __restrict__ owner,
                                                    int* __restrict__ bits, const int* __restrict__ prev, int* __restrict__ mig,
                                                    const unsigned char* __restrict__ built, int* __restrict__ missing) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= na) return;
  const int b = bases[i].x;                        // local plane index of the stencil base, relative to this rank's X0
  int o = me, w = 0;
  if (b < width) {
    w = kSlabHome | ((pol && pol[i] > T(0)) ? kSlabPolar : 0);
    // a filtered table (RowFilter) holds the rows of the atoms that were near this slab when it was built: a home atom
    // without a row means the list is older than its margin allows -- the evaluation is refused, not silently short a row
    if (built && !built[i] && !*(volatile int*)missing) atomicOr(missing, 1);
  } else {
    int gx = b + X0;
    if (gx >= K0) gx -= K0;
    o = slab_owner(gx, K0, N);
  }
  owner[i] = o;
  bits[i] = w;
  if (mig) {      // atoms that changed hands since the previous evaluation (see launch_slab_decompose)
    const int q = prev[i];
    mig[i] = (o == me && q != me) ? (1 << q) : ((q == me && o != me) ? (kSlabHome | (1 << o)) : 0);
  }
}

// import marks on the atoms other ranks own, export masks on the rank's own atoms; 8 lanes per row, every atom's row is
// visited (rows of other ranks leave at once)
__global__ __launch_bounds__(256) void k_slab_marks(int na, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    Topology top, const int* __restrict__ owner, int me,
                                                    int* __restrict__ bits) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t >> 3), sub = (int)(t & 7);
  const bool mine = i < na && owner[i] == me;
  int mask = 0;
  if (mine) {
    for (int k = rowptr[i] + sub; k < rowptr[i + 1]; k += 8) {
      const int j = col[k] & kColMask, o = owner[j];
      if (o != me) { mask |= 1 << o; bits[j] = 1 << o; }      // same value from every writer
    }
    if (sub == 0 && top.axis_type) {
      // frames: this site reads its axis atoms; the sites that use this atom as an axis atom read it
      const int type = top.axis_type[i];
      if (type != NoAxisType) {
        const int ax[3] = {top.axis_idx[3 * i], type != Zonly ? top.axis_idx[3 * i + 1] : -1,
                           (type == ZBisect || type == ThreeFold) ? top.axis_idx[3 * i + 2] : -1};
        for (int m = 0; m < 3; ++m)
          if (ax[m] >= 0) { const int o = owner[ax[m]]; if (o != me) bits[ax[m]] = 1 << o; }
      }
      if (top.inv_ptr)
        for (int k = top.inv_ptr[i]; k < top.inv_ptr[i + 1]; ++k) {
          const int o = owner[top.inv_idx[k]];
          if (o != me) mask |= 1 << o;
        }
    }
  }
  mask |= __shfl_xor(mask, 1, 64);
  mask |= __shfl_xor(mask, 2, 64);
  mask |= __shfl_xor(mask, 4, 64);
  if (mine && sub == 0 && mask) atomicOr(&bits[i], mask);     // (the import marks of other rows never touch a home word)
}

// ---- ordered compaction of several columns at once ------------------------------------------------------------------------
// column c keeps the x = seq_c[p] (or p itself), p = 0 .. len_c - 1, with (bits[x] & mask_c) == want_c, in the order of p.
// Three small launches for all columns: per-block counts, one scan per column, ranked writes.
constexpr int kCompactPer = 4, kCompactBlock = 256, kCompactSpan = kCompactPer * kCompactBlock;

__device__ __forceinline__ bool compact_pred(const SlabCols& cs, int c, const int* __restrict__ bits, int p, int& x) {
  if (p >= cs.len[c]) return false;
  x = cs.seq[c] ? cs.seq[c][p] : p;
  const int* __restrict__ w = cs.src[c] ? cs.src[c] : bits;
  return (w[x] & cs.mask[c]) == cs.want[c];
}

__global__ __launch_bounds__(kCompactBlock) void k_compact_count(SlabCols cs, const int* __restrict__ bits, int nblocks,
                                                                 int* __restrict__ counts) {
  __shared__ int wsum[kCompactBlock / 64];
  const int c = blockIdx.y, p0 = blockIdx.x * kCompactSpan;
  if (cs.binned[c]) return;                 // (workgroup-uniform) filled by k_slab_bins
  int n = 0;
  if (p0 < cs.len[c]) {
#pragma unroll
    for (int r = 0; r < kCompactPer; ++r) {
      int x;
      const bool keep = compact_pred(cs, c, bits, p0 + r * kCompactBlock + (int)threadIdx.x, x);
      n += __popcll(__ballot(keep));
    }
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) counts[(size_t)c * nblocks + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup per column: counts -> exclusive offsets (in place), column total
__global__ __launch_bounds__(1024) void k_compact_scan(int nblocks, int* __restrict__ counts, int* __restrict__ totals) {
  __shared__ int wsum[16];
  __shared__ int carry;
  int* cnt = counts + (size_t)blockIdx.x * nblocks;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += 1024) {
    const int i = b0 + t;
    const int v = i < nblocks ? cnt[i] : 0;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off, 64); if (lane >= off) inc += u; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (t < 16) {
      int w = wsum[t];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { const int u = __shfl_up(w, off, 64); if (t >= off) w += u; }
      wsum[t] = w;
    }
    __syncthreads();
    const int base = carry + (wave ? wsum[wave - 1] : 0);
    if (i < nblocks) cnt[i] = base + inc - v;
    __syncthreads();
    if (t == 0) carry += wsum[15];
    __syncthreads();
  }
  if (t == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kCompactBlock) void k_compact_write(SlabCols cs, const int* __restrict__ bits, int nblocks,
                                                                 const int* __restrict__ offsets, int* __restrict__ out,
                                                                 long col_stride) {
  __shared__ int wcnt[kCompactPer][kCompactBlock / 64];
  const int c = blockIdx.y, p0 = blockIdx.x * kCompactSpan;
  if (cs.binned[c] || p0 >= cs.len[c]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x[kCompactPer];
  bool keep[kCompactPer];
  unsigned long long m[kCompactPer];
#pragma unroll
  for (int r = 0; r < kCompactPer; ++r) {
    keep[r] = compact_pred(cs, c, bits, p0 + r * kCompactBlock + (int)threadIdx.x, x[r]);
    m[r] = __ballot(keep[r]);
    if (lane == 0) wcnt[r][wave] = __popcll(m[r]);
  }
  __syncthreads();
  int base = offsets[(size_t)c * nblocks + blockIdx.x];
  int* o = out + (size_t)c * col_stride;
#pragma unroll
  for (int r = 0; r < kCompactPer; ++r) {
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kCompactBlock / 64; ++w) { if (w < wave) before += wcnt[r][w]; total += wcnt[r][w]; }
    if (keep[r]) o[base + before + __popcll(m[r] & ((1ull << lane) - 1ull))] = x[r];
    base += total;
  }
}

// ---- all per-peer columns in one ordered pass --------------------------------------------------------------------------------
// A workgroup takes kCompactSpan consecutive atoms as 16 chunks of 64 (chunk q = r * 4 + wave holds atoms p0 + r * 256 + wave *
// 64 + lane: ascending in q).  visit() enumerates, wave-uniformly, the bins the 64 atoms of a chunk fall into (ballot per bin:
// a wave of a liquid holds home atoms and the imports / exports of one or two neighbours); rank inside the chunk = number of
// lower lanes in the same bin, so every bin's list comes out in ascending atom order -- the order both ends of an exchange
// rely on.  WRITE = false: per-workgroup counts of every bin; WRITE = true (after the scan): the lists.
template <class F>
__device__ __forceinline__ void slab_visit_bins(int w, int g, const SlabBins& sb, F&& f) {
  const bool home = (w & kSlabHome) != 0;
  const int peers = w & ((1 << kSlabMaxRanks) - 1);
  { const bool m = home; const unsigned long long b = __ballot(m); if (b) f(sb.c_home, m, b); }
  { const bool m = home && (w & kSlabPolar); const unsigned long long b = __ballot(m); if (b) f(sb.c_act, m, b); }
  const int imp = home ? 0 : peers, ex = home ? peers : 0;
  const bool gout = (g & kSlabHome) != 0;
  const int gp = g & ((1 << kSlabMaxRanks) - 1), gin = gout ? 0 : gp, go = gout ? gp : 0;
  // (which peers occur at all in this chunk: one OR over the wave, then one ballot per peer that does)
  int any = imp | ex | gp;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) any |= __shfl_xor(any, off, 64);
  while (any) {
    const int t = __builtin_ctz(any);
    any &= any - 1;
    { const bool m = (imp >> t) & 1; const unsigned long long b = __ballot(m); if (b) f(sb.c_imp[t], m, b); }
    { const bool m = (ex >> t) & 1; const unsigned long long b = __ballot(m); if (b) f(sb.c_exp[t], m, b); }
    { const bool m = (gin >> t) & 1; const unsigned long long b = __ballot(m); if (b && sb.c_min[t] >= 0) f(sb.c_min[t], m, b); }
    { const bool m = (go >> t) & 1; const unsigned long long b = __ballot(m); if (b && sb.c_mout[t] >= 0) f(sb.c_mout[t], m, b); }
  }
}

template <bool WRITE>
__global__ __launch_bounds__(kCompactBlock) void k_slab_bins(int na, int ncols, const int* __restrict__ bits,
                                                             const int* __restrict__ mig, SlabBins sb, int nblocks,
                                                             int* __restrict__ counts, int* __restrict__ lists, long col_stride) {
  __shared__ int cnt[kCompactPer * (kCompactBlock / 64)][kSlabMaxCols];
  constexpr int kChunks = kCompactPer * (kCompactBlock / 64);
  const int p0 = blockIdx.x * kCompactSpan, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = threadIdx.x; t < kChunks * kSlabMaxCols; t += kCompactBlock) (&cnt[0][0])[t] = 0;
  __syncthreads();
  int w[kCompactPer], g[kCompactPer];
#pragma unroll
  for (int r = 0; r < kCompactPer; ++r) {
    const int p = p0 + r * kCompactBlock + (int)threadIdx.x;
    w[r] = p < na ? bits[p] : 0;
    g[r] = (mig && p < na) ? mig[p] : 0;
    const int q = r * (kCompactBlock / 64) + wave;
    slab_visit_bins(w[r], g[r], sb, [&](int c, bool, unsigned long long b) { if (lane == 0 && c >= 0) cnt[q][c] = __popcll(b); });
  }
  __syncthreads();
  if (!WRITE) {
    for (int c = threadIdx.x; c < ncols; c += kCompactBlock) {
      int tot = 0;
      for (int q = 0; q < kChunks; ++q) tot += cnt[q][c];
      // (only the bins' columns: the others belong to k_compact_count)
      bool mine = c == sb.c_home || c == sb.c_act;
      for (int t = 0; t < sb.N && !mine; ++t) mine = c == sb.c_imp[t] || c == sb.c_exp[t] || c == sb.c_min[t] || c == sb.c_mout[t];
      if (mine) counts[(size_t)c * nblocks + blockIdx.x] = tot;
    }
    return;
  }
  for (int c = threadIdx.x; c < ncols; c += kCompactBlock) {      // chunk counts -> offsets of the chunks inside this workgroup
    int run = counts[(size_t)c * nblocks + blockIdx.x];           // (scanned: this workgroup's offset in column c)
    for (int q = 0; q < kChunks; ++q) { const int v = cnt[q][c]; cnt[q][c] = run; run += v; }
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < kCompactPer; ++r) {
    const int p = p0 + r * kCompactBlock + (int)threadIdx.x;
    const int q = r * (kCompactBlock / 64) + wave;
    slab_visit_bins(w[r], g[r], sb, [&](int c, bool member, unsigned long long b) {
      if (member && c >= 0) lists[(size_t)c * col_stride + cnt[q][c] + __popcll(b & lt)] = p;
    });
  }
}

int launch_slab_decompose(hipStream_t st, int na, const NbrTable& nb, const Topology& top, const int4* bases, const void* pol,
                          int prec, int width, int K0, int X0, int nranks, int me, int* owner, int* bits,
                          const SlabCols& cs, const SlabBins& sb, int* counts, int* totals, int* lists, const int* owner_prev,
                          int* mig) {
  if (na <= 0) return 0;
  const unsigned g1 = (unsigned)((na + 255) / 256);
  const int* pv = (owner_prev && mig) ? owner_prev : nullptr;
  int* mg = pv ? mig : nullptr;
  int* missing = totals + cs.ncols;                // (one word past the columns' totals)
  { const hipError_t e = hipMemsetAsync(missing, 0, sizeof(int), st); if (e != hipSuccess) return (int)e; }
  if (prec == 4) k_slab_owner<float><<<g1, 256, 0, st>>>(na, bases, (const float*)pol, width, K0, X0, nranks, me, owner, bits, pv, mg,
                                                         nb.built, missing);
  else k_slab_owner<double><<<g1, 256, 0, st>>>(na, bases, (const double*)pol, width, K0, X0, nranks, me, owner, bits, pv, mg,
                                                nb.built, missing);
  k_slab_marks<<<(unsigned)(((long)na * 8 + 255) / 256), 256, 0, st>>>(na, nb.rowptr, nb.col, top, owner, me, bits);
  int maxlen = 0;
  for (int c = 0; c < cs.ncols; ++c) maxlen = cs.len[c] > maxlen ? cs.len[c] : maxlen;
  const int nblocks = (maxlen + kCompactSpan - 1) / kCompactSpan;
  if (nblocks <= 0) return (int)hipMemsetAsync(totals, 0, sizeof(int) * cs.ncols, st);
  int generic = 0;
  for (int c = 0; c < cs.ncols; ++c) generic += cs.binned[c] ? 0 : 1;
  const int* migp = (owner_prev && mig) ? mig : nullptr;
  if (generic) k_compact_count<<<dim3(nblocks, cs.ncols), kCompactBlock, 0, st>>>(cs, bits, nblocks, counts);
  if (sb.N > 0) k_slab_bins<false><<<nblocks, kCompactBlock, 0, st>>>(na, cs.ncols, bits, migp, sb, nblocks, counts, lists, (long)na);
  k_compact_scan<<<cs.ncols, 1024, 0, st>>>(nblocks, counts, totals);
  if (generic) k_compact_write<<<dim3(nblocks, cs.ncols), kCompactBlock, 0, st>>>(cs, bits, nblocks, counts, lists, (long)na);
  if (sb.N > 0) k_slab_bins<true><<<nblocks, kCompactBlock, 0, st>>>(na, cs.ncols, bits, migp, sb, nblocks, counts, lists, (long)na);
  return (int)hipGetLastError();
}
int slab_compact_blocks(int maxlen) { return (maxlen + kCompactSpan - 1) / kCompactSpan; }

// the per-peer columns of one kind (imports or exports) joined into one list, peers in rank order
__global__ __launch_bounds__(256) void k_slab_concat(SlabSegs segs, const int* __restrict__ lists, long col_stride,
                                                     int* __restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= segs.off[segs.n]) return;
  int s = 0;
  while (k >= segs.off[s + 1]) ++s;
  out[k] = lists[(size_t)segs.col[s] * col_stride + (k - segs.off[s])];
}
void launch_slab_concat(hipStream_t st, const SlabSegs& segs, const int* lists, long col_stride, int* out) {
  const int n = segs.off[segs.n];
  if (n > 0) k_slab_concat<<<(n + 255) / 256, 256, 0, st>>>(segs, lists, col_stride, out);
}

// ---- halo rows -------------------------------------------------------------------------------------------------------------
// out[k][0..w) = src[idx[k]][0..w)
template <class T>
__global__ __launch_bounds__(256) void k_rows_gather(int n, int w, const int* __restrict__ idx, const T* __restrict__ src,
                                                     T* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n * w) return;
  const int k = t / w, c = t - k * w;
  out[t] = src[(size_t)idx[k] * w + c];
}
// mode 0: dst[idx[k]] = in[k]; mode 1: dst[idx[k]] += in[k]; mode 2: dst[idx[k]] = 0
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_rows_scatter(int n, int w, const int* __restrict__ idx, const T* __restrict__ in,
                                                      T* __restrict__ dst) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n * w) return;
  const int k = t / w, c = t - k * w;
  T* d = dst + (size_t)idx[k] * w + c;
  if (MODE == 0) *d = in[t];
  else if (MODE == 1) *d += in[t];
  else *d = T(0);
}
template <class T>
void launch_rows_gather(hipStream_t st, int n, int w, const int* idx, const T* src, T* out) {
  if (n > 0) k_rows_gather<T><<<(n * w + 255) / 256, 256, 0, st>>>(n, w, idx, src, out);
}
template <class T>
void launch_rows_scatter(hipStream_t st, int mode, int n, int w, const int* idx, const T* in, T* dst) {
  if (n <= 0) return;
  const unsigned g = (unsigned)((n * w + 255) / 256);
  if (mode == 0) k_rows_scatter<T, 0><<<g, 256, 0, st>>>(n, w, idx, in, dst);
  else if (mode == 1) k_rows_scatter<T, 1><<<g, 256, 0, st>>>(n, w, idx, in, dst);
  else k_rows_scatter<T, 2><<<g, 256, 0, st>>>(n, w, idx, in, dst);
}

// induced dipoles of the halo atoms.  what 0: the Cartesian dipoles of the listed atoms -> out (3 reals per atom);
// what 1: the last Jacobi step's change dU (pad words of the site rows, harmonic order; zero for non-polarizable sites,
// whose pad[0] carries the charge-only mark instead)
template <class T>
__global__ __launch_bounds__(256) void k_halo_u_pack(int n, int what, const int* __restrict__ idx, const T* __restrict__ Ucart,
                                                     const Site<T>* __restrict__ sites, T* __restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int i = idx[k];
  if (what == 0) {
    out[3 * k] = Ucart[3 * i]; out[3 * k + 1] = Ucart[3 * i + 1]; out[3 * k + 2] = Ucart[3 * i + 2];
  } else {
    const bool p = sites[i].p6 > T(0);
    out[3 * k] = p ? sites[i].pad[0] : T(0); out[3 * k + 1] = p ? sites[i].pad[1] : T(0); out[3 * k + 2] = p ? sites[i].pad[2] : T(0);
  }
}
// what 0: Ucart and the packed harmonic copy of the listed atoms <- in; what 1: dU into the pad words, U += dU (both copies)
template <class T>
__global__ __launch_bounds__(256) void k_halo_u_unpack(int n, int what, const int* __restrict__ idx, const T* __restrict__ in,
                                                       T* __restrict__ Ucart, Site<T>* __restrict__ sites) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int i = idx[k];
  if (what == 0) {
    const T ux = in[3 * k], uy = in[3 * k + 1], uz = in[3 * k + 2];
    Ucart[3 * i] = ux; Ucart[3 * i + 1] = uy; Ucart[3 * i + 2] = uz;
    sites[i].U[0] = uz; sites[i].U[1] = ux; sites[i].U[2] = uy;          // harmonic order (z, x, y)
  } else if (sites[i].p6 > T(0)) {
    const T dz = in[3 * k], dx = in[3 * k + 1], dy = in[3 * k + 2];      // harmonic order
    sites[i].pad[0] = dz; sites[i].pad[1] = dx; sites[i].pad[2] = dy;
    sites[i].U[0] += dz; sites[i].U[1] += dx; sites[i].U[2] += dy;
    Ucart[3 * i] += dx; Ucart[3 * i + 1] += dy; Ucart[3 * i + 2] += dz;
  }
}
template <class T>
void launch_halo_u_pack(hipStream_t st, int n, int what, const int* idx, const T* Ucart, const Site<T>* sites, T* out) {
  if (n > 0) k_halo_u_pack<T><<<(n + 255) / 256, 256, 0, st>>>(n, what, idx, Ucart, sites, out);
}
template <class T>
void launch_halo_u_unpack(hipStream_t st, int n, int what, const int* idx, const T* in, T* Ucart, Site<T>* sites) {
  if (n > 0) k_halo_u_unpack<T><<<(n + 255) / 256, 256, 0, st>>>(n, what, idx, in, Ucart, sites);
}

// ---- transposes of the distributed transform ---------------------------------------------------------------------------------
// x-slab layout spec[nx][K1][pitch >= nh] (rows padded to whole cache lines) <-> send / receive buffer: the block for peer t
// holds [nx][ny_t][nh] with ny_t = the y rows of rank t; one complex number per thread, z fastest (coalesced both sides).
// dir 0: spec -> buf (pack), dir 1: buf -> spec (unpack)
// A workgroup takes kPackRows consecutive y rows of one x plane: 16 lanes look up the rows' peers and offsets once (round 3
// did two integer divisions and the owner search per ELEMENT), then all threads sweep the rows' elements flat (row index by
// a float-reciprocal division), z fastest on both sides.
constexpr int kPackRows = 16;
template <class T>
__global__ __launch_bounds__(256) void k_transpose_pack(int nx, int K1, int nh, int pitch, int N, int dir, Cx<T>* __restrict__ spec,
                                                        Cx<T>* __restrict__ buf) {
  __shared__ long boff[kPackRows];
  const int ya = blockIdx.x * kPackRows, x = blockIdx.y, rows = min(kPackRows, K1 - ya);
  if (threadIdx.x < rows) {
    const int y = ya + threadIdx.x;
    int p = (int)(((long)y * N) / K1);                                   // owner of row y: floor(s K1 / N) <= y
    while ((int)(((long)p * K1) / N) > y) --p;
    while ((int)(((long)(p + 1) * K1) / N) <= y) ++p;
    const int y0 = (int)(((long)p * K1) / N), ny = (int)(((long)(p + 1) * K1) / N) - y0;
    boff[threadIdx.x] = (long)nx * y0 * nh + ((long)x * ny + (y - y0)) * nh;   // blocks of lower peers hold nx * y0 rows
  }
  __syncthreads();
  Cx<T>* __restrict__ s = spec + ((long)x * K1 + ya) * pitch;
  const int n = rows * nh;
  const float inv = 1.0f / (float)nh;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = fast_div(e, nh, inv), z = e - r * nh;
    if (dir == 0) buf[boff[r] + z] = s[(long)r * pitch + z];
    else s[(long)r * pitch + z] = buf[boff[r] + z];
  }
}
template <class T>
void launch_transpose_pack(hipStream_t st, int nx, int K1, int nh, int pitch, int nranks, int dir, T* spec, T* buf) {
  if (nx <= 0 || K1 <= 0 || nh <= 0) return;
  k_transpose_pack<T><<<dim3((unsigned)((K1 + kPackRows - 1) / kPackRows), (unsigned)nx), 256, 0, st>>>(
      nx, K1, nh, pitch, nranks, dir, reinterpret_cast<Cx<T>*>(spec), reinterpret_cast<Cx<T>*>(buf));
}

// the four energy parts of an evaluation in one block of words (what a SUM all-reduce over the ranks then takes)
__global__ void k_energy_pack(const double* __restrict__ e, int recip_slot, double* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double r = 0.0;
  if (recip_slot >= 0) r = e[recip_slot];
  else if (recip_slot == -1) for (int k = 0; k < E_PARTS; ++k) r += e[E_SLOTS + k];
  double er = e[E_REAL];
  for (int k = 0; k < E_PARTS; ++k) er += e[E_RPARTS + k];
  out[0] = er; out[1] = r; out[2] = e[E_SELF]; out[3] = e[E_PEN];
}
void launch_energy_pack(hipStream_t st, const double* e, int recip_slot, double* out) {
  k_energy_pack<<<1, 64, 0, st>>>(e, recip_slot, out);
}

#define INST(T)                                                                                                        \
  template void launch_rows_gather<T>(hipStream_t, int, int, const int*, const T*, T*);                                 \
  template void launch_rows_scatter<T>(hipStream_t, int, int, int, const int*, const T*, T*);                           \
  template void launch_halo_u_pack<T>(hipStream_t, int, int, const int*, const T*, const Site<T>*, T*);                 \
  template void launch_halo_u_unpack<T>(hipStream_t, int, int, const int*, const T*, T*, Site<T>*);                     \
  template void launch_transpose_pack<T>(hipStream_t, int, int, int, int, int, int, T*, T*);
INST(float)
INST(double)
#undef INST

}  // namespace admp
