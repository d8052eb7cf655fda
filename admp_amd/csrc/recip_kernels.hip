// Reciprocal-space kernels (reference admp/recip.py:21-431): B-spline spread of the multipoles onto
// the K1 x K2 x K3 mesh, the k-space multiply with the cached table G_k = 2 D C_k / theta_k^2, and the
// gather that produces dE/dQ and dE/dr from phi = c2r(G S) (the adjoint the reference leaves to jax.grad).
// The 3-D transforms themselves are rocFFT r2c / c2r plans driven from engine.hip.
#include <cstdlib>
#include <cstring>
#include <hipcub/hipcub.hpp>

#include "disp_math.h"
#include "launch.h"
#include "reduce.h"

namespace admp {

constexpr int kRecipBlock = 128;
bool spread_wants_bricks(int na, int ncell);

template <class T>
__device__ __forceinline__ void site_qtot(const Site<T>& s, int lpol, T r[3], T Q[9]) {
  r[0] = s.r[0]; r[1] = s.r[1]; r[2] = s.r[2];
#pragma unroll
  for (int k = 0; k < 9; ++k) Q[k] = s.Q[k];
  if (lpol) { Q[1] += s.U[0]; Q[2] += s.U[1]; Q[3] += s.U[2]; }   // Q_global_tot, admp/pme.py:236
}

// ---- LDS-brick spread -------------------------------------------------------------------------------
// Wave-aggregated counter update: lanes of the wavefront that target the same counter are combined into one
// global atomic (atoms arrive in a spatially coherent order, so a wave touches only a handful of bricks);
// returns this lane's slot (counter value before the add + rank among the lanes sharing the key), -1 if !pred.
__device__ __forceinline__ int wave_agg_add(int* __restrict__ counter, int key, bool pred) {
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

// stencil base indices + brick code of atom i: the compact record written by k_prepare_sites, or recomputed
template <class T>
__device__ __forceinline__ int4 atom_bases(const Site<T>* __restrict__ sites, const int4* __restrict__ bases,
                                           const RecipGeom<T>& g, const BrickGrid& bg, int i) {
  if (bases) return bases[i];
  T r[3] = {sites[i].r[0], sites[i].r[1], sites[i].r[2]};
  int b[3];
  for (int d = 0; d < 3; ++d) grid_ref(g, r, d, b[d]);
  const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
  return make_int4(b[0], b[1], b[2], brick_code(b, dims, bg));
}

// exclusive prefix sum of n <= 8192 ints by ONE workgroup (1024 threads x 8 elements): the brick counts of meshes up to
// 320^3; one dispatch where the library scan takes two
__global__ __launch_bounds__(1024) void k_scan_small(int n, const int* __restrict__ in, int* __restrict__ out) {
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[8], tot = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { const int i = t * 8 + k; v[k] = i < n ? in[i] : 0; tot += v[k]; }
  int inc = tot;                                     // inclusive scan of the per-thread totals within the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off, 64); if (lane >= off) inc += u; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (t < 16) {
    int w = wsum[t];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { const int u = __shfl_up(w, off, 64); if (t >= off) w += u; }
    wsum[t] = w;                                     // inclusive over waves
  }
  __syncthreads();
  int base = inc - tot + (wave ? wsum[wave - 1] : 0);
#pragma unroll
  for (int k = 0; k < 8; ++k) { const int i = t * 8 + k; if (i < n) out[i] = base; base += v[k]; }
}

// mode 0: count the (atom, brick) entries per brick; mode 1: write them (counter = running offsets)
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_bin(int na, const Site<T>* __restrict__ sites, RecipGeom<T> g, BrickGrid bg,
                                             int* __restrict__ counter, int* __restrict__ entries,
                                             const int* __restrict__ list, const int4* __restrict__ bases,
                                             const int* __restrict__ start) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  const int i = slot < na ? (list ? list[slot] : slot) : 0;
  int b[3][2] = {{0, 0}, {0, 0}, {0, 0}}, n[3] = {0, 0, 0};
  if (slot < na) {
    const int code = atom_bases(sites, bases, g, bg, i).w;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      b[d][0] = (code >> (9 * d)) & 511;
      b[d][1] = b[d][0] + 1 == bg.nb[d] ? 0 : b[d][0] + 1;   // periodic wrap
      n[d] = 1 + ((code >> (27 + d)) & 1);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {      // wave-uniform trip count: every lane takes part in the ballots
    const int x = e & 1, y = (e >> 1) & 1, z = e >> 2;
    const bool pred = x < n[0] && y < n[1] && z < n[2];
    const int cid = (b[0][x] * bg.nb[1] + b[1][y]) * bg.nb[2] + b[2][z];
    const int slot = wave_agg_add(counter, cid, pred);
    if (MODE == 1 && pred) entries[start[cid] + slot] = i;
  }
}

// One workgroup per brick: each listed atom adds the part of its stencil that falls inside the brick into a
// 16^3 LDS tile; the tile is then stored once -- no memset, no global atomics.  The spline weights are
// recomputed per (atom, brick) entry: ~0.5 kflop against up to 216 LDS atomics.
//
// The tile is FIXED POINT: integer LDS atomics run at twice (64-bit) to four times (32-bit) the rate of ds_add_f64 on the
// scattered addresses of this kernel (tools/ubench/lds_atomics_lanes.hip: ds_add_f32 193 CU-cycles per wave instruction,
// ds_add_f64 27.3, ds_add_u64 13.7, ds_add_u32 6.3), and integer sums do not depend on the order of the adds: the mesh is
// bitwise reproducible.  A first pass over the brick's entries finds bmax = max |q| + |c1|_1 + |c2|_1 of the folded
// multipoles; every spline weight and derivative is <= 1 in magnitude and a mesh word receives at most one term per entry,
// so no term exceeds bmax and no word bmax * cnt.
//   f64 meshes: 64-bit words, scale = the power of two that keeps 8 bmax cnt below 2^62 and bmax below 2^50 (the terms are
//               turned into integers by the 2^52 + 2^51 trick: one f64 add and a 64-bit subtract); finer than f64 on O(1).
//   f32 meshes: 32-bit words (round 3).  Round 3's scale kept bmax * cnt below 2^30 -- the worst case of EVERY entry of the
//               brick landing on one word with weight 1.  That quantum (2^-30 of 1.56 x 580 for liquid water at 1M atoms,
//               i.e. 2e-6 absolute per TERM, 14 terms per word) was 4e-5 of the mesh values: the f32 forces at 1M atoms went
//               from 2.0e-5 to 8.3e-5 of the f64 ones (round-3 verdict).  Round 4 bounds what a word can actually receive:
//                 (a) a term is q M M M + c1 (M' M M) + c2 (M'' M M | M' M' M) with max M = 0.55, max |M'| = 0.46004,
//                     max |M''| = 1 for order 6 (attained at u = 3; checked numerically): bm_e = 0.1664 |q| + 0.1392 |c1|_1
//                     + 0.3025 |c2|_1 instead of |q| + |c1|_1 + |c2|_1;
//                 (b) only entries whose stencil covers a word add to it: the stencil bases of the entries, relative to the
//                     brick (-5 .. 15 per axis), are counted into 7^3 cells of 3 positions (integer LDS adds: the bound
//                     does not depend on their order, so the mesh stays bitwise reproducible); the bases that reach a
//                     word span 6 positions = at most 3 cells per axis, so the largest 3 x 3 x 3 block count times the
//                     largest bm_e bounds sum_e |term_e| of every word.
//               Liquid water at 1M atoms: 5.5 x from (a), ~6 x from (b): the quantum drops 30-fold, below f32 round-off of
//               the values (force error of the f32 path against f64 at 1M atoms back at 2.0e-5, bench.py precision_check).
//
// The kernel is VALU bound (round 3: without its atomics it takes the same time; SQ_ACTIVE_INST_VALU 85 % of the SIMD
// cycles), so the work per entry is cut by what the entry carries: most sites of a force field are bare charges (the
// hydrogens of the water model: 2/3 of the entries), and the entries of an SCF increment are bare dipoles.  The entries of
// a chunk are counting-sorted by class (charge only | dipole only | general) in LDS and handed to the waves 64 in sorted
// order at a time, so that a wave runs one form:
//                      per entry                     per (x, y) row          per point
//   general            3 x (M, M', M''), fold        6 FMA (regrouped:       3 FMA + convert + add
//   dipole only        3 x (M, M'), fold of d        3       row factors     2 FMA + ...
//   charge only        3 x M                         1       per x)          1 MUL + ...
#ifndef ADMP_BRICK_ROW
#define ADMP_BRICK_ROW 17
#endif
#ifndef ADMP_SPREAD_CLASSES
#define ADMP_SPREAD_CLASSES 1     // 0: every entry takes the general form (A/B)
#endif
constexpr int kBrickRow = ADMP_BRICK_ROW;   // z-row pitch of the LDS tile in words (17: bank skew)
constexpr int kBrickChunk = 2048;           // entries sorted at a time
template <class T> struct BrickWord { using type = unsigned long long; };
template <> struct BrickWord<float> { using type = unsigned; };
__device__ __forceinline__ unsigned long long fixed_bits(double v) {     // v already scaled, |v| < 2^51
  const double t = v + 6755399441055744.0;                                // 2^52 + 2^51: the integer sits in the mantissa
  return (unsigned long long)__double_as_longlong(t) - 0x4338000000000000ull;
}
__device__ __forceinline__ unsigned fixed_bits(float v) {                // v already scaled, |v| < 2^30: floor(v + 0.5)
  int i;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(i) : "v"(v));
  return (unsigned)i;
}
__device__ __forceinline__ double brick_value(unsigned long long w, double inv) { return (double)(long long)w * inv; }
__device__ __forceinline__ float brick_value(unsigned w, double inv) { return (float)(int)w * (float)inv; }

enum { SK_GENERAL = 0, SK_DIPOLE = 1, SK_CHARGE = 2 };
// what an entry carries: SK_CHARGE if it has no dipole (permanent + induced) and no quadrupole, SK_DIPOLE if it has
// neither charge nor quadrupole
template <class T>
__device__ __forceinline__ int site_kind(const Site<T>& s, int lpol) {
  T d = m_abs(s.Q[1]) + m_abs(s.Q[2]) + m_abs(s.Q[3]);
  if (lpol) d += m_abs(s.U[0]) + m_abs(s.U[1]) + m_abs(s.U[2]);
  const T q2 = m_abs(s.Q[4]) + m_abs(s.Q[5]) + m_abs(s.Q[6]) + m_abs(s.Q[7]) + m_abs(s.Q[8]);
  if (q2 == T(0) && d == T(0)) return SK_CHARGE;
  if (q2 == T(0) && s.Q[0] == T(0)) return SK_DIPOLE;
  return SK_GENERAL;
}

// one entry (a site and this brick) into the tile
template <class T, int KIND, class W>
__device__ __forceinline__ void brick_add_entry(W* __restrict__ tile, const Site<T>& site, int lpol, const RecipGeom<T>& g,
                                                const int lo[3], const int n[3], T scale) {
  const T r[3] = {site.r[0], site.r[1], site.r[2]};
  // The stencil's position is expressed once per entry RELATIVE to the brick (off = base - lo, folded by the period when
  // the stencil reaches the brick across the periodic seam), so that point p of an axis sits at local index off + p with no
  // wrap, the LDS address is "row + constant", and validity is one bit of a 6-bit mask per axis.
  int off[3], ok[3];
  T M[3][6], D1[3][6], D2[3][6];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int base;
    T D3[6];
    const T f = grid_ref(g, r, d, base);
    bspline6(f, M[d], D1[d], D2[d], D3);              // (what a form does not use is never computed: all of it is inlined)
    const int period = d == 0 ? g.wrap0 : g.K[d];
    int o = base - lo[d];
    if (o + 5 < 0) o += period;                      // stencil starts before the seam, ends inside this brick
    else if (o >= n[d]) o -= period;                 // brick at the low end, stencil wraps around from the high end
    off[d] = o;
    int m = 0;
#pragma unroll
    for (int p6 = 0; p6 < 6; ++p6) m |= ((unsigned)(o + p6) < (unsigned)n[d]) << p6;
    ok[d] = m;
  }
  if (!(ok[0] && ok[1] && ok[2])) return;
  T q = T(0), c1[3] = {T(0), T(0), T(0)}, c2[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  if (KIND == SK_CHARGE) {
    q = scale * site.Q[0];
  } else {
    T Q[9];
    Q[0] = KIND == SK_DIPOLE ? T(0) : site.Q[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) Q[k] = site.Q[k] + (lpol ? site.U[k - 1] : T(0));      // Q_global_tot, admp/pme.py:236
#pragma unroll
    for (int k = 4; k < 9; ++k) Q[k] = KIND == SK_DIPOLE ? T(0) : site.Q[k];
    fold_multipole(g, Q, c1, c2);
    q = scale * Q[0];                                 // the power-of-two scale goes into the coefficients: exact
#pragma unroll
    for (int k = 0; k < 3; ++k) c1[k] *= scale;
#pragma unroll
    for (int k = 0; k < 6; ++k) c2[k] *= scale;
  }
  // z axis: no branch per point -- an out-of-brick point gets zero weights and a clamped (in-tile) index, so it adds an
  // exact integer 0 to a word of the row; x and y keep their `continue` (they skip 36 / 6 points at a time)
  T wz[6], w1z[6], w2z[6];
  int jz[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const bool in = (ok[2] >> c) & 1;
    wz[c] = in ? M[2][c] : T(0); w1z[c] = in ? D1[2][c] : T(0); w2z[c] = in ? D2[2][c] : T(0);
    const int j = off[2] + c;
    jz[c] = j < 0 ? 0 : (j >= n[2] ? n[2] - 1 : j);
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    if (!((ok[0] >> a) & 1)) continue;
    const int ja = off[0] + a;
    const T m0 = M[0][a], d0 = D1[0][a], e0 = D2[0][a];
    // value(a, b, c) = P0 M(z) + P1 M'(z) + P2 M''(z) with
    //   P0 = m1 A0 + d1 A1 + e1 A2,  P1 = m1 B0 + d1 B1,  P2 = m1 C0          (m1, d1, e1: M, M', M'' of the y axis)
    // -- spread_atom's (spline_math.h) sums regrouped by the y factor, the x factors taken out of the y loop
    const T A0 = q * m0 + c1[0] * d0 + c2[0] * e0, A1 = c1[1] * m0 + c2[3] * d0, A2 = c2[1] * m0;
    const T B0 = c1[2] * m0 + c2[4] * d0, B1 = c2[5] * m0, C0 = c2[2] * m0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      if (!((ok[1] >> b) & 1)) continue;
      const int jb = off[1] + b;
      const T m1 = M[1][b], d1 = D1[1][b], e1 = D2[1][b];
      W* row = tile + (ja * 16 + jb) * kBrickRow;
      if (KIND == SK_CHARGE) {
        const T P0 = m1 * A0;
#pragma unroll
        for (int c = 0; c < 6; ++c) atomicAdd(&row[jz[c]], fixed_bits(P0 * wz[c]));
      } else if (KIND == SK_DIPOLE) {
        const T P0 = m1 * A0 + d1 * A1, P1 = m1 * B0;
#pragma unroll
        for (int c = 0; c < 6; ++c) atomicAdd(&row[jz[c]], fixed_bits(P0 * wz[c] + P1 * w1z[c]));
      } else {
        const T P0 = m1 * A0 + d1 * A1 + e1 * A2, P1 = m1 * B0 + d1 * B1, P2 = m1 * C0;
#pragma unroll
        for (int c = 0; c < 6; ++c) atomicAdd(&row[jz[c]], fixed_bits(P0 * wz[c] + P1 * w1z[c] + P2 * w2z[c]));
      }
    }
  }
}

template <class T, bool TIGHT>
__global__ __launch_bounds__(256) void k_spread_bricks(const Site<T>* __restrict__ sites, int lpol, RecipGeom<T> g,
                                                       BrickGrid bg, const int* __restrict__ brick_start,
                                                       const int* __restrict__ entries, T* __restrict__ mesh,
                                                       int* __restrict__ clear_a, int* __restrict__ clear_b,
                                                       const int4* __restrict__ bases) {
  using W = typename BrickWord<T>::type;
  __shared__ W tile[16 * 16 * kBrickRow];
  __shared__ unsigned s_bmax;
  __shared__ unsigned short s_order[kBrickChunk];
  __shared__ int s_hist[4];
  __shared__ unsigned s_hist3[344];
  __shared__ unsigned char s_kind[kBrickChunk];     // class of the first chunk's entries, found while pass 1 has their rows
  const int bz = blockIdx.x % bg.nb[2], by = (blockIdx.x / bg.nb[2]) % bg.nb[1], bx = blockIdx.x / (bg.nb[2] * bg.nb[1]);
  const int bb[3] = {bx, by, bz};
  int lo[3], n[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = (bb[d] * g.dim(d)) / bg.nb[d];
    n[d] = ((bb[d] + 1) * g.dim(d)) / bg.nb[d] - lo[d];
  }
  for (int t = threadIdx.x; t < 16 * 16 * kBrickRow; t += 256) tile[t] = W(0);
  if (sizeof(T) == 4 && TIGHT)
    for (int t = threadIdx.x; t < 344; t += 256) s_hist3[t] = 0u;       // (word 343: the largest block count)
  if (threadIdx.x == 0) s_bmax = 0u;
  __syncthreads();
  const int beg = brick_start[blockIdx.x], cnt = brick_start[blockIdx.x + 1] - beg;
  // pass 1: magnitude bound of this brick's folded multipoles -> fixed-point scale
  // |c1|_1 <= amax |d|_1 and |c2|_1 <= 2 amax^2 |Theta/3|_1 with amax = the largest row (or column) sum of |Aop|
  // (fold_multipole): a bound from norms costs a dozen instructions per entry instead of the fold itself
  T amax = T(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    amax = fmax(amax, m_abs(g.Aop[j]) + m_abs(g.Aop[3 + j]) + m_abs(g.Aop[6 + j]));
    amax = fmax(amax, m_abs(g.Aop[3 * j]) + m_abs(g.Aop[3 * j + 1]) + m_abs(g.Aop[3 * j + 2]));
  }
  double bmax, bound;
  if (sizeof(T) == 4) {
    // (a) + (b) of the header comment: per-entry bound with the weight maxima, and a histogram of the entries' stencil bases
    // (counts: one LDS add per entry in the loop that finds the maximum -- the largest 3 x 3 x 3 block count times the largest
    // per-entry bound is what a word can receive).  TIGHT = false (the SCF increments: small dipole changes whose rounding
    // is far below the permanent mesh's) keeps (a) x cnt and skips the histogram.
    constexpr float kW0 = 0.55f * 0.55f * 0.55f, kW1 = 0.46004f * 0.55f * 0.55f, kW2 = 0.55f * 0.55f;   // see (a)
    float bm = 0.f;
    for (int e = threadIdx.x; e < cnt; e += 256) {
      T r[3], Q[9];
      const int atom = entries[beg + e];
      site_qtot(sites[atom], lpol, r, Q);
      const T d1 = m_abs(Q[1]) + m_abs(Q[2]) + m_abs(Q[3]);
      const T q2 = m_abs(Q[4]) + m_abs(Q[5]) + m_abs(Q[6]) + m_abs(Q[7]) + m_abs(Q[8]);
      if (e < kBrickChunk)      // (site_kind of the row: Q already holds Q + U)
        s_kind[e] = (unsigned char)((q2 == T(0) && d1 == T(0)) ? SK_CHARGE : ((q2 == T(0) && Q[0] == T(0)) ? SK_DIPOLE : SK_GENERAL));
      bm = fmaxf(bm, (float)(T(kW0) * m_abs(Q[0]) + T(kW1) * amax * d1 + T(kW2) * T(2) * amax * amax * q2));
      if (TIGHT) {
        int cl = 0;
        bool in = true;
        // (the stencil record of the atom when the caller has them -- k_prepare_sites / k_atom_bases wrote them with the same
        // grid_ref: 16 bytes the binning has just read, instead of the position words of the 80-byte row)
        int4 rec = make_int4(0, 0, 0, 0);
        if (bases) rec = bases[atom];
#pragma unroll
        for (int d = 0; d < 3; ++d) {                      // the stencil's base relative to the brick, as brick_add_entry sees it
          int base;
          if (bases) base = d == 0 ? rec.x : (d == 1 ? rec.y : rec.z);
          else (void)grid_ref(g, r, d, base);
          const int period = d == 0 ? g.wrap0 : g.K[d];
          int o = base - lo[d];
          if (o + 5 < 0) o += period;
          else if (o >= n[d]) o -= period;
          in = in && o + 5 >= 0 && o < n[d];
          cl = cl * 7 + (o + 5) / 3;
        }
        if (in) atomicAdd(&s_hist3[cl], 1u);               // (an entry that misses the brick adds no term)
      }
    }
    if (bm > 0.f) atomicMax(&s_bmax, __float_as_uint(bm * 1.0001f));    // non-negative floats order like their bits
    __syncthreads();
    bmax = (double)__uint_as_float(s_bmax);
    if (TIGHT) {
      if (threadIdx.x < 125) {
        const int wx = threadIdx.x / 25, wy = (threadIdx.x / 5) % 5, wz = threadIdx.x % 5;
        unsigned sum = 0u;
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b)
            for (int c = 0; c < 3; ++c) sum += s_hist3[((wx + a) * 7 + wy + b) * 7 + wz + c];
        atomicMax(&s_hist3[343], sum);
      }
      __syncthreads();
      const unsigned most = s_hist3[343];
      bound = bmax * (double)(most > 0u ? most : 1u);
    } else {
      bound = bmax * (double)(cnt > 0 ? cnt : 1);
    }
  } else {
    float bm = 0.f;
    for (int e = threadIdx.x; e < cnt; e += 256) {
      T r[3], Q[9];
      site_qtot(sites[entries[beg + e]], lpol, r, Q);
      const T d1 = m_abs(Q[1]) + m_abs(Q[2]) + m_abs(Q[3]);
      const T q2 = m_abs(Q[4]) + m_abs(Q[5]) + m_abs(Q[6]) + m_abs(Q[7]) + m_abs(Q[8]);
      bm = fmaxf(bm, (float)(m_abs(Q[0]) + amax * d1 + T(2) * amax * amax * q2));
      if (e < kBrickChunk)
        s_kind[e] = (unsigned char)((q2 == T(0) && d1 == T(0)) ? SK_CHARGE : ((q2 == T(0) && Q[0] == T(0)) ? SK_DIPOLE : SK_GENERAL));
    }
    if (bm > 0.f) atomicMax(&s_bmax, __float_as_uint(bm * 1.0001f));    // non-negative floats order like their bits
    __syncthreads();
    bmax = (double)__uint_as_float(s_bmax);
    bound = bmax * (double)(cnt > 0 ? cnt : 1);
  }
  int ex = 20;
  if (bmax > 0.0) {
    if (sizeof(T) == 4) {
      ex = 29 - ilogb(bound);                            // f32: 2^ex * bound < 2^30, no 32-bit word (and no term) overflows
    } else {
      ex = 58 - ilogb(bound);                            // f64: 2^ex * 8 bound < 2^62, no word can overflow ...
      const int e1 = 49 - ilogb(bmax);                   // ... and 2^ex * bmax < 2^50: every term fits the mantissa trick
      ex = ex < e1 ? ex : e1;
      ex = ex > 60 ? 60 : ex;
    }
  }
  const T scale = (T)ldexp(1.0, ex);
  const double inv_scale = ldexp(1.0, -ex);
  const int lane = threadIdx.x & 63;
  for (int c0 = 0; c0 < cnt; c0 += kBrickChunk) {
    const int nch = min(kBrickChunk, cnt - c0);
    // counting sort of the chunk by class: LDS atomics on three counters hand out the places within a class
    __syncthreads();                                   // the previous chunk's order is no longer read
    if (threadIdx.x < 4) s_hist[threadIdx.x] = 0;
    __syncthreads();
    int kind[kBrickChunk / 256], slot[kBrickChunk / 256];
#pragma unroll
    for (int k = 0; k < kBrickChunk / 256; ++k) {
      const int e = threadIdx.x + 256 * k;
      kind[k] = 0; slot[k] = 0;
      if (e < nch) {
        kind[k] = !ADMP_SPREAD_CLASSES ? SK_GENERAL : (c0 == 0 ? (int)s_kind[e] : site_kind(sites[entries[beg + c0 + e]], lpol));
        slot[k] = atomicAdd(&s_hist[kind[k]], 1);
      }
    }
    __syncthreads();
    const int n_gen = s_hist[SK_GENERAL], n_dip = s_hist[SK_DIPOLE];
#pragma unroll
    for (int k = 0; k < kBrickChunk / 256; ++k) {
      const int e = threadIdx.x + 256 * k;
      if (e < nch) s_order[(kind[k] == SK_GENERAL ? 0 : (kind[k] == SK_DIPOLE ? n_gen : n_gen + n_dip)) + slot[k]] = (unsigned short)e;
    }
    __syncthreads();
    // 64 consecutive places of the sorted chunk per wave instruction: one class, except where two classes meet (that
    // group takes the general form, which is right for every entry)
    const int groups = (nch + 63) >> 6;
    for (int grp = threadIdx.x >> 6; grp < groups; grp += 4) {
      const int first = grp << 6, last = min(first + 64, nch);
      const int place = first + lane;
      if (place >= nch) continue;
      const Site<T>& site = sites[entries[beg + c0 + s_order[place]]];
      if (first >= n_gen + n_dip) brick_add_entry<T, SK_CHARGE>(tile, site, lpol, g, lo, n, scale);
      else if (first >= n_gen && last <= n_gen + n_dip) brick_add_entry<T, SK_DIPOLE>(tile, site, lpol, g, lo, n, scale);
      else brick_add_entry<T, SK_GENERAL>(tile, site, lpol, g, lo, n, scale);
    }
  }
  __syncthreads();
  const int nyz = n[1] * n[2], ntot = n[0] * nyz;
  if (n[1] == 16 && n[2] == 16) {       // whole 16 x 16 rows (mesh sizes divisible by 16): shifts instead of two divisions per word
    for (int t = threadIdx.x; t < ntot; t += 256) {
      const int ja = t >> 8, jb = (t >> 4) & 15, jc = t & 15;
      mesh[((long)(lo[0] + ja) * g.K[1] + (lo[1] + jb)) * g.K[2] + lo[2] + jc] =
          (T)brick_value(tile[(ja * 16 + jb) * kBrickRow + jc], inv_scale);
    }
  } else {
    const float inv_yz = 1.0f / (float)nyz, inv_z = 1.0f / (float)n[2];
    for (int t = threadIdx.x; t < ntot; t += 256) {
      const int ja = fast_div(t, nyz, inv_yz), rem = t - ja * nyz, jb = fast_div(rem, n[2], inv_z), jc = rem - jb * n[2];
      mesh[((long)(lo[0] + ja) * g.K[1] + (lo[1] + jb)) * g.K[2] + lo[2] + jc] =
          (T)brick_value(tile[(ja * 16 + jb) * kBrickRow + jc], inv_scale);
    }
  }
  // this brick's binning counters are consumed: clear them for the next binning (no memset dispatches per step)
  if (threadIdx.x == 0 && clear_a) { clear_a[blockIdx.x] = 0; clear_b[blockIdx.x] = 0; }
}

// ---- column-task spread of an entry list into the brick's LDS tile (scan kernel; the binned kernel of large systems
// keeps one thread per entry: with ~580 entries per brick the column tasks' LDS reads cost more than the bank
// conflicts they avoid -- measured at 1M atoms 0.59 vs 0.31 ms).
// Entries are taken SUB at a time: 4 threads per entry stage its spline weights in LDS (one axis each, the fourth folds
// the multipoles), then every thread takes (entry, x, y) columns and issues the six z adds.  Lanes of a wavefront then
// hit runs of consecutive tile words instead of 64 unrelated atoms' points; the z-rows of the tile are padded to 17
// words so that the 36 columns of an entry fall into different LDS banks.
constexpr int kTileRow = 17;
constexpr int kTileWords = 16 * 16 * kTileRow;

template <class T, int SUB, class GetEntry>
__device__ __forceinline__ void spread_entry_list(double* tile, T (*wts)[64], int (*ebase)[3], int ne, GetEntry entry,
                                                  const Site<T>* __restrict__ sites, int lpol, const RecipGeom<T>& g,
                                                  const int lo[3], const int n[3]) {
  for (int sub = 0; sub < ne; sub += SUB) {
    const int cnt = min(SUB, ne - sub);
    if (threadIdx.x < 4 * cnt) {
      const int e = threadIdx.x >> 2, part = threadIdx.x & 3;
      const Site<T>& site = sites[entry(sub + e)];
      T* w = wts[e];
      if (part < 3) {
        T r[3] = {site.r[0], site.r[1], site.r[2]};
        int base;
        T M[6], D1[6], D2[6], D3[6];
        const T f = grid_ref(g, r, part, base);
        bspline6(f, M, D1, D2, D3);
        ebase[e][part] = base;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          w[part * 18 + k] = M[k];
          w[part * 18 + 6 + k] = D1[k];
          w[part * 18 + 12 + k] = D2[k];
        }
      } else {
        T r[3], Q[9], c1[3], c2[6];
        site_qtot(site, lpol, r, Q);
        fold_multipole(g, Q, c1, c2);
        w[54] = Q[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) w[55 + k] = c1[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) w[58 + k] = c2[k];
      }
    }
    __syncthreads();
    for (int task = threadIdx.x; task < cnt * 36; task += 256) {
      const int e = task / 36, ab = task - e * 36, a = ab / 6, b = ab - a * 6;
      const int ja = wrap_add(ebase[e][0], a, g.wrap0) - lo[0];
      const int jb = wrap_add(ebase[e][1], b, g.K[1]) - lo[1];
      if ((unsigned)ja >= (unsigned)n[0] || (unsigned)jb >= (unsigned)n[1]) continue;
      const T* w = wts[e];
      const T m0 = w[a], d0 = w[6 + a], e0 = w[12 + a];
      const T m1 = w[18 + b], d1 = w[24 + b], e1 = w[30 + b];
      const T mm = m0 * m1;
      const T P0 = w[54] * mm + w[55] * d0 * m1 + w[56] * m0 * d1 + w[58] * e0 * m1 + w[59] * m0 * e1 + w[61] * d0 * d1;
      const T P1 = w[57] * mm + w[62] * d0 * m1 + w[63] * m0 * d1;
      const T P2 = w[60] * mm;
      double* row = tile + (ja * 16 + jb) * kTileRow;
      const int bc = ebase[e][2];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int jc = wrap_add(bc, c, g.K[2]) - lo[2];
        if ((unsigned)jc < (unsigned)n[2])
          atomicAdd(&row[jc], (double)(P0 * w[36 + c] + P1 * w[42 + c] + P2 * w[48 + c]));
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void brick_range(const BrickGrid& bg, int dim0, int dim1, int dim2, int lo[3], int n[3]) {
  const int bz = blockIdx.x % bg.nb[2], by = (blockIdx.x / bg.nb[2]) % bg.nb[1], bx = blockIdx.x / (bg.nb[2] * bg.nb[1]);
  const int bb[3] = {bx, by, bz}, dims[3] = {dim0, dim1, dim2};
  for (int d = 0; d < 3; ++d) {
    lo[d] = (bb[d] * dims[d]) / bg.nb[d];
    n[d] = ((bb[d] + 1) * dims[d]) / bg.nb[d] - lo[d];
  }
}
template <class T>
__device__ __forceinline__ void store_tile(const double* tile, const RecipGeom<T>& g, const int lo[3], const int n[3],
                                           T* __restrict__ mesh) {
  const int nyz = n[1] * n[2], ntot = n[0] * nyz;
  const float inv_yz = 1.0f / (float)nyz, inv_z = 1.0f / (float)n[2];
  for (int t = threadIdx.x; t < ntot; t += 256) {
    const int ja = fast_div(t, nyz, inv_yz), rem = t - ja * nyz, jb = fast_div(rem, n[2], inv_z), jc = rem - jb * n[2];
    mesh[((long)(lo[0] + ja) * g.K[1] + (lo[1] + jb)) * g.K[2] + lo[2] + jc] = (T)tile[(ja * 16 + jb) * kTileRow + jc];
  }
}

// Small systems (a few thousand atoms): binning in separate launches costs more than it saves and global f64 atomics
// run at ~20 G/s (the mesh is shared by the 8 XCDs), so each brick's workgroup scans the stencil records of ALL atoms
// itself, keeps the ones whose stencil touches the brick, and spreads them as above.  One launch, no memset, no
// global atomics.
constexpr int kScanChunk = 3072;   // atoms scanned per round (bounds the LDS entry list; with the tile and the staged weights
                                   // just under the 64 KB a workgroup may declare: 1024 waters take one round)
constexpr int kScanSub = 32;       // entries whose weights are staged at a time

template <class T>
__global__ __launch_bounds__(256) void k_spread_scan(int na, const Site<T>* __restrict__ sites, int lpol, RecipGeom<T> g,
                                                     BrickGrid bg, T* __restrict__ mesh, const int* __restrict__ list,
                                                     const int4* __restrict__ bases) {
  sites += (size_t)blockIdx.y * na;                                            // batch of scalar channels (dispersion)
  mesh += (size_t)blockIdx.y * ((size_t)g.nloc0 * g.K[1] * g.K[2]);
  __shared__ double tile[kTileWords];
  __shared__ int ents[kScanChunk];
  __shared__ int nent;
  __shared__ T wts[kScanSub][64];
  __shared__ int ebase[kScanSub][3];
  int lo[3], n[3];
  brick_range(bg, g.dim(0), g.dim(1), g.dim(2), lo, n);
  for (int t = threadIdx.x; t < kTileWords; t += 256) tile[t] = 0.0;
  for (int c0 = 0; c0 < na; c0 += kScanChunk) {
    if (threadIdx.x == 0) nent = 0;
    __syncthreads();
    const int cend = min(na, c0 + kScanChunk);
    // all records of the round are fetched before the first test (12 independent loads per thread in flight)
    constexpr int kPer = kScanChunk / 256;
    int idx[kPer];
    int4 rec[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int s = c0 + (int)threadIdx.x + 256 * r;
      idx[r] = s < cend ? (list ? list[s] : s) : -1;
    }
#pragma unroll
    for (int r = 0; r < kPer; ++r) rec[r] = idx[r] >= 0 ? atom_bases(sites, bases, g, bg, idx[r]) : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      // the stencil [base, base+5] (periodic) meets this brick's [lo, lo+n) iff, with u = base - lo (mod dim),
      // u < n or u + 5 >= dim -- no integer division in the scan
      const int base[3] = {rec[r].x, rec[r].y, rec[r].z};
      bool hit = idx[r] >= 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        int u = base[d] - lo[d];
        if (u < 0) u += g.dim(d);
        hit = hit && (u < n[d] || u + 5 >= g.dim(d));
      }
      if (hit) ents[atomicAdd(&nent, 1)] = idx[r];
    }
    __syncthreads();
    spread_entry_list<T, kScanSub>(tile, wts, ebase, nent, [&](int k) { return ents[k]; }, sites, lpol, g, lo, n);
  }
  __syncthreads();
  store_tile(tile, g, lo, n, mesh);
}

// Small systems (too few atoms to fill the chip brick by brick, and launch-latency bound): global float
// atomics, 8-lane groups, lanes 0..5 each spread one x-plane (36 points) of the atom's stencil.
template <class T>
__global__ __launch_bounds__(256) void k_spread_planes(int na, const Site<T>* __restrict__ sites, int lpol,
                                                       RecipGeom<T> g, T* __restrict__ mesh,
                                                       const int* __restrict__ list) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int slot = t >> 3, a = t & 7;
  if (slot >= na || a >= 6) return;
  const int i = list ? list[slot] : slot;
  T r[3], Q[9];
  site_qtot(sites[i], lpol, r, Q);
  Stencil<T> st;
  st.init(g, r);
  T c1[3], c2[6];
  fold_multipole(g, Q, c1, c2);
  T m0 = 0, d0 = 0, e0 = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (k == a) { m0 = st.M[0][k]; d0 = st.D1[0][k]; e0 = st.D2[0][k]; }
  const int ia = wrap_add(st.base[0], a, g.wrap0);
  const T q = Q[0];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    const int ib = wrap_add(st.base[1], b, g.K[1]);
    const T m1 = st.M[1][b], d1 = st.D1[1][b], e1 = st.D2[1][b];
    const T mm = m0 * m1;
    const T P0 = q * mm + c1[0] * d0 * m1 + c1[1] * m0 * d1 + c2[0] * e0 * m1 + c2[1] * m0 * e1 + c2[3] * d0 * d1;
    const T P1 = c1[2] * mm + c2[4] * d0 * m1 + c2[5] * m0 * d1;
    const T P2 = c2[2] * mm;
    const long row = ((long)ia * g.K[1] + ib) * g.K[2];
#pragma unroll
    for (int c = 0; c < 6; ++c)
      atomicAdd(&mesh[row + wrap_add(st.base[2], c, g.K[2])], P0 * st.M[2][c] + P1 * st.D1[2][c] + P2 * st.D2[2][c]);
  }
}

// G table over the r2c half spectrum [K1][K2][K3/2+1].  Default: frequencies are assigned axis by axis
// (k_d <- mesh axis d).  ref_order != 0 reproduces the reference's table exactly (settings.REFERENCE_KPOINT_ORDER):
// its meshgrid(kz, kx, ky) with 'xy' indexing (recip.py:339-340) has shape (K1, K3, K2), and row t of the flattened
// table -- used for element t of the flattened (K1, K2, K3) spectrum (recip.py:410-426) -- is
// (kz[j], kx[i], ky[k]) with t = (i K3 + j) K2 + k; theta_k takes the same integer triple column by column against
// N = (K1, K2, K3) (recip.py:400-408).  For K1 = K2 = K3 this is "frequencies of mesh axes (1,0,2) in k-columns
// (0,1,2)"; for unequal meshes it also scrambles (j,k).  The resulting factor is not symmetric under k -> -k, which
// the half spectrum needs; since |S(-k)| = |S(k)| for a real mesh, the energy and all its derivatives only see the
// symmetrised factor (G(t) + G(-t)) / 2, which is what is stored.
__device__ inline double gfactor_at(int i0, int i1, int i2, int K0, int K1, int K2, const double* __restrict__ binv,
                                    double volume, double kappa, int which, int ref_order) {
  int m0, m1, m2;
  if (ref_order) {
    const long r = (long)i1 * K2 + i2;
    const int j = (int)(r / K1), k = (int)(r % K1);
    m0 = signed_freq(j, K2); m1 = signed_freq(i0, K0); m2 = signed_freq(k, K1);
  } else {
    m0 = signed_freq(i0, K0); m1 = signed_freq(i1, K1); m2 = signed_freq(i2, K2);
  }
  const double tp = 6.283185307179586;
  const double kx = tp * (m0 * binv[0] + m1 * binv[3] + m2 * binv[6]);
  const double ky = tp * (m0 * binv[1] + m1 * binv[4] + m2 * binv[7]);
  const double kz = tp * (m0 * binv[2] + m1 * binv[5] + m2 * binv[8]);
  const double ksq = kx * kx + ky * ky + kz * kz;
  const double th = theta_k_1d(m0, K0) * theta_k_1d(m1, K1) * theta_k_1d(m2, K2);
  if (which == 1)     // Ck_1 (recip.py:434-435), gamma point excluded (recip.py:413-415), x DIELECTRIC (:424)
    return (i0 == 0 && i1 == 0 && i2 == 0) ? 0.0
               : 2.0 * kDielectric * (tp / volume / ksq) * exp(-ksq / (4.0 * kappa * kappa)) / (th * th);
  return 2.0 * disp_ck(which, ksq, kappa, volume) / (th * th);   // dispersion: gamma point included (recip.py:416-426)
}

template <class T>
__global__ void k_gtab(int K0, int K1, int K2, int y0, int ny, const double* __restrict__ binv, double volume,
                       double kappa, int which, int ref_order, T* __restrict__ gtab, const int* __restrict__ fmap, int nhmap) {
  const int nh = fmap ? nhmap : K2 / 2 + 1;
  const long n = (long)K0 * ny * nh;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    int i2 = (int)(t % nh);
    int i1 = y0 + (int)((t / nh) % ny);
    int i0 = (int)(t / ((long)nh * ny));
    if (fmap) { i0 = fmap[i0]; i1 = fmap[K0 + i1]; i2 = fmap[K0 + K1 + i2]; }   // slot -> frequency (pfa_kernels.hip)
    double G = gfactor_at(i0, i1, i2, K0, K1, K2, binv, volume, kappa, which, ref_order);
    if (ref_order)
      G = 0.5 * (G + gfactor_at((K0 - i0) % K0, (K1 - i1) % K1, (K2 - i2) % K2, K0, K1, K2, binv, volume, kappa, which, 1));
    gtab[t] = (T)G;
  }
}

// ---- box gradient, k-space part -----------------------------------------------------------------------------------------
// dG/d(k^2) and the k vector of one spectrum point (same index conventions as gfactor_at)
__device__ inline double gprime_at(int i0, int i1, int i2, int K0, int K1, int K2, const double* __restrict__ binv,
                                   double volume, double kappa, int which, int ref_order, double kv[3]) {
  int m0, m1, m2;
  if (ref_order) {
    const long r = (long)i1 * K2 + i2;
    const int j = (int)(r / K1), k = (int)(r % K1);
    m0 = signed_freq(j, K2); m1 = signed_freq(i0, K0); m2 = signed_freq(k, K1);
  } else {
    m0 = signed_freq(i0, K0); m1 = signed_freq(i1, K1); m2 = signed_freq(i2, K2);
  }
  const double tp = 6.283185307179586;
  kv[0] = tp * (m0 * binv[0] + m1 * binv[3] + m2 * binv[6]);
  kv[1] = tp * (m0 * binv[1] + m1 * binv[4] + m2 * binv[7]);
  kv[2] = tp * (m0 * binv[2] + m1 * binv[5] + m2 * binv[8]);
  const double ksq = kv[0] * kv[0] + kv[1] * kv[1] + kv[2] * kv[2];
  const double th = theta_k_1d(m0, K0) * theta_k_1d(m1, K1) * theta_k_1d(m2, K2);
  if (which == 1) {   // G = 2 D (2 pi / V k^2) exp(-k^2 / 4 kappa^2) / theta^2 ; dG/dk^2 = -G (1/k^2 + 1/(4 kappa^2))
    if (ksq == 0.0 || (i0 == 0 && i1 == 0 && i2 == 0)) return 0.0;
    const double G = 2.0 * kDielectric * (tp / volume / ksq) * exp(-ksq / (4.0 * kappa * kappa)) / (th * th);
    return -G * (1.0 / ksq + 1.0 / (4.0 * kappa * kappa));
  }
  return 2.0 * disp_ck_dksq(which, ksq, kappa, volume) / (th * th);
}

// tk[0..5] += sum over the half spectrum of w dG/dk^2 |S|^2 k_a k_b, (a,b) = xx, yy, zz, xy, xz, yz -- run on the
// spectrum BEFORE k_kspace multiplies it by G.  With E = sum w G(k^2) |S|^2 and k = 2 pi m . box^-1 this is the part of
// dE/dbox that comes from the k vectors: -2 tk . box^-T (assembled on the host, engine.hip).
template <class T>
__global__ __launch_bounds__(256) void k_kspace_virial(int K0, int K1, int K2, const double* __restrict__ binv,
                                                       double volume, double kappa, int which, int ref_order,
                                                       const T* __restrict__ spec, double* tk, int y0, int ny) {
  // spec = [K0][ny][K2/2+1] holding the y rows y0 .. y0+ny-1 (one rank: all of them; a slab rank: its own, transposed layout)
  const int nh = K2 / 2 + 1;
  const long n = (long)K0 * ny * nh;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
    const int i2 = (int)(t % nh);
    const int i1 = y0 + (int)((t / nh) % ny);
    const int i0 = (int)(t / ((long)nh * ny));
    const double re = (double)spec[2 * t], im = (double)spec[2 * t + 1];
    const double w = ((i2 == 0 || ((K2 & 1) == 0 && i2 == K2 / 2)) ? 0.5 : 1.0) * (re * re + im * im);
    double kv[3];
    double gp = gprime_at(i0, i1, i2, K0, K1, K2, binv, volume, kappa, which, ref_order, kv);
    double f = ref_order ? 0.5 * w * gp : w * gp;
    acc[0] += f * kv[0] * kv[0]; acc[1] += f * kv[1] * kv[1]; acc[2] += f * kv[2] * kv[2];
    acc[3] += f * kv[0] * kv[1]; acc[4] += f * kv[0] * kv[2]; acc[5] += f * kv[1] * kv[2];
    if (ref_order) {   // the stored factor is the average over t and its mirror point (k_gtab)
      gp = gprime_at((K0 - i0) % K0, (K1 - i1) % K1, (K2 - i2) % K2, K0, K1, K2, binv, volume, kappa, which, 1, kv);
      f = 0.5 * w * gp;
      acc[0] += f * kv[0] * kv[0]; acc[1] += f * kv[1] * kv[1]; acc[2] += f * kv[2] * kv[2];
      acc[3] += f * kv[0] * kv[1]; acc[4] += f * kv[0] * kv[2]; acc[5] += f * kv[1] * kv[2];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double v = block_reduce_sum<256>(acc[k]);
    if (threadIdx.x == 0) atomicAdd(&tk[k], v);
  }
}

// Box gradient, mesh part: one thread per atom re-gathers its 20 F sums from phi and accumulates recip_box_terms
// (spline_math.h): xw[9] and y[9].  On request only -- the plain loop over the 216 stencil points is good enough.
template <class T>
__global__ __launch_bounds__(128) void k_gather_virial(int na, const Site<T>* __restrict__ sites, int lpol, RecipGeom<T> g,
                                                       const T* __restrict__ phi, double* xw, double* yy,
                                                       const int* __restrict__ list) {
  const int slot = blockIdx.x * 128 + threadIdx.x;
  const int i = (list && slot < na) ? list[slot] : slot;      // (slab rank: its home atoms)
  double ax[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ay[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (slot < na) {
    T r[3], Q[9], F[NF];
    site_qtot(sites[i], lpol, r, Q);
    gather_atom(g, r, [&](long idx) { return phi[idx]; }, F);
    recip_box_terms(g, r, Q, F, ax, ay);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double v = block_reduce_sum<128>(ax[k]);
    if (threadIdx.x == 0 && v != 0.0) atomicAdd(&xw[k], v);
    const double u = block_reduce_sum<128>(ay[k]);
    if (threadIdx.x == 0 && u != 0.0) atomicAdd(&yy[k], u);
  }
}

// spec <- G * spec, E += sum over the FULL spectrum of (G/2)|S|^2 (interior half-spectrum planes count twice)
template <class T>
__global__ __launch_bounds__(256) void k_kspace(int K0, int ny, int K2, const T* __restrict__ gtab,
                                                T* __restrict__ spec, double* energies, int slot) {
  const int nh = K2 / 2 + 1;
  const long n = (long)K0 * ny * nh;
  double e = 0.0;
  // the z index advances with the stride: ONE (64-bit) modulo per thread instead of one per element
  const long t0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
  int i2 = (int)(t0 % nh);
  const int di = (int)(stride % nh);
  for (long t = t0; t < n; t += stride) {
    const T G = gtab[t];
    T re = spec[2 * t], im = spec[2 * t + 1];
    const double w = (i2 == 0 || ((K2 & 1) == 0 && i2 == K2 / 2)) ? 0.5 : 1.0;
    e += w * (double)G * ((double)re * re + (double)im * im);
    spec[2 * t] = re * G;
    spec[2 * t + 1] = im * G;
    i2 += di;
    if (i2 >= nh) i2 -= nh;
  }
  e = block_reduce_sum<256>(e);
  if (threadIdx.x == 0) atomicAdd(&energies[slot], e);
}

// ---- staged gather ---------------------------------------------------------------------------------------------------
// The adjoint of the spread (the reference leaves it to jax.grad): dE/dQ, dE/dr of every atom from phi = c2r(G S).
// Round 1's form (8 lanes per atom, every lane its own splines, 60 shuffles per atom, one lane in eight converting: in the
// history) issued ~1100 instructions per wave of which a quarter were the 36 loads and their sums.  Here a workgroup takes
// 32 atoms with SIX lanes each (192 threads: no idle lanes in the load loop) and
//   0. wave d evaluates the dimension-d spline of the 32 atoms once and leaves the weights in LDS (W4 rows),
//   1. every lane sums its z-index over the 36 (x, y) with weights read from LDS,
//   2. the partial sums go to LDS (over the weight rows, after a barrier) and are folded by 192 threads, 6 words each,
//   3. 32 lanes of one wave convert the sums of the 32 atoms.
constexpr int kGsAtoms = 32, kGsBlock = 6 * kGsAtoms, kGsRow = 19 /* 18 W4 rows + 1: conflict-free 16-byte reads */;

template <class T>
struct GatherStage {
  union {
    W4<T> w[kGsAtoms][kGsRow];
    T part[NF][kGsBlock];
  };
  int base[kGsAtoms][4];
  T sum[kGsAtoms][NF + 1];
};

// step 0; every thread of the workgroup calls it (ends with a barrier)
template <class T, int NORD>
__device__ __forceinline__ void stage_splines(const RecipGeom<T>& g, const Site<T>* __restrict__ sites,
                                              const int* __restrict__ list, int slot0, int na, W4<T> (*w)[kGsRow],
                                              int (*base)[4]) {
  const int d = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), s = threadIdx.x & 63;
  if (d < 3 && s < kGsAtoms && slot0 + s < na) {
    const int i = list ? list[slot0 + s] : slot0 + s;
    const T r[3] = {sites[i].r[0], sites[i].r[1], sites[i].r[2]};
    int b;
    const T f = grid_ref(g, r, d, b);
    T M[6], D1[6], D2[6], D3[6];
    bspline6(f, M, D1, D2, D3);
    base[s][d] = b;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      W4<T> o;
      o.m = M[p]; o.d1 = D1[p];
      o.d2 = NORD > 2 ? D2[p] : T(0); o.d3 = NORD > 2 ? D3[p] : T(0);
      w[s][6 * d + p] = o;
    }
  }
  __syncthreads();
}

// the 20 sums of atom i -> dE/dQ, dE/dr (added to pot / grad), the reciprocal field, the atom's share of the mesh energy
// and, with the field epilogue, the total dE/dU and its magnitude
template <class T, bool ERECIP>
__device__ __forceinline__ void convert_sums(int i, const T* __restrict__ sums, const Site<T>* __restrict__ sites, int lpol,
                                             const RecipGeom<T>& g, T* __restrict__ pot, T* __restrict__ grad,
                                             T* __restrict__ fld, const FieldFin<T>& ff, T& er, double& fm) {
  T r[3], Q[9], S[NF];
  site_qtot(sites[i], lpol, r, Q);
#pragma unroll
  for (int k = 0; k < NF; ++k) S[k] = sums[k];
  T P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gr[3] = {0, 0, 0};
  unfold_potential(g, Q, S, P, gr);
#pragma unroll
  for (int k = 0; k < 9; ++k) pot[9 * i + k] += P[k];
  if (fld) { fld[3 * i] = P[2]; fld[3 * i + 1] = P[3]; fld[3 * i + 2] = P[1]; }   // harmonic (z,x,y) -> cartesian
  if (grad) {
    grad[3 * i] += gr[0]; grad[3 * i + 1] += gr[1]; grad[3 * i + 2] += gr[2];
  }
  if (ERECIP) {       // the mesh energy is a quadratic form of the multipoles: E = 1/2 sum_h Q_h dE/dQ_h
#pragma unroll
    for (int k = 0; k < 9; ++k) er += Q[k] * P[k];
  }
  if (ff.fmax_bits) {   // kernel-uniform
    const T al = ff.pol[i];
    T fx, fy, fz;
    const T fr[3] = {P[2], P[3], P[1]};
    total_field(sites[i], al, ff.Ucart + 3 * i, ff.fld_pair + 3 * i, fr, ff.kappa, fx, fy, fz);
    ff.field[3 * i] = fx; ff.field[3 * i + 1] = fy; ff.field[3 * i + 2] = fz;
    if (al > T(0.001)) fm = fmax(fm, fmax(fabs((double)fx), fmax(fabs((double)fy), fabs((double)fz))));
  }
}

// FIN (small molecular systems, round 4): the workgroup takes a run of WHOLE frame groups (Topology::gath_blk, <= 32 atoms) and
// its epilogue does the closing kernel's work for them -- self + penalty energies, frame adjoint, dE/dQ_local (k_finish_rows'
// arithmetic: every frame once, by its own site; positions in and gradient contributions out through LDS) -- so a step of a
// 3072-atom system has one dispatch fewer (the closing kernel took 10 us of a 0.21 ms step, all of it latency).
template <class T>
struct GatherFin {
  T spos[kGsAtoms][3];
  T sg[kGsAtoms][9];
  int sidx[kGsAtoms][3];
};
template <class T, bool ERECIP, bool FIN>
__global__ __launch_bounds__(kGsBlock) void k_gather_staged(int na, const Site<T>* __restrict__ sites, int lpol,
                                                            RecipGeom<T> g, const T* __restrict__ phi,
                                                            T* __restrict__ pot, T* __restrict__ grad,
                                                            const int* __restrict__ list, T* __restrict__ fld,
                                                            FieldFin<T> ff, double* e_recip, Topology top, Box<T> box,
                                                            FinishArgs<T> fin) {
  __shared__ GatherStage<T> L;
  __shared__ GatherFin<T> LF;
  const long blk = xcd_block(blockIdx.x, (unsigned)(FIN ? top.ngathblk : (na + kGsAtoms - 1) / kGsAtoms));
  if (blk < 0) return;                                   // workgroup-uniform
  int slot0 = (int)blk * kGsAtoms;
  if (FIN) { slot0 = top.gath_blk[blk]; na = top.gath_blk[blk + 1]; }      // this run's atoms: slot0 .. na - 1 (list == nullptr)
  stage_splines<T, 4>(g, sites, list, slot0, na, L.w, L.base);
  const int s = threadIdx.x / 6, c = threadIdx.x - 6 * s;
  T F[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) F[k] = T(0);
  if (slot0 + s < na) {
    const int base[3] = {L.base[s][0], L.base[s][1], L.base[s][2]};
    const W4<T> wz = L.w[s][12 + c];
    gather_zcol_w(g, base, &L.w[s][0], &L.w[s][6], wz, wrap_add(base[2], c, g.K[2]), [&](long idx) { return phi[idx]; }, F);
  }
  __syncthreads();                                       // the weight rows are dead: the partial sums take their place
#pragma unroll
  for (int k = 0; k < NF; ++k) L.part[k][threadIdx.x] = F[k];
  __syncthreads();
  for (int t = threadIdx.x; t < NF * kGsAtoms; t += kGsBlock) {
    const int k = t >> 5, a = t & 31;
    const T* q = &L.part[k][6 * a];
    L.sum[a][k] = ((q[0] + q[1]) + (q[2] + q[3])) + (q[4] + q[5]);
  }
  __syncthreads();
  if (!FIN && threadIdx.x >= 64) return;                  // step 3: one wave, lanes 0..31
  const int slot = slot0 + (int)threadIdx.x;
  const bool on = threadIdx.x < kGsAtoms && slot < na;
  T er = T(0);
  double fm = 0.0;
  if (!FIN) {
    if (on) convert_sums<T, ERECIP>(list ? list[slot] : slot, L.sum[threadIdx.x], sites, lpol, g, pot, grad, fld, ff, er, fm);
  } else {
    // the closing epilogue: the whole workgroup stays (barriers below), lanes 0 .. n-1 of the first wave work
    const int t = threadIdx.x, i = slot;
    double eself = 0.0, epen = 0.0;
    T gacc[3] = {T(0), T(0), T(0)};
    Site<T> st;
    if (t < kGsAtoms) LF.sidx[t][0] = LF.sidx[t][1] = LF.sidx[t][2] = -1;
    if (on) {
      st = sites[i];
      LF.spos[t][0] = st.r[0]; LF.spos[t][1] = st.r[1]; LF.spos[t][2] = st.r[2];
    }
    __syncthreads();
    if (on) {
      T r[3], Q[9], S[NF];
      site_qtot(st, lpol, r, Q);
#pragma unroll
      for (int k = 0; k < NF; ++k) S[k] = L.sum[t][k];
      T P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      unfold_potential(g, Q, S, P, gacc);               // reciprocal part of dE/dQ_global and of the gradient
      if (fld) { fld[3 * i] = P[2]; fld[3 * i + 1] = P[3]; fld[3 * i + 2] = P[1]; }
      if (ERECIP) {
#pragma unroll
        for (int k = 0; k < 9; ++k) er += Q[k] * P[k];
      }
      if (ff.fmax_bits) {
        const T al = ff.pol[i];
        T fx, fy, fz;
        const T fr[3] = {P[2], P[3], P[1]};
        total_field(st, al, ff.Ucart + 3 * i, ff.fld_pair + 3 * i, fr, ff.kappa, fx, fy, fz);
        ff.field[3 * i] = fx; ff.field[3 * i + 1] = fy; ff.field[3 * i + 2] = fz;
        if (al > T(0.001)) fm = fmax(fm, fmax(fabs((double)fx), fmax(fabs((double)fy), fabs((double)fz))));
      }
      // total potential: pair part (in pot) + reciprocal part + self term; self and penalty energies (k_finish_rows)
      T f[3];
      self_factors(fin.kappa, f);
      double es = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const T fl = k == 0 ? f[0] : (k < 4 ? f[1] : f[2]);
        es += (double)(fl * Q[k] * Q[k]);
        P[k] += pot[9 * (size_t)i + k];
        pot[9 * (size_t)i + k] = P[k];                   // (pair + reciprocal, as the separate gather leaves it)
        P[k] -= T(2.0 * kDielectric) * fl * Q[k];
      }
      eself = -kDielectric * es;
      if (lpol) {
        T al = fin.pol[i];
        al = al < T(1e-8) ? T(1e-8) : al;
        const double u2 = (double)st.U[0] * st.U[0] + (double)st.U[1] * st.U[1] + (double)st.U[2] * st.U[2];
        epen = kDielectric * 0.5 * u2 / (double)al;
      }
      if (fin.want_grad) {
        int type = top.axis_type[i];
        const int iz = top.axis_idx[3 * i], ix = top.axis_idx[3 * i + 1], iy = top.axis_idx[3 * i + 2];
        if (iz < 0) type = NoAxisType;
        if (type == NoAxisType) {
          if (fin.dQlocal) {
#pragma unroll
            for (int q = 0; q < 9; ++q) fin.dQlocal[9 * (size_t)i + q] = P[q];
          }
        } else {
          const bool usex = type != Zonly, usey = (type == ZBisect || type == ThreeFold);
          const int lz = iz - slot0, lx = usex && ix >= 0 ? ix - slot0 : -1, ly = usey && iy >= 0 ? iy - slot0 : -1;
          T pz[3] = {0, 0, 0}, px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
#pragma unroll
          for (int c2 = 0; c2 < 3; ++c2) {
            pz[c2] = LF.spos[lz][c2];
            if (ix >= 0) px[c2] = LF.spos[ix - slot0][c2];
            if (iy >= 0) py[c2] = LF.spos[iy - slot0][c2];
          }
          FrameWork<T> w;
          local_frame_fwd(type, box, st.r, pz, px, py, w);
          T tau[3], gp[3], gz[3], gx[3], gy[3];
          multipole_torque(P, st.Q, tau);
          local_frame_bwd(type, w, tau, gp, gz, gx, gy);
#pragma unroll
          for (int c2 = 0; c2 < 3; ++c2) {
            gacc[c2] += gp[c2];
            LF.sg[t][c2] = gz[c2]; LF.sg[t][3 + c2] = gx[c2]; LF.sg[t][6 + c2] = gy[c2];
          }
          LF.sidx[t][0] = lz; LF.sidx[t][1] = lx; LF.sidx[t][2] = ly;
          if (fin.dQlocal) {
            T dl[9];
            rot_harm(P, w.X, w.Y, w.Z, dl);
#pragma unroll
            for (int q = 0; q < 9; ++q) fin.dQlocal[9 * (size_t)i + q] = dl[q];
          }
        }
      }
    }
    __syncthreads();
    if (on && fin.want_grad && grad) {
      const int rec = top.grp_of[i], g0 = (rec >> 2) - slot0, gn = (rec & 3) + 1;
      for (int m = g0; m < g0 + gn; ++m) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (LF.sidx[m][k] == t) { gacc[0] += LF.sg[m][3 * k]; gacc[1] += LF.sg[m][3 * k + 1]; gacc[2] += LF.sg[m][3 * k + 2]; }
      }
      grad[3 * (size_t)i] += gacc[0]; grad[3 * (size_t)i + 1] += gacc[1]; grad[3 * (size_t)i + 2] += gacc[2];
    }
    if (threadIdx.x >= 64) return;
    eself = wave_reduce_sum(eself);
    epen = wave_reduce_sum(epen);
    if (threadIdx.x == 0) {
      if (eself != 0.0) atomicAdd(&fin.energies[E_SELF], eself);
      if (lpol && epen != 0.0) atomicAdd(&fin.energies[E_PEN], epen);
    }
  }
  if (ERECIP) {
    const double e = wave_reduce_sum((double)er);
    if (threadIdx.x == 0) atomicAdd(&e_recip[(blockIdx.x >> 3) & (E_PARTS - 1)], 0.5 * e);
  }
  if (ff.fmax_bits) {   // kernel-uniform
    fm = wave_reduce_max(fm);
    if (threadIdx.x == 0 && fm > 0.0) atomicMax(ff.fmax_bits, nonneg_bits(fm));
  }
}

template <class T>
__global__ __launch_bounds__(kGsBlock) void k_gather_field_staged(int na, const Site<T>* __restrict__ sites,
                                                                  RecipGeom<T> g, const T* __restrict__ phi,
                                                                  T* __restrict__ fld, const int* __restrict__ list,
                                                                  const int* __restrict__ n_dev,
                                                                  const int* __restrict__ add_to, FieldFin<T> ff) {
  __shared__ W4<T> w[kGsAtoms][kGsRow];
  __shared__ int sbase[kGsAtoms][4];
  __shared__ T part[3][kGsBlock];
  phi += (size_t)blockIdx.y * ((size_t)g.nloc0 * g.K[1] * g.K[2]);            // batch: same atoms, another mesh
  fld += (size_t)blockIdx.y * 3 * na;
  if (n_dev) na = min(na, *n_dev);
  const long blk = xcd_block(blockIdx.x, (unsigned)((na + kGsAtoms - 1) / kGsAtoms));
  if (blk < 0) return;
  const int slot0 = (int)blk * kGsAtoms;
  stage_splines<T, 2>(g, sites, list, slot0, na, w, sbase);
  const int s = threadIdx.x / 6, c = threadIdx.x - 6 * s;
  T f[3] = {0, 0, 0};
  if (slot0 + s < na) {
    const int base[3] = {sbase[s][0], sbase[s][1], sbase[s][2]};
    const W4<T> wz = w[s][12 + c];
    gather_zcol_field_w(g, base, &w[s][0], &w[s][6], wz, wrap_add(base[2], c, g.K[2]), [&](long idx) { return phi[idx]; }, f);
  }
  part[0][threadIdx.x] = f[0]; part[1][threadIdx.x] = f[1]; part[2][threadIdx.x] = f[2];
  __syncthreads();
  if (threadIdx.x >= 64) return;                         // one wave converts the 32 atoms
  const int slot = slot0 + (int)threadIdx.x;
  const bool on = threadIdx.x < kGsAtoms && slot < na;
  double fm = 0.0;
  if (on) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const T* q = &part[k][6 * threadIdx.x];
      f[k] = ((q[0] + q[1]) + (q[2] + q[3])) + (q[4] + q[5]);
    }
    const T* A = g.Aop;
    const int i = list ? list[slot] : slot;
    const int owner = add_to ? add_to[slot] : i;
    T* o = fld + 3 * (size_t)owner;
    T r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const T v = A[3 * k + 0] * f[0] + A[3 * k + 1] * f[1] + A[3 * k + 2] * f[2];
      r[k] = add_to ? o[k] + v : v;     // compact rows (incremental SCF): accumulate into the owning atom's entry
      o[k] = r[k];
    }
    if (ff.fmax_bits) {   // kernel-uniform: the SCF residual of these atoms and its maximum (k_field_finish's work)
      const T al = ff.pol[owner];
      T fx, fy, fz;
      total_field(ff.sites[owner], al, ff.Ucart + 3 * owner, ff.fld_pair + 3 * owner, r, ff.kappa, fx, fy, fz);
      ff.field[3 * owner] = fx; ff.field[3 * owner + 1] = fy; ff.field[3 * owner + 2] = fz;
      if (al > T(0.001)) fm = fmax(fabs((double)fx), fmax(fabs((double)fy), fabs((double)fz)));
    }
  }
  if (ff.fmax_bits) {
    fm = wave_reduce_max(fm);
    if (threadIdx.x == 0 && fm > 0.0) atomicMax(ff.fmax_bits, nonneg_bits(fm));
  }
}

template <class T>
__global__ __launch_bounds__(256) void k_mesh_add(long n, T* __restrict__ a, const T* __restrict__ b) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) a[t] += b[t];
}
template <class T>
void launch_mesh_add(hipStream_t st, long n, T* a, const T* b) {
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  k_mesh_add<T><<<(int)blocks, 256, 0, st>>>(n, a, b);
}
template void launch_mesh_add<float>(hipStream_t, long, float*, const float*);
template void launch_mesh_add<double>(hipStream_t, long, double*, const double*);

static inline int nblk(int n, int b) { return (n + b - 1) / b; }

#define RC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
template <class T>
int launch_spread(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, BinScratch& bs,
                  T* mesh, const int* list, const int4* bases, int nb, int reuse_bins, int tight) {
  const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
  const BrickGrid bg = make_bricks(dims);
  if (na <= 0) {      // nothing to spread (a slab rank without atoms of this kind): the mesh is zero
    RC(hipMemsetAsync(mesh, 0, sizeof(T) * (size_t)g.nloc0 * g.K[1] * g.K[2] * (size_t)(nb > 0 ? nb : 1), st));
    return 0;
  }
  // the binned brick kernel expresses a stencil as ONE run of local indices per axis, which needs >= 2 bricks per axis
  // (a stencil that wraps around inside a single brick is two runs): meshes of <= 16 points per axis take the scan kernel
  const bool one_brick_axis = bg.nb[0] == 1 || bg.nb[1] == 1 || bg.nb[2] == 1;
  if (!spread_wants_bricks(na, bg.ncell) || one_brick_axis) {
    // measured (f32, reference K rule): 12 288 atoms scan 0.052 / bricks 0.066 / global atomics 0.130 ms; 18 000 atoms
    // 0.070 / 0.083 / 0.189; 30 000 atoms 0.144 / 0.109 -- the scan kernel serves everything below the brick threshold
    static const int scan_max = [] { const char* e = getenv("ADMP_SPREAD_SCAN_MAX"); return e ? atoi(e) : 20000; }();
    if (na <= scan_max || one_brick_axis) {
      k_spread_scan<T><<<dim3(bg.ncell, nb), 256, 0, st>>>(na, sites, lpol, g, bg, mesh, list, bases);
      return 0;
    }
    if (nb != 1) return (int)hipErrorInvalidValue;
    RC(hipMemsetAsync(mesh, 0, sizeof(T) * (size_t)g.nloc0 * g.K[1] * g.K[2], st));
    k_spread_planes<T><<<nblk(na * 8, 256), 256, 0, st>>>(na, sites, lpol, g, mesh, list);
    return 0;
  }
  if (nb != 1) return (int)hipErrorInvalidValue;      // batches exist for the scan kernel only
  if (reuse_bins) {   // same positions as the previous call (next dispersion power): the brick lists are still valid
    if (tight) k_spread_bricks<T, true><<<bg.ncell, 256, 0, st>>>(sites, lpol, g, bg, bs.cell_start, bs.sorted, mesh, nullptr, nullptr, bases);
    else k_spread_bricks<T, false><<<bg.ncell, 256, 0, st>>>(sites, lpol, g, bg, bs.cell_start, bs.sorted, mesh, nullptr, nullptr, nullptr);
    return 0;
  }
  { const int rc = launch_bin_bricks<T>(st, na, sites, g, bs, list, bases); if (rc != 0) return rc; }
  if (tight) k_spread_bricks<T, true><<<bg.ncell, 256, 0, st>>>(sites, lpol, g, bg, bs.cell_start, bs.sorted, mesh, bs.cursor, bs.fillcur, bases);
  else k_spread_bricks<T, false><<<bg.ncell, 256, 0, st>>>(sites, lpol, g, bg, bs.cell_start, bs.sorted, mesh, bs.cursor, bs.fillcur, nullptr);
  bs.counters_zero = true;      // element ncell of both arrays is never written: it stays zero
  return 0;
}
template <class T>
int launch_bin_bricks(hipStream_t st, int na, const Site<T>* sites, const RecipGeom<T>& g, BinScratch& bs, const int* list,
                      const int4* bases) {
  const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
  const BrickGrid bg = make_bricks(dims);
  if (!bs.counters_zero) {
    RC(hipMemsetAsync(bs.cursor, 0, sizeof(int) * (bg.ncell + 1), st));
    RC(hipMemsetAsync(bs.fillcur, 0, sizeof(int) * (bg.ncell + 1), st));
  }
  bs.counters_zero = false;
  if (na > 0) k_bin<T, 0><<<nblk(na, 256), 256, 0, st>>>(na, sites, g, bg, bs.cursor, nullptr, list, bases, nullptr);
  if (bg.ncell + 1 <= 8192) {
    k_scan_small<<<1, 1024, 0, st>>>(bg.ncell + 1, bs.cursor, bs.cell_start);
  } else {
    size_t need = bs.scan_bytes;
    RC(hipcub::DeviceScan::ExclusiveSum(bs.scan_tmp, need, bs.cursor, bs.cell_start, bg.ncell + 1, st));
  }
  if (na > 0) k_bin<T, 1><<<nblk(na, 256), 256, 0, st>>>(na, sites, g, bg, bs.fillcur, bs.sorted, list, bases, bs.cell_start);
  return 0;
}
#undef RC

template <class T>
bool spread_uses_bricks(int na, const RecipGeom<T>& g) {
  const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
  const BrickGrid bg = make_bricks(dims);
  return spread_wants_bricks(na, bg.ncell) && bg.nb[0] > 1 && bg.nb[1] > 1 && bg.nb[2] > 1;
}
template bool spread_uses_bricks<float>(int, const RecipGeom<float>&);
template bool spread_uses_bricks<double>(int, const RecipGeom<double>&);

// The scan kernel costs (bricks x atoms) record tests, the binned kernel two binning passes: bricks win above ~20 000 atoms on
// a whole mesh (measured, see launch_spread) -- and on a slab rank's local mesh, or for the polarizable subset of an SCF
// increment, whenever a brick still gets a few dozen atoms (16 384 polarizable home atoms on 320 local bricks took the scan
// kernel in round 2's rule: 0.076 ms against 0.060 ms for the full spread of 49 152 atoms).
bool spread_wants_bricks(int na, int ncell) {
  const int m = spread_brick_min_atoms();
  return na >= m || (m > 0 && na >= 4096 && (long)na >= 24l * ncell);
}
int spread_brick_min_atoms() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("ADMP_SPREAD_BRICK_MIN");
    v = e ? atoi(e) : 20000;
    if (v < 0) v = 0;
  }
  return v;
}

// bytes of hipcub scan scratch for a mesh (used by the engine to size BinScratch)
size_t spread_scan_bytes(int ncell) {
  size_t need = 0;
  int* p = nullptr;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, need, p, p, ncell + 1, (hipStream_t)0);
  return need + 256;
}
template <class T>
void launch_gtab(hipStream_t st, const int K[3], int y0, int ny, const double* box_inv, double volume, double kappa,
                 int which, T* gtab, int ref_order, const int* fmap, int nh) {
  const long n = (long)K[0] * ny * (fmap ? nh : K[2] / 2 + 1);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  k_gtab<T><<<blocks, 256, 0, st>>>(K[0], K[1], K[2], y0, ny, box_inv, volume, kappa, which, ref_order, gtab, fmap, nh);
}
template <class T>
void launch_kspace(hipStream_t st, const int K[3], int ny, const T* gtab, T* spec, double* energies, int slot) {
  const long n = (long)K[0] * ny * (K[2] / 2 + 1);
  // few, fat workgroups: every workgroup ends in one f64 atomic on the same energy word
  int blocks = (int)((n + 2047) / 2048);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  k_kspace<T><<<blocks, 256, 0, st>>>(K[0], ny, K[2], gtab, spec, energies, slot);
}

template <class T>
void launch_kspace_virial(hipStream_t st, const int K[3], const double* box_inv, double volume, double kappa, int which,
                          int ref_order, const T* spec, double* tk, int y0, int ny) {
  if (ny <= 0) { y0 = 0; ny = K[1]; }
  const long n = (long)K[0] * ny * (K[2] / 2 + 1);
  int blocks = (int)((n + 2047) / 2048);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  k_kspace_virial<T><<<blocks, 256, 0, st>>>(K[0], K[1], K[2], box_inv, volume, kappa, which, ref_order, spec, tk, y0, ny);
}
template <class T>
void launch_gather_virial(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, const T* phi,
                          double* xw, double* yy, const int* list) {
  if (na > 0) k_gather_virial<T><<<(na + 127) / 128, 128, 0, st>>>(na, sites, lpol, g, phi, xw, yy, list);
}

template <class T>
void launch_gather(hipStream_t st, int na, const Site<T>* sites, int lpol, const RecipGeom<T>& g, const T* phi, T* pot,
                   T* grad, const int* list, T* fld, const FieldFin<T>& ff, double* e_recip, const Topology* top,
                   const Box<T>* box, const FinishArgs<T>& fin) {
  if (na <= 0) return;
  if (fin.energies && top && box && top->gath_blk && !list) {      // closing work in the epilogue: one workgroup per run of groups
    const unsigned gs = xcd_grid((unsigned)top->ngathblk);
    if (e_recip)
      k_gather_staged<T, true, true><<<gs, kGsBlock, 0, st>>>(na, sites, lpol, g, phi, pot, grad, list, fld, ff, e_recip, *top, *box, fin);
    else
      k_gather_staged<T, false, true><<<gs, kGsBlock, 0, st>>>(na, sites, lpol, g, phi, pot, grad, list, fld, ff, e_recip, *top, *box, fin);
    return;
  }
  const unsigned gs = xcd_grid((unsigned)nblk(na, kGsAtoms));
  const Topology t0;
  const Box<T> b0 = Box<T>();
  if (e_recip)
    k_gather_staged<T, true, false><<<gs, kGsBlock, 0, st>>>(na, sites, lpol, g, phi, pot, grad, list, fld, ff, e_recip, t0, b0, FinishArgs<T>());
  else
    k_gather_staged<T, false, false><<<gs, kGsBlock, 0, st>>>(na, sites, lpol, g, phi, pot, grad, list, fld, ff, e_recip, t0, b0, FinishArgs<T>());
}
template <class T>
void launch_gather_field(hipStream_t st, int na, const Site<T>* sites, const RecipGeom<T>& g, const T* phi, T* fld,
                         const int* list, int nb, const int* n_dev, const int* add_to, const FieldFin<T>& ff) {
  if (na <= 0) return;
  k_gather_field_staged<T><<<dim3(xcd_grid((unsigned)nblk(na, kGsAtoms)), nb), kGsBlock, 0, st>>>(na, sites, g, phi, fld, list,
                                                                                                n_dev, add_to, ff);
}
#define INST(T)                                                                                                       \
  template int launch_bin_bricks<T>(hipStream_t, int, const Site<T>*, const RecipGeom<T>&, BinScratch&, const int*,   \
                                    const int4*);                                                                     \
  template int launch_spread<T>(hipStream_t, int, const Site<T>*, int, const RecipGeom<T>&, BinScratch&, T*,          \
                                const int*, const int4*, int, int, int);                                              \
  template void launch_gtab<T>(hipStream_t, const int*, int, int, const double*, double, double, int, T*, int,             \
                               const int*, int);                                                                      \
  template void launch_kspace<T>(hipStream_t, const int*, int, const T*, T*, double*, int);                           \
  template void launch_gather<T>(hipStream_t, int, const Site<T>*, int, const RecipGeom<T>&, const T*, T*, T*,        \
                                 const int*, T*, const FieldFin<T>&, double*, const Topology*, const Box<T>*,          \
                                 const FinishArgs<T>&);                                                                \
  template void launch_gather_field<T>(hipStream_t, int, const Site<T>*, const RecipGeom<T>&, const T*, T*,           \
                                       const int*, int, const int*, const int*, const FieldFin<T>&);                  \
  template void launch_kspace_virial<T>(hipStream_t, const int*, const double*, double, double, int, int, const T*,   \
                                        double*, int, int);                                                           \
  template void launch_gather_virial<T>(hipStream_t, int, const Site<T>*, int, const RecipGeom<T>&, const T*, double*, \
                                        double*, const int*);
INST(float)
INST(double)
#undef INST

}  // namespace admp
