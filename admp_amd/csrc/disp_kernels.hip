// Dispersion PME, reciprocal part at scale (reference admp/disp_pme.py:80-123: three scalar reciprocal passes, one per power
// 6 / 8 / 10, each through generate_pme_recip with lmax = 0).  The three channels share the atoms' positions, hence their
// stencils and spline weights; only the coefficient differs.  So the channels go through the spread and the gather TOGETHER:
//   k_atom_bases            stencil base records of all atoms (what the binning and the slab decomposition read)
//   k_spread_bricks_scalar  one binning, one pass over every brick's entries, NCH LDS tiles: the weights of a stencil point
//                           are formed once and added to every channel's tile
//   k_gather_scalar         one pass over the atoms: the mesh index arithmetic and the weights are shared, the channels'
//                           first-derivative sums are combined with the atom's coefficients before the fold and the
//                           result goes straight into the gradient (no per-channel field arrays, no scale-add pass); the
//                           self term (disp_pme.py:254-279) rides along
// (round 2 ran spread / gather_field / scale_add once per channel over 80-byte site rows: 0.65 + 0.39 + 0.21 ms of the
// 2.5 ms dispersion call at 1M atoms).  The meshes themselves still take one transform pair each.
#include "launch.h"
#include "reduce.h"

namespace admp {

template <class T>
__global__ __launch_bounds__(256) void k_atom_bases(int na, const T* __restrict__ pos, RecipGeom<T> g, int4* __restrict__ bases) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= na) return;
  const T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  int b[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) grid_ref(g, r, d, b[d]);
  const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
  bases[i] = make_int4(b[0], b[1], b[2], brick_code(b, dims, make_bricks(dims)));
}
template <class T>
void launch_atom_bases(hipStream_t st, int na, const T* pos, const RecipGeom<T>& g, int4* bases) {
  if (na > 0) k_atom_bases<T><<<(na + 255) / 256, 256, 0, st>>>(na, pos, g, bases);
}

// ---- spread -------------------------------------------------------------------------------------------------------------------
// Fixed-point LDS tiles as in k_spread_bricks (recip_kernels.hip): integer LDS atomics, order-independent sums.  f32 meshes
// take 32-bit words (three 17 KB tiles: three workgroups per CU): a mesh word receives at most ONE term per entry of the
// brick's list and a term is c M(x) M(y) M(z) with M <= 0.55, so |word| <= 0.17 cnt max|c| -- the scale is the power of two
// that keeps that below 2^30; the resolution, 2^-30 of that bound, is ~5e-8 max|c| for the ~600 entries of a brick of liquid
// water, below f32 rounding of the values it is added to.  f64 meshes keep 64-bit words (one channel per launch).
constexpr int kScalarRow = 17;                          // z pitch of a tile in words (bank skew)
constexpr int kScalarTile = 16 * 16 * kScalarRow;
template <class T> struct ScalarWord { using type = unsigned; };
template <> struct ScalarWord<double> { using type = unsigned long long; };
__device__ __forceinline__ unsigned scalar_fixed(float v) { return (unsigned)__float2int_rn(v); }
__device__ __forceinline__ unsigned long long scalar_fixed(double v) {        // |v| < 2^51
  const double t = v + 6755399441055744.0;
  return (unsigned long long)__double_as_longlong(t) - 0x4338000000000000ull;
}
__device__ __forceinline__ float scalar_value(unsigned w, double inv) { return (float)((double)(int)w * inv); }
__device__ __forceinline__ double scalar_value(unsigned long long w, double inv) { return (double)(long long)w * inv; }

extern __shared__ __align__(16) unsigned char scalar_smem[];

template <class T, int NCH>
__global__ __launch_bounds__(256) void k_spread_bricks_scalar(const T* __restrict__ pos, const T* __restrict__ vals, int stride,
                                                              int chan0, RecipGeom<T> g, BrickGrid bg,
                                                              const int* __restrict__ brick_start,
                                                              const int* __restrict__ entries, T* __restrict__ mesh,
                                                              long mesh_stride, int* __restrict__ clear_a,
                                                              int* __restrict__ clear_b) {
  using W = typename ScalarWord<T>::type;
  W* tile = reinterpret_cast<W*>(scalar_smem);          // [NCH][kScalarTile]
  __shared__ unsigned s_bmax[NCH];
  const int bz = blockIdx.x % bg.nb[2], by = (blockIdx.x / bg.nb[2]) % bg.nb[1], bx = blockIdx.x / (bg.nb[2] * bg.nb[1]);
  const int bb[3] = {bx, by, bz};
  int lo[3], n[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = (bb[d] * g.dim(d)) / bg.nb[d];
    n[d] = ((bb[d] + 1) * g.dim(d)) / bg.nb[d] - lo[d];
  }
  for (int t = threadIdx.x; t < NCH * kScalarTile; t += 256) tile[t] = W(0);
  if (threadIdx.x < NCH) s_bmax[threadIdx.x] = 0u;
  __syncthreads();
  const int beg = brick_start[blockIdx.x], cnt = brick_start[blockIdx.x + 1] - beg;
  {   // pass 1: largest coefficient of every channel in this brick
    float bm[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) bm[c] = 0.f;
    for (int e = threadIdx.x; e < cnt; e += 256) {
      const int i = entries[beg + e];
#pragma unroll
      for (int c = 0; c < NCH; ++c) bm[c] = fmaxf(bm[c], (float)m_abs(vals[(long)stride * i + chan0 + c]));
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (bm[c] > 0.f) atomicMax(&s_bmax[c], __float_as_uint(bm[c] * 1.0001f));     // non-negative floats order like their bits
  }
  __syncthreads();
  T scale[NCH];
  double inv_scale[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const double bmax = (double)__uint_as_float(s_bmax[c]), bound = 0.17 * bmax * (double)(cnt > 0 ? cnt : 1);
    int ex = 20;
    if (bmax > 0.0) {
      if (sizeof(T) == 4) {
        ex = 29 - ilogb(bound);                            // 2^ex bound < 2^30: no 32-bit word can overflow
      } else {
        ex = 61 - ilogb(bound);                            // 2^ex bound < 2^62 ...
        const int e1 = 49 - ilogb(bmax);                   // ... and every term below 2^50 (mantissa trick of scalar_fixed)
        ex = ex < e1 ? ex : e1;
        ex = ex > 60 ? 60 : ex;
      }
    }
    scale[c] = (T)ldexp(1.0, ex);
    inv_scale[c] = ldexp(1.0, -ex);
  }
  // entries are read transposed (lane l, pass p -> entry l * rows + p): the 64 entries of an instruction are far apart
  const int rows = (cnt + 63) >> 6, lane = threadIdx.x & 63;
  for (int pass = threadIdx.x >> 6; pass < rows; pass += 4) {
    const int e = lane * rows + pass;
    if (e >= cnt) continue;
    const int i = entries[beg + e];
    const T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    T q[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) q[c] = scale[c] * vals[(long)stride * i + chan0 + c];
    int base[3];
    T M[3][6];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      T D1[6], D2[6], D3[6];
      const T f = grid_ref(g, r, d, base[d]);
      bspline6(f, M[d], D1, D2, D3);
    }
    // the stencil's position relative to the brick, one validity bit per point and axis (see k_spread_bricks)
    int off[3], ok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int period = d == 0 ? g.wrap0 : g.K[d];
      int o = base[d] - lo[d];
      if (o + 5 < 0) o += period;
      else if (o >= n[d]) o -= period;
      off[d] = o;
      int m = 0;
#pragma unroll
      for (int p6 = 0; p6 < 6; ++p6) m |= ((unsigned)(o + p6) < (unsigned)n[d]) << p6;
      ok[d] = m;
    }
    if (!(ok[0] && ok[1] && ok[2])) continue;
    T wz[6];
    int jz[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      wz[c] = ((ok[2] >> c) & 1) ? M[2][c] : T(0);          // an out-of-brick point adds an exact 0 to a word of the row
      const int j = off[2] + c;
      jz[c] = j < 0 ? 0 : (j >= n[2] ? n[2] - 1 : j);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (!((ok[0] >> a) & 1)) continue;
      const int ja = off[0] + a;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        if (!((ok[1] >> b) & 1)) continue;
        const int jb = off[1] + b;
        const T mm = M[0][a] * M[1][b];
        const int rowo = (ja * 16 + jb) * kScalarRow;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const T w = mm * wz[c];
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) atomicAdd(&tile[ch * kScalarTile + rowo + jz[c]], scalar_fixed(q[ch] * w));
        }
      }
    }
  }
  __syncthreads();
  const int nyz = n[1] * n[2], ntot = n[0] * nyz;
  const float inv_yz = 1.0f / (float)nyz, inv_z = 1.0f / (float)n[2];
  for (int t = threadIdx.x; t < ntot; t += 256) {
    int ja, jb, jc;
    if (n[1] == 16 && n[2] == 16) { ja = t >> 8; jb = (t >> 4) & 15; jc = t & 15; }
    else { ja = fast_div(t, nyz, inv_yz); const int rem = t - ja * nyz; jb = fast_div(rem, n[2], inv_z); jc = rem - jb * n[2]; }
    const long mi = ((long)(lo[0] + ja) * g.K[1] + (lo[1] + jb)) * g.K[2] + lo[2] + jc;
    const int ti = (ja * 16 + jb) * kScalarRow + jc;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) mesh[(long)ch * mesh_stride + mi] = scalar_value(tile[ch * kScalarTile + ti], inv_scale[ch]);
  }
  if (threadIdx.x == 0 && clear_a) { clear_a[blockIdx.x] = 0; clear_b[blockIdx.x] = 0; }
}

// ---- typed form (round 4) -------------------------------------------------------------------------------------------------
// When the coefficient rows (C6, C8, C10) of the atoms take only a few distinct values -- atom TYPES: water has two -- the
// structure factor of channel p is a combination of the types' structure factors, S_p = sum_t c_p,t S_t, and what an atom of
// type t gathers is the potential psi_t = sum_p c_p,t phi_p.  So the meshes are per TYPE: an atom spreads its bare stencil
// weights into the mesh of its type (216 LDS atomics instead of 216 per channel), NT meshes go through the transforms
// instead of one per channel, the x pass combines them at every k (fftx_kernels.hip k_fftx_mix), and an atom gathers from
// its type's mesh alone.  Same sums regrouped (the reference spreads c_p,i per channel: admp/disp_pme.py:80-123).
template <class T, int NT>
__global__ __launch_bounds__(256) void k_spread_bricks_typed(const T* __restrict__ pos, const int* __restrict__ types,
                                                             RecipGeom<T> g, BrickGrid bg, const int* __restrict__ brick_start,
                                                             const int* __restrict__ entries, T* __restrict__ mesh,
                                                             long mesh_stride, int* __restrict__ clear_a,
                                                             int* __restrict__ clear_b) {
  using W = typename ScalarWord<T>::type;
  W* tile = reinterpret_cast<W*>(scalar_smem);          // [NT][kScalarTile]
  const int bz = blockIdx.x % bg.nb[2], by = (blockIdx.x / bg.nb[2]) % bg.nb[1], bx = blockIdx.x / (bg.nb[2] * bg.nb[1]);
  const int bb[3] = {bx, by, bz};
  int lo[3], n[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = (bb[d] * g.dim(d)) / bg.nb[d];
    n[d] = ((bb[d] + 1) * g.dim(d)) / bg.nb[d] - lo[d];
  }
  for (int t = threadIdx.x; t < NT * kScalarTile; t += 256) tile[t] = W(0);
  __syncthreads();
  const int beg = brick_start[blockIdx.x], cnt = brick_start[blockIdx.x + 1] - beg;
  // unit sources: a word receives at most cnt * 0.17 (the largest product of three order-6 spline values)
  int ex;
  {
    const double bound = 0.17 * (double)(cnt > 0 ? cnt : 1);
    ex = sizeof(T) == 4 ? 29 - ilogb(bound) : 49;
  }
  const T scale = (T)ldexp(1.0, ex);
  const double inv_scale = ldexp(1.0, -ex);
  const int rows = (cnt + 63) >> 6, lane = threadIdx.x & 63;
  for (int pass = threadIdx.x >> 6; pass < rows; pass += 4) {
    const int e = lane * rows + pass;
    if (e >= cnt) continue;
    const int i = entries[beg + e];
    const T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    int ty = types[i];
    ty = ty < 0 ? 0 : (ty >= NT ? NT - 1 : ty);            // (the host checks the table; never index outside the tiles)
    W* mytile = tile + ty * kScalarTile;
    int base[3];
    T M[3][6];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      T D1[6], D2[6], D3[6];
      const T f = grid_ref(g, r, d, base[d]);
      bspline6(f, M[d], D1, D2, D3);
    }
    int off[3], ok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int period = d == 0 ? g.wrap0 : g.K[d];
      int o = base[d] - lo[d];
      if (o + 5 < 0) o += period;
      else if (o >= n[d]) o -= period;
      off[d] = o;
      int m = 0;
#pragma unroll
      for (int p6 = 0; p6 < 6; ++p6) m |= ((unsigned)(o + p6) < (unsigned)n[d]) << p6;
      ok[d] = m;
    }
    if (!(ok[0] && ok[1] && ok[2])) continue;
    T wz[6];
    int jz[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      wz[c] = ((ok[2] >> c) & 1) ? scale * M[2][c] : T(0);
      const int j = off[2] + c;
      jz[c] = j < 0 ? 0 : (j >= n[2] ? n[2] - 1 : j);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (!((ok[0] >> a) & 1)) continue;
      const int ja = off[0] + a;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        if (!((ok[1] >> b) & 1)) continue;
        const int jb = off[1] + b;
        const T mm = M[0][a] * M[1][b];
        const int rowo = (ja * 16 + jb) * kScalarRow;
#pragma unroll
        for (int c = 0; c < 6; ++c) atomicAdd(&mytile[rowo + jz[c]], scalar_fixed(mm * wz[c]));
      }
    }
  }
  __syncthreads();
  const int nyz = n[1] * n[2], ntot = n[0] * nyz;
  const float inv_yz = 1.0f / (float)nyz, inv_z = 1.0f / (float)n[2];
  for (int t = threadIdx.x; t < ntot; t += 256) {
    int ja, jb, jc;
    if (n[1] == 16 && n[2] == 16) { ja = t >> 8; jb = (t >> 4) & 15; jc = t & 15; }
    else { ja = fast_div(t, nyz, inv_yz); const int rem = t - ja * nyz; jb = fast_div(rem, n[2], inv_z); jc = rem - jb * n[2]; }
    const long mi = ((long)(lo[0] + ja) * g.K[1] + (lo[1] + jb)) * g.K[2] + lo[2] + jc;
    const int ti = (ja * 16 + jb) * kScalarRow + jc;
#pragma unroll
    for (int ty = 0; ty < NT; ++ty) mesh[(long)ty * mesh_stride + mi] = scalar_value(tile[ty * kScalarTile + ti], inv_scale);
  }
  if (threadIdx.x == 0 && clear_a) { clear_a[blockIdx.x] = 0; clear_b[blockIdx.x] = 0; }
}
// the caller's type table against the coefficient rows it stands for (admp_disp_set_types): any row that is not bit for bit
// its type's row bumps *bad (read back with the energies; the call then fails instead of returning a wrong energy)
template <class T>
__global__ __launch_bounds__(256) void k_types_check(int na, const T* __restrict__ vals, int stride, const int* __restrict__ types,
                                                     MixTab mix, double* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool wrong = false;
  if (i < na) {
    const int ty = types[i];
    wrong = ty < 0 || ty >= mix.nt;
    if (!wrong)
      for (int ch = 0; ch < mix.nch; ++ch) wrong = wrong || vals[(long)stride * i + ch] != (T)mix.c[ch][ty];
  }
  if (__syncthreads_or(wrong) && threadIdx.x == 0) atomicAdd(bad, 1.0);
}
template <class T>
void launch_types_check(hipStream_t st, int na, const T* vals, int stride, const int* types, const MixTab& mix, double* bad) {
  if (na > 0) k_types_check<T><<<(na + 255) / 256, 256, 0, st>>>(na, vals, stride, types, mix, bad);
}
template <class T>
__global__ __launch_bounds__(256) void k_onehot(int na, int nt, const int* __restrict__ types, T* __restrict__ w) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= na) return;
  const int ty = types[i];
  for (int t = 0; t < nt; ++t) w[(long)nt * i + t] = t == ty ? T(1) : T(0);
}
// atoms per type (once per type table: the self term of the typed form is a sum over types, formed on the host)
__global__ __launch_bounds__(256) void k_type_counts(int na, const int* __restrict__ types, int* __restrict__ counts) {
  __shared__ int c[4];
  if (threadIdx.x < 4) c[threadIdx.x] = 0;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < na; i += gridDim.x * 256) {
    const int ty = types[i];
    if (ty >= 0 && ty < 4) atomicAdd(&c[ty], 1);
  }
  __syncthreads();
  if (threadIdx.x < 4 && c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], c[threadIdx.x]);
}
void launch_type_counts(hipStream_t st, int na, const int* types, int* counts4) {
  int blocks = (na + 2047) / 2048;
  if (blocks > 256) blocks = 256;
  if (na > 0) k_type_counts<<<blocks, 256, 0, st>>>(na, types, counts4);
}
template <class T>
void launch_onehot(hipStream_t st, int na, int nt, const int* types, T* w) {
  if (na > 0) k_onehot<T><<<(na + 255) / 256, 256, 0, st>>>(na, nt, types, w);
}
// nt = 1..3 type meshes (single precision: three 17 KB tiles per workgroup)
template <class T>
int launch_spread_typed(hipStream_t st, int nt, const T* pos, const int* types, const RecipGeom<T>& g, const BinScratch& bs,
                        T* mesh, long mesh_stride) {
  const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
  const BrickGrid bg = make_bricks(dims);
  using W = typename ScalarWord<T>::type;
  const size_t sh = (size_t)nt * kScalarTile * sizeof(W);
  if (sh > 64 * 1024 || nt < 1 || nt > 3) return (int)hipErrorInvalidValue;
  if (nt == 3)
    k_spread_bricks_typed<T, 3><<<bg.ncell, 256, sh, st>>>(pos, types, g, bg, bs.cell_start, bs.sorted, mesh, mesh_stride, bs.cursor, bs.fillcur);
  else if (nt == 2)
    k_spread_bricks_typed<T, 2><<<bg.ncell, 256, sh, st>>>(pos, types, g, bg, bs.cell_start, bs.sorted, mesh, mesh_stride, bs.cursor, bs.fillcur);
  else
    k_spread_bricks_typed<T, 1><<<bg.ncell, 256, sh, st>>>(pos, types, g, bg, bs.cell_start, bs.sorted, mesh, mesh_stride, bs.cursor, bs.fillcur);
  return (int)hipGetLastError();
}

template <class T>
int launch_spread_scalar(hipStream_t st, int nch, const T* pos, const T* vals, int stride, const RecipGeom<T>& g,
                         const BinScratch& bs, T* mesh, long mesh_stride) {
  const int dims[3] = {g.nloc0, g.K[1], g.K[2]};
  const BrickGrid bg = make_bricks(dims);
  using W = typename ScalarWord<T>::type;
  if (sizeof(T) == 4 && nch == 3) {
    k_spread_bricks_scalar<T, 3><<<bg.ncell, 256, 3 * kScalarTile * sizeof(W), st>>>(pos, vals, stride, 0, g, bg, bs.cell_start,
                                                                                    bs.sorted, mesh, mesh_stride, bs.cursor, bs.fillcur);
  } else if (sizeof(T) == 4 && nch == 2) {
    k_spread_bricks_scalar<T, 2><<<bg.ncell, 256, 2 * kScalarTile * sizeof(W), st>>>(pos, vals, stride, 0, g, bg, bs.cell_start,
                                                                                    bs.sorted, mesh, mesh_stride, bs.cursor, bs.fillcur);
  } else {
    for (int c = 0; c < nch; ++c)      // (the last launch clears the binning counters the lists were built with)
      k_spread_bricks_scalar<T, 1><<<bg.ncell, 256, kScalarTile * sizeof(W), st>>>(
          pos, vals, stride, c, g, bg, bs.cell_start, bs.sorted, mesh + (long)c * mesh_stride, mesh_stride,
          c == nch - 1 ? bs.cursor : nullptr, c == nch - 1 ? bs.fillcur : nullptr);
  }
  return (int)hipGetLastError();
}

// ---- gather -------------------------------------------------------------------------------------------------------------------
// The staged form of recip_kernels.hip (32 atoms per workgroup of 192 threads: wave d evaluates the dimension-d spline of the
// 32 atoms once into LDS, six lanes per atom sum their z-index over the 36 (x, y) points) over NCH meshes at once.
constexpr int kSgAtoms = 32, kSgBlock = 6 * kSgAtoms, kSgRow = 19;

template <class T, int NCH> struct ChanVec { T v[NCH]; };

template <class T, int NCH>
__global__ __launch_bounds__(256) void k_interleave(long n, const T* __restrict__ src, long stride, T* __restrict__ dst) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    ChanVec<T, NCH> o;
#pragma unroll
    for (int c = 0; c < NCH; ++c) o.v[c] = src[(long)c * stride + i];
    reinterpret_cast<ChanVec<T, NCH>*>(dst)[i] = o;
  }
}
template <class T>
void launch_interleave(hipStream_t st, int nch, long n, const T* src, long stride, T* dst) {
  if (n <= 0) return;
  long blocks = (n + 255) / 256;
  if (blocks > 32768) blocks = 32768;
  if (nch == 3) k_interleave<T, 3><<<(unsigned)blocks, 256, 0, st>>>(n, src, stride, dst);
  else if (nch == 2) k_interleave<T, 2><<<(unsigned)blocks, 256, 0, st>>>(n, src, stride, dst);
  else k_interleave<T, 1><<<(unsigned)blocks, 256, 0, st>>>(n, src, stride, dst);
}

// IL: phi holds the NCH channels of every mesh point side by side (k_interleave).  The gather is bound by the number of
// load INSTRUCTIONS its lanes issue (36 per lane and mesh: the address unit of the CU, round-3 counters), not by the bytes
// they return -- so the channels of a point are fetched with one 4 NCH-byte load and folded with the atom's coefficients at
// once: sum_c q_c phi_c takes the place of phi in the separable sums (they are linear in phi).
template <class T, int NCH, bool IL>
__global__ __launch_bounds__(kSgBlock) void k_gather_scalar(int na, const T* __restrict__ pos, const T* __restrict__ vals,
                                                            int stride, RecipGeom<T> g, const T* __restrict__ phi,
                                                            long mesh_stride, T* __restrict__ grad,
                                                            const int* __restrict__ list, const int* __restrict__ types) {
  // types (NCH = 1, not interleaved): atom i gathers from the mesh of its type, which already carries the coefficients
  __shared__ W4<T> w[kSgAtoms][kSgRow];                   // the four orders of the 18 stencil indices of every atom
  __shared__ int sbase[kSgAtoms][4];
  __shared__ T part[3][kSgBlock];
  const long blk = xcd_block(blockIdx.x, (unsigned)((na + kSgAtoms - 1) / kSgAtoms));
  if (blk < 0) return;
  const int slot0 = (int)blk * kSgAtoms;
  {
    const int d = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), s = threadIdx.x & 63;
    if (d < 3 && s < kSgAtoms && slot0 + s < na) {
      const int i = list ? list[slot0 + s] : slot0 + s;
      const T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
      int b;
      const T f = grid_ref(g, r, d, b);
      T M[6], D1[6], D2[6], D3[6];
      bspline6(f, M, D1, D2, D3);
      sbase[s][d] = b;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        W4<T> o;
        o.m = M[p]; o.d1 = D1[p]; o.d2 = T(0); o.d3 = T(0);
        w[s][6 * d + p] = o;
      }
    }
  }
  __syncthreads();
  const int s = threadIdx.x / 6, c = threadIdx.x - 6 * s;
  T f[3] = {0, 0, 0};
  if (slot0 + s < na) {
    const int i = list ? list[slot0 + s] : slot0 + s;
    const int base[3] = {sbase[s][0], sbase[s][1], sbase[s][2]};
    const W4<T> wz = w[s][12 + c];
    const int ic = wrap_add(base[2], c, g.K[2]);
    if (IL) {
      T q[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) q[ch] = vals[(long)stride * i + ch];
      const ChanVec<T, NCH>* __restrict__ ph = reinterpret_cast<const ChanVec<T, NCH>*>(phi);
      gather_zcol_field_w(g, base, &w[s][0], &w[s][6], wz, ic, [&](long idx) {
        const ChanVec<T, NCH> p = ph[idx];
        T v = q[0] * p.v[0];
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch) v += q[ch] * p.v[ch];
        return v;
      }, f);
    } else {
      // one channel at a time (its 36 loads in flight together), the channels' sums combined with the atom's coefficients
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const T* __restrict__ ph = phi + (long)(types ? types[i] : ch) * mesh_stride;
        T fc[3];
        gather_zcol_field_w(g, base, &w[s][0], &w[s][6], wz, ic, [&](long idx) { return ph[idx]; }, fc);
        const T q = types ? T(1) : vals[(long)stride * i + ch];
        f[0] += q * fc[0]; f[1] += q * fc[1]; f[2] += q * fc[2];
      }
    }
  }
  part[0][threadIdx.x] = f[0]; part[1][threadIdx.x] = f[1]; part[2][threadIdx.x] = f[2];
  __syncthreads();
  if (threadIdx.x >= 64) return;                         // one wave converts the 32 atoms
  const int slot = slot0 + (int)threadIdx.x;
  if (threadIdx.x < kSgAtoms && slot < na) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const T* p = &part[k][6 * threadIdx.x];
      f[k] = ((p[0] + p[1]) + (p[2] + p[3])) + (p[4] + p[5]);
    }
    const int i = list ? list[slot] : slot;
    const T* A = g.Jac;                                  // scalar sites: dE/dr = c Jac . F1
#pragma unroll
    for (int k = 0; k < 3; ++k) grad[3 * i + k] += A[3 * k + 0] * f[0] + A[3 * k + 1] * f[1] + A[3 * k + 2] * f[2];
  }
}
// energies[E_SELF] += sum over the listed atoms and the channels of self_coefs[c] vals[i][c]^2 (admp/disp_pme.py:254-279).
// Few, fat workgroups: thousands of workgroups adding doubles into ONE word serialise at the memory side (0.27 ms per launch
// when this sum rode in the 32 768 workgroups of the gather at 1M atoms).
template <class T>
__global__ __launch_bounds__(256) void k_scalar_self(int na, int nch, const T* __restrict__ vals, int stride,
                                                     const int* __restrict__ list, SelfCoefs self_coefs, double* energies) {
  double es = 0.0;
  for (int s = blockIdx.x * 256 + threadIdx.x; s < na; s += gridDim.x * 256) {
    const int i = list ? list[s] : s;
    for (int ch = 0; ch < nch; ++ch) { const double q = (double)vals[(long)stride * i + ch]; es += self_coefs.c[ch] * q * q; }
  }
  es = block_reduce_sum<256>(es);
  if (threadIdx.x == 0 && es != 0.0) atomicAdd(&energies[E_SELF], es);
}

template <class T>
void launch_gather_scalar(hipStream_t st, int nch, int na, const T* pos, const T* vals, int stride, const RecipGeom<T>& g,
                          const T* phi, long mesh_stride, T* grad, const int* list, int interleaved, const int* types) {
  if (na <= 0) return;
  const unsigned grid = xcd_grid((unsigned)((na + kSgAtoms - 1) / kSgAtoms));
  if (types) {      // one mesh per atom, chosen by its type
    k_gather_scalar<T, 1, false><<<grid, kSgBlock, 0, st>>>(na, pos, vals, stride, g, phi, mesh_stride, grad, list, types);
    return;
  }
  if (interleaved) {
    if (nch == 3) k_gather_scalar<T, 3, true><<<grid, kSgBlock, 0, st>>>(na, pos, vals, stride, g, phi, mesh_stride, grad, list, nullptr);
    else if (nch == 2) k_gather_scalar<T, 2, true><<<grid, kSgBlock, 0, st>>>(na, pos, vals, stride, g, phi, mesh_stride, grad, list, nullptr);
    else k_gather_scalar<T, 1, true><<<grid, kSgBlock, 0, st>>>(na, pos, vals, stride, g, phi, mesh_stride, grad, list, nullptr);
    return;
  }
  if (nch == 3) k_gather_scalar<T, 3, false><<<grid, kSgBlock, 0, st>>>(na, pos, vals, stride, g, phi, mesh_stride, grad, list, nullptr);
  else if (nch == 2) k_gather_scalar<T, 2, false><<<grid, kSgBlock, 0, st>>>(na, pos, vals, stride, g, phi, mesh_stride, grad, list, nullptr);
  else k_gather_scalar<T, 1, false><<<grid, kSgBlock, 0, st>>>(na, pos, vals, stride, g, phi, mesh_stride, grad, list, nullptr);
}
template <class T>
void launch_scalar_self(hipStream_t st, int nch, int na, const T* vals, int stride, const int* list, const double* self_coefs,
                        double* energies) {
  if (na <= 0) return;
  SelfCoefs sc;
  for (int b = 0; b < nch && b < 3; ++b) sc.c[b] = self_coefs[b];
  int blocks = (na + 4095) / 4096;
  if (blocks > 256) blocks = 256;
  k_scalar_self<T><<<blocks, 256, 0, st>>>(na, nch, vals, stride, list, sc, energies);
}

// out[i * stride + chan] += phi(r_i) + extra * vals[i * stride + chan]: the mesh potential at every atom (the stencil weights
// without derivatives) -- dE_recip/dc_i of a scalar channel -- plus the derivative of the self term.  One thread per atom;
// on request only (parameter gradients).
template <class T>
__global__ __launch_bounds__(128) void k_gather_value(int na, const T* __restrict__ pos, const T* __restrict__ vals, int stride,
                                                      int chan, RecipGeom<T> g, const T* __restrict__ phi, T extra,
                                                      T* __restrict__ out, const int* __restrict__ list) {
  const int slot = blockIdx.x * 128 + threadIdx.x;
  if (slot >= na) return;
  const int i = list ? list[slot] : slot;                 // (slab rank: its home atoms)
  const T r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  int base[3];
  T M[3][6];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    T D1[6], D2[6], D3[6];
    const T f = grid_ref(g, r, d, base[d]);
    bspline6(f, M[d], D1, D2, D3);
  }
  T v = T(0);
  for (int a = 0; a < 6; ++a) {
    const long ia = wrap_add(base[0], a, g.wrap0);
    for (int b = 0; b < 6; ++b) {
      const long row = (ia * g.K[1] + wrap_add(base[1], b, g.K[1])) * g.K[2];
      T s = T(0);
#pragma unroll
      for (int c = 0; c < 6; ++c) s += M[2][c] * phi[row + wrap_add(base[2], c, g.K[2])];
      v += M[0][a] * M[1][b] * s;
    }
  }
  out[(long)stride * i + chan] += v + extra * vals[(long)stride * i + chan];
}
template <class T>
void launch_gather_value(hipStream_t st, int na, const T* pos, const T* vals, int stride, int chan, const RecipGeom<T>& g,
                         const T* phi, double extra, T* out, const int* list) {
  if (na > 0) k_gather_value<T><<<(na + 127) / 128, 128, 0, st>>>(na, pos, vals, stride, chan, g, phi, (T)extra, out, list);
}

#define INST(T)                                                                                                          \
  template void launch_atom_bases<T>(hipStream_t, int, const T*, const RecipGeom<T>&, int4*);                            \
  template int launch_spread_scalar<T>(hipStream_t, int, const T*, const T*, int, const RecipGeom<T>&, const BinScratch&, \
                                       T*, long);                                                                        \
  template void launch_gather_scalar<T>(hipStream_t, int, int, const T*, const T*, int, const RecipGeom<T>&, const T*,    \
                                        long, T*, const int*, int, const int*);                                          \
  template int launch_spread_typed<T>(hipStream_t, int, const T*, const int*, const RecipGeom<T>&, const BinScratch&, T*, \
                                      long);                                                                             \
  template void launch_types_check<T>(hipStream_t, int, const T*, int, const int*, const MixTab&, double*);              \
  template void launch_onehot<T>(hipStream_t, int, int, const int*, T*);                                                  \
  template void launch_interleave<T>(hipStream_t, int, long, const T*, long, T*);                                        \
  template void launch_scalar_self<T>(hipStream_t, int, int, const T*, int, const int*, const double*, double*);         \
  template void launch_gather_value<T>(hipStream_t, int, const T*, const T*, int, int, const RecipGeom<T>&, const T*,     \
                                       double, T*, const int*);
INST(float)
INST(double)
#undef INST

}  // namespace admp
