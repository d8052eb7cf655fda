"""Tracing front-end for user-defined pair kernels -- what `jax.numpy` + `jax.jit/vmap/grad` are to the reference's
`generate_pairwise_interaction(pair_int_kernel, ...)` (admp/pairwise.py:45-91).

A pair kernel is an ordinary Python function of scalars,

    def kernel(dr, m, p1i, p1j, p2i, p2j, ...):   # distance, topological scale, atomic parameters of the two atoms
        ...
        return energy

written against this module instead of jax.numpy (`from admp_amd import xp as jnp`): it is called ONCE with symbolic
arguments, every arithmetic operation appends one line of HIP source, and the resulting expression is compiled at run
time (hiprtc) into a row-per-atom pair kernel of libadmp_hip's usual shape.  The derivative with respect to `dr` -- the
only thing the gradient needs -- is carried by forward-mode dual numbers in the generated code, so there is no autodiff
at run time either.  Values that do not depend on `dr` (products of parameters, the scale factor) stay plain reals.

Supported: + - * / ** (integer or real exponent, symbolic exponent), unary minus, abs, sqrt, exp, log, erf, erfc, sin, cos,
tanh, power, square, minimum / maximum and `where(cond, a, b)` with comparisons (< <= > >=) as conditions.
"""
import math as _math

__all__ = ['sqrt', 'exp', 'log', 'erf', 'erfc', 'sin', 'cos', 'tanh', 'abs', 'power', 'square', 'minimum', 'maximum',
           'where', 'pi', 'trace_pair_kernel']

pi = _math.pi
_ctx = None


class _Trace:
    def __init__(self):
        self.lines = []
        self.n = 0

    def new(self, expr, dual):
        name = 't%d' % self.n
        self.n += 1
        self.lines.append('const %s %s = %s;' % ('D' if dual else 'R', name, expr))
        return Sym(name, dual)


def _lit(x):
    return 'R(%s)' % repr(float(x))


class Sym:
    """a value of the traced kernel: `code` is a HIP identifier or literal, `dual` says whether it depends on dr"""
    __slots__ = ('code', 'dual')
    __array_priority__ = 1000

    def __init__(self, code, dual):
        self.code, self.dual = code, dual

    # -- helpers
    @staticmethod
    def wrap(x):
        if isinstance(x, Sym):
            return x
        if isinstance(x, Cond):
            raise TypeError('a comparison can only be used as the condition of where()')
        return Sym(_lit(x), False)

    def _bin(self, other, op, swap=False):
        o = Sym.wrap(other)
        a, b = (o, self) if swap else (self, o)
        return _ctx.new('(%s %s %s)' % (a.code, op, b.code), a.dual or b.dual)

    def __add__(self, o): return self._bin(o, '+')
    def __radd__(self, o): return self._bin(o, '+', True)
    def __sub__(self, o): return self._bin(o, '-')
    def __rsub__(self, o): return self._bin(o, '-', True)
    def __mul__(self, o): return self._bin(o, '*')
    def __rmul__(self, o): return self._bin(o, '*', True)
    def __truediv__(self, o): return self._bin(o, '/')
    def __rtruediv__(self, o): return self._bin(o, '/', True)
    def __neg__(self): return _ctx.new('(-%s)' % self.code, self.dual)
    def __pos__(self): return self
    def __abs__(self): return _call1('d_abs', self)

    def __pow__(self, e):
        if isinstance(e, int) or (isinstance(e, float) and e == int(e) and builtins_abs(e) <= 64):
            return _ipow(self, int(e))
        return power(self, e)

    def __rpow__(self, base):
        return power(base, self)

    def _cmp(self, o, op):
        o = Sym.wrap(o)
        return Cond('(d_val(%s) %s d_val(%s))' % (self.code, op, o.code))

    def __lt__(self, o): return self._cmp(o, '<')
    def __le__(self, o): return self._cmp(o, '<=')
    def __gt__(self, o): return self._cmp(o, '>')
    def __ge__(self, o): return self._cmp(o, '>=')

    def __bool__(self):
        raise TypeError('the truth value of a traced quantity is not known while tracing: use xp.where(cond, a, b)')

    def __float__(self):
        raise TypeError('a traced quantity has no numeric value while tracing')


builtins_abs = abs


class Cond:
    __slots__ = ('code',)

    def __init__(self, code):
        self.code = code

    def __and__(self, o): return Cond('(%s && %s)' % (self.code, o.code))
    def __or__(self, o): return Cond('(%s || %s)' % (self.code, o.code))
    def __invert__(self): return Cond('(!%s)' % self.code)

    def __bool__(self):
        raise TypeError('a traced comparison cannot steer Python control flow: use xp.where(cond, a, b)')


def _ipow(x, n):
    if n == 0:
        return Sym(_lit(1.0), False)
    if n < 0:
        return Sym(_lit(1.0), False) / _ipow(x, -n)
    result, base = None, x
    while n:                                    # binary powering: a handful of multiplications
        if n & 1:
            result = base if result is None else result * base
        n >>= 1
        if n:
            base = base * base
    return result


def _call1(fn, x):
    x = Sym.wrap(x)
    if not isinstance(x.code, str):
        raise TypeError
    return _ctx.new('%s(%s)' % (fn, x.code), x.dual)


def sqrt(x): return _call1('d_sqrt', x) if isinstance(x, Sym) else _math.sqrt(x)
def exp(x): return _call1('d_exp', x) if isinstance(x, Sym) else _math.exp(x)
def log(x): return _call1('d_log', x) if isinstance(x, Sym) else _math.log(x)
def erf(x): return _call1('d_erf', x) if isinstance(x, Sym) else _math.erf(x)
def erfc(x): return _call1('d_erfc', x) if isinstance(x, Sym) else _math.erfc(x)
def sin(x): return _call1('d_sin', x) if isinstance(x, Sym) else _math.sin(x)
def cos(x): return _call1('d_cos', x) if isinstance(x, Sym) else _math.cos(x)
def tanh(x): return _call1('d_tanh', x) if isinstance(x, Sym) else _math.tanh(x)
def abs(x): return _call1('d_abs', x) if isinstance(x, Sym) else builtins_abs(x)      # noqa: A001
def square(x): return x * x


def power(x, e):
    if not isinstance(x, Sym) and not isinstance(e, Sym):
        return x ** e
    if isinstance(e, int) and isinstance(x, Sym):
        return _ipow(x, e)
    x, e = Sym.wrap(x), Sym.wrap(e)
    return _ctx.new('d_pow(%s, %s)' % (x.code, e.code), x.dual or e.dual)


def where(cond, a, b):
    if not isinstance(cond, Cond):
        return a if cond else b
    a, b = Sym.wrap(a), Sym.wrap(b)
    dual = a.dual or b.dual
    ty = 'D' if dual else 'R'
    return _ctx.new('(%s ? %s(%s) : %s(%s))' % (cond.code, ty, a.code, ty, b.code), dual)


def minimum(a, b):
    a, b = Sym.wrap(a), Sym.wrap(b)
    return where(a < b, a, b)


def maximum(a, b):
    a, b = Sym.wrap(a), Sym.wrap(b)
    return where(a > b, a, b)


def trace_pair_kernel(fn, n_params):
    """Run `fn(dr, m, p0i, p0j, p1i, p1j, ...)` on symbols; returns (body lines, name of the result, result is dual)."""
    global _ctx
    if _ctx is not None:
        raise RuntimeError('nested tracing')
    _ctx = _Trace()
    try:
        args = [Sym('dr', True), Sym('m', False)]
        for k in range(n_params):
            args += [Sym('pi%d' % k, False), Sym('pj%d' % k, False)]
        out = fn(*args)
        out = Sym.wrap(out)
        if not out.dual:                                 # an energy that does not depend on the distance: zero force
            out = _ctx.new('D(%s)' % out.code, True)
        return list(_ctx.lines), out.code
    finally:
        _ctx = None


# --------------------------------------------------------------------------------------------------------- code generation
_MATH = r'''
typedef REAL_T R;
struct D { R v, d; ADMP_DEV D() {} ADMP_DEV D(R v_) : v(v_), d(R(0)) {} ADMP_DEV D(R v_, R d_) : v(v_), d(d_) {} };
ADMP_DEV inline R d_val(R a) { return a; }
ADMP_DEV inline R d_val(D a) { return a.v; }
ADMP_DEV inline D operator+(D a, D b) { return D(a.v + b.v, a.d + b.d); }
ADMP_DEV inline D operator+(D a, R b) { return D(a.v + b, a.d); }
ADMP_DEV inline D operator+(R a, D b) { return D(a + b.v, b.d); }
ADMP_DEV inline D operator-(D a, D b) { return D(a.v - b.v, a.d - b.d); }
ADMP_DEV inline D operator-(D a, R b) { return D(a.v - b, a.d); }
ADMP_DEV inline D operator-(R a, D b) { return D(a - b.v, -b.d); }
ADMP_DEV inline D operator-(D a) { return D(-a.v, -a.d); }
ADMP_DEV inline D operator*(D a, D b) { return D(a.v * b.v, a.v * b.d + a.d * b.v); }
ADMP_DEV inline D operator*(D a, R b) { return D(a.v * b, a.d * b); }
ADMP_DEV inline D operator*(R a, D b) { return D(a * b.v, a * b.d); }
ADMP_DEV inline D operator/(D a, D b) { R i = R(1) / b.v; R q = a.v * i; return D(q, (a.d - q * b.d) * i); }
ADMP_DEV inline D operator/(D a, R b) { R i = R(1) / b; return D(a.v * i, a.d * i); }
ADMP_DEV inline D operator/(R a, D b) { R i = R(1) / b.v; R q = a * i; return D(q, -q * b.d * i); }
#define ADMP_FN1(name, f, df) \
  ADMP_DEV inline R name(R x) { return f; } \
  ADMP_DEV inline D name(D a) { const R x = a.v; const R y = f; return D(y, (df) * a.d); }
ADMP_FN1(d_sqrt, sqrt(x), R(0.5) / y)
ADMP_FN1(d_exp, exp(x), y)
ADMP_FN1(d_log, log(x), R(1) / x)
ADMP_FN1(d_erf, erf(x), R(1.1283791670955126) * exp(-x * x))
ADMP_FN1(d_erfc, erfc(x), R(-1.1283791670955126) * exp(-x * x))
ADMP_FN1(d_sin, sin(x), cos(x))
ADMP_FN1(d_cos, cos(x), -sin(x))
ADMP_FN1(d_tanh, tanh(x), R(1) - y * y)
ADMP_FN1(d_abs, fabs(x), (x < R(0) ? R(-1) : R(1)))
ADMP_DEV inline R d_pow(R a, R b) { return pow(a, b); }
ADMP_DEV inline D d_pow(D a, R b) { const R y = pow(a.v, b); return D(y, b * y / a.v * a.d); }
ADMP_DEV inline D d_pow(R a, D b) { const R y = pow(a, b.v); return D(y, y * log(a) * b.d); }
ADMP_DEV inline D d_pow(D a, D b) { const R y = pow(a.v, b.v); return D(y, y * (b.v / a.v * a.d + log(a.v) * b.d)); }

// the user's kernel (admp/pairwise.py:88: pair_int_kernel(dr, mscales, *pair_params)); returns energy and d(energy)/d(dr)
ADMP_DEV inline D pair_kernel(const D dr, const R m, const R* __restrict__ pi, const R* __restrict__ pj) {
%(unpack)s
%(body)s
  return %(result)s;
}
'''

_KERNEL = r'''
// admp/spatial.py:13-32: ds = dr . box^-1 ; ds -= floor(ds + 1/2) ; dr = ds . box
__device__ inline void min_image(const R* __restrict__ h, const R* __restrict__ hi, R d[3]) {
  R s0 = d[0] * hi[0] + d[1] * hi[3] + d[2] * hi[6];
  R s1 = d[0] * hi[1] + d[1] * hi[4] + d[2] * hi[7];
  R s2 = d[0] * hi[2] + d[1] * hi[5] + d[2] * hi[8];
  s0 -= floor(s0 + R(0.5)); s1 -= floor(s1 + R(0.5)); s2 -= floor(s2 + R(0.5));
  d[0] = s0 * h[0] + s1 * h[3] + s2 * h[6];
  d[1] = s0 * h[1] + s1 * h[4] + s2 * h[7];
  d[2] = s0 * h[2] + s1 * h[5] + s2 * h[8];
}

// One row of the i-grouped neighbour table per 8 lanes (both directions of every pair are stored: the row atom's side is
// evaluated, energy halved), gradient folded across the lanes and written once, energy by one f64 atomic per workgroup.
extern "C" __global__ __launch_bounds__(256) void admp_pair_custom(int na, const int* __restrict__ rowptr,
    const int* __restrict__ col, const int* __restrict__ order, const R* __restrict__ pos, const R* __restrict__ par,
    const R* __restrict__ box, const R* __restrict__ mtab, R* __restrict__ grad, double* __restrict__ energy) {
  constexpr int NP = %(np)d, NPS = NP > 0 ? NP : 1;
  __shared__ double part[4];
  __shared__ R s_box[18], s_m[16];
  if (threadIdx.x < 18) s_box[threadIdx.x] = box[threadIdx.x];
  if (threadIdx.x < 16) s_m[threadIdx.x] = mtab[threadIdx.x];
  __syncthreads();
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int slot = (int)(t >> 3), sub = (int)(t & 7);
  const int row = slot < na ? (order ? order[slot] : slot) : na;
  R g[3] = {0, 0, 0};
  double e = 0.0;
  if (row < na) {
    const R ri[3] = {pos[3 * row], pos[3 * row + 1], pos[3 * row + 2]};
    R pi[NPS], pj[NPS];
    for (int q = 0; q < NP; ++q) pi[q] = par[(long)NP * row + q];
    const int end = rowptr[row + 1];
    for (int k = rowptr[row] + sub; k < end; k += 8) {
      const int c = col[k];
      const int j = c & 0x0fffffff;
      for (int q = 0; q < NP; ++q) pj[q] = par[(long)NP * j + q];
      R d[3] = {ri[0] - pos[3 * j], ri[1] - pos[3 * j + 1], ri[2] - pos[3 * j + 2]};
      min_image(s_box, s_box + 9, d);
      const R r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      const D f = pair_kernel(D(r, R(1)), s_m[(c >> 28) & 7], pi, pj);
      e += (double)f.v;
      const R s = f.d / r;
      g[0] += s * d[0]; g[1] += s * d[1]; g[2] += s * d[2];
    }
  }
  for (int off = 4; off > 0; off >>= 1) {
    g[0] += __shfl_xor(g[0], off, 64); g[1] += __shfl_xor(g[1], off, 64); g[2] += __shfl_xor(g[2], off, 64);
  }
  if (row < na && sub == 0 && grad) { grad[3 * row] = g[0]; grad[3 * row + 1] = g[1]; grad[3 * row + 2] = g[2]; }
  for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(energy, 0.5 * (part[0] + part[1] + part[2] + part[3]));
}
'''


def _fill(fn, n_params):
    lines, result = trace_pair_kernel(fn, n_params)
    unpack = '\n'.join('  const R pi%d = pi[%d], pj%d = pj[%d];' % (k, k, k, k) for k in range(n_params))
    body = '\n'.join('  ' + ln for ln in lines)
    return dict(unpack=unpack, body=body, result=result, np=n_params)


def generate_source(fn, n_params):
    """HIP source of the pair kernel that evaluates `fn`; compile with -DREAL_T=float|double (hiprtc, engine.hip)."""
    d = _fill(fn, n_params)
    return '#include <hip/hip_runtime.h>\n#define ADMP_DEV __device__\n' + _MATH % d + _KERNEL % d


def generate_host_source(fn, n_params):
    """The same traced expression as a host C++ function `eval(r, m, pi, pj, out[2])` (value, d/dr): lets the tests check the
    tracer and the dual arithmetic with g++ on a machine without a GPU.  Not used by the product."""
    d = _fill(fn, n_params)
    return ('#include <cmath>\nusing namespace std;\n#define ADMP_DEV\n#define REAL_T double\n' + _MATH % d +
            'extern "C" void eval(double r, double m, const double* pi, const double* pj, double* out) {\n'
            '  const D f = pair_kernel(D(r, 1.0), m, pi, pj);\n  out[0] = f.v; out[1] = f.d;\n}\n')
