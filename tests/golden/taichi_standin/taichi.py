"""A minimal SERIAL stand-in for the `taichi` package -- TEST INFRASTRUCTURE, build container only.

Purpose: the reference (lyd405121/wcsph) is Python + Taichi and `import taichi` fails in this
image, so nothing the reference itself computes could pin the oracle.  With this module first on sys.path the reference's
OWN files (HashGrid.py, ParticleData.py, kernels/*.py, dfsph.py, sesph.py -- imported unmodified from /root/reference)
run as plain Python: `@ti.kernel` / `@ti.func` bodies execute serially, fields are NumPy arrays, scalars are NumPy
float32 / int32 values so that every arithmetic operation rounds to the field types.  tests/golden/
make_reference_fixtures.py uses it to produce golden vectors from the reference's text.

What this pins: the reference's FORMULAS, loop structure, host control flow and constants, as written -- independently
of the reader who restated them in oracle/wcsph_ref.c.  What it does NOT pin: Taichi's execution semantics -- parallel
schedule and atomic order (here: ascending particle index, one valid order), fast-math / FMA contraction, f32 constant
folding (Python-float constants are combined in double before they meet an f32 operand), the closed form
ti.Matrix.inverse() uses, and out-of-bounds field reads (dense SNodes of the 2020-21 Taichi the scripts target pad to a
power of two and mask indices; here an out-of-bounds READ returns 0 and is counted, an out-of-bounds WRITE raises).

Only what the five files above use is implemented.  Nothing of this travels to the GPU box; only the .npz fixtures do.
"""
import builtins as _b
import collections
import itertools
import math

import numpy as np

f32, i32, f64, i64 = np.float32, np.int32, np.float64, np.int64
gpu, cpu = "gpu", "cpu"
i, j, k, ij, ijk = (0,), (1,), (2,), (0, 1), (0, 1, 2)   # axis selectors of ti.root.dense

_depth = 0                      # > 0 while a @ti.kernel / @ti.func body runs ("Taichi scope")
_log = collections.deque(maxlen=32)   # recent field reads, for atomic_add
oob_reads = collections.Counter()     # field name -> out-of-bounds reads served as 0


def init(**kw):
    pass


def data_oriented(cls):
    return cls


def template():
    return None


def _scoped(fn):
    def run(*a, **kw):
        global _depth
        _depth += 1
        try:
            return fn(*a, **kw)
        finally:
            _depth -= 1
    run.__name__ = getattr(fn, "__name__", "kernel")
    run.__wrapped__ = fn
    return run


kernel = _scoped
func = _scoped


# ---- vectors / matrices -----------------------------------------------------------------------------------------
def _arr(x):
    if isinstance(x, (Vector, Matrix)):
        return x.v
    return x


class Vector:
    __slots__ = ("v",)
    __array_ufunc__ = None      # NumPy scalars on the left defer to __r*__

    def __init__(self, vals, dt=None):
        if isinstance(vals, np.ndarray):
            self.v = vals
        else:
            vals = list(vals)
            if dt is None:
                dt = i32 if all(isinstance(t, (int, np.integer)) and not isinstance(t, bool) for t in vals) else f32
            self.v = np.array(vals, dtype=dt)

    # components
    x = property(lambda s: s.v[0], lambda s, t: s.v.__setitem__(0, t))
    y = property(lambda s: s.v[1], lambda s, t: s.v.__setitem__(1, t))
    z = property(lambda s: s.v[2], lambda s, t: s.v.__setitem__(2, t))
    w = property(lambda s: s.v[3], lambda s, t: s.v.__setitem__(3, t))

    def __getitem__(self, n):
        return self.v[n]

    def __setitem__(self, n, t):
        self.v[n] = t

    def __len__(self):
        return len(self.v)

    def __iter__(self):
        return iter(self.v)

    def __add__(s, o): return Vector(s.v + _arr(o))          # noqa: E704
    def __radd__(s, o): return Vector(_arr(o) + s.v)         # noqa: E704
    def __sub__(s, o): return Vector(s.v - _arr(o))          # noqa: E704
    def __rsub__(s, o): return Vector(_arr(o) - s.v)         # noqa: E704
    def __mul__(s, o): return Vector(s.v * _arr(o))          # noqa: E704
    def __rmul__(s, o): return Vector(_arr(o) * s.v)         # noqa: E704
    def __truediv__(s, o): return Vector(s.v / _arr(o))      # noqa: E704
    def __rtruediv__(s, o): return Vector(_arr(o) / s.v)     # noqa: E704
    def __neg__(s): return Vector(-s.v)                      # noqa: E704

    def dot(s, o):
        a, b = s.v, o.v
        r = a[0] * b[0]
        for n in range(1, len(a)):
            r = r + a[n] * b[n]
        return r

    def norm_sqr(s):
        return s.dot(s)

    def norm(s):
        return np.sqrt(s.dot(s))

    def normalized(s):
        return Vector(s.v / s.norm())

    def cross(s, o):
        a, b = s.v, o.v
        return Vector(np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=a.dtype))

    def outer_product(s, o):
        return Matrix(np.outer(s.v, o.v))

    def to_numpy(s):
        return s.v.copy()

    def __repr__(s):
        return "Vector(%r)" % (s.v,)

    @staticmethod
    def field(n, dtype=f32, shape=None):
        return Field(dtype, shape, inner=(n,))


class Matrix:
    __slots__ = ("v",)
    __array_ufunc__ = None

    def __init__(self, rows, dt=None):
        self.v = rows if isinstance(rows, np.ndarray) else np.array(rows, dtype=dt or f32)

    def __getitem__(self, rc):
        return self.v[rc]

    def __setitem__(self, rc, t):
        self.v[rc] = t

    def __add__(s, o): return Matrix(s.v + _arr(o))          # noqa: E704
    def __radd__(s, o): return Matrix(_arr(o) + s.v)         # noqa: E704
    def __sub__(s, o): return Matrix(s.v - _arr(o))          # noqa: E704
    def __rsub__(s, o): return Matrix(_arr(o) - s.v)         # noqa: E704
    def __mul__(s, o): return Matrix(s.v * _arr(o))          # noqa: E704   (element-wise, as in Taichi)
    def __rmul__(s, o): return Matrix(_arr(o) * s.v)         # noqa: E704
    def __truediv__(s, o): return Matrix(s.v / _arr(o))      # noqa: E704
    def __neg__(s): return Matrix(-s.v)                      # noqa: E704

    def __matmul__(s, o):
        m = s.v
        if isinstance(o, Vector):
            out = np.empty(m.shape[0], dtype=m.dtype)
            for r in range(m.shape[0]):
                t = m[r, 0] * o.v[0]
                for c in range(1, m.shape[1]):
                    t = t + m[r, c] * o.v[c]
                out[r] = t
            return Vector(out)
        b = o.v
        out = np.empty((m.shape[0], b.shape[1]), dtype=m.dtype)
        for r in range(m.shape[0]):
            for c in range(b.shape[1]):
                t = m[r, 0] * b[0, c]
                for q in range(1, m.shape[1]):
                    t = t + m[r, q] * b[q, c]
                out[r, c] = t
        return Matrix(out)

    def transpose(s):
        return Matrix(s.v.T.copy())

    def determinant(s):
        a = s.v
        return (a[0, 0] * (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) - a[0, 1] * (a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
                + a[0, 2] * (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]))

    def inverse(s):
        """3x3: adjugate / determinant in the matrix's own precision (Taichi expands the same closed form; the order
        of its operations is not pinned here)"""
        a = s.v
        if a.shape != (3, 3):
            return Matrix(np.linalg.inv(a.astype(np.float64)).astype(a.dtype))
        inv_det = a.dtype.type(1.0) / s.determinant()
        out = np.empty((3, 3), dtype=a.dtype)
        for r in range(3):
            for c in range(3):
                r1, r2, c1, c2 = (c + 1) % 3, (c + 2) % 3, (r + 1) % 3, (r + 2) % 3     # cofactor of (c, r)
                out[r, c] = (a[r1, c1] * a[r2, c2] - a[r1, c2] * a[r2, c1]) * inv_det
        return Matrix(out)

    def to_numpy(s):
        return s.v.copy()

    @staticmethod
    def field(n, m, dtype=f32, shape=None):
        return Field(dtype, shape, inner=(n, m))


# ---- fields ---------------------------------------------------------------------------------------------------
class Field:
    """ti.field / ti.Vector.field / ti.Matrix.field: a dense NumPy array, allocated at creation (shape=) or by
    ti.root.dense(...).place()"""

    def __init__(self, dtype, shape=None, inner=()):
        self.dtype, self.inner, self.a, self.shape, self.name = dtype, tuple(inner), None, None, "?"
        if shape is not None:
            self._alloc(shape)

    def _alloc(self, shape):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(int(s) for s in shape)
        self.a = np.zeros(self.shape + self.inner, dtype=self.dtype)

    def _name(self):
        return self.name

    def __getitem__(self, key):
        try:
            if isinstance(key, tuple):
                if _b.min(key) < 0:
                    raise IndexError
            elif key < 0:
                raise IndexError
            v = self.a[key]
        except IndexError:
            if _depth == 0:
                raise
            oob_reads[self._name()] += 1
            v = np.zeros(self.inner, dtype=self.dtype) if self.inner else self.dtype(0)
        if self.inner:
            v = Vector(v.copy()) if len(self.inner) == 1 else Matrix(v.copy())
            if _depth:
                _log.append((self, key, v))
            return v
        if _depth == 0:           # Python scope: Taichi hands out Python numbers
            return v.item()
        _log.append((self, key, v))
        return v

    def __setitem__(self, key, val):
        if isinstance(key, tuple):
            if _b.min(key) < 0 or any(q >= n for q, n in zip(key, self.shape)):
                raise IndexError("out-of-bounds store to field %s%r" % (self._name(), (key,)))
        elif key < 0 or key >= self.shape[0]:
            raise IndexError("out-of-bounds store to field %s[%r]" % (self._name(), key))
        self.a[key] = _arr(val)

    def __iter__(self):          # struct-for: `for i in field` / `for i, j in field`
        if len(self.shape) == 1:
            return iter(range(self.shape[0]))
        return itertools.product(*[range(n) for n in self.shape])

    def to_numpy(self):
        return self.a.copy()

    def from_numpy(self, arr):
        self.a[...] = np.asarray(arr).reshape(self.a.shape)

    def fill(self, v):
        self.a[...] = v


def field(dtype=f32, shape=None):
    return Field(dtype, shape)


class _Dense:
    def __init__(self, shape):
        self.shape = shape

    def place(self, *fields):
        for f in fields:
            f._alloc(self.shape)
        return self

    def dense(self, axes, shape):
        raise NotImplementedError("nested SNodes are not used by the reference")


class _Root:
    def dense(self, axes, shape):
        return _Dense(shape)


root = _Root()


# ---- functions ------------------------------------------------------------------------------------------------
def cast(x, dt):
    if isinstance(x, Vector):
        return Vector((np.trunc(x.v) if np.issubdtype(dt, np.integer) else x.v).astype(dt))
    if np.issubdtype(dt, np.integer):
        return dt(math.trunc(x))
    return dt(x)


def atomic_add(ref, val):
    """`old = ti.atomic_add(field[idx], v)`: the first argument is the VALUE the field read just returned; the read is
    found in the log by object identity and the field element is updated in place."""
    for f, key, obj in reversed(_log):
        if obj is ref:
            _log.clear()
            f.a[key] = f.a[key] + _arr(val)
            return ref
    raise RuntimeError("ti.atomic_add: the first argument is not a field element read")


def sqrt(x):
    return np.sqrt(x) if isinstance(x, np.generic) else math.sqrt(x)


def max(a, b):            # noqa: A001
    return a if a > b else b


def min(a, b):            # noqa: A001
    return a if a < b else b


def abs(a):               # noqa: A001
    return -a if a < 0 else a


def _m(fn_np, fn_py):
    return lambda x: fn_np(x) if isinstance(x, np.generic) else fn_py(x)


sin, cos, tan, asin, acos, exp, log, floor = (_m(np.sin, math.sin), _m(np.cos, math.cos), _m(np.tan, math.tan),
                                              _m(np.arcsin, math.asin), _m(np.arccos, math.acos), _m(np.exp, math.exp),
                                              _m(np.log, math.log), _m(np.floor, math.floor))
_rng = np.random.default_rng(20240901)


def random(dt=f32):
    return dt(_rng.random())


def ndrange(*ns):
    return itertools.product(*[range(n) for n in ns])


def static(x):
    return x


def svd(m, dt=f32):
    u, s, vt = np.linalg.svd(m.v.astype(np.float64))
    return Matrix(u.astype(m.v.dtype)), Matrix(np.diag(s).astype(m.v.dtype)), Matrix(vt.T.astype(m.v.dtype))


def imwrite(*a, **kw):
    pass


class GUI:
    """no window: `while gui.running` never runs, the scripts stop after their set-up code"""
    running = False

    def __init__(self, *a, **kw):
        pass

    def set_image(self, img):
        pass

    def show(self, *a):
        pass
