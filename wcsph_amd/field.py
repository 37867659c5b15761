"""Minimal stand-in for the Taichi field objects the reference drivers touch from Python
scope: .to_numpy(), .from_numpy(a), f[i], f[i] = v, .shape (SURVEY 8(b) "Field protocol",
ref: dfsph.py:98,113-114,129,141,162,626,645; ParticleData.py:182; HashGrid.py:50-52).
Data lives on the GPU in the library's cell-sorted layout; these calls always present the
caller's ORIGINAL particle order."""
import ctypes as C

import numpy as np

from . import _lib


class Field:
    def __init__(self, owner, name, ncomp, dtype=np.float32, full_count=False):
        self._owner, self.name, self._ncomp, self.dtype, self._full = owner, name, ncomp, dtype, full_count

    @property
    def shape(self):
        n = self._owner.count if self._full else self._owner.liquid_count
        return (n,)

    def _np_shape(self):
        n = self.shape[0]
        return {1: (n,), 3: (n, 3), 9: (n, 3, 3)}[self._ncomp]

    def to_numpy(self):
        ctx = self._owner._need_ctx()
        out = np.empty(self._np_shape(), dtype=np.float32 if self.dtype != np.int32 else np.int32)
        _lib.check(ctx, _lib.load().wcsph_field_to_host(ctx, _lib.FIELD_ID[self.name], out.ctypes.data, out.nbytes))
        return out

    def from_numpy(self, a):
        ctx = self._owner._need_ctx()
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape != self._np_shape():
            raise ValueError("%s.from_numpy: expected shape %s, got %s" % (self.name, self._np_shape(), a.shape))
        _lib.check(ctx, _lib.load().wcsph_field_from_host(ctx, _lib.FIELD_ID[self.name], a.ctypes.data, a.nbytes))

    def _row_ok(self, i):
        return isinstance(i, (int, np.integer)) and self._ncomp in (1, 3) and 0 <= int(i) < self._owner.liquid_count

    def __getitem__(self, i):
        if not self._row_ok(i):          # slices, negative indices, solid rows of pos, matrices: whole-field path
            return self.to_numpy()[i]
        ctx = self._owner._need_ctx()
        out = np.empty(self._ncomp, dtype=np.float32 if self.dtype != np.int32 else np.int32)
        _lib.check(ctx, _lib.load().wcsph_field_rows_to_host(ctx, _lib.FIELD_ID[self.name], int(i), 1, out.ctypes.data))
        return out if self._ncomp == 3 else out[0]

    def __setitem__(self, i, v):
        if not self._row_ok(i) or self.dtype == np.int32:
            a = self.to_numpy()
            a[i] = v
            self.from_numpy(a)
            return
        ctx = self._owner._need_ctx()
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32), (self._ncomp,)))
        _lib.check(ctx, _lib.load().wcsph_field_rows_from_host(ctx, _lib.FIELD_ID[self.name], int(i), 1, a.ctypes.data))


class ScalarField:
    """shape-(1) field (deltaT, avg_density_err, cg_delta ...); the value lives host-side in the context."""

    def __init__(self, owner, name):
        self._owner, self.name = owner, name
        self.shape = (1,)

    def _get(self):
        ctx = self._owner._need_ctx()
        v = C.c_float()
        _lib.check(ctx, _lib.load().wcsph_get_scalar(ctx, _lib.SCALAR_ID[self.name], C.byref(v)))
        return np.float32(v.value)

    def _set(self, v):
        ctx = self._owner._need_ctx()
        _lib.check(ctx, _lib.load().wcsph_set_scalar(ctx, _lib.SCALAR_ID[self.name], float(v)))

    def to_numpy(self):
        return np.array([self._get()], dtype=np.float32)

    def from_numpy(self, a):
        self._set(np.asarray(a).reshape(-1)[0])

    def __getitem__(self, i):
        if i != 0:
            raise IndexError(i)
        return self._get()

    def __setitem__(self, i, v):
        if i != 0:
            raise IndexError(i)
        self._set(v)
