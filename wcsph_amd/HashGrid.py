"""HashGrid with the reference's constructor, attributes and methods (ref: HashGrid.py:8-152).

The reference keeps a `count`-bucket hash table (64 slots each) and an explicit 2048-entry
neighbour list per liquid particle, both cleared and refilled every step.  Here
update_grid() is one call into libwcsph_hip.so which counting-sorts the particles into a
dense z-fastest cell table; neighbours are re-walked from that table inside every pass.
The reference table's aliasing (Q1) is reproduced in "faithful" mode through static
per-cell multiplicity tables; `neighborCount` is the reference's raw list length.
`neighbor`, `gridCount` and `grid` are materialised on demand for inspection only.
"""
import ctypes as C

import numpy as np

from . import _lib
from .field import Field


class _HostVec:
    """shape-(1) ti.Vector.field stand-in held on the host (blockSize / min_boundary / max_boundary)."""

    def __init__(self, dtype):
        self._a = np.zeros((1, 3), dtype=dtype)
        self.shape = (1,)

    def to_numpy(self):
        return self._a.copy()

    def from_numpy(self, a):
        self._a[...] = np.asarray(a).reshape(1, 3)

    def __getitem__(self, i):
        return self._a[i]


class _NeighborList:
    """HashGrid.neighbor[i, k] (ref: HashGrid.py:25,88-106): explicit list in ORIGINAL indices, -1 padded.
    Faithful mode: the reference's row entry for entry, read from the bucket table the device materialises on demand
    (walked cells x outermost, bucket slots in ascending id, far aliased particles included).  Clean mode: the
    particles of the 5x5x5 walked cells."""

    def __init__(self, grid):
        self._g = grid

    @property
    def shape(self):
        return (self._g.particle_data.liquid_count, self._g.maxNeighbour)

    def rows(self, i0, n, cap=None):
        pd = self._g.particle_data
        ctx = pd._need_ctx()
        cap = cap or self._g.maxNeighbour
        counts = np.empty(n, np.int32)
        lists = np.empty((n, cap), np.int32)
        _lib.check(ctx, _lib.load().wcsph_neighbors_to_host(ctx, i0, n, cap, counts.ctypes.data, lists.ctypes.data))
        return counts, lists

    def compact_rows(self, i0, n, cap=256):
        """The compact per-step lists the neighbour passes iterate (in-support subset, multiplicity kept), decoded from
        the device's 16-bit delta streams: (counts[n,3] = liquid entries, solid entries, hop slots; lists[n,cap])."""
        pd = self._g.particle_data
        ctx = pd._need_ctx()
        counts = np.empty((n, 3), np.int32)
        lists = np.empty((n, cap), np.int32)
        _lib.check(ctx, _lib.load().wcsph_compact_lists_to_host(ctx, i0, n, cap, counts.ctypes.data, lists.ctypes.data))
        return counts, lists

    def to_numpy(self):
        n = self._g.particle_data.liquid_count
        if n * self._g.maxNeighbour > (1 << 28):
            raise MemoryError("HashGrid.neighbor.to_numpy(): use .rows(i0, n, cap) on large scenes")
        return self.rows(0, n)[1]

    def __getitem__(self, ik):
        i, k = ik
        return self.rows(int(i), 1)[1][0, k]


class _GridTable:
    """HashGrid.grid[h, k] (ref: HashGrid.py:21,76): the reference's bucket table, materialised on demand from the
    device (faithful mode): global particle ids, ascending inside a bucket, -1 padded to maxInGrid."""

    def __init__(self, grid):
        self._g = grid

    @property
    def shape(self):
        pd = self._g.particle_data
        return (pd.hash_table_size or pd.count, self._g.maxInGrid)

    def rows(self, h0, n):
        ctx = self._g.particle_data._need_ctx()
        out = np.empty((n, self._g.maxInGrid), np.int32)
        _lib.check(ctx, _lib.load().wcsph_hashgrid_rows_to_host(ctx, h0, n, self._g.maxInGrid, out.ctypes.data))
        return out

    def to_numpy(self):
        return self.rows(0, self.shape[0])

    def __getitem__(self, hk):
        h, k = hk
        return self.rows(int(h), 1)[0, k]


class HashGrid:
    def __init__(self, gridR, maxInGrid, maxNeighbour, particle_data):
        self.maxInGrid = maxInGrid
        self.maxNeighbour = maxNeighbour
        self.particle_data = particle_data
        self.invGridR = 1.0 / gridR
        self.gridR = gridR
        self.searchR = gridR * 2.0
        self.neighborCount = Field(particle_data, "neighborCount", 1, np.int32)
        self.neighbor = _NeighborList(self)
        self.grid = _GridTable(self)
        self.blockSize = _HostVec(np.int32)
        self.min_boundary = _HostVec(np.float32)
        self.max_boundary = _HostVec(np.float32)

    def setup_grid_gpu(self):
        """ref: HashGrid.py:34-40 -- table allocation; device memory is owned by the library context."""
        self.particle_data._need_ctx()

    def setup_grid_cpu(self, maxboundarynp, minboundarynp):
        """ref: HashGrid.py:44-54 -- fixes the grid box (never updated afterwards, Q4) and uploads positions."""
        pd = self.particle_data
        ctx = pd._need_ctx()
        lib = _lib.load()
        self.max_boundary.from_numpy(maxboundarynp)
        self.min_boundary.from_numpy(minboundarynp)
        pos = pd._initial_positions()
        maxb = np.ascontiguousarray(maxboundarynp, np.float32).reshape(3)
        minb = np.ascontiguousarray(minboundarynp, np.float32).reshape(3)
        _lib.check(ctx, lib.wcsph_setup(ctx, pos.ctypes.data, maxb.ctypes.data, minb.ctypes.data))
        bs = (C.c_int32 * 3)()
        _lib.check(ctx, lib.wcsph_get_block_size(ctx, C.byref(bs)))
        self.blockSize.from_numpy(np.array(list(bs), np.int32))
        print("serach grid szie:", int(bs[0]) * int(bs[1]) * int(bs[2]))

    def update_grid(self):
        """ref: HashGrid.py:56-85"""
        ctx = self.particle_data._need_ctx()
        _lib.check(ctx, _lib.load().wcsph_grid_update(ctx))

    # ---- host-side inspection helpers (not on the hot path) ----------------------------------
    def process_neighbour(self, index):
        """ref: HashGrid.py:126-142 -- one level of the reference's diagnostic "tree": every slot i that is a multiple of
        `index` takes max(gridCount[i], gridCount[i + index // 2]).  (It re-reads gridCount, not the previous level, so
        the tree does not propagate; reproduced as written.  Slots whose partner lies past the table keep their value --
        the reference reads out of bounds there.)"""
        gc = self.gridCount
        if getattr(self, "maxCurNeighbour", None) is None or len(self.maxCurNeighbour) != len(gc):
            self.maxCurNeighbour = np.zeros(len(gc), np.int32)
        i = np.arange(0, len(gc), int(index))
        i = i[i + int(index) // 2 < len(gc)]
        self.maxCurNeighbour[i] = np.maximum(gc[i], gc[i + int(index) // 2])

    def get_max_neighbour(self):
        """ref: HashGrid.py:146-152 (never called by the reference's drivers); prints and returns slot 0"""
        size = 2
        while size < self.particle_data.liquid_count:
            self.process_neighbour(size)
            size = size * 2
        v = int(self.maxCurNeighbour[0]) if getattr(self, "maxCurNeighbour", None) is not None else 0
        print(v)
        return v

    def _cells(self, pos):
        mn = self.min_boundary.to_numpy()[0]
        f = (pos.astype(np.float32) - mn) * np.float32(self.invGridR)
        return np.trunc(f).astype(np.int64)

    def get_cell_hash(self, a):
        """ref: HashGrid.py:108-114 (int32 wrap-around), vectorised; `a` is [...,3] integer cell coordinates."""
        a = np.asarray(a, dtype=np.int64)
        def wrap(v):
            return ((v + 2 ** 31) % 2 ** 32) - 2 ** 31
        p1, p2, p3 = wrap(73856093 * a[..., 0]), wrap(19349663 * a[..., 1]), wrap(83492791 * a[..., 2])
        x = (p1.astype(np.int32) ^ p2.astype(np.int32) ^ p3.astype(np.int32)).astype(np.int64)
        n = self.particle_data.hash_table_size or self.particle_data.count
        return np.mod(x, n)  # numpy mod is already the non-negative residue

    def check_in_box(self, index):
        bs = self.blockSize.to_numpy()[0]
        index = np.asarray(index)
        return np.all((index >= 0) & (index < bs), axis=-1).astype(np.int32)

    @property
    def gridCount(self):
        """Reference bucket occupancy (ref: HashGrid.py:20,71).  Faithful mode: the device's own table of this step
        (the counts the raw neighbour counts are built from); clean mode has no bucket table and the counts are
        rebuilt on the host for inspection."""
        pd = self.particle_data
        if pd.neighbor_mode == "faithful":
            ctx = pd._need_ctx()
            n = pd.hash_table_size or pd.count
            out = np.empty(n, np.int32)
            _lib.check(ctx, _lib.load().wcsph_hashgrid_count_to_host(ctx, out.ctypes.data, n))
            return out
        pos = self.particle_data.pos.to_numpy()
        cells = self._cells(pos)
        inb = self.check_in_box(cells).astype(bool)
        n = self.particle_data.hash_table_size or self.particle_data.count
        return np.bincount(self.get_cell_hash(cells[inb]), minlength=n).astype(np.int32)
