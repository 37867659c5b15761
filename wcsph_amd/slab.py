"""1-D slab decomposition of the DFSPH step across the GPUs of one node.

The reference is single-device; this module is new work with no reference counterpart.  One process per
GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU tests).  The fluid
domain is cut along x on hash-cell boundaries; rank r owns the liquid particles whose cell column lies in
[cut[r], cut[r+1]) and holds GHOST copies of the neighbours' particles within two cell columns (= the
kernel support h) of each cut, plus the static boundary particles of its columns +- 2.

Per step (the reference main-loop body, dfsph.py:606-617, with the exchanges it needs once the neighbour
loops read particles another rank owns):
  migrate owners that crossed a cut -> rebuild ghosts -> grid update -> [phase, halo refresh of the field
  the NEXT phase reads from neighbours j]* -> scalar all-reduces where the reference reads a global sum / max.
Every exchange is a pair of point-to-point messages with the two slab neighbours (batch_isend_irecv:
ncclSend/ncclRecv grouped), 0.1-2 MB each; the only collectives are all-reduces of one double.

The choreography is written against a small backend protocol so that the CPU tests can drive it with a
checker backend; the product backend is HipBackend (libwcsph_hip.so through the C ABI) and nothing else.
"""
import ctypes as C
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, scene

STATE_W = 15  # pos3 vel3 omega3 vel_guess3 kappa kappa_v (the fields the reference carries across steps) + global id (int bits)
GHOST_CELLS = 2  # support h = 2 hash cells (HashGrid.py:18: searchR = 2 * gridR)
MIN_SLAB = 2 * GHOST_CELLS + 2  # cell columns: a particle that has just crossed one cut is never within reach of the other
FIELD_W = {"vel": 3, "vel_guess": 3, "cg_dir": 3, "omega": 3, "normal": 3}   # everything else is a scalar


class _DevArray:
    """device memory owned by the library, presented to torch (zero copy) through the CUDA array interface"""

    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


# ---------------------------------------------------------------------------------------------------------
class HipBackend:
    """Compute backend of one rank: a libwcsph_hip.so context with a slab-local cell window."""

    def __init__(self, capacity, solid_local, bounds, window, particleR=0.025, device=None, neighbor_mode="clean",
                 owned=None, n_global=(0, 0), stream=None, **attrs):
        from .ParticleData import ParticleData
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.load()
        self.faithful = neighbor_mode == "faithful"
        pd = ParticleData(particleR, neighbor_mode=neighbor_mode)
        pd.tension_fixed = 1   # the literal compute_tension reads the reference's raw bucket rows, which span all ranks
        for k, v in attrs.items():
            setattr(pd, k, v)
        if self.faithful:   # the reference hash table has `count` buckets: the GLOBAL particle count (HashGrid.py:114)
            pd.hash_table_size = int(n_global[0]) + int(n_global[1])
        # capacity rows of liquid; positions are loaded per step (load_state)
        pd.liquid_count = pd.count = int(capacity)
        pd._chunks = [np.zeros((int(capacity), 3))]
        pd.add_solid_points(solid_local)
        pd.maxboundarynp[0] = np.asarray(bounds[1], np.float32)
        pd.minboundarynp[0] = np.asarray(bounds[0], np.float32)
        pd.setup_data_gpu()
        self.ctx = pd._need_ctx()
        _lib.check(self.ctx, self.lib.wcsph_set_cell_window(self.ctx, int(window[0]), int(window[1])))
        if owned is not None:
            _lib.check(self.ctx, self.lib.wcsph_set_owned_columns(self.ctx, int(owned[0]), int(owned[1])))
        if self.faithful:
            _lib.check(self.ctx, self.lib.wcsph_set_global_liquid(self.ctx, int(n_global[0])))
        # one HIP stream for everything this rank does: the library's kernels, torch's row shuffling and the
        # RCCL calls torch.distributed enqueues are ordered by it (SlabSim.step runs under torch.cuda.stream).  SlabSim
        # passes ITS stream, so a backend rebuilt by a rebalance keeps the stream the rows were written on.
        self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
        _lib.check(self.ctx, self.lib.wcsph_set_stream(self.ctx, C.c_void_p(self.stream.cuda_stream)))
        pd.setup_data_cpu()
        # reduction slots in a torch tensor: the slab driver all-reduces them in place on the device (SlabSim.reduce_acc)
        self.acc = torch.zeros(8 * _lib.ACC_STRIPES, dtype=torch.float64, device=self.device)
        _lib.check(self.ctx, self.lib.wcsph_set_acc_buffer(self.ctx, C.c_void_p(self.acc.data_ptr())))
        torch.cuda.synchronize(self.device)
        self.pd = pd
        self.capacity = int(capacity)
        self.n_owned = 0

    def _chk(self, rc):
        _lib.check(self.ctx, rc)

    def load_state_split(self, owned, ghosts):
        """owned rows and ghost rows as two tensors (no concatenation of the whole state)"""
        owned = owned.contiguous(); ghosts = [g.contiguous() for g in ghosts if g.shape[0]]
        ghost = ghosts[0] if len(ghosts) == 1 else (torch.cat(ghosts, 0) if ghosts else owned[:0])
        if owned.shape[0] + ghost.shape[0] > self.capacity:
            raise RuntimeError("slab capacity %d exceeded (%d rows)" % (self.capacity, owned.shape[0] + ghost.shape[0]))
        self.n_owned = int(owned.shape[0])
        self._chk(self.lib.wcsph_load_state_split(self.ctx, self.n_owned, C.c_void_p(owned.data_ptr()), int(ghost.shape[0]),
                                                  C.c_void_p(ghost.data_ptr())))
        self._keep = (owned, ghost)  # keep alive until the stream has consumed them

    def load_state(self, n_owned, rows):
        rows = rows.contiguous()
        if rows.shape[0] > self.capacity:
            raise RuntimeError("slab capacity %d exceeded (%d rows)" % (self.capacity, rows.shape[0]))
        self.n_owned = int(n_owned)
        self._chk(self.lib.wcsph_load_state(self.ctx, int(n_owned), int(rows.shape[0]), C.c_void_p(rows.data_ptr())))
        self._keep = rows  # keep alive until the stream has consumed it

    def classify_rows(self, rows, lo, hi, ghost, has_l, has_r):
        """wcsph_slab_classify: ((near_l, near_r, go_l, go_r, lost) counts, four int64 index lists) in one launch and one
        host wait; None when a list did not fit its buffer (the caller falls back to the torch form)"""
        n = int(rows.shape[0])
        cap = max(65536, n // 8)
        if getattr(self, "_cls_idx", None) is None or self._cls_idx.shape[1] < cap:
            self._cls_idx = torch.empty((4, cap), dtype=torch.int64, device=self.device)
            self._cls_cnt = torch.empty(5, dtype=torch.int32, device=self.device)
        rows = rows.contiguous()
        self._chk(self.lib.wcsph_slab_classify(self.ctx, C.c_void_p(rows.data_ptr()), n, int(lo), int(hi), int(ghost), int(has_l), int(has_r),
                                               C.c_void_p(self._cls_idx.data_ptr()), int(self._cls_idx.shape[1]), C.c_void_p(self._cls_cnt.data_ptr())))
        c = self._cls_cnt.tolist()
        if max(c[:4]) > self._cls_idx.shape[1]:
            return None
        return c, [self._cls_idx[k, :c[k]] for k in range(4)]

    # ---- state resident in the library between steps (include/wcsph.h "slab state resident in the library") -------------
    def classify_state(self, lo, hi, ghost, has_l, has_r, grow=False):
        """classify_rows over the context's OWN owned rows (no row table in the driver): counts + four int64 lists of
        original indices; None when a list did not fit its buffer (grow: a larger buffer and once more instead).  One launch and
        one host wait -- or neither (round 6): the library takes the same classification in the tail of wcsph_slab_step and its
        counts arrive with the step's last mail (`self.classify_waited` says which it was)."""
        n = int(self.n_owned)
        cap = max(65536, n // 8)
        if getattr(self, "_cls_idx", None) is None or self._cls_idx.shape[1] < cap:
            self._cls_idx = torch.empty((4, cap), dtype=torch.int64, device=self.device)
            self._cls_cnt = torch.empty(5, dtype=torch.int32, device=self.device)
            self._tc_key = None
        c = None
        self.classify_waited = False
        for _attempt in range(2):
            key = (int(lo), int(hi), int(ghost), int(has_l), int(has_r), int(self._cls_idx.shape[1]), self._cls_idx.data_ptr())
            if c is None and os.environ.get("WCSPH_SLAB_TAIL_CLASSIFY", "1") != "0":
                if getattr(self, "_tc_key", None) != key:       # (re)register: the next step's tail classifies with these parameters
                    self._chk(self.lib.wcsph_slab_set_tail_classify(self.ctx, key[0], key[1], key[2], key[3], key[4], C.c_void_p(self._cls_idx.data_ptr()), key[5],
                                                                    C.c_void_p(self._cls_cnt.data_ptr())))
                    self._tc_key = key
                else:
                    got = (C.c_int32 * 5)()
                    rc = self.lib.wcsph_slab_tail_classify_result(self.ctx, C.byref(got))
                    if rc < 0:
                        self._chk(rc)
                    if rc in (0, 2):                            # 2: the step ended on its plain path, the call has waited for the counts
                        c = list(got)
                        self.classify_waited = rc == 2
            if c is None:
                self._chk(self.lib.wcsph_state_classify(self.ctx, int(lo), int(hi), int(ghost), int(has_l), int(has_r),
                                                        C.c_void_p(self._cls_idx.data_ptr()), int(self._cls_idx.shape[1]), C.c_void_p(self._cls_cnt.data_ptr())))
                c = self._cls_cnt.tolist()
                self.classify_waited = True
            if max(c[:4]) <= self._cls_idx.shape[1]:
                break
            if not grow:
                return None
            # a list did not fit (the lists were cut at the buffer's end): a larger buffer, the classification once more
            self._cls_idx = torch.empty((4, int(max(c[:4]) * 1.25) + 4096), dtype=torch.int64, device=self.device)
            self._tc_key = None
            c = None
        return c, [self._cls_idx[k, :c[k]] for k in range(4)]

    # ---- migration + ghost rebuild inside the library (csrc/api_migrate.h) ------------------------------------------------
    def migrate_begin(self, counts):
        """the rows that travel, gathered into the library's send buffer: (to_left, to_right) as torch views of it"""
        c5 = (C.c_int32 * 5)(*[int(v) for v in counts])
        pl, pr = C.c_void_p(), C.c_void_p()
        self._chk(self.lib.wcsph_slab_migrate_begin(self.ctx, C.c_void_p(self._cls_idx.data_ptr()), int(self._cls_idx.shape[1]), C.byref(c5), C.byref(pl), C.byref(pr)))
        n_l, n_r = int(counts[2] + counts[0]), int(counts[3] + counts[1])
        view = lambda ptr, n: (torch.as_tensor(_DevArray(ptr.value, n * STATE_W, "<f4"), device=self.device).view(n, STATE_W) if n   # noqa: E731
                               else torch.empty((0, STATE_W), dtype=torch.float32, device=self.device))
        return view(pl, n_l), view(pr, n_r)

    def _adopt_migration(self, out):
        out = [int(v) for v in out]
        self.n_owned = out[0]
        return out

    def halo_list_views(self):
        """the halo lists the library has installed (wcsph_slab_halo_lists) as torch int32 views: (send list, ghost list)"""
        ps, pg, ns, ng = C.c_void_p(), C.c_void_p(), C.c_int32(), C.c_int32()
        self._chk(self.lib.wcsph_slab_halo_lists(self.ctx, C.byref(ps), C.byref(ns), C.byref(pg), C.byref(ng)))
        i32 = lambda ptr, n: (torch.as_tensor(_DevArray(ptr.value, n, "<i4"), device=self.device) if n else torch.zeros(0, dtype=torch.int32, device=self.device))   # noqa: E731
        return i32(ps, ns.value), i32(pg, ng.value)

    def migrate_end(self, in_l, a_l, in_r, a_r):
        """what the neighbours sent (the first a_l / a_r rows of a message become this rank's): plan, rows and halo lists on the
        device.  Returns [n_owned, n_send_left, n_send_right, n_ghost_left, n_ghost_right]; the halo lists are installed (halo_list_views)"""
        in_l, in_r = in_l.contiguous(), in_r.contiguous()
        out = (C.c_int32 * 5)()
        self._chk(self.lib.wcsph_slab_migrate_end(self.ctx, C.c_void_p(in_l.data_ptr()), int(in_l.shape[0]), int(a_l),
                                                  C.c_void_p(in_r.data_ptr()), int(in_r.shape[0]), int(a_r), C.byref(out)))
        self._keep = (in_l, in_r)            # alive until the stream has consumed them
        return self._adopt_migration(out)

    def tail_neighbour_counts(self):
        """native data plane: the slab neighbours' counts of the classification taken in the last step's tail, or None"""
        l, r = (C.c_int32 * 5)(), (C.c_int32 * 5)()
        rc = self.lib.wcsph_slab_tail_neighbour_counts(self.ctx, C.byref(l), C.byref(r))
        if rc < 0:
            self._chk(rc)
        return (list(l), list(r)) if rc == 0 else None

    def migrate_native(self, counts, nbr_l, nbr_r):
        """begin + one grouped ncclSend / ncclRecv of exact sizes + end inside the library (wcsph_slab_migrate)"""
        c5, l5, r5 = (C.c_int32 * 5)(*[int(v) for v in counts]), (C.c_int32 * 5)(*nbr_l), (C.c_int32 * 5)(*nbr_r)
        out = (C.c_int32 * 5)()
        self._chk(self.lib.wcsph_slab_migrate(self.ctx, C.c_void_p(self._cls_idx.data_ptr()), int(self._cls_idx.shape[1]), C.byref(c5), C.byref(l5), C.byref(r5), C.byref(out)))
        return self._adopt_migration(out)

    def gather_state_rows(self, idx):
        """the STATE_W-float rows of the listed original indices (int64, device)"""
        idx = idx.contiguous()
        out = torch.empty((int(idx.numel()), STATE_W), dtype=torch.float32, device=self.device)
        if idx.numel():
            self._chk(self.lib.wcsph_state_gather_rows(self.ctx, C.c_void_p(idx.data_ptr()), int(idx.numel()), C.c_void_p(out.data_ptr())))
        return out

    def apply_state(self, n_owned_new, fold_dst, fold_src, write_idx, write_rows, ghosts):
        """ownership changes applied to the resident state (wcsph_state_apply): relabels, arrivals, the new ghost rows"""
        fold_dst, fold_src, write_idx = fold_dst.contiguous(), fold_src.contiguous(), write_idx.contiguous()
        write_rows, ghosts = write_rows.contiguous(), ghosts.contiguous()
        if n_owned_new + ghosts.shape[0] > self.capacity:
            raise RuntimeError("slab capacity %d exceeded (%d rows)" % (self.capacity, n_owned_new + ghosts.shape[0]))
        self._chk(self.lib.wcsph_state_apply(self.ctx, int(n_owned_new), C.c_void_p(fold_dst.data_ptr()), C.c_void_p(fold_src.data_ptr()), int(fold_dst.numel()),
                                             C.c_void_p(write_idx.data_ptr()), C.c_void_p(write_rows.data_ptr()), int(write_idx.numel()),
                                             C.c_void_p(ghosts.data_ptr()), int(ghosts.shape[0])))
        self.n_owned = int(n_owned_new)
        self._keep = (fold_dst, fold_src, write_idx, write_rows, ghosts)   # alive until the stream has consumed them

    def store_state(self):
        # a view of a slightly larger buffer: rows that arrive next step are appended in place (SlabSim._swap_rows)
        slack = max(4096, self.n_owned // 64)
        base = torch.empty((self.n_owned + slack, STATE_W), dtype=torch.float32, device=self.device)
        self._chk(self.lib.wcsph_store_state(self.ctx, C.c_void_p(base.data_ptr())))
        return base[:self.n_owned]

    def grid_update(self):
        self._chk(self.lib.wcsph_grid_update(self.ctx))

    def grid_update_begin(self):
        self._chk(self.lib.wcsph_grid_update_begin(self.ctx))

    def grid_update_end(self):
        self._chk(self.lib.wcsph_grid_update_end(self.ctx))

    def bucket_hist(self, which):
        """the library's bucket histogram (0: liquid of this step, 1: static boundary) as a torch int32 view"""
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.wcsph_bucket_hist(self.ctx, int(which), C.byref(p), C.byref(n)))
        return torch.as_tensor(_DevArray(p.value, n.value), device=self.device)

    def bucket_hist_packed(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.wcsph_bucket_hist_pack(self.ctx, C.byref(p), C.byref(n)))
        return torch.as_tensor(_DevArray(p.value, n.value), device=self.device)

    def bucket_hist_unpack(self):
        self._chk(self.lib.wcsph_bucket_hist_unpack(self.ctx))

    def pack_rows(self, fields, idx, out):
        """rows idx of several fields side by side into `out` ([n, sum of widths]); one launch"""
        ids = (C.c_int32 * len(fields))(*[_lib.FIELD_ID[f] for f in fields])
        if idx.numel():
            self._chk(self.lib.wcsph_halo_pack(self.ctx, len(fields), ids, C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(out.data_ptr())))

    def unpack_rows(self, fields, idx, inp):
        ids = (C.c_int32 * len(fields))(*[_lib.FIELD_ID[f] for f in fields])
        if idx.numel():
            self._chk(self.lib.wcsph_halo_unpack(self.ctx, len(fields), ids, C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(inp.data_ptr())))

    def phase(self, name):
        self._chk(self.lib.wcsph_phase(self.ctx, _lib.PHASE_ID[name]))

    def cfl_local_max(self):
        self._chk(self.lib.wcsph_dfsph_cfl_reduce(self.ctx))
        v = C.c_float()
        self._chk(self.lib.wcsph_get_scalar(self.ctx, _lib.SCALAR_ID["vel_max0"], C.byref(v)))
        return float(v.value)

    def set_part(self, mode, x_lo=0, x_hi=0):
        """restrict the neighbour passes launched next: 0 all, 1 interior columns [x_lo, x_hi), 2 the rest"""
        self._chk(self.lib.wcsph_set_part(self.ctx, int(mode), int(x_lo), int(x_hi)))

    def get_acc(self, slot):
        v = C.c_double()
        self._chk(self.lib.wcsph_get_acc(self.ctx, slot, C.byref(v)))
        return v.value

    def set_acc(self, slot, v):
        self._chk(self.lib.wcsph_set_acc(self.ctx, slot, float(v)))

    def set_scalar(self, name, v):
        self._chk(self.lib.wcsph_set_scalar(self.ctx, _lib.SCALAR_ID[name], float(v)))

    def gather_rows(self, field, idx):
        w = 3 if field in ("vel", "vel_guess", "cg_dir", "omega", "normal") else 1
        out = torch.empty((idx.numel(), w), dtype=torch.float32, device=self.device)
        if idx.numel():
            self._chk(self.lib.wcsph_field_gather_rows(self.ctx, _lib.FIELD_ID[field], C.c_void_p(idx.data_ptr()),
                                                       idx.numel(), C.c_void_p(out.data_ptr())))
        return out

    def scatter_rows(self, field, idx, rows):
        if idx.numel():
            rows = rows.contiguous()
            self._chk(self.lib.wcsph_field_scatter_rows(self.ctx, _lib.FIELD_ID[field], C.c_void_p(idx.data_ptr()),
                                                        idx.numel(), C.c_void_p(rows.data_ptr())))
            self._keep2 = rows

    def sync(self):
        self._chk(self.lib.wcsph_sync(self.ctx))

    # ---- the native RCCL data plane (include/wcsph.h "native data plane", csrc/api_comm.h) ---------------------------
    native_comm = False

    def comm_init(self, unique_id, rank, world, neighbours=None):
        """ncclCommInitRank inside the library; neighbours = (left, right) overrides rank -+ 1 (-1: none; both == rank: loop-back)"""
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        self._chk(self.lib.wcsph_comm_init(self.ctx, buf, int(rank), int(world)))
        if neighbours is not None:
            self._chk(self.lib.wcsph_comm_set_neighbours(self.ctx, int(neighbours[0]), int(neighbours[1])))
        self.native_comm = True

    def comm_set_mode(self, mode):
        """1: halo refreshes on the communicator's own stream (beside the interior pass); 0: inline on the context's stream"""
        self._chk(self.lib.wcsph_comm_set_mode(self.ctx, int(mode)))

    def comm_detach(self):
        h = C.c_void_p()
        self._chk(self.lib.wcsph_comm_detach(self.ctx, C.byref(h)))
        self.native_comm = False
        return h

    def comm_attach(self, handle):
        self._chk(self.lib.wcsph_comm_attach(self.ctx, handle))
        self.native_comm = True

    def comm_probe(self, rank, world):
        """one small grouped exchange with the slab neighbours and one all-reduce of each kind through the library's own
        communicator, verified; raises on a mismatch (bench: before the data plane is trusted with the run)"""
        lib, ctx, dev = self.lib, self.ctx, self.device
        has_l, has_r, n = rank > 0, rank < world - 1, 128
        ns = n * (int(has_l) + int(has_r))
        with torch.cuda.stream(self.stream):
            idx = torch.zeros(max(ns, 1), dtype=torch.int32, device=dev)
            send = torch.full((max(ns, 1), 3), float(rank), dtype=torch.float32, device=dev)
            recv = torch.full((max(ns, 1), 3), -1.0, dtype=torch.float32, device=dev)
            self._chk(lib.wcsph_slab_set_halo(ctx, C.c_void_p(idx.data_ptr()), ns, C.c_void_p(idx.data_ptr()), ns))
            self._chk(lib.wcsph_comm_set_halo_split(ctx, n * int(has_l), n * int(has_l)))
            self._chk(lib.wcsph_halo_exchange(ctx, 3, C.c_void_p(send.data_ptr()), ns, C.c_void_p(recv.data_ptr()), ns))
            w = torch.tensor([rank, 7], dtype=torch.int32, device=dev)
            dbl = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
            mx = torch.tensor([rank], dtype=torch.int32, device=dev)
            self._chk(lib.wcsph_comm_allreduce(ctx, C.c_void_p(w.data_ptr()), 2, 0))
            self._chk(lib.wcsph_comm_allreduce(ctx, C.c_void_p(dbl.data_ptr()), 64, 1))
            self._chk(lib.wcsph_comm_allreduce(ctx, C.c_void_p(mx.data_ptr()), 1, 2))
            self.sync()
        torch.cuda.synchronize(dev)
        exp = ([float(rank - 1)] * n if has_l else []) + ([float(rank + 1)] * n if has_r else [])
        got = recv[:ns, 0].tolist()
        if got != exp or (ns and recv[:ns].min().item() != min(exp)):
            raise RuntimeError("native halo exchange returned wrong rows")
        if w.tolist() != [world * (world - 1) // 2, 7 * world] or float(dbl[0].item()) != world * (world + 1) / 2.0 or int(mx.item()) != world - 1:
            raise RuntimeError("native all-reduce returned %r / %r / %r" % (w.tolist(), float(dbl[0].item()), int(mx.item())))
        self._chk(lib.wcsph_slab_set_halo(ctx, None, 0, None, 0))
        self._chk(lib.wcsph_comm_set_halo_split(ctx, 0, 0))

    def comm_stats(self):
        out = (C.c_int64 * 2)()
        self._chk(self.lib.wcsph_comm_stats(self.ctx, C.byref(out)))
        return int(out[0]), int(out[1])

    # ---- the whole step in one library call (include/wcsph.h wcsph_slab_step) --------------------------------------
    def native_step(self, sim):
        """The choreography of SlabSim._step inside the library; `sim` supplies the communicator through callbacks.
        Returns (dt, vs_iter, dv_iter, pr_iter, avg_density_err)."""
        lib, ctx = self.lib, self.ctx
        if getattr(self, "_cbs_for", None) is not sim:
            self._cbs, self._cbs_for = sim._make_callbacks(), sim
        lib_halo = sim.__dict__.get("_halo_lib") is not None      # installed by wcsph_slab_migrate(_end), split included
        if not lib_halo:
            self._chk(lib.wcsph_slab_set_halo(ctx, C.c_void_p(sim.send_all.data_ptr()), int(sim.send_all.numel()),
                                              C.c_void_p(sim.ghost_all.data_ptr()), int(sim.ghost_all.numel())))
        self._chk(lib.wcsph_set_scalar(ctx, _lib.SCALAR_ID["deltaT"], float(sim.dt)))
        self._chk(lib.wcsph_set_scalar(ctx, _lib.SCALAR_ID["avg_density_err"], float(sim.avg_density_err)))
        self._chk(lib.wcsph_set_iters(ctx, int(sim.vs_iter), int(sim.dv_iter), int(sim.pr_iter)))
        stats = (C.c_int32 * 2)()
        flags = (1 if sim.overlap else 0) | (2 if sim.tension else 0) | (0 if sim.speculate else 4) | (8 if getattr(sim, "force_comm", False) else 0)
        sim._cb_exc = None
        if self.native_comm:
            # the library's own communicator: no callback, no Python between the library and RCCL
            if not lib_halo:
                self._chk(lib.wcsph_comm_set_halo_split(ctx, int(sim.send_l.numel()), int(sim.ghost_l.numel())))
            cbp = None
        else:
            cbp = C.byref(self._cbs)
        rc = lib.wcsph_slab_step(ctx, cbp, int(sim.n_global), int(max(sim.interior[0], -(1 << 30))),
                                 int(min(sim.interior[1], 1 << 30)), flags, C.byref(stats))
        if sim._cb_exc is not None:
            raise sim._cb_exc
        self._chk(rc)
        sim.n_host_waits += stats[1]       # (the exchanges of a callback step are counted where they are posted: SlabSim._post)
        if self.native_comm:
            sim.n_exchanges += stats[0]
        dt, err = C.c_float(), C.c_float()
        vs, dv, pr = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(lib.wcsph_get_scalar(ctx, _lib.SCALAR_ID["deltaT"], C.byref(dt)))
        self._chk(lib.wcsph_get_scalar(ctx, _lib.SCALAR_ID["avg_density_err"], C.byref(err)))
        self._chk(lib.wcsph_get_iters(ctx, C.byref(vs), C.byref(dv), C.byref(pr)))
        return np.float32(dt.value), vs.value, dv.value, pr.value, np.float32(err.value)

    def close(self):
        self.pd.close()


# ---------------------------------------------------------------------------------------------------------
def balanced_cuts(cellx_hist, world, min_width=GHOST_CELLS):
    """Cell-column cuts [world+1] giving ~equal particle counts per slab; cut[0] = 0, cut[-1] = nx; every slab at least
    min_width columns wide (SlabSim uses MIN_SLAB)."""
    nx = len(cellx_hist)
    if nx < world * min_width:
        raise ValueError("the domain is %d cell columns wide: %d slabs need at least %d columns each (%d in all); use fewer "
                         "ranks or a wider box" % (nx, world, min_width, world * min_width))
    cum = np.concatenate([[0], np.cumsum(cellx_hist)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(cum, total * r / world, side="left"))
        c = max(c, cuts[-1] + min_width)
        c = min(c, nx - (world - r) * min_width)   # leave room for the slabs to the right
        cuts.append(max(min(c, nx), cuts[-1]))
    cuts.append(nx)
    return cuts


class _HaloList:
    """one of SlabSim's six halo index lists (int32 device tensors).  After a migration inside the library (csrc/api_migrate.h) the
    lists live there and are installed already: a torch view is only made when somebody asks for one (the step itself does not:
    two torch.as_tensor() of a __cuda_array_interface__ cost more host time per step than the migration's own calls)."""

    def __init__(self, name):
        self.key = "_h_" + name

    def __get__(self, obj, typ=None):
        if obj is None:
            return self
        v = obj.__dict__.get(self.key)
        if v is None and obj.__dict__.get("_halo_lib") is not None:
            obj._materialise_halo()
            v = obj.__dict__.get(self.key)
        return v

    def __set__(self, obj, v):
        obj.__dict__[self.key] = v
        obj.__dict__["_halo_lib"] = None          # the driver's own lists: wcsph_slab_set_halo hands them over before the step


class SlabSim:
    """The DFSPH step of dfsph.py:606-617 on one slab, with halo exchanges between phases."""
    send_l, send_r, ghost_l, ghost_r, send_all, ghost_all = (_HaloList(k) for k in ("send_l", "send_r", "ghost_l", "ghost_r", "send_all", "ghost_all"))

    def _halo_counts(self):
        """(rows sent per refresh, ghost rows, of them to / from the left neighbour) without touching the lists"""
        lib = self.__dict__.get("_halo_lib")
        if lib is not None:
            return lib[0] + lib[1], lib[2] + lib[3], lib[0], lib[2]
        return int(self.send_all.numel()), int(self.ghost_all.numel()), int(self.send_l.numel()), int(self.ghost_l.numel())

    def _materialise_halo(self):
        n_sl, n_sr, n_gl, n_gr = self._halo_lib
        send_all, ghost_all = self.B.halo_list_views()
        d = self.__dict__
        d["_h_send_all"], d["_h_ghost_all"] = send_all, ghost_all
        d["_h_send_l"], d["_h_send_r"] = send_all[:n_sl], send_all[n_sl:]
        d["_h_ghost_l"], d["_h_ghost_r"] = ghost_all[:n_gl], ghost_all[n_gl:]

    def _adopt_library_halo(self, out):
        """the lists wcsph_slab_migrate(_end) left installed: out = [n_owned, n_send_left, n_send_right, n_ghost_left, n_ghost_right]"""
        d = self.__dict__
        for k in ("send_l", "send_r", "ghost_l", "ghost_r", "send_all", "ghost_all"):
            d["_h_" + k] = None
        d["_halo_lib"] = (out[1], out[2], out[3], out[4])
        self.n_owned = out[0]

    def __init__(self, backend_factory, liquid_local, solid_all, bounds, cuts, n_global, rank, world,
                 particleR=0.025, device=None, capacity_factor=1.6, group=None, comm_device=None,
                 neighbor_mode="clean", gid_local=None, rebalance_every=0):
        self.rank, self.world, self.group = rank, world, group
        self.device = device if device is not None else torch.device("cpu")
        # tensors handed to torch.distributed live here: the compute device for nccl (RCCL moves them over
        # xGMI directly), the host for gloo (tests: two ranks sharing one GPU, or CPU-only checker runs)
        self.comm_device = comm_device if comm_device is not None else self.device
        self.minb = np.asarray(bounds[0], np.float32)
        self.maxb = np.asarray(bounds[1], np.float32)
        self.particleR = particleR
        grid_r = particleR * 2.0
        self.inv_grid = np.float32(1.0 / grid_r)
        self.nx = int(float(self.maxb[0] - self.minb[0]) / grid_r + 1.0)   # HashGrid.py:46-48
        self.n_global = int(n_global)
        self.neighbor_mode = neighbor_mode
        self.solid_all = np.asarray(solid_all)
        self._factory, self._cap_factor = backend_factory, capacity_factor
        self.rebalance_every, self.steps_done, self.n_rebalanced = int(rebalance_every), 0, 0
        n_local = len(liquid_local)
        rows = np.zeros((n_local, STATE_W), np.float32)
        rows[:, 0:3] = np.asarray(liquid_local, np.float64).astype(np.float32)
        if gid_local is None:   # global ids: rank-major numbering (callers that care -- CFL prefix, Q5 -- pass the scene's own)
            counts = [n_local]
            if world > 1:
                t = torch.tensor([n_local], dtype=torch.int64, device=self.comm_device)
                allc = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(allc, t, group=group)
                counts = [int(c.item()) for c in allc]
            gid_local = sum(counts[:rank]) + np.arange(n_local, dtype=np.int64)
        rows[:, 14] = np.asarray(gid_local, np.int32).view(np.float32)
        self._rows = None
        self.rows = torch.from_numpy(rows).to(self.device)
        self.B = None
        # ONE stream for the life of the simulation, handed to every backend (a rebalance rebuilds the backend)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._set_cuts(list(cuts))
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        # reference module globals (dfsph.py:36-41,178)
        self.dt = np.float32(0.001)
        self.vs_iter = self.dv_iter = self.pr_iter = 0
        self.avg_density_err = np.float32(0.0)
        self.cg_delta = self.cg_delta_zero = np.float32(0.0)
        self.eps, self.viscosity_err = 1e-5, 0.05
        self.user_max_t, self.user_min_t = 0.005, 0.0001
        self.tension = False
        self.n_exchanges = self.n_host_waits = 0
        # WCSPH_SLAB_NATIVE=0: drive the step phase by phase from Python (the form the checker backend of the CPU tests
        # runs, kept in step with wcsph_slab_step by tests/test_gpu_slab.py)
        self.native = os.environ.get("WCSPH_SLAB_NATIVE", "1") != "0"
        self.resident = os.environ.get("WCSPH_SLAB_RESIDENT", "1") != "0"
        # the native step queues the common-case continuation behind every solver decision (wcsph_slab_step); False: one host
        # wait per decision (tests compare the two)
        self.speculate = os.environ.get("WCSPH_SLAB_SPECULATE", "1") != "0"
        self._words_work = None
        self._cb_exc = None
        # flags bit 3 of wcsph_slab_step: call the communicator at every exchange point even with one rank (tools/slab_overhead.py)
        self.force_comm = os.environ.get("WCSPH_SLAB_FORCE_COMM", "0") == "1"
        # the rows of a migration over the library's own RCCL communicator when it is on (wcsph_slab_migrate); False: over this
        # rank's torch.distributed group with a header round trip, as on the callback plane
        self.native_migrate = os.environ.get("WCSPH_SLAB_NATIVE_MIGRATE", "1") != "0"
        # swap plan, row writes and halo lists of a migration inside the library (csrc/api_migrate.h); False: the torch form (rounds 5-6)
        self.library_migrate = os.environ.get("WCSPH_SLAB_LIBRARY_MIGRATE", "1") != "0"
        self.n_rows_arrived = 0

    # ---- the owned rows (STATE_W floats each) ------------------------------------------------------------------------------
    # With the HIP backend and the native step the state LIVES IN THE LIBRARY between steps (WCSPH_SLAB_RESIDENT=0 switches it
    # off): `rows` is then materialised on demand (wcsph_store_state) -- by a test, a tool, the rebalance -- and a
    # materialised table is the truth again: the next redistribute() loads it in full, as rounds 2-4 did every step
    # (0.052 + 0.047 ms of the driver's 0.115 ms per step at C3, profiles/r05_slab_overhead.txt).
    @property
    def rows(self):
        if self._rows is None:
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    self._rows = self.B.store_state()
            else:
                self._rows = self.B.store_state()
        return self._rows

    @rows.setter
    def rows(self, value):
        self._rows = value

    # ---- the native data plane: RCCL inside the library instead of torch.distributed behind callbacks ----------------------
    def _agree(self, ok):
        """AND of a rank-local verdict over every rank of the job, over the control plane (the job's default process group,
        whatever its backend): decisions about the data plane are taken by all ranks or by none"""
        if self.world == 1:
            return bool(ok)
        flags = [None] * self.world
        dist.all_gather_object(flags, bool(ok))
        return all(flags)

    def enable_native_comm(self, neighbours=None):
        """Every rank of the job calls this (collective): rank 0 draws the ncclUniqueId, it travels over the job's default
        process group (the control plane), every rank joins with ncclCommInitRank inside the library.  The halo refreshes and
        all-reduces of wcsph_slab_step then never leave the library; migration and ghost rebuild (redistribute) stay on
        torch.distributed.
        Collective in its FAILURES too: the ranks agree over the control plane (a) that librccl is loadable
        everywhere BEFORE any of them enters ncclCommInitRank -- a rank that cannot load it would otherwise leave the others
        blocked in there for good -- and (b) that every rank came out of it with a communicator; either every rank returns
        with the native plane on, or every rank raises with it off."""
        if not hasattr(self.B, "comm_init"):
            raise RuntimeError("the native data plane needs the HIP backend")
        lib = self.B.lib
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = lib.wcsph_comm_unique_id(buf)         # every rank: this is also the test that librccl loads here (only rank 0's id is used)
        why = (lib.wcsph_comm_last_error() or b"").decode(errors="replace") if rc else ""
        if not self._agree(rc == 0):
            raise RuntimeError("native data plane: RCCL is not usable on at least one rank%s" % (" (here: %s)" % why if why else ""))
        ident = [bytes(buf.raw) if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(ident, src=0)
        err = None
        try:
            with torch.cuda.device(self.device):
                self.B.comm_init(ident[0], self.rank, self.world, neighbours)
        except Exception as e:      # noqa: BLE001 -- the verdict must be collective
            err = e
        if not self._agree(err is None):
            if err is None:
                self.disable_native_comm()
            raise RuntimeError("native data plane: wcsph_comm_init failed on at least one rank%s" % (" (here: %s)" % err if err else ""))

    def set_overlap(self, on):
        """split the neighbour passes into interior / rest around the halo refreshes (kept across a rebalance)"""
        self._overlap_choice = bool(on)
        self.overlap = bool(on)

    HALO_SCHEDULES = [("exchange inline on the context's stream, passes not split", 0, False),
                      ("exchange on the halo stream beside the interior pass", 1, True),
                      ("exchange + unpack + boundary pass on the halo stream beside the interior pass", 2, True)]

    def _max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        vals = [None] * self.world
        dist.all_gather_object(vals, float(x))
        return max(vals)

    def tune_halo_schedule(self, steps=3, reps=2):
        """Native data plane only, collective: `steps` steps (after one to settle) of each way of scheduling a halo refresh
        (wcsph_comm_set_mode x the interior / boundary split), the maximum over the ranks decides -- the same numbers on every
        rank -- and the fastest is left switched on.  Advances the simulation by 2 + reps x (steps + 1) x 3 steps."""
        if not getattr(self.B, "native_comm", False):
            raise RuntimeError("tune_halo_schedule: the native data plane is not on")
        sync = (lambda: torch.cuda.synchronize(self.device)) if self.device.type == "cuda" else (lambda: None)
        for _ in range(2):                       # whoever is measured first must not pay for a cold start
            self.step()
        times = [float("inf")] * len(self.HALO_SCHEDULES)
        for _rep in range(reps):                 # round robin, the best of `reps` windows per candidate
            for k, (_label, m, ov) in enumerate(self.HALO_SCHEDULES):
                self.B.comm_set_mode(m); self.set_overlap(ov)
                self.step()
                sync(); self._max_over_ranks(0.0); t0 = time.perf_counter()      # (the exchange of the maximum is the barrier)
                for _ in range(steps):
                    self.step()
                sync()
                times[k] = min(times[k], self._max_over_ranks(time.perf_counter() - t0) / steps)
        best = int(np.argmin(times))
        self.B.comm_set_mode(self.HALO_SCHEDULES[best][1]); self.set_overlap(self.HALO_SCHEDULES[best][2])
        return {"candidates_ms_per_step": {c[0]: round(t * 1e3, 3) for c, t in zip(self.HALO_SCHEDULES, times)},
                "chosen": self.HALO_SCHEDULES[best][0], "steps_spent": 2 + reps * (steps + 1) * len(self.HALO_SCHEDULES)}

    def disable_native_comm(self):
        if self.B is not None and getattr(self.B, "native_comm", False):
            _lib.check(self.B.ctx, self.B.lib.wcsph_comm_destroy(self.B.ctx))     # (drains the context's stream first)
            self.B.native_comm = False

    def _set_cuts(self, cuts):
        """(re)build this rank's backend for the slab [cuts[rank], cuts[rank+1]): cell window, static boundary particles
        of the window, and in faithful mode the global static bucket histogram"""
        rank, world = self.rank, self.world
        keep = (getattr(self, "cuts", None), getattr(self, "lo", None), getattr(self, "hi", None), getattr(self, "window", None))
        try:
            self._set_cuts_(list(cuts))
        except Exception:
            if self.B is not None:                    # the old backend is whole and keeps serving the old decomposition
                self.cuts, self.lo, self.hi, self.window = keep
            raise

    def _set_cuts_(self, cuts):
        rank, world = self.rank, self.world
        self.cuts = list(cuts)
        self.lo, self.hi = self.cuts[rank], self.cuts[rank + 1]
        w0, w1 = max(self.lo - GHOST_CELLS, 0), min(self.hi + GHOST_CELLS, self.nx)
        self.window = (w0, w1)
        sol32 = np.asarray(self.solid_all, np.float64).astype(np.float32)
        scx = self._cellx_np(sol32[:, 0]) if len(sol32) else np.zeros(0, np.int64)
        solid_local = self.solid_all[(scx >= w0) & (scx < w1)]
        capacity = int(max(self.rows.shape[0], self.n_global // world) * self._cap_factor) + 4096
        # Everything that can fail comes BEFORE the old backend and its communicator are touched (a ValueError or a
        # failed allocation after the detach lost the ncclComm, its streams and buffers, and left the other ranks with a
        # communicator whose peer had gone): the width check, then the NEW backend, built beside the old one.
        if self.hi - self.lo < MIN_SLAB and world > 1:
            raise ValueError("rank %d: slab [%d, %d) is narrower than %d cell columns" % (rank, self.lo, self.hi, MIN_SLAB))
        kw = dict(capacity=capacity, solid_local=solid_local, bounds=(self.minb, self.maxb), window=self.window,
                  particleR=self.particleR, device=self.device)
        if self.stream is not None:
            kw["stream"] = self.stream
        if self.neighbor_mode == "faithful":
            big = 1 << 30
            kw.update(neighbor_mode="faithful", owned=(self.lo if rank > 0 else -big, self.hi if rank < world - 1 else big),
                      n_global=(self.n_global, len(self.solid_all)))
        old = self.B
        if old is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)       # nothing of the old backend may still be in flight when it is freed
        new = self._factory(**kw)
        if old is not None:
            if getattr(old, "native_comm", False):    # the communicator outlives the context (a rebalance rebuilds it)
                handle = old.comm_detach()
                try:
                    new.comm_attach(handle)
                except Exception:
                    old.comm_attach(handle)           # still whole: the old context keeps it
                    new.close()
                    raise
            old.close()
        self.B = new
        if self.neighbor_mode == "faithful" and world > 1:
            self._allreduce_tensor(self.B.bucket_hist(1))      # static boundary particles: once per backend
        # halo/compute overlap: a neighbour pass that needs a halo refresh runs on the slab interior (columns at least
        # GHOST_CELLS away from a cut with a neighbour: no ghost within reach) while the exchange is in flight, then on
        # the rest.  WCSPH_SLAB_OVERLAP=0 switches it off.
        self.overlap = (world > 1 and os.environ.get("WCSPH_SLAB_OVERLAP", "1") != "0") if getattr(self, "_overlap_choice", None) is None else self._overlap_choice
        big = 1 << 20
        self.interior = (self.lo + GHOST_CELLS if rank > 0 else -big, self.hi - GHOST_CELLS if rank < world - 1 else big)
        # preallocated halo buffers (sized for the widest exchange: two vec3 fields), grown on demand
        self._hb = {}
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)   # backend construction used the default stream (setup copies)

    # ---- geometry -------------------------------------------------------------------------------------
    def _cellx_np(self, x):
        return np.trunc((x.astype(np.float32) - self.minb[0]) * self.inv_grid).astype(np.int64)

    def _cellx(self, rows):
        # ref: HashGrid.py:68 -- f32 subtract, f32 multiply, truncation toward zero
        f = (rows[:, 0] - float(self.minb[0])) * float(self.inv_grid)
        return torch.trunc(f).to(torch.int64)

    # ---- communication ----------------------------------------------------------------------------------
    def _neighbours(self):
        return (self.rank - 1 if self.rank > 0 else None, self.rank + 1 if self.rank < self.world - 1 else None)

    def _host(self, t):
        """device values the host needs to decide something: one wait"""
        self.n_host_waits += 1
        return t.tolist()

    def _allreduce_tensor(self, t, op=None):
        op = op if op is not None else dist.ReduceOp.SUM
        if self.world == 1:                        # (the communicator path forced on one rank: wcsph_slab_step flags bit 3)
            return
        if self.comm_device != self.device:        # gloo dry run: stage through the host
            h = t.to(self.comm_device)
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=self.group)

    def _exchange_var(self, to_left, to_right, head_l=(), head_r=()):
        """Variable-size row exchange with the two slab neighbours ONLY (no world-wide collective): a fixed-size header
        (the row count and the caller's integers) travels first, point to point, then the rows.
        Returns (from_left, from_right, head_from_left, head_from_right)."""
        left, right = self._neighbours()
        w = to_left.shape[1]
        nh = 1 + max(len(head_l), len(head_r))
        hs_l = torch.tensor([to_left.shape[0]] + list(head_l) + [0] * (nh - 1 - len(head_l)), dtype=torch.int64, device=self.comm_device)
        hs_r = torch.tensor([to_right.shape[0]] + list(head_r) + [0] * (nh - 1 - len(head_r)), dtype=torch.int64, device=self.comm_device)
        hr_l, hr_r = torch.zeros_like(hs_l), torch.zeros_like(hs_r)
        ops = []
        if left is not None:
            ops += [dist.P2POp(dist.isend, hs_l, left, group=self.group), dist.P2POp(dist.irecv, hr_l, left, group=self.group)]
        if right is not None:
            ops += [dist.P2POp(dist.isend, hs_r, right, group=self.group), dist.P2POp(dist.irecv, hr_r, right, group=self.group)]
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        hl, hr = self._host(torch.stack([hr_l, hr_r]))
        from_left = torch.empty((int(hl[0]), w), dtype=torch.float32, device=self.device)
        from_right = torch.empty((int(hr[0]), w), dtype=torch.float32, device=self.device)
        self._sendrecv(to_left, to_right, from_left, from_right)
        return from_left, from_right, hl[1:], hr[1:]

    def _sendrecv(self, to_left, to_right, from_left, from_right):
        self._complete(self._post(to_left, to_right, from_left, from_right))

    def _post(self, to_left, to_right, from_left, from_right):
        """one tensor per direction; everything goes out as one batch.  Returns what _complete needs.  With nccl the
        transfers run on the process group's own stream: work queued on the current stream after this call overlaps with
        them until _complete makes the stream wait."""
        left, right = self._neighbours()
        if getattr(self, "loopback", False):
            # one rank whose left and right neighbour are itself (the one-GPU test of the data plane, tests/test_gpu_native_comm.py):
            # what goes out to the left comes in from the right
            from_right.copy_(to_left); from_left.copy_(to_right)
            self.n_exchanges += 1
            return [], None, (from_left, from_right), (to_left, to_right)
        staged = self.comm_device != self.device
        dst = None
        if staged:
            dst = (from_left, from_right)
            to_left, to_right = to_left.to(self.comm_device), to_right.to(self.comm_device)
            from_left, from_right = torch.empty_like(from_left, device=self.comm_device), torch.empty_like(from_right, device=self.comm_device)
        ops = []
        if left is not None:
            if to_left.numel():
                ops.append(dist.P2POp(dist.isend, to_left.contiguous(), left, group=self.group))
            if from_left.numel():
                ops.append(dist.P2POp(dist.irecv, from_left, left, group=self.group))
        if right is not None:
            if to_right.numel():
                ops.append(dist.P2POp(dist.isend, to_right.contiguous(), right, group=self.group))
            if from_right.numel():
                ops.append(dist.P2POp(dist.irecv, from_right, right, group=self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        self.n_exchanges += 1
        return reqs, dst, (from_left, from_right), (to_left, to_right)      # the send buffers stay referenced until done

    def _complete(self, pending):
        reqs, dst, src, _keep = pending
        for req in reqs:
            req.wait()
        if dst is not None:
            dst[0].copy_(src[0]); dst[1].copy_(src[1])

    def _halo_buffers(self, W):
        """send / receive buffers of a halo refresh of total row width W, allocated once per (W, ghost layout)"""
        n_send, n_ghost = self._halo_counts()[:2]
        key = (W, n_send, n_ghost)
        b = self._hb.get(W)
        if b is None or b[0] != key:
            b = (key, torch.empty((n_send, W), dtype=torch.float32, device=self.device),
                 torch.empty((n_ghost, W), dtype=torch.float32, device=self.device))
            self._hb[W] = b
        return b[1], b[2]

    def xchg(self, *fields):
        """Refresh the ghost rows of the given fields from their owners: ONE packed buffer per slab neighbour however many
        fields travel (the cost of an exchange is its latency, not its bytes)."""
        self.xchg_end(self.xchg_begin(*fields))

    def xchg_begin(self, *fields):
        """pack + post the sends/receives; returns a handle for xchg_end (None when there is nobody to talk to)"""
        if self.world == 1:
            return None
        B = self.B
        W = sum(FIELD_W.get(f, 1) for f in fields)
        out, inc = self._halo_buffers(W)
        nl, ngl = self._halo_counts()[2:]
        if hasattr(B, "pack_rows"):
            B.pack_rows(fields, self.send_all, out)                        # ONE pack kernel for all fields, both neighbours
        else:
            c = 0
            for f in fields:
                w = FIELD_W.get(f, 1)
                out[:, c:c + w] = B.gather_rows(f, self.send_all); c += w
        pending = self._post(out[:nl], out[nl:], inc[:ngl], inc[ngl:])
        return fields, inc, pending

    def xchg_end(self, h):
        if h is None:
            return
        fields, inc, pending = h
        self._complete(pending)
        B = self.B
        if hasattr(B, "unpack_rows"):
            B.unpack_rows(fields, self.ghost_all, inc)                     # ONE unpack kernel
        else:
            c = 0
            for f in fields:
                w = FIELD_W.get(f, 1)
                B.scatter_rows(f, self.ghost_all, inc[:, c:c + w].contiguous()); c += w

    def xp(self, fields, phase):
        """halo refresh of `fields`, then the neighbour phase that reads them from the ghosts -- overlapped: the phase runs
        on the slab interior while the exchange is in flight, on the rest once it has landed"""
        if isinstance(fields, str):
            fields = (fields,)
        if not self.overlap:
            self.xchg(*fields)
            self.B.phase(phase)
            return
        h = self.xchg_begin(*fields)
        self.B.set_part(1, *self.interior); self.B.phase(phase)
        self.xchg_end(h)
        self.B.set_part(2, *self.interior); self.B.phase(phase)
        self.B.set_part(0)

    def _allreduce(self, v, op):
        if self.world == 1:
            self.n_host_waits += 1
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.comm_device)
        dist.all_reduce(t, op=op, group=self.group)
        self.n_host_waits += 1
        return float(t.item())

    def reduce_acc(self, slot, fetch=True):
        """Global total of reduction slot `slot`.  With a device-resident slot buffer (HipBackend.acc) the stripes are
        all-reduced in place on the stream -- kernels launched next read the global total, no host round trip -- and the
        host value, when asked for, costs one wait.  Other backends: local value -> all-reduce -> written back."""
        B = self.B
        acc = getattr(B, "acc", None)
        if acc is None or self.world == 1:
            if fetch or self.world > 1:
                self.n_host_waits += 1
            v = self.allsum(B.get_acc(slot)) if self.world > 1 else B.get_acc(slot)
            if self.world > 1:
                B.set_acc(slot, v)
            return v
        sl = acc[slot * _lib.ACC_STRIPES:(slot + 1) * _lib.ACC_STRIPES]
        self._allreduce_tensor(sl)
        if fetch:
            self.n_host_waits += 1
        return B.get_acc(slot) if fetch else None

    def _make_callbacks(self):
        """ctypes callbacks behind wcsph_slab_step: this rank's side of the communicator (torch.distributed)"""
        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except BaseException as e:   # never let an exception unwind through the C frames
                    self._cb_exc = e
                    return -2
            return run

        def buffers(_u, W, p_send, p_recv):
            out, inc = self._halo_buffers(int(W))
            self._cb_buf = (out, inc)
            p_send[0], p_recv[0] = out.data_ptr(), inc.data_ptr()

        def post(_u, W):
            out, inc = self._cb_buf
            nl, ngl = self._halo_counts()[2:]
            self._cb_pending = self._post(out[:nl], out[nl:], inc[:ngl], inc[ngl:])

        def wait(_u):
            self._complete(self._cb_pending)
            self._cb_pending = None

        def acc(_u, slot):
            self._allreduce_tensor(self.B.acc[slot * _lib.ACC_STRIPES:(slot + 1) * _lib.ACC_STRIPES])

        def vmax(_u, p):
            p[0] = self._allreduce(p[0], dist.ReduceOp.MAX)
            self.n_host_waits -= 1          # counted by the library (the CFL wait)

        def words(_u, ptr, n):
            self._allreduce_tensor(torch.as_tensor(_DevArray(ptr, n), device=self.device))

        def maxbits(_u, ptr):     # the CFL maximum as an int32 bit pattern on the device: no host value, no extra wait
            self._allreduce_tensor(torch.as_tensor(_DevArray(ptr, 1), device=self.device), dist.ReduceOp.MAX)

        def words_post(_u, ptr, n):
            # the bucket histogram's all-reduce, started here and awaited in words_wait: with RCCL it runs on the process
            # group's own stream, ordered behind what the context's stream holds NOW (the packed histogram), while the
            # library queues the scan, the sort and the list build.  Host-staged (gloo) runs have nothing to overlap with.
            if self.world == 1:
                self._words_work = None
                return
            t = torch.as_tensor(_DevArray(ptr, n), device=self.device)
            if self.comm_device != self.device:
                self._allreduce_tensor(t)
                self._words_work = None
            else:
                self._words_work = (dist.all_reduce(t, group=self.group, async_op=True), t)

        def words_wait(_u):
            if self._words_work is not None:
                self._words_work[0].wait()          # the current (= the context's) stream waits; the host does not
                self._words_work = None

        cb = _lib.SlabCallbacks()
        cb.user, cb.world = None, int(self.world)
        self._cb_keep = [_lib.CB_BUFFERS(guard(buffers)), _lib.CB_POST(guard(post)), _lib.CB_WAIT(guard(wait)),
                         _lib.CB_ACC(guard(acc)), _lib.CB_MAX(guard(vmax)), _lib.CB_WORDS(guard(words)), _lib.CB_MAXBITS(guard(maxbits)),
                         _lib.CB_WORDS(guard(words_post)), _lib.CB_WAIT(guard(words_wait))]
        (cb.halo_buffers, cb.halo_post, cb.halo_wait, cb.allreduce_acc, cb.allreduce_max, cb.allreduce_words,
         cb.allreduce_max_bits, cb.allreduce_words_post, cb.allreduce_words_wait) = self._cb_keep
        return cb

    def allsum(self, v): return self._allreduce(v, dist.ReduceOp.SUM)   # noqa: E704
    def allmax(self, v): return self._allreduce(v, dist.ReduceOp.MAX)   # noqa: E704

    # ---- ownership ----------------------------------------------------------------------------------------
    def redistribute(self):
        """Migration and ghost rebuild in ONE neighbour-only exchange, then the backend is loaded.

        To the left neighbour go [my particles that crossed the left cut | my remaining particles within GHOST_CELLS of
        it]; the first block becomes the neighbour's property, the second its ghosts.  What I handed over I keep as a
        ghost myself (it sits just beyond the cut), so nothing travels twice.  The index lists of the per-phase halo
        refreshes follow from the same bookkeeping on both sides:
            my send list to L  = my near-cut rows (in sent order) ++ the rows L just handed to me (in received order)
            L's ghost rows of me = what it received as ghosts       ++ what it handed to me."""
        if self._rows is None:
            return self._redistribute_resident()
        rows = self.rows
        i32 = torch.int32
        if self.world == 1:
            empty = torch.zeros(0, dtype=i32, device=self.device)
            self.send_l = self.send_r = self.ghost_l = self.ghost_r = self.send_all = self.ghost_all = empty
            self.n_owned = rows.shape[0]
            self.B.load_state(self.n_owned, rows)
            return
        G = GHOST_CELLS
        n = rows.shape[0]
        has_l, has_r = self.rank > 0, self.rank < self.world - 1
        # classification of every owned row by its cell column: 1 / 2 = stays within G columns of the left / right cut (a
        # ghost over there), 3 / 4 = crossed the left / right cut (changes owner), 5 = more than a ghost layer beyond a cut
        # in one step (CFL broken).  ONE pass over the rows, ONE compaction of the ~6 % that are not interior, ONE host wait.
        native = self.B.classify_rows(rows, self.lo, self.hi, G, has_l, has_r) if (rows.is_cuda and hasattr(self.B, "classify_rows")) else None
        if native is not None:       # one launch of the library + one host wait (the lists come in arbitrary order)
            (n_nl, n_nr, n_ml, n_mr, n_lost), (idx_nl, idx_nr, idx_gl, idx_gr) = native
            self.n_host_waits += 1
        else:
            cx = self._cellx(rows)
            code = torch.zeros(n, dtype=torch.uint8, device=self.device)
            if has_l:
                code = torch.where(cx < self.lo + G, 1, code)
                code = torch.where(cx < self.lo, 3, code)
                code = torch.where(cx < self.lo - G, 5, code)
            if has_r:
                code = torch.where(cx >= self.hi - G, 2, code)
                code = torch.where(cx >= self.hi, 4, code)
                code = torch.where(cx >= self.hi + G, 5, code)
            idx = torch.nonzero(code).flatten()
            sub = code[idx].to(torch.int64)
            c = self._host(torch.bincount(sub, minlength=6))
            n_nl, n_nr, n_ml, n_mr, n_lost = int(c[1]), int(c[2]), int(c[3]), int(c[4]), int(c[5])
            idx = idx[torch.argsort(sub, stable=True)]           # by class, ascending row index inside a class
            idx_nl, idx_nr = idx[:n_nl], idx[n_nl:n_nl + n_nr]
            idx_gl, idx_gr = idx[n_nl + n_nr:n_nl + n_nr + n_ml], idx[n_nl + n_nr + n_ml:n_nl + n_nr + n_ml + n_mr]
        if n_lost:
            raise RuntimeError("rank %d: %d particles moved more than %d cell columns past a cut in one step (CFL violated); "
                               "they would land outside the neighbour's cell window" % (self.rank, n_lost, G))
        mig_l, mig_r = rows.index_select(0, idx_gl), rows.index_select(0, idx_gr)
        to_l = torch.cat([mig_l, rows.index_select(0, idx_nl)], 0)
        to_r = torch.cat([mig_r, rows.index_select(0, idx_nr)], 0)
        in_l, in_r, hl, hr = self._exchange_var(to_l, to_r, (n_ml,), (n_mr,))
        a_l, a_r = int(hl[0]), int(hr[0])                                        # rows that became mine
        # ownership changes IN PLACE: the rows that arrived fill the slots of the rows that left; what is left over is
        # appended, or the tail is folded into the remaining holes -- O(rows that moved), never a copy of the whole slab
        arrivals = torch.cat([in_l[:a_l], in_r[:a_r]], 0) if (a_l or a_r) else rows[:0]
        rows, pos_arr, idx_nl, idx_nr = self._swap_rows(rows, torch.cat([idx_gl, idx_gr]), arrivals, idx_nl, idx_nr)
        n = rows.shape[0]
        i32 = torch.int32
        self.send_l = torch.cat([idx_nl, pos_arr[:a_l]]).to(i32)
        self.send_r = torch.cat([idx_nr, pos_arr[a_l:]]).to(i32)
        gl = torch.cat([in_l[a_l:], mig_l], 0)
        gr = torch.cat([in_r[a_r:], mig_r], 0)
        self.ghost_l = torch.arange(n, n + gl.shape[0], dtype=i32, device=self.device)
        self.ghost_r = torch.arange(n + gl.shape[0], n + gl.shape[0] + gr.shape[0], dtype=i32, device=self.device)
        self.send_all = torch.cat([self.send_l, self.send_r])
        self.ghost_all = torch.cat([self.ghost_l, self.ghost_r])
        self.rows = rows
        self.n_owned = n
        if hasattr(self.B, "load_state_split"):
            self.B.load_state_split(rows, (gl, gr))
        else:
            self.B.load_state(n, torch.cat([rows, gl, gr], 0))

    def _redistribute_resident(self):
        """redistribute() on the state that stayed in the library (the previous step ran through wcsph_slab_step and nobody
        has asked for `rows` since): the same bookkeeping, but only the rows that travel -- ~6 % of a slab -- leave the
        library, and what arrives is written into the slots of what left (wcsph_state_apply); no row table in the driver."""
        B = self.B
        i32 = torch.int32
        self.n_resident = getattr(self, "n_resident", 0) + 1
        if self.world == 1 and not getattr(self, "loopback", False):
            return                                           # nothing moves, nothing to reload: the state is where the step left it
        G = GHOST_CELLS
        n = int(self.n_owned)
        has_l, has_r = self.rank > 0, self.rank < self.world - 1
        in_library = hasattr(B, "migrate_begin") and self.library_migrate
        native = B.classify_state(self.lo, self.hi, G, has_l, has_r, grow=True) if in_library else B.classify_state(self.lo, self.hi, G, has_l, has_r)
        self.n_host_waits += 1 if getattr(B, "classify_waited", True) else 0
        if native is None:                                   # a list overflowed its buffer: the classic path on a materialised table
            _ = self.rows
            return self.redistribute()
        (n_nl, n_nr, n_ml, n_mr, n_lost), (idx_nl, idx_nr, idx_gl, idx_gr) = native
        if n_lost:
            raise RuntimeError("rank %d: %d particles moved more than %d cell columns past a cut in one step (CFL violated); "
                               "they would land outside the neighbour's cell window" % (self.rank, n_lost, G))
        if in_library:
            # Round 6: gather, swap plan, row writes and the halo lists of the coming step inside the library (csrc/api_migrate.h: five
            # launches, no host wait) instead of ~20 small torch launches with the GPU idle behind them.  On the library's own RCCL
            # plane the neighbours' counts came with the step's last mail (every rank has them or none: the tail of
            # wcsph_slab_step is the same launch sequence on every rank), so the rows travel in ONE grouped ncclSend / ncclRecv of
            # exact sizes; otherwise header + rows over this rank's communicator as before.
            nbr = B.tail_neighbour_counts() if (getattr(B, "native_comm", False) and self.native_migrate) else None
            if nbr is not None:
                out = B.migrate_native(native[0], nbr[0], nbr[1])
                self.n_exchanges += 1
                self.n_native_migrations = getattr(self, "n_native_migrations", 0) + 1
                self.n_rows_arrived += (nbr[0][3] if has_l else 0) + (nbr[1][2] if has_r else 0)
            else:
                to_l, to_r = B.migrate_begin(native[0])
                in_l, in_r, hl, hr = self._exchange_var(to_l, to_r, (n_ml,), (n_mr,))
                out = B.migrate_end(in_l, int(hl[0]), in_r, int(hr[0]))
                self.n_rows_arrived += int(hl[0]) + int(hr[0])
            self._adopt_library_halo(out)
            return
        out = B.gather_state_rows(torch.cat([idx_gl, idx_nl, idx_gr, idx_nr]))
        to_l, to_r = out[:n_ml + n_nl], out[n_ml + n_nl:]
        mig_l, mig_r = to_l[:n_ml], to_r[:n_mr]
        in_l, in_r, hl, hr = self._exchange_var(to_l, to_r, (n_ml,), (n_mr,))
        a_l, a_r = int(hl[0]), int(hr[0])
        self.n_rows_arrived += a_l + a_r
        arrivals = torch.cat([in_l[:a_l], in_r[:a_r]], 0) if (a_l or a_r) else out[:0]
        n_new, pos_arr, idx_nl, idx_nr, fold_dst, fold_src = self._swap_plan(n, torch.cat([idx_gl, idx_gr]), int(arrivals.shape[0]), idx_nl, idx_nr)
        n_gl, n_gr = int(in_l.shape[0]) - a_l + int(mig_l.shape[0]), int(in_r.shape[0]) - a_r + int(mig_r.shape[0])
        B.apply_state(n_new, fold_dst, fold_src, pos_arr, arrivals, torch.cat([in_l[a_l:], mig_l, in_r[a_r:], mig_r], 0))
        n = n_new
        n_sl = int(idx_nl.shape[0]) + a_l
        self.send_all = torch.cat([idx_nl, pos_arr[:a_l], idx_nr, pos_arr[a_l:]]).to(i32)
        self.send_l, self.send_r = self.send_all[:n_sl], self.send_all[n_sl:]
        self.ghost_all = torch.arange(n, n + n_gl + n_gr, dtype=i32, device=self.device)
        self.ghost_l, self.ghost_r = self.ghost_all[:n_gl], self.ghost_all[n_gl:]
        self.n_owned = n

    def _swap_plan(self, n, holes, a, *index_lists):
        """_swap_rows as a PLAN on indices (the rows themselves live in the library): a table of n rows loses the rows `holes`
        (k indices) and gains `a` arrivals.  Returns (n', positions of the arrivals, the index lists translated, fold_dst,
        fold_src): a >= k: the first k arrivals take the holes, the rest is appended at n ..; a < k: the arrivals take the first
        a holes, the table shrinks by L = k - a, and the rows of its last L slots that stay are relabelled (fold_src -> fold_dst)
        into the remaining holes below the new end -- an arrival that took a hole in the tail is folded like any other row."""
        k = int(holes.shape[0])
        dev = holes.device
        none = torch.zeros(0, dtype=torch.int64, device=dev)
        if a >= k:
            pos = torch.cat([holes, torch.arange(n, n + a - k, dtype=torch.int64, device=dev)])
            return (n + a - k, pos) + index_lists + (none, none)
        rest, L = holes[a:], k - a
        n_new = n - L
        in_tail = rest >= n_new
        hole_tail = torch.zeros(L + 1, dtype=torch.bool, device=dev)
        hole_tail[torch.where(in_tail, rest - n_new, L)] = True          # slot L: dummy for the holes below the new end
        keep = ~hole_tail[:L]
        below = ~in_tail
        target = torch.full((L + 1,), n, dtype=torch.int64, device=dev)
        target[torch.where(below, torch.cumsum(below, 0) - 1, L)] = rest
        tail = n_new + torch.arange(L, device=dev)
        dest = torch.where(keep, target[:L][(torch.cumsum(keep, 0) - 1).clamp_(min=0)], tail)
        out = []
        for lst in (holes[:a],) + index_lists:
            moved = lst >= n_new
            out.append(torch.where(moved, dest[(lst - n_new).clamp_(min=0)], lst))
        return (n_new,) + tuple(out) + (dest[keep], tail[keep])

    def _swap_rows(self, rows, holes, arrivals, *index_lists):
        """rows (n, W) loses the rows `holes` (k indices) and gains `arrivals` (a rows), in place where it can.
        Returns (rows', positions of the arrivals in rows', the index lists translated to rows').  a >= k: the first k
        arrivals take the holes, the rest is appended (into the slack behind a store_state() view, else one concatenation).
        a < k: the arrivals take the first a holes; the table shrinks by L = k - a, and the rows of its last L slots that
        stay are folded into the remaining holes below the new end.  No host wait: every size is known on the host."""
        n, k, a = rows.shape[0], int(holes.shape[0]), int(arrivals.shape[0])
        dev = rows.device
        if k == 0 and a == 0:
            return (rows, holes[:0]) + index_lists
        if a >= k:
            if k:
                rows.index_copy_(0, holes, arrivals[:k])
            extra = a - k
            if extra:
                base = rows._base
                if (base is not None and base.dim() == 2 and base.shape[1] == rows.shape[1] and base.is_contiguous() and rows.is_contiguous()
                        and base.data_ptr() == rows.data_ptr() and base.shape[0] >= n + extra):
                    rows = base[:n + extra]
                    rows[n:] = arrivals[k:]
                else:
                    rows = torch.cat([rows, arrivals[k:]], 0)
            pos = torch.cat([holes, torch.arange(n, n + extra, dtype=torch.int64, device=dev)])
            return (rows, pos) + index_lists
        if a:
            rows.index_copy_(0, holes[:a], arrivals)
        rest, L = holes[a:], k - a                  # L holes stay open: the table shrinks to n - L rows
        n_new = n - L
        in_tail = rest >= n_new
        # tail slot j (row n_new + j) stays unless it is a hole itself
        hole_tail = torch.zeros(L + 1, dtype=torch.bool, device=dev)
        hole_tail[torch.where(in_tail, rest - n_new, L)] = True          # slot L: dummy for the holes below the new end
        keep = ~hole_tail[:L]
        # the r-th staying tail row goes to the r-th hole below the new end (there are equally many of both)
        below = ~in_tail
        target = torch.full((L + 1,), n, dtype=torch.int64, device=dev)   # n: a row index nobody reads afterwards
        target[torch.where(below, torch.cumsum(below, 0) - 1, L)] = rest
        dest = torch.where(keep, target[:L][(torch.cumsum(keep, 0) - 1).clamp_(min=0)], n_new + torch.arange(L, device=dev))
        # (a tail row that is a hole "moves" onto itself: beyond the new end, dropped with it)
        rows.index_copy_(0, dest, rows[n_new:].clone())
        out = []
        for lst in (holes[:a],) + index_lists:       # an arrival that took a hole in the tail was folded like any other row
            moved = lst >= n_new
            out.append(torch.where(moved, dest[(lst - n_new).clamp_(min=0)], lst))
        return (rows[:n_new],) + tuple(out)

    def rebalance(self, force=False):
        """between steps only (never from inside _step: the backend, and everything queued on it, is replaced)"""
        if self.stream is None:
            return self._rebalance(force)
        with torch.cuda.stream(self.stream):      # the rows were written on this stream
            return self._rebalance(force)

    def _rebalance(self, force=False):
        """New cuts from the all-reduced cell-column histogram ("rebalance cuts every K steps"): as the dam
        breaks along x the liquid leaves the initial columns, and static cuts would pile it onto the edge ranks.  Rows
        move to their new owners point to point, and every rank rebuilds its backend for the new cell window."""
        if self.world == 1:
            return False
        cx = self._cellx(self.rows).clamp_(0, self.nx - 1)
        hist = torch.bincount(cx, minlength=self.nx).to(torch.int64)
        self._allreduce_tensor(hist)
        h = np.asarray(self._host(hist), np.int64)
        cuts = balanced_cuts(h, self.world, MIN_SLAB)
        cum = np.concatenate([[0], np.cumsum(h)])
        load = lambda c: max(cum[c[r + 1]] - cum[c[r]] for r in range(self.world))   # noqa: E731
        if not force and load(self.cuts) <= 1.1 * load(cuts):
            return False                                    # the present cuts are within 10 % of the best ones
        inner = torch.tensor(cuts[1:-1], dtype=torch.int64, device=self.device)
        owner = torch.bucketize(cx, inner, right=True)       # cut[r] <= cx < cut[r+1]  ->  r
        parts = [self.rows.index_select(0, torch.nonzero(owner == r).flatten()) for r in range(self.world)]
        counts = torch.tensor([p.shape[0] for p in parts], dtype=torch.int64, device=self.comm_device)
        allc = [torch.zeros_like(counts) for _ in range(self.world)]
        dist.all_gather(allc, counts, group=self.group)
        incoming = [int(allc[r][self.rank]) for r in range(self.world)]
        self.n_host_waits += 1
        staged = self.comm_device != self.device
        recv = [torch.empty((incoming[r], STATE_W), dtype=torch.float32, device=self.comm_device) for r in range(self.world)]
        send = [p.to(self.comm_device).contiguous() if staged else p.contiguous() for p in parts]
        ops = []
        for r in range(self.world):
            if r == self.rank:
                continue
            if send[r].numel():
                ops.append(dist.P2POp(dist.isend, send[r], r, group=self.group))
            if recv[r].numel():
                ops.append(dist.P2POp(dist.irecv, recv[r], r, group=self.group))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        recv[self.rank] = send[self.rank]
        self.rows = torch.cat([t.to(self.device) for t in recv], 0)
        self._set_cuts(cuts)
        self.n_rebalanced += 1
        return True

    # ---- the step -------------------------------------------------------------------------------------------
    def step(self):
        # the periodic rebalance runs BEFORE the step and outside its stream context: it closes the backend and builds a
        # new one, and nothing of the old one may still be in flight or referenced by the step's locals
        if self.rebalance_every and self.steps_done and self.steps_done % self.rebalance_every == 0:
            self.rebalance()
        self.steps_done += 1
        if self.stream is None:
            return self._step()
        with torch.cuda.stream(self.stream):
            return self._step()

    def _step(self):
        B, ph, x = self.B, self.B.phase, self.xchg
        ACC_ERR, ACC_DAD, ACC_DNEW, ACC_DZERO = _lib.ACC_ERR, _lib.ACC_DAD, _lib.ACC_DNEW, _lib.ACC_DZERO
        N = float(self.n_global)
        self.redistribute()
        if self.native and hasattr(B, "native_step"):
            # the rest of the step -- every phase, halo refresh, solver loop and the dt rule below -- runs inside the
            # library (wcsph_slab_step); this rank's communicator is reached through callbacks
            self.dt, self.vs_iter, self.dv_iter, self.pr_iter, self.avg_density_err = B.native_step(self)
            # the state stays in the library (`rows` materialises on demand); otherwise one copy out per step
            self.rows = None if (self.resident and hasattr(B, "apply_state")) else B.store_state()
            return float(self.dt)
        B.set_scalar("deltaT", self.dt)
        if self.neighbor_mode == "faithful" and self.world > 1:
            # particle_data.hash_grid.update_grid() with the reference's GLOBAL bucket table (HashGrid.py:108-114): every
            # rank counts its own liquid particles per bucket, the histogram is summed over the ranks on the device
            B.grid_update_begin()
            self._allreduce_tensor(B.bucket_hist_packed())     # two 16-bit counts per int32 word: count/2 words on the wire
            B.bucket_hist_unpack()
            B.grid_update_end()                                 # (runs the sort and the list build too: no explicit _mid here)
        else:
            B.grid_update()                               # particle_data.hash_grid.update_grid()
        # compute_density(); compute_dfsph_coff(); and the first pass of solve_vel_divergence() (dfsph.py:131-146), which
        # reads pos_j / vel_j only, in one walk; rho_j then travels with kappa_v
        self.dv_iter = 0
        xp = self.xp
        ph("DENSITY_COFF_WARM"); xp(("rho", "kappa_v"), "VEL_WARM")
        err = -0.1
        xp("vel", "DRHO_DIV_BEGIN")
        dt = float(self.dt)
        while float(self.avg_density_err) > err and self.dv_iter < 10:
            ph("MAKE_K_DIV"); xp("kbuf", "VEL_ITER")
            xp("vel", "DRHO_DIV_ITER")
            self.avg_density_err = np.float32(self.reduce_acc(ACC_ERR))
            err = 0.001 * N / dt
            self.dv_iter += 1
        # end of solve_vel_divergence + compute_nonpressure_force()'s clear_nonpressure and v_guess update (dfsph.py:84-103)
        # in one elementwise launch: each touches particle i alone, tension only adds to d_vel afterwards
        ph("END_DIV_BEGIN_NONPRESSURE")
        if self.tension:
            ph("TENSION_NORMAL"); x("normal"); ph("TENSION_ACCEL")
        # the sums of compute_vorticity() (dfsph.py:307-328) ride on the residual walk: vel_j, omega_j, rho_j do not change
        # during the viscosity solve
        xp("vel_guess", "VISC_MINV_RESIDUAL_VORT")
        self.cg_delta_zero = np.float32(self.reduce_acc(ACC_DZERO))
        self.cg_delta = self.cg_delta_zero
        self.vs_iter = 0
        while self.vs_iter < 100:
            B.set_scalar("cg_delta", self.cg_delta)
            xp("cg_dir", "VISC_AD")
            self.reduce_acc(ACC_DAD, fetch=False)          # k_visc_update reads the global dAd on the device
            ph("VISC_UPDATE")
            dnew = self.reduce_acc(ACC_DNEW)               # k_visc_dir reads the global delta' on the device
            self.cg_delta = np.float32(dnew)
            self.vs_iter += 1
            if float(self.cg_delta) <= self.viscosity_err * float(self.cg_delta_zero) or float(self.cg_delta_zero) < self.eps:
                break
            ph("VISC_DIR")                                # the next search direction, only if there is a next iteration
        ph("VISC_END")
        ph("VORTICITY_INT")

        # optimize_time_step()  (dfsph.py:107-129)
        vmax = np.float32(self.allmax(B.cfl_local_max()))
        if vmax > self.eps:
            time_step = 0.5 * 0.4 * self.particleR * 2.0 / math.sqrt(vmax)
            time_step = max(min(time_step, self.user_max_t), self.user_min_t)
            it = max(self.vs_iter, max(self.pr_iter, self.vs_iter))
            d = self.dt
            # Python double on the float32 value, rounded once (dfsph.py:124-126 under the NumPy 1.x of the reference's time;
            # NumPy >= 2 would form the product in float32: one ulp off)
            if it > 10:
                d = np.float32(float(d) * 0.9)
            elif it < 5:
                d = np.float32(float(d) * 1.1)
            self.dt = np.float32(min(float(d), time_step))
            B.set_scalar("deltaT", self.dt)
        # update_vel + solve_pressure()  (dfsph.py:150-164); the warm start is a no-op in the reference (Q7): its gate
        # adv_rho > rho_L0 never opens, so kappa_j of ghosts is not refreshed for it and it changes no velocity
        ph("UPDATE_VEL_WARM_SCALE"); ph("WARM_PRESS_VEL")
        self.pr_iter = 0
        err = 0.0
        xp("vel", "DRHO_PRESS_BEGIN")
        while (err > 0.001 or self.pr_iter < 2) and self.pr_iter < 100:
            ph("MAKE_K_PRESS"); xp("kbuf", "VEL_ITER")
            xp("vel", "DRHO_PRESS_ITER")
            self.pr_iter += 1
            if self.pr_iter < 2:      # `pr_iter < 2` forces the second iteration whatever the first reports:
                err = 1.0             # no device wait, no all-reduce for a value that cannot change the control flow
                continue
            self.avg_density_err = np.float32(self.reduce_acc(ACC_ERR))
            err = float(self.avg_density_err) / N
        ph("END_PRESS_UPDATE_POS")
        self.rows = B.store_state()
        return float(self.dt)


# ---------------------------------------------------------------------------------------------------------
def dam_break_slab(n, rank, world, particleRadius=scene.PARTICLE_RADIUS, cuts=None):
    """This rank's share of the n^3 dam break (dfsph.py:59-82 recipe) without building the whole scene:
    returns (liquid_local, solid_all, bounds, cuts, n_global, gid_local); gid = the particle's index in the reference's
    own creation order (dfsph.py:69-73: i = (x * ny + y) * nz + z), which the faithful CFL prefix (Q5) is defined on.
    cuts: explicit cell-column cuts instead of the balanced ones (tests start from skewed cuts and rebalance)."""
    dis = particleRadius * 2.0
    L = n * dis
    solid = scene.box_shell(L)
    minb = np.array([-L, 0.0, -L], np.float32)
    maxb = np.array([L, 2 * L, L], np.float32)
    inv = np.float32(1.0 / dis)
    xs = (np.arange(n) - n / 2).astype(np.float64) * dis + dis * 0.5            # lattice x planes
    cellx = np.trunc((xs.astype(np.float32) - minb[0]) * inv).astype(np.int64)
    nx = int(float(maxb[0] - minb[0]) / dis + 1.0)
    hist = np.bincount(cellx, minlength=nx) * (n * n)
    cuts = balanced_cuts(hist, world, MIN_SLAB) if cuts is None else list(cuts)
    mine = np.nonzero((cellx >= cuts[rank]) & (cellx < cuts[rank + 1]))[0]
    j = np.arange(n * n)
    y = (j // n).astype(np.float64) * dis + 0.2
    z = (j % n - n / 2).astype(np.float64) * dis + dis * 0.5
    liquid = np.empty((len(mine) * n * n, 3))
    gid = np.empty(len(mine) * n * n, np.int64)
    for k, ix in enumerate(mine):
        blk = liquid[k * n * n:(k + 1) * n * n]
        blk[:, 0] = xs[ix]; blk[:, 1] = y; blk[:, 2] = z
        gid[k * n * n:(k + 1) * n * n] = ix * n * n + j
    return liquid, solid, (minb, maxb), cuts, n ** 3, gid


def _rccl_probe(rank, world, dev, group):
    """the kinds of calls a slab step makes over RCCL, once, small: raises what RCCL / torch raise"""
    t = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
    dist.all_reduce(t, group=group)
    w = torch.tensor([rank], dtype=torch.int32, device=dev)
    dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
    ops, keep = [], []
    for peer in (rank - 1, rank + 1):
        if 0 <= peer < world:
            a = torch.full((128, STATE_W), float(rank), dtype=torch.float32, device=dev)
            b = torch.empty((128, STATE_W), dtype=torch.float32, device=dev)
            keep.append((peer, a, b))
            ops += [dist.P2POp(dist.isend, a, peer, group=group), dist.P2POp(dist.irecv, b, peer, group=group)]
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
    torch.cuda.synchronize(dev)
    if float(t[0].item()) != world * (world + 1) / 2.0 or int(w.item()) != world - 1:
        raise RuntimeError("all_reduce returned %r / %r" % (float(t[0].item()), int(w.item())))
    for peer, _a, b in keep:
        if float(b[0, 0].item()) != float(peer) or float(b[-1, -1].item()) != float(peer):
            raise RuntimeError("send/recv with rank %d returned wrong data" % peer)


class _fd_stdout_to_stderr:
    """C++ layers under torch.distributed (gloo's "[Gloo] Rank r is connected to ..." lines) write to file descriptor 1 directly:
    while a process group is being set up, fd 1 points at stderr, so that bench.py's stdout carries the ONE JSON line only"""

    def __enter__(self):
        sys.stdout.flush()
        self._keep = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._keep, 1)
        os.close(self._keep)
        return False


def bench_main(args):
    """bench.py --gpus N (N > 1): weak scaling, ~2 M liquid particles per GPU (config C4 at N = 8).  Rank 0 returns the
    JSON object (bench.py adds the cpu_baseline leg, measured once on rank 0's host after the ranks have left)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    # WCSPH_BENCH_COMM=gloo: dry run of this path with every rank on ONE GPU (halo buffers staged through the
    # host); the real thing is RCCL (backend "nccl") with one GPU per rank
    # a wedged collective must not hold the node: dump the stacks and exit after 20 minutes
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("WCSPH_BENCH_WATCHDOG_S", "1200")), exit=True)
    comm = os.environ.get("WCSPH_BENCH_COMM", "nccl")
    if comm == "gloo":
        local = 0
    local = int(os.environ.get("WCSPH_BENCH_LOCAL_GPU", local))     # tests: every rank on the one GPU of the box
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # Control plane: a gloo group over the same ranks, always there -- barriers, the timing maximum and the agreement below
    # travel over it as host tensors; it is never on the data path of an RCCL run.
    with _fd_stdout_to_stderr():
        dist.init_process_group("gloo")
        dist.barrier()                          # (gloo connects its full mesh lazily: here, not in the middle of the run)
    grp = None
    one_gpu = comm == "gloo"
    if not one_gpu:
        # Data plane: RCCL (backend "nccl"), one GPU per rank.  RCCL has never run with more than one rank on the development
        # pool (single-GPU boxes; two ranks on one GPU are refused: tools/rccl_probe.py): before the run every rank creates
        # the group and tries the kinds of calls a step makes -- all_reduce (f64 SUM, i32 MAX), a grouped isend / irecv with
        # its slab neighbours -- and the ranks agree over gloo whether all of them came through.  If not, the SAME HIP
        # computation runs with the halo buffers staged through the host over gloo (slower, still one GPU per rank, and
        # the JSON line says so) instead of the job dying without a number.  A rank that HANGS inside RCCL is not
        # recoverable here: the watchdog above ends the job.
        ok, why, ng = 1, "", None
        try:
            if os.environ.get("WCSPH_BENCH_FAIL_RCCL_PROBE"):          # test hook: the fall-back path on a healthy box
                raise RuntimeError("WCSPH_BENCH_FAIL_RCCL_PROBE is set")
            with _fd_stdout_to_stderr():
                ng = dist.new_group(backend="nccl", device_id=dev)
                _rccl_probe(rank, world, dev, ng)
        except Exception as e:      # noqa: BLE001 -- whatever RCCL / torch raise here, the decision must be collective
            ok, why = 0, "%s: %s" % (type(e).__name__, str(e).strip().splitlines()[0][:200] if str(e).strip() else "")
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            grp = ng
        else:
            if why:
                print("[bench rank %d] RCCL probe failed (%s): halos staged through the host over gloo" % (rank, why), file=sys.stderr, flush=True)
            comm = "gloo fallback (the RCCL probe failed on at least one rank): one GPU per rank, halo buffers staged through the host"
    staged = comm != "nccl"
    import __graft_entry__ as g
    if rank == 0:
        g.build()
    dist.barrier()
    # weak scaling: n^3 ~ world * 126^3  (C3 per GPU; world = 8 gives the 252^3 of config C4)
    n_base = scene.CONFIGS.get(args.config) or int(args.config)
    n = int(round(n_base * world ** (1.0 / 3.0)))
    liquid, solid, bounds, cuts, n_global, gid = dam_break_slab(n, rank, world)
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, cuts, n_global, rank, world, device=dev,
                  comm_device=torch.device("cpu") if staged else dev, neighbor_mode=args.mode, gid_local=gid, group=grp,
                  rebalance_every=int(os.environ.get("WCSPH_REBALANCE_EVERY", "200")))
    # Data plane of the step's halo refreshes and all-reduces: the library's own RCCL calls (csrc/api_comm.h: no Python between
    # wcsph_slab_step and the wire) when every rank can join and a probe of the real neighbour pattern comes back right;
    # otherwise this rank's torch.distributed group behind wcsph_slab_callbacks, as before.  WCSPH_BENCH_NATIVE=0 keeps the callbacks.
    plane = "torch.distributed (%s) behind wcsph_slab_callbacks" % ("RCCL" if not staged else "gloo, host-staged")
    if not staged and sim.native and os.environ.get("WCSPH_BENCH_NATIVE", "1") != "0":
        # Every stage is agreed over the control plane before the next RCCL collective is entered: a rank that drops out
        # early must not leave the others blocked inside ncclCommInitRank or the probe's all-reduce.  enable_native_comm() is
        # collective in its failures (every rank raises, or none); the probe's own verdict is agreed below.  A rank that HANGS
        # inside RCCL is not recoverable: a shorter watchdog covers these first calls.
        ok, why = 1, ""
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(int(os.environ.get("WCSPH_BENCH_PROBE_WATCHDOG_S", "300")), exit=True)
        try:
            sim.enable_native_comm()
        except Exception as e:      # noqa: BLE001 -- raised on every rank alike
            ok, why = 0, "%s: %s" % (type(e).__name__, (str(e).strip().splitlines() or [""])[0][:200])
        if ok:
            try:
                if os.environ.get("WCSPH_BENCH_FAIL_NATIVE_PROBE"):    # test hook: the fall-back to the callbacks on a healthy box
                    raise RuntimeError("WCSPH_BENCH_FAIL_NATIVE_PROBE is set")
                sim.B.comm_probe(rank, world)
            except Exception as e:  # noqa: BLE001 -- the decision must be collective
                ok, why = 0, "%s: %s" % (type(e).__name__, (str(e).strip().splitlines() or [""])[0][:200])
            ok = 1 if sim._agree(ok) else 0
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(int(os.environ.get("WCSPH_BENCH_WATCHDOG_S", "1200")), exit=True)
        if ok:
            plane = ("native RCCL inside the library (wcsph_comm_init / wcsph_halo_exchange: grouped ncclSend/ncclRecv on the communicator's halo stream "
                     "beside the interior pass, ncclAllReduce on the context's stream; migration + ghost rebuild by wcsph_slab_migrate: message sizes from "
                     "the step's last mail, one grouped ncclSend/ncclRecv, plan and row writes on the device)")
        else:
            if why:
                print("[bench rank %d] native data plane not used (%s): torch.distributed callbacks" % (rank, why), file=sys.stderr, flush=True)
            try:
                sim.disable_native_comm()
            except Exception:       # noqa: BLE001
                sim.B.native_comm = False
    # How a halo refresh is scheduled is MEASURED, not assumed (round 6, profiles/r06_overlap.md): on one GPU, where an exchange is a
    # 16 us copy, the inline form without the interior / boundary split is fastest; with real neighbours it depends on what an
    # exchange costs over xGMI.  Three steps of each candidate before the warm-up, the maximum over the ranks decides (the same
    # numbers on every rank: all-reduced over the control plane).
    tuning = None
    if getattr(sim.B, "native_comm", False) and os.environ.get("WCSPH_BENCH_TUNE", "1") != "0":
        tuning = sim.tune_halo_schedule()
    iters = []
    for _ in range(args.warmup):
        sim.step()
    dist.barrier(); torch.cuda.synchronize()
    # in-situ timing of the north-star kernel pair on rank 0: HIP events on the context's stream around every
    # pressure_iter of the timed steps (wcsph_slab_step brackets them; the halo refreshes in between are inside)
    prof = rank == 0 and sim.native
    if prof:
        _lib.check(sim.B.ctx, sim.B.lib.wcsph_profile_begin(sim.B.ctx, 16 * args.steps + 16))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.step()
        iters.append((sim.vs_iter, sim.dv_iter, sim.pr_iter))
    dist.barrier(); torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)       # control plane: host tensors over gloo
    insitu_ms, insitu_n = None, 0
    if prof:
        ms_, n_ = C.c_float(), C.c_int32()
        with torch.cuda.stream(sim.stream):
            _lib.check(sim.B.ctx, sim.B.lib.wcsph_profile_end(sim.B.ctx, C.byref(ms_), C.byref(n_)))
        insitu_ms, insitu_n = ms_.value, n_.value
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([sim.n_owned], dtype=torch.int64)
    allc = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(allc, cnt)
    result = None
    if rank == 0:
        elapsed = float(el.item())
        # dominant kernel pair on this rank's slab, timed live on the context's stream (the unfused launches of the
        # phase path: k_make_k + k_vel_from_k + k_drho_press); algorithmic bytes as in bench.py
        ms = C.c_float()
        with torch.cuda.stream(sim.B.stream):
            _lib.check(sim.B.ctx, sim.B.lib.wcsph_time_kernel(sim.B.ctx, b"pressure_iter", 10, C.byref(ms)))
        ns_local = sim.B.pd.solid_count
        alg_bytes = 84.0 * sim.n_owned + 24.0 * ns_local
        launch_ms = insitu_ms if insitu_ms else ms.value       # live measurement when the library drove the step
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        it = np.array(iters, dtype=np.float64)
        out = {
            "metric": "M particle-steps/sec, DFSPH dam-break", "value": round(n_global * args.steps / elapsed / 1e6, 3),
            "unit": "M particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DFSPH 3D dam-break %d^3 = %d liquid particles over %d x-slabs (C3 load per GPU; C4 at 8 "
                                   "GPUs), ghost halos over %s, viscosity PCG + vorticity on" % (n, n_global, world, "RCCL send/recv" if not staged else "gloo (host-staged)"),
                       "neighbor_mode": args.mode, "n_liquid": n_global, "cuts": sim.cuts, "comm": comm, "data_plane": plane,
                       "step_driver": ("wcsph_slab_step (library, its own RCCL communicator)" if getattr(sim.B, "native_comm", False) else
                                       "wcsph_slab_step (library, communicator behind callbacks)") if sim.native else "python phase driver",
                       "halo_schedule_tuning": tuning,
                       "host_waits_per_step": round(sim.n_host_waits / float(max(sim.steps_done, 1)), 1),
                       "migrations_inside_the_library_over_rccl": getattr(sim, "n_native_migrations", 0),
                       "rebalance_every": sim.rebalance_every, "rebalanced": sim.n_rebalanced,
                       "owned_per_rank": [int(c.item()) for c in allc],
                       "halo_exchanges_per_step": round(sim.n_exchanges / float(max(sim.steps_done, 1)), 1),
                       "mean_iters": {"vs": round(it[:, 0].mean(), 2), "dv": round(it[:, 1].mean(), 2), "pr": round(it[:, 2].mean(), 2)}},
            "roofline": {"bound": "hbm", "kernel": "pressure_iter on rank 0's slab (k_make_k + k_vel_from_k + k_drho_press<ITER>), ref dfsph.py:518-547",
                         "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": None,
                         "avg_launch_us": round(launch_ms * 1e3, 1), "launches_timed": insitu_n,
                         "timing": ("HIP events on rank 0's stream around every pressure_iter of the timed steps, the two halo refreshes of an iteration included"
                                    if insitu_ms else "wcsph_time_kernel replay after the run"),
                         "replay_us": round(ms.value * 1e3, 1), "algorithmic_bytes_per_launch": int(alg_bytes)},
        }
        result = out
    dist.barrier()
    sim.B.close()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    return result
