"""csrc/api_migrate.h against the torch form it replaces (SlabSim._swap_plan + wcsph_state_apply, rounds 5-6): random classification
lists and random incoming messages on two identical resident states -- every branch of the plan (more arrivals than holes: appended;
fewer: the tail folded by a swap of labels, holes in the tail, arrivals that took a hole up there; none of either), the gathered rows,
the halo lists and, after the sort that drops the dead slots, every owned row and every ghost row bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(n_rows, seed, bounds, x_cols):
    from wcsph_amd.slab import STATE_W
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(bounds[0], np.float64), np.asarray(bounds[1], np.float64)
    rows = np.zeros((n_rows, STATE_W), np.float32)
    rows[:, 0] = lo[0] + rng.uniform(x_cols[0] + 0.01, x_cols[1] - 0.01, n_rows) * 0.05
    rows[:, 1] = rng.uniform(lo[1] + 0.06, hi[1] - 0.06, n_rows)
    rows[:, 2] = rng.uniform(lo[2] + 0.06, hi[2] - 0.06, n_rows)
    rows[:, 3:14] = rng.normal(size=(n_rows, 11)).astype(np.float32)
    return rows


@pytest.mark.parametrize("n,g_old,n_ml,n_mr,a_l,a_r,n_nl,n_nr,g_l,g_r", [
    (3000, 400, 0, 0, 0, 0, 0, 0, 0, 0),            # nothing moves, no ghosts
    (3000, 400, 40, 30, 40, 30, 200, 180, 150, 170),    # as many arrive as leave
    (3000, 400, 10, 5, 80, 60, 200, 180, 150, 170),     # more arrive: appended behind the table, into old ghost slots and beyond them
    (3000, 50, 10, 5, 300, 200, 200, 180, 400, 300),    # ... well beyond the rows the table had
    (3000, 400, 90, 70, 20, 10, 200, 180, 150, 170),    # fewer arrive: the tail is folded
    (3000, 400, 120, 100, 0, 0, 0, 0, 0, 0),            # nobody arrives, no halo at all
    (300, 40, 150, 140, 3, 2, 5, 5, 7, 9),              # almost everybody leaves: most of the tail are holes themselves
    (4000, 0, 0, 700, 600, 0, 0, 300, 250, 0),          # one-sided traffic
])
def test_the_librarys_migration_equals_the_torch_plan(n, g_old, n_ml, n_mr, a_l, a_r, n_nl, n_nr, g_l, g_r):
    import slab_util
    from wcsph_amd.slab import HipBackend, SlabSim, STATE_W
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _liquid, solid, bounds = slab_util.test_scene(8)
    seed = n + 7 * n_ml + 13 * a_l + g_old
    rng = np.random.default_rng(seed)
    base = _state(n + g_old, seed, bounds, (2, 14))
    base[:, 14] = np.arange(n + g_old, dtype=np.int32).view(np.float32)
    # the classification: disjoint random subsets of the owned rows, tail rows over-represented so that holes fall beyond the new end
    k = n_ml + n_mr
    pool = rng.permutation(n)
    tail_first = np.concatenate([pool[pool >= n - 2 * max(k, 1)], pool[pool < n - 2 * max(k, 1)]])
    pick = np.concatenate([tail_first[:k // 2], tail_first[::-1][:k - k // 2]])
    rest = np.setdiff1d(pool, pick)
    rng.shuffle(pick); rng.shuffle(rest)
    idx = [rest[:n_nl], rest[n_nl:n_nl + n_nr], pick[:n_ml], pick[n_ml:k]]           # near_l, near_r, go_l, go_r
    counts = [n_nl, n_nr, n_ml, n_mr, 0]
    n_fl, n_fr = a_l + g_l, a_r + g_r
    inc = _state(n_fl + n_fr, seed + 1, bounds, (2, 14))
    inc[:, 14] = (100000 + np.arange(n_fl + n_fr, dtype=np.int32)).view(np.float32)
    got = {}
    for form in ("library", "torch"):
        B = HipBackend(capacity=12000, solid_local=solid, bounds=bounds, window=(0, 17), particleR=0.025, device=dev)
        with torch.cuda.stream(B.stream):
            B.load_state(n, torch.from_numpy(base).to(dev))
            B.grid_update()                                   # a sorted state: where / oid are a real permutation
            cap = 4096
            B._cls_idx = torch.zeros((4, cap), dtype=torch.int64, device=dev)
            for q in range(4):
                B._cls_idx[q, :len(idx[q])] = torch.from_numpy(idx[q].astype(np.int64)).to(dev)
            lists = [B._cls_idx[q, :len(idx[q])] for q in range(4)]
            in_l, in_r = torch.from_numpy(inc[:n_fl]).to(dev), torch.from_numpy(inc[n_fl:]).to(dev)
            if form == "library":
                to_l, to_r = B.migrate_begin(counts)
                sent = torch.cat([to_l, to_r]).clone()
                out = B.migrate_end(in_l, a_l, in_r, a_r)
                send_all, ghost_all = B.halo_list_views()
                n_new, n_sl, n_gl = out[0], out[1], out[3]
                assert out[2] == n_nr + a_r and out[4] == g_r + n_mr
            else:
                idx_nl, idx_nr, idx_gl, idx_gr = lists
                sent = B.gather_state_rows(torch.cat([idx_gl, idx_nl, idx_gr, idx_nr]))
                mig_l, mig_r = sent[:n_ml], sent[n_ml + n_nl:n_ml + n_nl + n_mr]
                arrivals = torch.cat([in_l[:a_l], in_r[:a_r]], 0)
                n_new, pos_arr, idx_nl, idx_nr, fold_dst, fold_src = SlabSim._swap_plan(None, n, torch.cat([idx_gl, idx_gr]), a_l + a_r, idx_nl, idx_nr)
                B.apply_state(n_new, fold_dst, fold_src, pos_arr, arrivals, torch.cat([in_l[a_l:], mig_l, in_r[a_r:], mig_r], 0))
                n_sl, n_gl = n_nl + a_l, g_l + n_ml
                send_all = torch.cat([idx_nl, pos_arr[:a_l], idx_nr, pos_arr[a_l:]]).to(torch.int32)
                ghost_all = torch.arange(n_new, n_new + g_l + n_ml + g_r + n_mr, dtype=torch.int32, device=dev)
            send_all, ghost_all = send_all.clone(), ghost_all.clone()
            B.grid_update()                                   # the sort drops the dead slots; the accessors work again
            owned = B.store_state().clone()
            gh = {f: B.gather_rows(f, ghost_all).clone() for f in ("vel", "omega", "vel_guess")} if ghost_all.numel() else {}
            snd = B.gather_rows("vel", send_all).clone() if send_all.numel() else None
            torch.cuda.synchronize()
        got[form] = (n_new, n_sl, n_gl, sent.cpu().numpy(), send_all.cpu().numpy(), ghost_all.cpu().numpy(), owned.cpu().numpy(),
                     {f: v.cpu().numpy() for f, v in gh.items()}, None if snd is None else snd.cpu().numpy())
        B.close()
    a, b = got["library"], got["torch"]
    assert a[:3] == b[:3] and a[0] == n - k + a_l + a_r
    for q in (3, 4, 5, 6):
        np.testing.assert_array_equal(a[q].view(np.int32), b[q].view(np.int32))       # gathered rows, halo lists, every owned row: bit for bit
    for f in a[7]:
        np.testing.assert_array_equal(a[7][f], b[7][f])
    # and against what was asked for: the owned rows are exactly the stayers + the arrivals; the ghosts the rest of the messages + what left
    gid = a[6][:, 14].copy().view(np.int32)
    want = np.concatenate([np.setdiff1d(np.arange(n), pick), 100000 + np.arange(a_l), 100000 + n_fl + np.arange(a_r)])
    np.testing.assert_array_equal(np.sort(gid), np.sort(want))
    if a[7]:
        exp = np.concatenate([inc[a_l:n_fl, 3:6], base[idx[2], 3:6], inc[n_fl + a_r:, 3:6], base[idx[3], 3:6]])
        np.testing.assert_array_equal(a[7]["vel"], exp)
    if a[8] is not None:
        exp = np.concatenate([base[idx[0], 3:6], inc[:a_l, 3:6], base[idx[1], 3:6], inc[n_fl:n_fl + a_r, 3:6]])
        np.testing.assert_array_equal(a[8], exp)               # the send list names the rows the neighbours hold ghosts of, in their order


def test_a_classification_list_longer_than_its_buffer_gets_a_larger_buffer():
    """the in-library migration never falls back to the row-table path (its messages would not match the neighbours'): a list that
    did not fit is taken again into a larger buffer (HipBackend.classify_state(grow=True)); without `grow` the caller is told (None)"""
    import slab_util
    from wcsph_amd.slab import HipBackend, GHOST_CELLS as G, STATE_W
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _liquid, solid, bounds = slab_util.test_scene(8)
    n = 600000
    rng = np.random.default_rng(11)
    rows = np.zeros((n, STATE_W), np.float32)
    rows[:, 0] = bounds[0][0] + rng.uniform(2.01, 13.99, n).astype(np.float32) * np.float32(0.05)     # cell columns 2 .. 13
    rows[:, 1] = 0.3; rows[:, 2] = 0.0
    lo, hi = 4, 12                                            # near_l: columns 4-5, near_r: 10-11 (100 k each of 600 k: more than the n / 8 = 75 000 the default buffer holds); go_l: 2-3, go_r: 12-13
    cx = np.trunc((rows[:, 0] - np.float32(bounds[0][0])) * np.float32(1.0 / 0.05)).astype(np.int64)
    want = [np.nonzero((cx >= lo) & (cx < lo + G))[0], np.nonzero((cx >= hi - G) & (cx < hi))[0], np.nonzero(cx < lo)[0], np.nonzero(cx >= hi)[0]]
    B = HipBackend(capacity=n + 4096, solid_local=solid, bounds=bounds, window=(0, 17), particleR=0.025, device=dev)
    with torch.cuda.stream(B.stream):
        B.load_state(n, torch.from_numpy(rows).to(dev))
        assert B.classify_state(lo, hi, G, 1, 1) is None      # the default buffer holds max(65536, n / 8) entries per list
        got = B.classify_state(lo, hi, G, 1, 1, grow=True)
        torch.cuda.synchronize()
    assert got is not None and B.classify_waited
    counts, lists = got
    assert counts == [len(w) for w in want] + [0] and max(counts) > 75000 and B._cls_idx.shape[1] >= max(counts)
    for w, have in zip(want, lists):
        np.testing.assert_array_equal(np.sort(have.cpu().numpy()), w)
    B.close()


def test_migration_entry_points_refuse_what_they_cannot_do():
    """wrong call order, lists longer than their buffer, a CFL violation, no communicator: error codes with a message, no launch"""
    import slab_util
    from wcsph_amd import _lib
    from wcsph_amd.slab import HipBackend, STATE_W
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _liquid, solid, bounds = slab_util.test_scene(8)
    B = HipBackend(capacity=4000, solid_local=solid, bounds=bounds, window=(0, 17), particleR=0.025, device=dev)
    lib, ctx = B.lib, B.ctx
    with torch.cuda.stream(B.stream):
        B.load_state(1000, torch.from_numpy(_state(1200, 5, bounds, (2, 14))).to(dev))
        B._cls_idx = torch.zeros((4, 64), dtype=torch.int64, device=dev)
        out = (C.c_int32 * 5)()
        z = torch.zeros((4, STATE_W), dtype=torch.float32, device=dev)
        rc = lib.wcsph_slab_migrate_end(ctx, C.c_void_p(z.data_ptr()), 0, 0, C.c_void_p(z.data_ptr()), 0, 0, C.byref(out))
        assert rc == -1 and b"without wcsph_slab_migrate_begin" in lib.wcsph_last_error(ctx)
        with pytest.raises(_lib.WcsphError, match="longer than its buffer"):
            B.migrate_begin([65, 0, 0, 0, 0])
        with pytest.raises(_lib.WcsphError, match="CFL"):
            B.migrate_begin([1, 0, 0, 0, 3])
        five = (C.c_int32 * 5)(0, 0, 0, 0, 0)
        rc = lib.wcsph_slab_migrate(ctx, C.c_void_p(B._cls_idx.data_ptr()), 64, C.byref(five), C.byref(five), C.byref(five), C.byref(out))
        assert rc == -1 and b"no native communicator" in lib.wcsph_last_error(ctx)
        # a valid begin / end with nothing travelling leaves the state as it was, and only a grid update may follow
        to_l, to_r = B.migrate_begin([0, 0, 0, 0, 0])
        assert to_l.shape == (0, STATE_W) and to_r.shape == (0, STATE_W)
        rc = lib.wcsph_slab_migrate_end(ctx, None, 0, 1, None, 0, 0, C.byref(out))          # more arrivals than rows in the message
        assert rc == -1
        res = B.migrate_end(z[:0], 0, z[:0], 0)
        assert res == [1000, 0, 0, 0, 0]
        with pytest.raises(_lib.WcsphError, match="pending changes"):
            B.store_state()
        B.grid_update()
        assert B.store_state().shape[0] == 1000
        torch.cuda.synchronize()
    B.close()
