"""SlabSim._swap_rows: ownership changes of a slab in place (rows that leave, rows that arrive) -- host logic, CPU."""
import numpy as np
import pytest
import torch

from wcsph_amd.slab import SlabSim


@pytest.mark.parametrize("n,k,a,slack", [(50, 0, 0, 0), (50, 5, 5, 0), (50, 3, 9, 0), (50, 3, 9, 16), (50, 9, 3, 0), (50, 12, 0, 0),
                                         (50, 50, 4, 0), (40, 17, 2, 8), (1000, 130, 7, 0), (1000, 7, 130, 256)])
def test_swap_rows_keeps_the_right_rows_and_translates_indices(n, k, a, slack):
    rng = np.random.default_rng(n * 1000 + k * 10 + a)
    W = 15
    base = torch.zeros((n + slack, W))
    base[:n, 0] = torch.arange(n, dtype=torch.float32)               # row id in column 0
    rows = base[:n] if slack else base[:n].clone()
    holes = torch.from_numpy(rng.permutation(n)[:k].astype(np.int64))
    arrivals = torch.zeros((a, W)); arrivals[:, 0] = 1000 + torch.arange(a, dtype=torch.float32)
    stay = np.setdiff1d(np.arange(n), holes.numpy())
    lists = [torch.from_numpy(rng.choice(stay, size=min(len(stay), m), replace=False).astype(np.int64)) if len(stay) else torch.zeros(0, dtype=torch.int64)
             for m in (7, 0, 11)]
    out, pos, l0, l1, l2 = SlabSim._swap_rows(None, rows, holes, arrivals, *lists)
    assert out.shape[0] == n - k + a
    ids = out[:, 0].numpy()
    assert sorted(ids.tolist()) == sorted(stay.tolist() + [1000 + j for j in range(a)])     # exactly the staying rows + the arrivals
    np.testing.assert_array_equal(ids[pos.numpy()], 1000 + np.arange(a))                     # arrival j sits at pos[j]
    for old, new in zip(lists, (l0, l1, l2)):
        np.testing.assert_array_equal(ids[new.numpy()], old.numpy().astype(np.float32))     # a listed row is found at its new index
    if slack and a - k > 0 and a - k <= slack:
        assert out.data_ptr() == base.data_ptr()                                            # appended in place


@pytest.mark.parametrize("n,k,a", [(50, 0, 0), (50, 5, 5), (50, 3, 9), (50, 9, 3), (50, 12, 0), (50, 50, 4), (40, 17, 2), (1000, 130, 7), (1000, 7, 130)])
def test_swap_plan_is_swap_rows_on_indices(n, k, a):
    """SlabSim._swap_plan (round 5: the rows stay in the library, wcsph_state_apply) = _swap_rows as a plan: relabel the folded tail
    rows, then write the arrivals at their final positions -- emulated here on a row table with the library's slot semantics
    (original index -> slot through `where`; a relabel swaps two labels, a write goes to the slot the index has)."""
    rng = np.random.default_rng(n * 1000 + k * 10 + a + 1)
    W = 15
    rows = torch.zeros((n, W)); rows[:, 0] = torch.arange(n, dtype=torch.float32)
    holes = torch.from_numpy(rng.permutation(n)[:k].astype(np.int64))
    arrivals = torch.zeros((a, W)); arrivals[:, 0] = 1000 + torch.arange(a, dtype=torch.float32)
    stay = np.setdiff1d(np.arange(n), holes.numpy())
    lists = [torch.from_numpy(rng.choice(stay, size=min(len(stay), m), replace=False).astype(np.int64)) if len(stay) else torch.zeros(0, dtype=torch.int64)
             for m in (7, 11)]
    ref_rows, ref_pos, r0, r1 = SlabSim._swap_rows(None, rows.clone(), holes, arrivals, *lists)
    n_new, pos, l0, l1, fold_dst, fold_src = SlabSim._swap_plan(None, n, holes, a, *lists)
    assert n_new == ref_rows.shape[0]
    for x, y in ((pos, ref_pos), (l0, r0), (l1, r1)):
        np.testing.assert_array_equal(x.numpy(), y.numpy())
    # the library's semantics: slots in a shuffled order, labels swapped by the fold, arrivals written through `where`
    perm = rng.permutation(n)                                    # where[o] = slot
    slots = torch.zeros((max(n, n_new), W)); slots[torch.from_numpy(perm[:n].astype(np.int64))] = rows
    where = np.concatenate([perm, np.arange(n, max(n, n_new))]).astype(np.int64)
    assert len(np.intersect1d(fold_dst.numpy(), fold_src.numpy())) == 0
    for h, t in zip(fold_dst.tolist(), fold_src.tolist()):
        where[h], where[t] = where[t], where[h]
    for j, o in enumerate(pos.tolist()):
        slots[where[o]] = arrivals[j]
    got = slots[torch.from_numpy(where[:n_new])]
    np.testing.assert_array_equal(got[:, 0].numpy(), ref_rows[:, 0].numpy())
