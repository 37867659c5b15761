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
