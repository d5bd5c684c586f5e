"""The units kernel's deal across the eight XCDs (csrc/mbk_units.h: units_plan, units_lookup; MBK_OPT_XCD_BALANCE): the
share arithmetic runs on the host through two diagnostic entry points of the C ABI, so its one invariant can be checked
without a GPU -- whatever the fractions and the list sizes, every entry of the heavy list, the middle list and the row-unit
list is computed by exactly one workgroup id, ids beyond an XCD's shares compute nothing, and the even deal is the
identity (id u takes heavy entry u)."""
import ctypes as C

import numpy as np
import pytest

from distributedmandelbrot_amd import _lib


def plan_of(lib, n_h, n_v, n_m, fractions):
    f = (C.c_double * 8)(*fractions)
    plan = (C.c_uint32 * 40)()
    assert lib.mbk_units_plan(n_h, n_v, n_m, f, plan) == 0
    return plan


def walk(lib, plan):
    """(list, index) of every id, per XCD in the order the hardware deals them."""
    total = plan[2]
    out = np.zeros((total, 2), dtype=np.int64)
    li, ix = C.c_uint32(), C.c_uint32()
    for u in range(total):
        assert lib.mbk_units_lookup(plan, u, C.byref(li), C.byref(ix)) == 0
        out[u] = (li.value, ix.value)
    return out


EVEN = [0.125] * 8
UNEVEN = [0.110, 0.140, 0.125, 0.120, 0.130, 0.125, 0.115, 0.135]


@pytest.mark.parametrize("n_h,n_v,n_m", [(47683, 21072, 48828), (0, 2048, 0), (16384, 0, 0), (3, 1, 2), (0, 0, 0), (1, 0, 0),
                                         (9000, 10, 3), (5, 4000, 4000), (8, 8, 8), (4097, 513, 7)])
@pytest.mark.parametrize("fractions", [EVEN, UNEVEN, [0.14, 0.11, 0.14, 0.11, 0.14, 0.11, 0.14, 0.11], [1, 0, 0, 0, 0, 0, 0, 0],
                                       [0, 0, 0, 0, 0, 0, 0, 1]])
def test_every_entry_exactly_once(n_h, n_v, n_m, fractions):
    lib = _lib.load()
    plan = plan_of(lib, n_h, n_v, n_m, fractions)
    assert plan[2] % 8 == 0 and plan[2] >= n_h + n_v + n_m
    assert sum(plan[8:16]) == n_h and sum(plan[16:24]) == n_v + n_m
    got = walk(lib, plan)
    for code, n in ((1, n_h), (2, n_m), (3, n_v)):
        idx = np.sort(got[got[:, 0] == code, 1])
        assert np.array_equal(idx, np.arange(n)), (code, n, idx[:10])
    # per XCD: heavy entries first, then light ones, then nothing -- and once nothing, nothing any more (the kernel leaves)
    for x in range(8):
        lists = got[x::8, 0]
        h_x, l_x = plan[8 + x], plan[16 + x]
        assert (lists[:h_x] == 1).all() and (lists[h_x:h_x + l_x] >= 2).all() and (lists[h_x + l_x:] == 0).all()
        light = got[x::8][h_x:h_x + l_x]
        order = light[:, 0] * (1 << 32) + light[:, 1]
        assert (np.diff(order) > 0).all()           # middle entries before row units, each in list order


def test_even_deal_is_the_identity():
    lib = _lib.load()
    n_h, n_v, n_m = 47683, 21072, 48828
    got = walk(lib, plan_of(lib, n_h, n_v, n_m, EVEN))
    assert (got[:n_h - 8, 0] == 1).all() and np.array_equal(got[:n_h - 8, 1], np.arange(n_h - 8))


def test_shares_follow_the_fractions():
    lib = _lib.load()
    plan = plan_of(lib, 47683, 21072, 48828, UNEVEN)
    h = np.array(plan[8:16], dtype=np.float64)
    assert np.abs(h / h.sum() - np.array(UNEVEN)).max() < 1e-4
    # an XCD with fewer heavy entries takes more light ones: the same number of ids for all (up to the rounding at the end)
    ids = h + np.array(plan[16:24])
    assert ids.max() - ids.min() <= 8 and ids.max() == plan[2] // 8


def test_random_fractions_and_sizes():
    lib = _lib.load()
    rs = np.random.RandomState(5)
    for _ in range(60):
        f = rs.uniform(0.05, 0.2, 8)
        f /= f.sum()
        n_h, n_v, n_m = (int(rs.randint(0, 3000)) for _ in range(3))
        plan = plan_of(lib, n_h, n_v, n_m, list(f))
        got = walk(lib, plan)
        for code, n in ((1, n_h), (2, n_m), (3, n_v)):
            assert np.array_equal(np.sort(got[got[:, 0] == code, 1]), np.arange(n))
