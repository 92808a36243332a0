"""The planner of the mid-size factorisation schedule (madqp_jl_amd/csrc/mid_plan.inc, compiled for the CPU by
tests/csrc): every plan it hands to chol_mid_step_kernel must apply each panel to each trailing tile exactly once, in
order, in time, and never touch a tile twice in one step."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QMAX = 2


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "csrc")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmadqp_mid_plan.so"))
    lib.mid_plan_rows.restype = C.c_int
    lib.mid_plan_rows.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    return lib


def plan(lib, nblk, cap):
    rows = np.zeros((nblk * nblk * nblk // 4 + 64, 5), dtype=np.int32)
    units = np.zeros(max(nblk, 1), dtype=np.int32)
    n = C.c_int64(0)
    rounds = lib.mid_plan_rows(nblk, cap, rows.ctypes.data, rows.shape[0], C.byref(n), units.ctypes.data)
    assert rounds >= 0 and n.value <= rows.shape[0]
    return rounds, rows[: n.value], units


def check(nblk, cap, rounds, rows, units):
    assert rounds >= 1
    upto = np.zeros((nblk, nblk), dtype=np.int64)  # panels applied to tile (i, j), i >= j
    for k in range(nblk):
        mine = rows[rows[:, 0] == k]
        assert len(mine) == units[k] <= rounds * cap
        touched = set()
        for _, i, j, p0, q in mine:
            assert k <= j <= i < nblk and (i, j) != (k, k), "a trailing tile; the diagonal tile of the step is not a unit"
            assert (i, j) not in touched, "a tile is updated by one workgroup per step"
            touched.add((i, j))
            assert 1 <= q <= QMAX and p0 + q <= k, "only solved panels, at most QMAX of them in one product"
            assert p0 == upto[i, j], "panels are applied in order, none skipped, none twice"
            upto[i, j] += q
        if k > 0:
            # the panel of this step is solved next: its column is complete; the diagonal tile lacks panel k-1 alone
            # (the diagonal workgroup's own SYRK), and so will the next one
            assert (upto[k + 1:, k] == k).all()
            assert upto[k, k] == k - 1
            upto[k, k] = k
            if k + 1 < nblk:
                assert (upto[k + 1:, k + 1] == k).all()


@pytest.mark.parametrize("cap", [254, 255, 127, 63, 7])
def test_plans_are_complete_and_in_order(lib, cap):
    for nblk in list(range(1, 34)) + [40, 47, 56, 63, 64, 79, 80]:
        rounds, rows, units = plan(lib, nblk, cap)
        if rounds == 0:  # no plan within 64 rounds per step (a sliver of a GPU): the library keeps its older schedule
            assert cap == 7 and nblk >= 40 and len(rows) == 0
            continue
        check(nblk, cap, rounds, rows, units)


def test_one_round_per_step_up_to_forty_blocks_on_a_whole_gpu(lib):
    # n = 5 000 (configs[1]): every step fits one round of tiles, i.e. the shadow of the diagonal chain
    for nblk in (16, 24, 32, 40):
        assert plan(lib, nblk, 254)[0] == 1
    assert plan(lib, 80, 254)[0] <= 4


def test_neighbouring_units_share_operands(lib):
    # most runs of 32 units of a full step at n = 5 000 touch far fewer operand blocks than 32 separate tiles would (64);
    # a column visited alone -- column k, one panel -- has nothing to share but its own rows
    _, rows, units = plan(lib, 40, 254)
    mine = rows[rows[:, 0] == 10]
    assert len(mine) > 200
    counts = []
    for a in range(0, len(mine) - 32, 32):
        run = mine[a: a + 32]
        counts.append(len({(i, p0, q) for _, i, j, p0, q in run} | {(j, p0, q) for _, i, j, p0, q in run}))
    assert sorted(counts)[len(counts) // 2] <= 16 and max(counts) <= 34
