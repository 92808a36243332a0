"""GPU: the random soak of round 1 (tests/soak_random.py, ~1 100 problems outside pytest) as collected tests.

`seed0 = 9000` replays the stretch that contains seed 9195 (n_x = 186, m = 78, LP through the condensed form), the
example DESIGN.md gave for "one iteration more than the oracle"; `seed0 = 1000` is the soak's default start.  Every
problem runs through the Python driver, the native driver and the batched engine against the oracle with the
acceptance rule of tests/parity.py: identical iteration counts, except threshold ties -- LPs only, counted and bounded --
objective to 1e-9 (relative) and solution to 1e-7 (the stated bar), or SENS_FACTOR = 4 x the noise floor
measured over an ensemble of valid CPU executions of the oracle (tests/parity.py: ensemble_floor) where the problem's
conditioning does not support the bar (round 2 had 1e-7 / 1e-5 for every problem here, round 3 16 x a two-run distance).
Round 5: the PER-ITERATION traces (alpha_p, alpha_d, inf_pr, inf_du, inf_compl, mu) are held to the same rule for the
native and python drivers -- with the default library: sweeps that substitute and the AUTO refinement rule of
madqp_jl_amd/options.py (every problem here has order <= 260).
"""
import numpy as np
import pytest

import madqp_jl_amd as M
from oracle import mpc
from oracle import qp as Q
from parity import assert_parity, iteration_parity

pytestmark = pytest.mark.gpu
REG, OREG = M.FixedRegularization(1e-8, -1e-8), mpc.FixedRegularization(1e-8, -1e-8)


def soak_cases(seed0, count, only_lp=False):
    """The problem stream of tests/soak_random.py --mode drivers --seed0 seed0."""
    rng = np.random.default_rng(seed0)
    for t in range(count):
        n = int(rng.integers(1, 260))
        m = int(rng.integers(0, max(1, n)))
        lp = bool(rng.integers(0, 4) == 0)
        if lp or not only_lp:
            yield seed0 + t, n, m, lp


def run_case(hip, seed, n, m, lp, drivers):
    qp = Q.random_qp(seed, n, m, lp)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG)
    floor = {}  # the ensemble floor of this problem, computed at most once and only if a driver misses the stated bar
    dq = M.DeviceQP.from_numpy(hip.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
    ties = []
    for name in drivers:
        if name == "batched":  # no per-iteration trace: residuals at its last iteration, and at the oracle's if needed
            s = M.BatchedMPCSolver([dq], hip, regularization=REG)
            r = s.solve()[0]
            r["trace"] = {r["iter"]: r}
            if r["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"] + 1:
                s.close()
                s = M.BatchedMPCSolver([dq], hip, regularization=REG, max_iter=ref["iter"])
                r["trace"][ref["iter"]] = s.solve()[0]
        else:
            s = M.MPCSolver(dq, hip, regularization=REG, driver=name)
            r = s.solve()
        s.close()
        what = (seed, n, m, lp, name)
        assert r["status"] == ref["status"], (what, r["status"], ref["status"])
        if ref["status"] != M.SOLVE_SUCCEEDED:
            continue
        tie = iteration_parity(r, ref, 1e-8, what, lp=lp) == "tie"
        if tie:
            ties.append((what, r["iter"], ref["iter"]))
        if tie:  # the two points are different iterates, both of which satisfy the termination test to 1e-8
            dobj = abs(r["objective"] - ref["objective"]) / max(1.0, abs(ref["objective"]))
            assert dobj <= 1e-7, (what, "objective after a tie", dobj)
            continue
        from parity import ensemble_floor, exceeds_stated_bar

        # round 5: the per-iteration traces are compared too (native and python drivers; the batched engine keeps no trace):
        # SURVEY.md 8d states the tolerance per iteration, and round 4's soak passed only because it looked at x and the
        # objective
        trace = name != "batched"
        if exceeds_stated_bar(r, ref, trace=trace) and not floor:
            floor.update(ensemble_floor(qp, ref, regularization=OREG))
        assert_parity(r, ref, qp, what, trace=trace, floor=floor or None, regularization=OREG)
    return ties


def test_seed_9195_condensed_lp(hip):
    """The example of DESIGN.md (round 1): every device driver took 13 iterations, the oracle 12 on the GPU box's host
    (13 in the build container).  All drivers must agree with each other bit for bit in the count, and with the oracle
    up to a threshold tie."""
    qp = Q.random_qp(9195, 186, 78, True)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG)
    dq = M.DeviceQP.from_numpy(hip.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
    iters = []
    for driver in ("python", "native"):
        s = M.MPCSolver(dq, hip, regularization=REG, driver=driver)
        r = s.solve()
        s.close()
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED
        tie = iteration_parity(r, ref, 1e-8, driver, lp=True) == "tie"
        assert abs(r["objective"] - ref["objective"]) <= (1e-7 if tie else 1e-9) * max(1.0, abs(ref["objective"]))
        if not tie:
            assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
        iters.append(r["iter"])
    assert iters[0] == iters[1]
    # the same LP through the formulations that are not at the edge of fp64: identical counts, no tie allowed
    for ksys, oksys, reg, oreg in (("augmented", "K2", REG, OREG),
                                   ("normal", "normal", M.FixedRegularization(1e-8, 0.0), mpc.FixedRegularization(1e-8, 0.0))):
        kref = mpc.solve(qp, kkt_system=oksys, regularization=oreg)
        s = M.MPCSolver(dq, hip, regularization=reg, kkt_system=ksys, driver="native")
        r = s.solve()
        s.close()
        assert r["status"] == kref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == kref["iter"], (ksys, r["iter"], kref["iter"])


@pytest.mark.parametrize("seed0,count,only_lp,drivers", [
    (9000, 200, False, ("native",)),            # the stretch around seed 9195, all problem kinds
    (9000, 200, True, ("python", "batched")),   # its LPs through the other two drivers
    (1000, 150, False, ("native", "batched")),  # the soak's default start
    (31000, 400, True, ("native",)),            # LPs only: where the condensed form sits at the edge of fp64
])
def test_soak(hip, seed0, count, only_lp, drivers):
    ties, cases = [], 0
    for seed, n, m, lp in soak_cases(seed0, count, only_lp):
        ties += run_case(hip, seed, n, m, lp, drivers)
        cases += len(drivers)
    print(f"soak seed0={seed0}: {cases} solves, {len(ties)} threshold ties: {ties}")
    assert all(lp for (_, _, _, lp, _), _, _ in ties), ("a threshold tie on a QP", ties)
    assert len(ties) <= max(2, cases // 20), ties  # a few per cent of the LPs at most


@pytest.mark.parametrize("driver", ["python", "native"])
def test_refinement_step_on_the_edge_of_fp64(hip, driver):
    """`refine_steps=1` (extension, off by default): one step of iterative refinement with the residual that
    solve_system! forms anyway.  On seed 9195 -- the LP where the explicit block inverses cost the device path an
    iteration against the oracle on the GPU box's host -- the refined solves leave a smaller residual and the solver
    needs no more iterations than the oracle on any host (12 or 13)."""
    qp = Q.random_qp(9195, 186, 78, True)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG)
    dq = M.DeviceQP.from_numpy(hip.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
    out = {}
    for steps in (0, 1):
        s = M.MPCSolver(dq, hip, regularization=REG, driver=driver, refine_steps=steps)
        out[steps] = s.solve()
        out[steps]["resid"] = s.last_residual_ratio
        s.close()
    assert out[0]["status"] == out[1]["status"] == M.SOLVE_SUCCEEDED
    assert out[1]["iter"] <= min(out[0]["iter"], ref["iter"])
    assert abs(out[1]["objective"] - ref["objective"]) <= 1e-7 * max(1.0, abs(ref["objective"]))


def test_refinement_inside_solve_is_the_drivers_refinement(hip):
    """`madqp_kkt_set_refine` (round 5): the refinement step run INSIDE `MadNLP.solve!(kkt, w)` -- what the Julia glue asks for,
    because MadIPM's own solve_system! (src/linear_solver.jl:19-45) calls solve! once -- is operation for operation the step
    the drivers of this package run in THEIR solve_system with the residual they form anyway: bitwise the same iterates.
    Also: with the AUTO rule (-1) a problem of this order refines, and the solver follows the oracle's traces."""
    from parity import assert_parity

    qp = Q.random_qp(9030, 204, 51, False)  # (78 x the CPU noise floor without refinement in round 4's table)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG)
    dq = M.DeviceQP.from_numpy(hip.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
    out = {}
    for name, kkt_steps in (("driver", 0), ("inside_solve", 1), ("inside_solve_auto", -1)):
        s = M.MPCSolver(dq, hip, regularization=REG, driver="python", refine_steps=0)
        s.initialize()  # (the start point's two solves run before the switch: not refined in any of the three)
        if name == "driver":
            s.opt.refine_steps = 1  # (read by solve_system at every call)
        s.kkt.set_refine(kkt_steps)
        while s.iteration_head() is None:
            s.iteration_body()
        out[name] = dict(iter=s.k, x=s.st.x.cpu().numpy().copy(), trace=[dict(t) for t in s.trace])
        s.close()
    assert out["driver"]["iter"] == out["inside_solve"]["iter"] == out["inside_solve_auto"]["iter"]
    assert np.array_equal(out["driver"]["x"], out["inside_solve"]["x"])
    assert np.array_equal(out["driver"]["x"], out["inside_solve_auto"]["x"])
    for a, b in zip(out["driver"]["trace"], out["inside_solve"]["trace"]):
        assert all(a[k] == b[k] for k in ("alpha_p", "alpha_d", "inf_pr", "inf_du", "mu")), (a, b)
    # the default library as a whole (AUTO in the driver, start point included) against the oracle, traces and all
    s = M.MPCSolver(dq, hip, regularization=REG, driver="native")
    r = s.solve()
    s.close()
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED
    assert_parity(r, ref, qp, "soak9030 default", trace=True, regularization=OREG)
