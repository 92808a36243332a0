"""CPU: bench.py's bookkeeping (tests/fake_backend.py drives the product's host loop).  The factorisation count of the
timed region must survive the re-initialisations that happen when the solve converges inside it -- round 1 read a
field of the KKT object that initialize() replaces and reported 1 factorisation for 21."""
import bench
import madqp_jl_amd as M
from fake_backend import FakeBackend
from oracle import mpc
from oracle import qp as Q


def make_solver(max_ncorr=0):
    qp = Q.synthetic_qp(20250614, 40, 16)
    dq = M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
    s = M.MPCSolver(dq, FakeBackend(), regularization=M.FixedRegularization(1e-8, -1e-8), max_iter=300,
                    step_rule=M.AdaptiveStep(0.995), mu_min=1e-12, max_ncorr=max_ncorr)
    s.initialize()
    return s


def test_factorizations_counted_across_reinitializations():
    s = make_solver()
    iters = mpc.solve(Q.synthetic_qp(20250614, 40, 16), kkt_system="condensed",
                      regularization=mpc.FixedRegularization(1e-8, -1e-8), step_rule=mpc.AdaptiveStep(0.995),
                      mu_min=1e-12, max_iter=300)["iter"]
    loop = bench.StepLoop(s)
    for _ in range(3):  # warm-up, as bench.measure does
        loop.step()
    loop.reset()
    steps = 3 * iters + 2  # converges (at least) twice inside the timed loop
    totals = []
    for _ in range(steps):
        loop.step()
        totals.append(s.n_factorizations_total)
    assert loop.steps == steps
    assert loop.reinits >= 2
    # one factorisation per iteration plus one start-point factorisation per re-initialisation (no retries here)
    assert loop.factorizations() == steps + loop.reinits
    assert all(b >= a for a, b in zip(totals, totals[1:])), "the counter must be monotone"
    # the object-level counter restarts with every initialize(): this is what bench.py must NOT read
    assert s.kkt.n_factorizations < loop.factorizations()
    assert loop.excluded > 0.0


def test_total_counter_without_reinit_equals_kkt_counter():
    s = make_solver(max_ncorr=3)
    loop = bench.StepLoop(s)
    loop.step()
    loop.step()
    assert loop.reinits == 0 and loop.factorizations() == 2
    assert s.n_factorizations_total == s.kkt.n_factorizations == 3  # start point + 2 iterations
