"""GPU parity: every fused vector kernel of include/madqp.h against its numpy restatement
(tests/fake_backend.py, which follows src/kernels.jl line by line) on the same seeded state.
Elementwise results and max/min/arg-min reductions must be BIT-EXACT; sums within 1e-13 relative."""
import numpy as np
import pytest
import torch

from fake_backend import FakeBackend
from madqp_jl_amd.backend import State

pytestmark = pytest.mark.gpu

FIELDS = ("x", "xl", "xu", "zl", "zu", "f", "jacl", "reg", "pr_diag", "y", "c", "du_diag", "rhs",
          "d", "p", "w1", "w2", "correction_lb", "correction_ub", "l_diag", "l_lower", "u_diag", "u_lower")


def make_states(hip, n, m, seed, identity=False):
    rng = np.random.default_rng(seed)
    if identity:
        ind_lb, ind_ub = np.arange(n), np.arange(n)
    else:
        ind_lb = np.sort(rng.choice(n, size=max(1, (2 * n) // 3), replace=False)) if n else np.arange(0)
        ind_ub = np.sort(rng.choice(n, size=max(1, n // 2), replace=False)) if n else np.arange(0)
    cpu = State(n, m, ind_lb, ind_ub, "cpu")
    xl = rng.uniform(-2, 0, n)
    xu = xl + rng.uniform(0.5, 3, n)
    vals = dict(
        xl=xl, xu=xu, x=xl + (xu - xl) * rng.uniform(0.01, 0.99, n),
        zl=rng.uniform(0.01, 2, n), zu=rng.uniform(0.01, 2, n), f=rng.standard_normal(n),
        jacl=rng.standard_normal(n), reg=rng.uniform(0, 1, n), pr_diag=rng.uniform(0.5, 2, n),
        y=rng.standard_normal(m), c=rng.standard_normal(m), du_diag=-rng.uniform(0.1, 1, m),
        rhs=rng.standard_normal(m), d=rng.standard_normal(cpu.ntot), p=rng.standard_normal(cpu.ntot),
        w1=rng.standard_normal(cpu.ntot), w2=rng.standard_normal(cpu.ntot),
        correction_lb=rng.standard_normal(cpu.nlb), correction_ub=rng.standard_normal(cpu.nub),
        l_diag=-rng.uniform(0.1, 1, cpu.nlb), l_lower=rng.uniform(0.1, 1, cpu.nlb),
        u_diag=-rng.uniform(0.1, 1, cpu.nub), u_lower=rng.uniform(0.1, 1, cpu.nub))
    gpu = hip.new_state(n, m, ind_lb, ind_ub)
    for k, v in vals.items():
        getattr(cpu, k).copy_(torch.as_tensor(v))
        getattr(gpu, k).copy_(torch.as_tensor(v))
    return cpu, gpu


def assert_states_equal(cpu, gpu, what):
    for k in FIELDS:
        a, b = getattr(cpu, k).numpy(), getattr(gpu, k).cpu().numpy()
        assert np.array_equal(a, b, equal_nan=True), f"{what}: field {k} differs, max {np.max(np.abs(a - b))}"


SIZES = [(1, 1), (7, 3), (300, 120), (5000, 1700), (300000, 100000)]


@pytest.mark.parametrize("n,m", SIZES)
def test_elementwise_kernels_bit_exact(hip, n, m):
    fake = FakeBackend()
    calls = [
        ("set_aug_diagonal_reg", (1e-8, -1e-8)), ("set_predictive_rhs", ()), ("set_correction_rhs", (0.37,)),
        ("set_initial_primal_rhs", ()), ("set_initial_dual_rhs", ()), ("get_correction", ()),
        ("set_extra_correction", (0.9, 0.8, 0.1, 10.0, 0.05)), ("update_iterates", (0.7, 0.6)),
        ("adjust_boundary", (1e-3,)), ("sp_init_duals", ()), ("sp_shift", (0.25, 1.5)),
        ("sp_project", (1e-2,)),
    ]
    for name, args in calls:
        cpu, gpu = make_states(hip, n, m, seed=hash(name) % 1000)
        getattr(fake, name)(cpu, *args)
        getattr(hip, name)(gpu, *args)
        assert_states_equal(cpu, gpu, name)
    for name, args in [("reduce_rhs", ()), ("finish_aug_solve", ())]:
        cpu, gpu = make_states(hip, n, m, seed=3)
        getattr(fake, name)(cpu, cpu.w1, *args)
        getattr(hip, name)(gpu, gpu.w1, *args)
        assert_states_equal(cpu, gpu, name)
    cpu, gpu = make_states(hip, n, m, seed=4)
    fake.kktmul(cpu, cpu.w1, cpu.d, -1.0, 1.0)
    hip.kktmul(gpu, gpu.w1, gpu.d, -1.0, 1.0)
    assert_states_equal(cpu, gpu, "kktmul")


def test_adjust_boundary_triggers(hip):
    cpu, gpu = make_states(hip, 500, 10, seed=9)
    for st in (cpu, gpu):
        st.x[st.ind_lb[:50]] = st.xl[st.ind_lb[:50]] + 1e-20
        st.x[st.ind_ub[-50:]] = st.xu[st.ind_ub[-50:]] - 1e-20
    before = cpu.xl.clone()
    FakeBackend().adjust_boundary(cpu, 1e-3)
    hip.adjust_boundary(gpu, 1e-3)
    assert not torch.equal(before, cpu.xl)
    assert_states_equal(cpu, gpu, "adjust_boundary(triggered)")


@pytest.mark.parametrize("n,m", SIZES)
def test_reductions(hip, n, m):
    fake = FakeBackend()
    cpu, gpu = make_states(hip, n, m, seed=21)
    # order-independent -> bit exact
    assert fake.get_inf(cpu) == hip.get_inf(gpu)
    assert fake.sp_mins(cpu) == hip.sp_mins(gpu)
    assert fake.norm_inf3(cpu.w1, cpu.p, cpu.d) == hip.norm_inf3(gpu.w1, gpu.p, gpu.d)
    assert fake.norm_inf(cpu.f) == hip.norm_inf(gpu.f)
    assert fake.sp_check(cpu) == hip.sp_check(gpu) is True
    for tau in (1.0, 0.995):
        (a0, i0), (a1, i1) = fake.get_alpha_max(cpu, tau), hip.get_alpha_max(gpu, tau)
        assert a0 == a1 and i0 == i1, (tau, a0, a1, i0, i1)
    # sums -> tolerance
    rel = lambda a, b: abs(a - b) / max(abs(a), abs(b), 1e-300)
    assert rel(fake.get_complementarity_measure(cpu), hip.get_complementarity_measure(gpu)) < 1e-13
    assert rel(fake.get_affine_complementarity_measure(cpu, 0.9, 0.8),
               hip.get_affine_complementarity_measure(gpu, 0.9, 0.8)) < 1e-13
    for a, b in zip(fake.sp_sums(cpu), hip.sp_sums(gpu)):
        assert rel(a, b) < 1e-12, (a, b)


def test_alpha_max_ties_and_nothing_blocks(hip):
    """Exact ties resolve to the LAST index, as the reference's reducer `elem1[1] < elem2[1] ? elem1 : elem2` folded from the
    left does (src/kernels.jl:248); no blocking bound -> (1.0, -1) (src/kernels.jl:243-251)."""
    n = 4000
    cpu, gpu = make_states(hip, n, 5, seed=2, identity=True)
    for st in (cpu, gpu):
        st.d.zero_()
        st.primal(st.d)[:] = 1e-9  # dx > 0 tiny: no lower blocking, upper far away
    a, ib = hip.get_alpha_max(gpu, 1.0)
    assert a == [1.0, 1.0, 1.0, 1.0] and ib == [-1, -1, -1, -1]
    for st in (cpu, gpu):
        st.x[:] = 0.5
        st.xl[:] = 0.0
        st.xu[:] = 1.0
        st.primal(st.d)[:] = 0.0
        st.primal(st.d)[[3000, 1234, 77]] = -2.0  # three exact ties: alpha = 0.25
    a, ib = hip.get_alpha_max(gpu, 1.0)
    a_ref, ib_ref = FakeBackend().get_alpha_max(cpu, 1.0)
    assert a == a_ref and ib == ib_ref and ib[0] == 3000 and a[0] == 0.25


def test_nan_propagates_through_max_norms(hip):
    """norm(w, Inf) must return NaN so that solve_system! throws (src/linear_solver.jl:41-43)."""
    v = torch.ones(100000, dtype=torch.float64, device=hip.device)
    v[54321] = float("nan")
    out = hip.norm_inf3(v, v, v)
    assert all(np.isnan(o) for o in out)
    assert np.isnan(hip.norm_inf(v))


def test_empty_bound_lists(hip):
    st = hip.new_state(10, 4, np.arange(0), np.arange(0))
    assert hip.get_complementarity_measure(st) == 0.0  # src/kernels.jl:173-174
    assert hip.get_alpha_max(st, 1.0) == ([1.0] * 4, [-1] * 4)
    hip.set_aug_diagonal_reg(st, 0.5, -0.25)
    assert torch.all(st.pr_diag == 0.5) and torch.all(st.du_diag == -0.25)
    assert hip.sp_check(st)
