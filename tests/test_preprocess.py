"""CPU: the step in front of the path (madqp_jl_amd/preprocess.py; reference: scripts/common.jl): QPS reader,
Ruiz scaling, standard form.  Pins: hand-written instance files of problems with known models / optima
(HS21 of Maros-Meszaros, the simple LP of test/runtests.jl:24-55), structural identities of
scripts/common.jl:109-288 and equivalence of the optima before / after each transformation (oracle)."""
import numpy as np
import pytest
import scipy.sparse as sp

from madqp_jl_amd import preprocess as P
from oracle import mpc
from oracle import qp as Q

HS21_QPS = """NAME          HS21
ROWS
 N  OBJ.FUNC
 G  R------1
COLUMNS
    C------1  R------1  0.100000e+02
    C------2  R------1  -.100000e+01
RHS
    RHS1      OBJ.FUNC  0.100000e+03
    RHS1      R------1  0.100000e+02
RANGES
BOUNDS
 LO BOUNDS    C------1  0.200000e+01
 UP BOUNDS    C------1  0.500000e+02
 LO BOUNDS    C------2  -.500000e+02
 UP BOUNDS    C------2  0.500000e+02
QUADOBJ
    C------1  C------1  0.200000e-01
    C------2  C------2  0.200000e+01
ENDATA
"""

MIXED_MPS = """* free format, every row / bound flavour
NAME MIXED
OBJSENSE
    MAX
ROWS
 N COST
 N SPARE
 E EQ1
 L LE1
 G GE1
 E EQR
 L LER
 G GER
COLUMNS
 X1 COST 1.0 EQ1 1.0
 X1 LE1 2.0 SPARE 9.0
 MARKER 'MARKER' 'INTORG'
 X2 COST -2.0 GE1 1.0
 X2 EQR 1.0
 MARKER 'MARKER' 'INTEND'
 X3 LER 1.0 GER -1.0
 X4 EQ1 3.0
 X5 COST 0.5
 X6 LE1 1.0
RHS
 B EQ1 4.0 LE1 10.0
 B GE1 -1.0 EQR 2.0
 B LER 5.0 GER 1.0
 B COST -7.0
RANGES
 R EQR -3.0 LER 2.0
 R GER 4.0
BOUNDS
 UP BND X1 4.0
 UP BND X2 -1.0
 LO BND X3 -5.0
 FX BND X4 2.5
 FR BND X5
 MI BND X6
ENDATA
"""


def dense(qp):
    return Q.DenseQP(H=qp.H.toarray(), q=qp.c, A=qp.A.toarray(), lvar=qp.lvar, uvar=qp.uvar, lcon=qp.lcon,
                     ucon=qp.ucon, x0=qp.x0, c0=qp.c0, name=qp.name)


def solve(qp, **kw):
    kw.setdefault("regularization", mpc.FixedRegularization(1e-8, -1e-8))
    return mpc.solve(dense(qp), kkt_system="condensed", **kw)


def test_reader_hs21_known_model_and_optimum(tmp_path):
    qp = P.read_qps(HS21_QPS)
    ref = Q.hs21()
    assert qp.name == "HS21" and qp.nvar == 2 and qp.ncon == 1 and qp.nnzj == 2 and qp.nnzh == 2
    assert np.array_equal(qp.H.toarray(), ref.H) and np.array_equal(qp.c, ref.q) and qp.c0 == ref.c0 == -100.0
    assert np.array_equal(qp.A.toarray(), ref.A)
    for k in ("lvar", "uvar", "lcon", "ucon"):
        assert np.array_equal(getattr(qp, k), getattr(ref, k)), k
    r = solve(qp)
    assert r["status"] == mpc.SOLVE_SUCCEEDED and abs(r["objective"] + 99.96) < 1e-7
    # from a (gzipped) file, as import_mps does (scripts/common.jl:21-36)
    import gzip

    path = tmp_path / "HS21.SIF.gz"
    with gzip.open(path, "wt") as f:
        f.write(HS21_QPS)
    q2 = P.read_qps(str(path))
    assert np.array_equal(q2.A.toarray(), qp.A.toarray()) and np.array_equal(q2.uvar, qp.uvar)


def test_reader_rows_ranges_bounds_and_sense():
    qp = P.read_qps(MIXED_MPS)
    assert qp.varnames == ["X1", "X2", "X3", "X4", "X5", "X6"]
    assert qp.connames == ["EQ1", "LE1", "GE1", "EQR", "LER", "GER"]  # N rows dropped, order kept
    # OBJSENSE MAX: everything of the objective negated; RHS on the objective row = -constant
    assert np.array_equal(qp.c, -np.array([1.0, -2.0, 0, 0, 0.5, 0])) and qp.c0 == -7.0
    A = np.zeros((6, 6))
    A[0, 0], A[1, 0], A[2, 1], A[3, 1], A[4, 2], A[5, 2], A[0, 3], A[1, 5] = 1, 2, 1, 1, 1, -1, 3, 1
    assert np.array_equal(qp.A.toarray(), A)
    inf = np.inf
    #                 EQ1  LE1   GE1  EQR(R<0)  LER(R=2)  GER(R=4)
    assert np.array_equal(qp.lcon, [4.0, -inf, -1.0, -1.0, 3.0, 1.0])
    assert np.array_equal(qp.ucon, [4.0, 10.0, inf, 2.0, 5.0, 5.0])
    #                 X1   X2 (UP<0 -> lo=-inf)  X3    X4 (FX)  X5 (FR)  X6 (MI)
    assert np.array_equal(qp.lvar, [0.0, -inf, -5.0, 2.5, -inf, -inf])
    assert np.array_equal(qp.uvar, [4.0, -1.0, inf, 2.5, inf, inf])
    with pytest.raises(ValueError):
        P.read_qps("NAME X\nFOO\n")


def test_ruiz_scaling_equilibrates_and_preserves_the_optimum():
    qp0 = Q.random_qp(5, 30, 14)
    qp0.A[:, 11] = 0.0  # an empty column
    rng = np.random.default_rng(1)
    dr = 10.0 ** rng.uniform(-3, 3, 14)
    bad = sp.diags(dr) @ sp.csr_matrix(qp0.A)  # badly scaled rows (bounds scaled with them)
    qp = P.HostQP(qp0.c0, qp0.q, sp.csr_matrix(qp0.H), bad, qp0.lvar, qp0.uvar, qp0.lcon * dr, qp0.ucon * dr)
    qs, Dr, Dc = P.ruiz_scale(qp)
    As = abs(qs.A)
    rmax, cmax = As.max(axis=1).toarray().ravel(), As.max(axis=0).toarray().ravel()
    assert np.max(np.abs(rmax[rmax > 0] - 1.0)) < 1e-6 and np.max(np.abs(cmax[cmax > 0] - 1.0)) < 1e-6
    assert np.all(Dc[cmax == 0] == 1.0)  # empty columns keep the factor 1
    # the transformation of scripts/common.jl:68-94
    assert np.allclose(qs.A.toarray(), bad.toarray() / np.outer(Dr, Dc)) and np.allclose(qs.c, qp.c / Dc)
    assert np.allclose(qs.H.toarray(), qp.H.toarray() / np.outer(Dc, Dc))
    assert np.allclose(qs.uvar, qp.uvar * Dc) and np.allclose(qs.ucon, qp.ucon / Dr)
    r, rs = solve(qp, max_iter=500), solve(qs, max_iter=500)
    assert r["status"] == rs["status"] == mpc.SOLVE_SUCCEEDED
    assert abs(r["objective"] - rs["objective"]) <= 1e-6 * max(1.0, abs(r["objective"]))
    assert np.max(np.abs(rs["solution"] / Dc - r["solution"])) < 1e-5


def test_standard_form_structure_and_equivalence():
    """scripts/common.jl:109-288: counts, blocks and bounds of the reformulation; same optimum."""
    qp = P.read_qps(MIXED_MPS)
    qp.lvar[4] = -3.0  # X5: lower bound only
    qp.uvar[5] = 6.0   # X6: upper bound only
    sf = P.standard_form(qp)
    n, m = 6, 6
    ineq = [1, 2, 3, 4, 5]            # rows with lcon < ucon (EQR became a range through RANGES)
    rng_x = [0]                       # X1 is the only variable with two finite, different bounds
    rng_s = [k for k, i in enumerate(ineq) if np.isfinite(qp.lcon[i]) and np.isfinite(qp.ucon[i])]  # EQR, LER, GER
    ns, nw = len(ineq), len(rng_x) + len(rng_s)
    assert (sf.nvar, sf.ncon) == (n + ns + nw, m + nw) and sf.nnzj == qp.nnzj + ns + 2 * nw
    A = sf.A.toarray()
    assert np.array_equal(A[:m, :n], qp.A.toarray())
    for k, i in enumerate(ineq):  # A x - s = 0
        assert A[i, n + k] == -1.0 and sf.lcon[i] == sf.ucon[i] == 0.0
    assert sf.lcon[0] == sf.ucon[0] == 4.0  # equality row kept
    for k, j in enumerate(rng_x + [n + k for k in rng_s]):  # x + w = xu
        row = A[m + k]
        assert row[j] == 1.0 and row[n + ns + k] == 1.0 and np.count_nonzero(row) == 2
        assert sf.lcon[m + k] == sf.ucon[m + k] == (qp.uvar[j] if j < n else qp.ucon[ineq[j - n]])
        assert sf.uvar[j] == np.inf and sf.lvar[n + ns + k] == 0.0 and sf.uvar[n + ns + k] == np.inf
    assert sf.lvar[3] == sf.uvar[3] == 2.5  # fixed variable kept in the formulation
    assert sf.uvar[5] == 6.0                # upper bound only: stays a bound
    # equivalence on a problem without fixed variables (the oracle does not treat them)
    qp2 = Q.random_qp(5, 12, 7, lp=False)
    h = P.HostQP(qp2.c0, qp2.q, sp.csr_matrix(qp2.H), sp.csr_matrix(qp2.A), qp2.lvar, qp2.uvar, qp2.lcon, qp2.ucon)
    a, b = solve(h), solve(P.standard_form(h))
    assert a["status"] == b["status"] == mpc.SOLVE_SUCCEEDED
    assert abs(a["objective"] - b["objective"]) <= 1e-6 * max(1.0, abs(a["objective"]))
    assert np.max(np.abs(b["solution"][:12] - a["solution"])) < 1e-5


def test_benchmark_row():
    qp = P.read_qps(HS21_QPS)
    row = P.benchmark_row(qp, dict(status=1, iter=7, objective=-99.96), 0.5, 0.25)
    assert row == (2, 1, 2, 2, 1, 7, -99.96, 0.5, 0.25)


# ------------------------------------------------------------------------------------------------ presolve
def kkt_violation(qp, s):
    """Largest violation of the optimality conditions of ``qp`` at the primal-dual point ``s`` (x, y, zl, zu):
    stationarity H x + c + A'y - zl + zu = 0, feasibility, signs, complementarity -- no solver involved."""
    x, y, zl, zu = s["x"], s["y"], s["zl"], s["zu"]
    Ax = qp.A @ x
    v = [np.max(np.abs(qp.H @ x + qp.c + qp.A.T @ y - zl + zu), initial=0.0),
         np.max(np.maximum(qp.lvar - x, 0), initial=0.0), np.max(np.maximum(x - qp.uvar, 0), initial=0.0),
         np.max(np.maximum(qp.lcon - Ax, 0), initial=0.0), np.max(np.maximum(Ax - qp.ucon, 0), initial=0.0),
         np.max(np.maximum(-zl, 0), initial=0.0), np.max(np.maximum(-zu, 0), initial=0.0)]
    fin = lambda b, d: np.where(np.isfinite(b), d, 0.0)
    v += [np.max(np.abs(zl * fin(qp.lvar, x - qp.lvar)), initial=0.0), np.max(np.abs(zu * fin(qp.uvar, qp.uvar - x)), initial=0.0),
          np.max(np.abs(np.where(np.isfinite(qp.lvar), 0.0, zl)), initial=0.0),
          np.max(np.abs(np.where(np.isfinite(qp.uvar), 0.0, zu)), initial=0.0),
          np.max(np.abs(np.maximum(y, 0) * fin(qp.ucon, qp.ucon - Ax)), initial=0.0),  # y > 0: upper side active
          np.max(np.abs(np.minimum(y, 0) * fin(qp.lcon, Ax - qp.lcon)), initial=0.0),
          np.max(np.abs(np.where(np.isfinite(qp.ucon), 0.0, np.maximum(y, 0))), initial=0.0),
          np.max(np.abs(np.where(np.isfinite(qp.lcon), 0.0, np.minimum(y, 0))), initial=0.0)]
    return max(v)


def planted_qp(seed, lp=False):
    """A random sparse QP with every structure the presolve removes planted into it."""
    rng = np.random.default_rng(seed)
    n, m = 30, 22
    A = sp.random(m, n, density=0.25, random_state=rng, data_rvs=rng.standard_normal).tolil()
    R = sp.random(n, n, density=0.1, random_state=rng, data_rvs=rng.standard_normal)
    H = (R @ R.T + sp.identity(n)).tolil() if not lp else sp.lil_matrix((n, n))
    c = rng.standard_normal(n)
    lvar, uvar = -1.0 - rng.random(n), 1.0 + rng.random(n)
    xf = rng.uniform(-0.5, 0.5, n)  # a strictly feasible point for the rows
    fixed, free_cols = [2, 11, 19], [5, 23]
    lvar[fixed] = uvar[fixed] = xf[fixed]
    for j in free_cols:  # variables in no row and no quadratic term
        A[:, j] = 0.0
        H[j, :] = 0.0
        H[:, j] = 0.0
    c[5], c[23] = 0.7, -0.4
    lvar[7] = -np.inf  # a one-sided variable
    empty, single, single_eq, single_neg, redundant = [3, 14], 6, 9, 17, 20
    for i in empty:
        A[i, :] = 0.0
    A[redundant, 7] = 0.0  # keep the redundant row's activity range finite
    for i, j, a in ((single, 8, 2.0), (single_eq, 12, -1.5), (single_neg, 21, -0.5)):
        A[i, :] = 0.0
        A[i, j] = a
    A = A.tocsr()
    Ax = A @ xf
    lcon, ucon = Ax - 0.3 - rng.random(m), Ax + 0.3 + rng.random(m)
    eq = [0, 10]
    lcon[eq] = ucon[eq] = Ax[eq]
    lcon[empty], ucon[empty] = [-1.0, 0.0], [0.0, np.inf]
    lcon[single], ucon[single] = 2.0 * (xf[8] - 0.05), np.inf  # tightens the lower bound of x8 (likely active)
    lcon[single_eq] = ucon[single_eq] = -1.5 * xf[12]  # fixes x12 through a row
    lcon[single_neg], ucon[single_neg] = -np.inf, -0.5 * (xf[21] - 0.02)  # a < 0: an upper row bound -> lower bound
    lcon[redundant], ucon[redundant] = -1e3, 1e3
    c[8], c[21] = 3.0, 3.0  # push x8, x21 against the bounds the singleton rows created
    return P.HostQP(0.25, c, sp.csr_matrix(H), A, lvar, uvar, lcon, ucon, name=f"planted{seed}")


@pytest.mark.parametrize("seed,lp", [(0, False), (1, False), (2, True), (3, True)])
def test_presolve_reductions_and_postsolve(seed, lp):
    qp = planted_qp(seed, lp)
    ps = P.presolve(qp)
    assert ps.flag and ps.status == "reduced"
    gone_v, gone_c = set(range(qp.nvar)) - set(ps.keep_var), set(range(qp.ncon)) - set(ps.keep_con)
    assert {2, 11, 19, 5, 23, 12} <= gone_v  # fixed, unconstrained linear, fixed by a singleton equality row
    assert {3, 14, 6, 9, 17, 20} <= gone_c  # empty, singleton, redundant rows
    assert ps.qp.nvar == qp.nvar - len(gone_v) and ps.qp.ncon == qp.ncon - len(gone_c)
    assert ps.x_removed[5] == qp.lvar[5] and ps.x_removed[23] == qp.uvar[23]  # cost sign picks the bound
    r = mpc.solve(dense(ps.qp), kkt_system="K2", tol=1e-9)
    assert r["status"] == mpc.SOLVE_SUCCEEDED
    full = ps.postsolve(r["solution"], r["multipliers"], r["multipliers_L"], r["multipliers_U"])
    assert abs(full["objective"] - r["objective"]) <= 1e-8 * max(1.0, abs(r["objective"]))  # c0 bookkeeping
    assert kkt_violation(qp, full) <= 2e-6  # optimal for the ORIGINAL model, duals of the removed rows included
    assert abs(full["y"][6]) > 1e-3 or abs(full["y"][17]) > 1e-3  # a singleton row's multiplier was recovered
    if lp:
        from scipy.optimize import linprog

        cons = sp.vstack([qp.A, -qp.A]).tocsr()
        rhs = np.concatenate([qp.ucon, -qp.lcon])
        ok = np.isfinite(rhs)
        lp_ref = linprog(qp.c, A_ub=cons[ok], b_ub=rhs[ok], bounds=list(zip(qp.lvar, qp.uvar)), method="highs")
        assert lp_ref.status == 0 and abs(lp_ref.fun + qp.c0 - full["objective"]) <= 1e-6
    ps2 = P.presolve(ps.qp)  # idempotent
    assert ps2.status == "unchanged" and ps2.qp is ps.qp


def test_presolve_flags():
    """The three ways presolve_qp returns flag = false (scripts/common.jl:121-124): solved, infeasible, unbounded."""
    z = lambda r, c: sp.csr_matrix((r, c))
    inf = np.inf
    # everything eliminated: x0 fixed, x1 unconstrained linear, the row a singleton turned into a bound
    qp = P.HostQP(1.0, np.array([2.0, -1.0]), z(2, 2), sp.csr_matrix([[0.0, 1.0]]), np.array([3.0, 0.0]),
                  np.array([3.0, inf]), np.array([-inf]), np.array([4.0]))
    ps = P.presolve(qp)
    assert not ps.flag and ps.status == "solved"
    full = ps.postsolve()
    assert np.allclose(full["x"], [3.0, 4.0]) and full["objective"] == pytest.approx(1.0 + 6.0 - 4.0)
    # an empty row that cannot hold
    qp = P.HostQP(0.0, np.ones(2), z(2, 2), z(1, 2), np.zeros(2), np.ones(2), np.array([1.0]), np.array([2.0]))
    assert P.presolve(qp).status == "infeasible" and not P.presolve(qp).flag
    # a singleton row against the variable's bounds
    qp = P.HostQP(0.0, np.ones(1), z(1, 1), sp.csr_matrix([[1.0]]), np.zeros(1), np.ones(1), np.array([2.0]), np.array([3.0]))
    assert P.presolve(qp).status == "infeasible"
    # a free column whose cost pushes it to infinity
    qp = P.HostQP(0.0, np.array([-1.0, 1.0]), z(2, 2), sp.csr_matrix([[0.0, 1.0]]), np.zeros(2), np.array([inf, 1.0]),
                  np.array([0.0]), np.array([1.0]))
    assert P.presolve(qp).status == "unbounded"
    # nothing to do
    qp = P.HostQP(0.0, np.ones(2), sp.identity(2), sp.csr_matrix([[1.0, 1.0]]), np.zeros(2), np.ones(2),
                  np.array([0.5]), np.array([1.5]))
    ps = P.presolve(qp)
    assert ps.flag and ps.status == "unchanged" and ps.qp is qp


def test_write_qps_round_trip(tmp_path):
    """write_qps is the inverse of read_qps (values written with repr): every array comes back, ranged rows to
    one rounding of ucon - (ucon - lcon); gz and plain."""
    for seed, lp, ext in ((0, False, ".qps"), (3, True, ".mps.gz")):
        qp = planted_qp(seed, lp)
        path = str(tmp_path / f"inst{seed}{ext}")
        P.write_qps(qp, path)
        r = P.read_qps(path)
        assert (r.nvar, r.ncon, r.c0) == (qp.nvar, qp.ncon, qp.c0)
        assert abs(r.A - qp.A).max() == 0.0 and (abs(r.H - qp.H).max() if qp.H.nnz else 0.0) == 0.0
        assert np.array_equal(r.c, qp.c) and np.array_equal(r.lvar, qp.lvar) and np.array_equal(r.uvar, qp.uvar)
        assert np.array_equal(r.ucon, qp.ucon) and np.allclose(r.lcon, qp.lcon, rtol=0.0, atol=1e-15)
    hs = P.read_qps(HS21_QPS)
    P.write_qps(hs, str(tmp_path / "hs21.qps"))
    back = P.read_qps(str(tmp_path / "hs21.qps"))
    assert back.varnames == hs.varnames and back.connames == hs.connames and back.c0 == hs.c0
    assert abs(back.H - hs.H).max() == 0.0 and np.array_equal(back.lcon, hs.lcon)


def random_structured_qp(seed):
    """Random small QP / LP with fixed variables, free columns, empty and singleton rows, one-sided and equality rows
    thrown in at random (a strictly feasible point exists by construction)."""
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(2, 25)), int(rng.integers(0, 20))
    A = sp.random(m, n, density=rng.uniform(0.05, 0.5), random_state=rng, data_rvs=rng.standard_normal).tolil()
    lp = rng.random() < 0.4
    R = sp.random(n, n, density=0.15, random_state=rng, data_rvs=rng.standard_normal)
    H = (R @ R.T + 0.5 * sp.identity(n)).tolil() if not lp else sp.lil_matrix((n, n))
    c = rng.standard_normal(n)
    xf = rng.uniform(-0.5, 0.5, n)
    lvar, uvar = xf - rng.uniform(0.1, 1.5, n), xf + rng.uniform(0.1, 1.5, n)
    for j in range(n):
        u = rng.random()
        if u < 0.12:
            lvar[j] = uvar[j] = xf[j]
        elif u < 0.2:
            lvar[j] = -np.inf
        elif u < 0.28:
            uvar[j] = np.inf
        elif u < 0.36 and not lp:
            H[j, :] = 0
            H[:, j] = 0
            A[:, j] = 0
    for i in range(m):
        u = rng.random()
        if u < 0.1:
            A[i, :] = 0
        elif u < 0.3:
            j, a = int(rng.integers(0, n)), rng.choice([-2.0, 0.7, 1.5])
            A[i, :] = 0
            A[i, j] = a
    A = A.tocsr()
    Ax = A @ xf
    lcon, ucon = Ax - rng.uniform(0.05, 1.0, m), Ax + rng.uniform(0.05, 1.0, m)
    for i in range(m):
        u = rng.random()
        if u < 0.2:
            lcon[i] = ucon[i] = Ax[i]
        elif u < 0.3:
            lcon[i] = -np.inf
        elif u < 0.4:
            ucon[i] = np.inf
    if lp:  # keep the LP bounded
        lvar, uvar = np.where(np.isfinite(lvar), lvar, xf - 2), np.where(np.isfinite(uvar), uvar, xf + 2)
    return P.HostQP(0.1, c, sp.csr_matrix(H), A, lvar, uvar, lcon, ucon, name=f"fuzz{seed}")


def test_presolve_fuzz():
    """40 random structured models: whatever the presolve removes, the postsolved point satisfies the optimality
    conditions of the original model (120 seeds of the same generator: 105 reduced, none violated)."""
    reduced = 0
    for seed in range(40):
        qp = random_structured_qp(seed)
        ps = P.presolve(qp)
        assert ps.status in ("reduced", "unchanged", "solved"), (seed, ps.status)
        if ps.status == "solved":
            x = ps.postsolve()["x"]
            assert np.all(x >= qp.lvar - 1e-9) and np.all(x <= qp.uvar + 1e-9)
            assert np.all(qp.A @ x >= qp.lcon - 1e-7) and np.all(qp.A @ x <= qp.ucon + 1e-7)
            continue
        reduced += ps.status == "reduced"
        r = mpc.solve(dense(ps.qp), kkt_system="K2", tol=1e-9, fixed_variable_treatment="relax_bound")
        assert r["status"] == mpc.SOLVE_SUCCEEDED, seed
        full = ps.postsolve(r["solution"], r["multipliers"], r["multipliers_L"], r["multipliers_U"])
        assert kkt_violation(qp, full) <= 5e-6, seed
    assert reduced >= 25
