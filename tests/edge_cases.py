"""Problems at the edges of the index lists of SURVEY.md 8a-0 (ind_lb / ind_ub / ind_ineq), shared by the GPU suites:
no constraint row at all, an empty upper or lower bound list, one-sided rows, a 1 x 1 problem, every kind of variable and
row at once, and no bound on anything (both lists empty, every row an equality)."""
import numpy as np

from oracle import qp as Q

EDGE_CASES = ("no_constraints", "lower_bounds_only", "upper_bounds_only", "one_by_one", "mixed", "no_bounds")


def edge_qp(case):
    if case == "no_constraints":  # a box-constrained QP: m = 0, the KKT system is H + Sigma alone
        qp = Q.synthetic_qp(301, 50, 1)
        return Q.DenseQP(H=qp.H, q=qp.q, A=np.zeros((0, 50)), lvar=qp.lvar, uvar=qp.uvar, lcon=np.zeros(0), ucon=np.zeros(0),
                         x0=qp.x0, name="edge-no-constraints")
    if case == "lower_bounds_only":  # nub = 0 on the variables, one-sided rows (slacks bounded below only)
        qp = Q.synthetic_qp(302, 40, 15)
        qp.uvar[:] = np.inf
        qp.ucon[:] = np.inf
        return qp
    if case == "upper_bounds_only":  # nlb = 0 on the variables, rows bounded above only
        qp = Q.synthetic_qp(303, 40, 15)
        qp.lvar[:] = -np.inf
        qp.lcon[:] = -np.inf
        qp.x0[:] = 0.5
        return qp
    if case == "one_by_one":  # a single variable, a single row
        return Q.DenseQP(H=np.array([[2.0]]), q=np.array([-1.0]), A=np.array([[1.0]]), lvar=np.array([0.0]),
                         uvar=np.array([1.0]), lcon=np.array([0.0]), ucon=np.array([0.25]), x0=np.array([0.0]),
                         name="edge-1x1")
    if case == "mixed":  # free, lower-only, upper-only and boxed variables; equality, one-sided and ranged rows
        qp = Q.synthetic_qp(304, 48, 18)
        qp.lvar[0::4], qp.uvar[0::4] = -np.inf, np.inf
        qp.uvar[1::4] = np.inf
        qp.lvar[2::4] = -np.inf
        qp.ucon[0::3] = qp.lcon[0::3] = 0.1
        qp.ucon[1::3] = np.inf
        return qp
    if case == "no_bounds":  # nlb = nub = 0, no slacks: the start point's 0 / 0 (src/solver.jl:93-94)
        qp = Q.synthetic_qp(123, 60, 20)
        qp.lvar[:], qp.uvar[:] = -np.inf, np.inf
        qp.ucon[:] = qp.lcon[:] = 0.25
        return qp
    raise ValueError(case)
