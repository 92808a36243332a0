"""Shared acceptance rule of the whole-solve parity tests (test infrastructure).

Stated bar (test/runtests.jl:110-114): identical status, identical iteration count, objective and
solution within tolerance.  One refinement, with its justification:

The termination test `max(inf_pr, inf_du, inf_compl) <= tol` (src/solver.jl:279) is a threshold on quantities that
carry rounding noise of the linear solves.  On LPs solved through the condensed form (K = delta_w I + A' Theta A with
Theta up to 1e8: conditioned at the edge of fp64) that noise reaches tens of per cent of `tol` in the last iterations,
so whether iteration k or k+1 is the last one can depend on the summation order inside the Cholesky factorisation --
for ANY two correct implementations: the oracle itself (LAPACK through scipy) stops after 12 iterations on the GPU
box's host CPU and after 13 in the build container on the same problem (tests/test_gpu_soak.py, seed 9195), and a
numpy emulation of a blocked Cholesky with plain substitution differs from LAPACK in ~1 % of such LPs, in both
directions.  Such a *threshold tie* is accepted, and only it: the counts differ by exactly one,
the run that stopped first satisfied the test at iteration k, and the run that went on missed it at that same
iteration by less than TIE_FACTOR.  Everything else (objective, solution, multipliers) is still compared.
Ties are an LP matter: `iteration_parity` refuses one on a problem with a Hessian unless the caller says the problem
is an LP.

Second refinement (round 3): WHERE the stated per-iteration tolerance (1e-9 while mu >= 1e-4, 1e-6 after) is not what
the conditioning of a problem supports, the tolerance is not loosened by hand any more; it is MEASURED on the CPU: the
oracle runs the problem twice, as the reference does (LAPACK solves) and with one step of iterative refinement of
every solve (`refine_steps=1`, oracle/mpc.py) -- two equally valid executions of the same algorithm that differ only
in the rounding of the linear solves.  `sensitivity(ref, ref2)` is their distance per iteration and quantity;
`trace_tolerances` allows the device SENS_FACTOR times that distance where it exceeds the stated bar (e.g. the
condensed LP of tests/test_gpu_dist2d.py: 9e-7 at iteration 16 between the two CPU runs, where round 2 had set 1e-5
by hand; 1e-12 everywhere before iteration 12, where the bar stays 1e-9).  SENS_FACTOR = 16: the distance between the
two CPU runs is the error of ONE LAPACK solve sequence; the device's blocked factorisation multiplies with explicit
inverses of its 128 x 128 diagonal blocks (panel solves and sweeps) where LAPACK substitutes, which costs a factor
cond(L_kk)-ish in the error bound -- the largest ratio measured in this suite is 9 (the start point of the QP with
equality rows at Theta = 1e8, cond(K) = 3.6e7: 4.9e-7 against 1.2e-7; soak seed 31315: 1.06e-7 against 1.2e-8).
"""
TIE_FACTOR = 4.0
SENS_FACTOR = 16.0
TRACE_KEYS = ("alpha_p", "alpha_d", "inf_pr", "inf_du", "inf_compl", "mu")


def close(a, b, tol):
    return abs(a - b) <= tol * max(1.0, abs(a), abs(b))


def stated_bar(mu_a, mu_b):
    """1e-9 while mu >= 1e-4, 1e-6 afterwards (conditioning ~ 1/mu)."""
    return 1e-9 if min(mu_a, mu_b) >= 1e-4 else 1e-6


def sensitivity(ref, ref2):
    """Per iteration, the largest relative distance over TRACE_KEYS between two CPU runs of the oracle (same iteration
    count required: otherwise the problem is a threshold tie for the oracle itself and the caller must treat it so)."""
    assert len(ref) == len(ref2), ("the two oracle runs stop at different iterations", len(ref) - 1, len(ref2) - 1)
    return [max(abs(a[k] - b[k]) / max(1.0, abs(a[k]), abs(b[k])) for k in TRACE_KEYS) for a, b in zip(ref, ref2)]


def trace_tolerances(ref, ref2):
    """Tolerance per iteration: the stated bar, or SENS_FACTOR x the distance between the two oracle runs where that is
    larger."""
    return [max(stated_bar(a["mu"], b["mu"]), SENS_FACTOR * s) for a, b, s in zip(ref, ref2, sensitivity(ref, ref2))]


def compare_traces_measured(tr, ref, ref2, name):
    """Device trace against the oracle's with the measured tolerances."""
    assert len(tr) == len(ref), f"{name}: iteration count {len(tr) - 1} vs {len(ref) - 1}"
    for t, r, tol in zip(tr, ref, trace_tolerances(ref, ref2)):
        for key in TRACE_KEYS:
            assert close(t[key], r[key], tol), f"{name}: iter {t['k']} {key}: {t[key]!r} vs {r[key]!r} (tolerance {tol:.1e})"


def worst_residual(t):
    return max(t["inf_pr"], t["inf_du"], t["inf_compl"])


def iteration_parity(r, ref, tol, what="", lp=False):
    """'equal' when the iteration counts agree, 'tie' for a threshold tie (see module docstring) -- on an LP only;
    raises otherwise."""
    if r["iter"] == ref["iter"]:
        return "equal"
    assert lp, (what, "iteration counts differ on a problem with a Hessian: no tie allowance", r["iter"], ref["iter"])
    assert abs(r["iter"] - ref["iter"]) == 1, (what, "iteration counts", r["iter"], ref["iter"])
    longer, shorter = (r, ref) if r["iter"] > ref["iter"] else (ref, r)
    k = shorter["iter"]
    wl, ws = worst_residual(longer["trace"][k]), worst_residual(shorter["trace"][k])
    assert ws <= tol < wl <= TIE_FACTOR * tol, (what, "not a threshold tie", r["iter"], ref["iter"], ws, wl, tol)
    return "tie"
