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
by hand; 1e-12 everywhere before iteration 12, where the bar stays 1e-9).  (Round 3 allowed 16 x that two-run distance.)

Round 4 replaced the two-run distance -- ONE sample of the noise -- by the floor over an ENSEMBLE of valid executions
(below) and set SENS_FACTOR = 4.  Round 5: the device's sweeps substitute (unit block substitution over 16 x 16
sub-blocks, csrc/chol.hip) instead of multiplying with stored 128 x 128 inverses, which is what had put the
per-iteration traces of ill-conditioned problems tens of times above the floor (profiles/r04_parity_ratios_default.json
against profiles/r05_parity_ratios_*.json); the soak now compares traces too (tests/test_gpu_soak.py).
"""
TIE_FACTOR = 4.0
SENS_FACTOR = 4.0  # round 4: times the ENSEMBLE floor (below); round 3 had 16 x a two-run distance
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


# ---- round 4: the noise floor from an ENSEMBLE of equally valid CPU executions ---------------------------------------
# Two runs (LAPACK with / without a refinement step) give ONE sample of the rounding noise of a problem; on the condensed
# LPs of the soak that single sample is, per case, anywhere between the typical distance of two valid executions and a
# twentieth of it -- tools/numerics/blockchol_emul.py: a blocked Cholesky that SUBSTITUTES everywhere (no inverse at all)
# sits beyond 4 x that sample on half of such LPs.  A multiplier large enough for the unlucky samples (16 in round 3) is
# then a hand-set tolerance again.  The floor is therefore taken over several executions of the oracle that differ only in
# how the dense solves round:
#   "refine"      LAPACK + one step of iterative refinement per solve          (round 3's second run)
#   "reverse"     LAPACK on the symmetrically permuted system (variables in reverse order)
#   "blocked48"   a blocked left-looking Cholesky with substitution, blocks of 48 (numpy; another summation order)
#   "blocked80"   the same in blocks of 80
#   "reverse+refine"
# floor = the largest distance of any of them from the LAPACK run, per iteration / for the solution / objective;
# the device is allowed SENS_FACTOR times the floor where that exceeds the stated bar.
class _ReversedLapack:
    def __init__(self, K):
        import scipy.linalg as sla

        self.sla = sla
        self.c = sla.cho_factor(K[::-1, ::-1].copy(), lower=True)

    def solve(self, rhs):
        return self.sla.cho_solve(self.c, rhs[::-1].copy())[::-1].copy()


class _BlockedSubstitution:
    def __init__(self, K, nb):
        import numpy as np
        import scipy.linalg as sla

        n = K.shape[0]
        self.np, self.sla = np, sla
        L = np.tril(K).copy()
        self.blocks = [(j, min(n, j + nb)) for j in range(0, n, nb)]
        for j0, j1 in self.blocks:
            if j0:
                L[j0:, j0:j1] -= L[j0:, :j0] @ L[j0:j1, :j0].T
            Ljj = np.linalg.cholesky(L[j0:j1, j0:j1])  # LinAlgError as scipy's cho_factor
            L[j0:j1, j0:j1] = Ljj
            if j1 < n:
                L[j1:, j0:j1] = sla.solve_triangular(Ljj, L[j1:, j0:j1].T, lower=True).T
        self.L = np.tril(L)

    def solve(self, rhs):
        L, sla = self.L, self.sla
        y = rhs.astype(float).copy()
        for j0, j1 in self.blocks:
            y[j0:j1] = sla.solve_triangular(L[j0:j1, j0:j1], y[j0:j1] - L[j0:j1, :j0] @ y[:j0], lower=True)
        for j0, j1 in reversed(self.blocks):
            y[j0:j1] = sla.solve_triangular(L[j0:j1, j0:j1], y[j0:j1] - L[j1:, j0:j1].T @ y[j1:], lower=True, trans=1)
        return y


class _SlaProxy:
    """scipy.linalg as oracle/mpc.py sees it, with the Cholesky pair swapped for another valid execution."""

    def __init__(self, kind):
        import scipy.linalg as sla

        self._sla, self.kind = sla, kind

    def __getattr__(self, name):
        return getattr(self._sla, name)

    def cho_factor(self, K, lower=True):
        if self.kind == "reverse":
            return _ReversedLapack(K)
        return _BlockedSubstitution(K, int(self.kind[7:]))

    def cho_solve(self, c, rhs):
        return c.solve(rhs)


ENSEMBLE = ("refine", "reverse", "blocked48", "blocked80", "reverse+refine")


def oracle_execution(qp, kind, **opts):
    """One run of the oracle's condensed path with its dense solves executed as `kind` (None: LAPACK, the reference)."""
    from oracle import mpc

    opts = dict(opts)
    if kind and kind.endswith("refine"):
        opts["refine_steps"] = 1
        kind = kind[:-len("refine")].rstrip("+") or None
    old = mpc.sla
    try:
        if kind:
            mpc.sla = _SlaProxy(kind)
        return mpc.solve(qp, kkt_system="condensed", **opts)
    finally:
        mpc.sla = old


def rel_dist(a, b):
    return abs(a - b) / max(1.0, abs(a), abs(b))


def ensemble_floor(qp, ref, members=ENSEMBLE, **opts):
    """dict(trace=[per iteration], dx, dy, obj, stopped_elsewhere): the largest distance of any member from `ref`; a
    member that stops at another iteration says the problem is a threshold tie for the oracle itself and is only counted."""
    import numpy as np

    fl = dict(trace=[0.0] * len(ref["trace"]), dx=0.0, dy=0.0, obj=0.0, stopped_elsewhere=0, members=len(members))
    for kind in members:
        e = oracle_execution(qp, kind, **opts)
        if e["iter"] != ref["iter"] or e["status"] != ref["status"]:
            fl["stopped_elsewhere"] += 1
            continue
        for i, (a, b) in enumerate(zip(ref["trace"], e["trace"])):
            fl["trace"][i] = max(fl["trace"][i], max(rel_dist(a[k], b[k]) for k in TRACE_KEYS))
        fl["dx"] = max(fl["dx"], float(np.max(np.abs(e["solution"] - ref["solution"]), initial=0.0)))
        fl["dy"] = max(fl["dy"], float(np.max(np.abs(e["multipliers"] - ref["multipliers"]), initial=0.0)))
        fl["obj"] = max(fl["obj"], rel_dist(e["objective"], ref["objective"]))
    return fl


def ratios_to_floor(r, ref, fl):
    """How far a result is from `ref` in units of the floor, where it exceeds the stated bar at all (else 0):
    dict(trace, dx, obj) -- None when the iteration counts differ."""
    import numpy as np

    if r["iter"] != ref["iter"]:
        return None
    out = dict(trace=0.0, dx=0.0, obj=0.0)
    for t, a, f in zip(r["trace"], ref["trace"], fl["trace"]):
        d = max(rel_dist(t[k], a[k]) for k in TRACE_KEYS)
        if d > stated_bar(t["mu"], a["mu"]):
            out["trace"] = max(out["trace"], d / max(f, 1e-300))
    dx = float(np.max(np.abs(r["solution"] - ref["solution"]), initial=0.0))
    if dx > 1e-7:
        out["dx"] = dx / max(fl["dx"], 1e-300)
    do = rel_dist(r["objective"], ref["objective"])
    if do > 1e-9:
        out["obj"] = do / max(fl["obj"], 1e-300)
    return out


def exceeds_stated_bar(r, ref, trace=True, multipliers=False):
    """True when `r` is further from `ref` than the stated bar somewhere (per-iteration quantities 1e-9 while
    mu >= 1e-4 and 1e-6 after, objective 1e-9 relative, |dx| 1e-7, multipliers 1e-6)."""
    import numpy as np

    if trace:
        for t, a in zip(r["trace"], ref["trace"]):
            if max(rel_dist(t[k], a[k]) for k in TRACE_KEYS) > stated_bar(t["mu"], a["mu"]):
                return True
    if rel_dist(r["objective"], ref["objective"]) > 1e-9:
        return True
    if float(np.max(np.abs(np.asarray(r["solution"]) - np.asarray(ref["solution"])), initial=0.0)) > 1e-7:
        return True
    if multipliers and float(np.max(np.abs(np.asarray(r["multipliers"]) - np.asarray(ref["multipliers"])), initial=0.0)) > 1e-6:
        return True
    return False


def assert_parity(r, ref, qp, what, trace=True, multipliers=False, floor=None, **oracle_opts):
    """The acceptance rule of the whole-solve parity tests for a result with the oracle's iteration count: the stated bar --
    or, where the problem's conditioning does not support it, SENS_FACTOR x the ensemble floor (computed only then: five
    more oracle runs).  Returns "bar" or "floor"; `floor`: a precomputed ensemble_floor."""
    import numpy as np

    assert r["iter"] == ref["iter"], (what, "iteration counts", r["iter"], ref["iter"])
    if not exceeds_stated_bar(r, ref, trace, multipliers):
        return "bar"
    fl = floor if floor is not None else ensemble_floor(qp, ref, **oracle_opts)
    if trace:
        for t, a, f in zip(r["trace"], ref["trace"], fl["trace"]):
            tol = max(stated_bar(t["mu"], a["mu"]), SENS_FACTOR * f)
            for key in TRACE_KEYS:
                assert close(t[key], a[key], tol), f"{what}: iter {t['k']} {key}: {t[key]!r} vs {a[key]!r} (tolerance {tol:.1e}, floor {f:.1e})"
    do = rel_dist(r["objective"], ref["objective"])
    assert do <= max(1e-9, SENS_FACTOR * fl["obj"]), (what, "objective", do, fl["obj"])
    dx = float(np.max(np.abs(np.asarray(r["solution"]) - np.asarray(ref["solution"])), initial=0.0))
    assert dx <= max(1e-7, SENS_FACTOR * fl["dx"]), (what, "solution", dx, fl["dx"])
    if multipliers:
        dy = float(np.max(np.abs(np.asarray(r["multipliers"]) - np.asarray(ref["multipliers"])), initial=0.0))
        assert dy <= max(1e-6, SENS_FACTOR * fl["dy"]), (what, "multipliers", dy, fl["dy"])
    return "floor"
