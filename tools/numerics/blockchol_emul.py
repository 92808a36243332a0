#!/usr/bin/env python3
"""Which use of the explicit 128 x 128 block inverses carries the device's distance from LAPACK?  

A numpy emulation of the device's blocked Cholesky (csrc/chol.hip: 128-column blocks, panel solve and the diagonal step
of both sweeps as products with W = L_kk^-1) is put behind the ORACLE's condensed KKT system, with each use switchable:

    panel = "inv"   L[below, k] = C[below, k] W_k'            (what chol.hip did up to round 3)
            "sub"   substitution with L_kk                     (what LAPACK's dtrsm does)
            "ref"   X0 = C W', X1 = X0 + (C - X0 L_kk') W'     (one refinement step with the same inverse)
    sweep = "inv" | "sub" | "ref"                              (the same three for y_r = L_rr^-1 v)

and the resulting solves are compared with the LAPACK oracle in units of the distance between two CPU runs of the oracle
(LAPACK with / without one refinement step per solve: tests/parity.py).  Test tooling only; imports oracle/."""
import argparse
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mpc  # noqa: E402
from oracle import qp as Q  # noqa: E402

NB = 128
KEYS = ("alpha_p", "alpha_d", "inf_pr", "inf_du", "inf_compl", "mu")


class ReversedLapack:
    """LAPACK on the symmetrically permuted system (variables in reverse order): the same algorithm, other roundings."""

    def __init__(self, K):
        self.c = sla.cho_factor(K[::-1, ::-1].copy(), lower=True)

    def solve(self, rhs):
        return sla.cho_solve(self.c, rhs[::-1].copy())[::-1].copy()


class BlockChol:
    def __init__(self, K, panel, sweep, nb=NB):
        n = K.shape[0]
        self.n, self.panel, self.sweep = n, panel, sweep
        L = np.tril(K).copy()
        self.blocks = [(j, min(n, j + nb)) for j in range(0, n, nb)]
        self.W = []
        for j0, j1 in self.blocks:
            if j0:
                L[j0:, j0:j1] -= L[j0:, :j0] @ L[j0:j1, :j0].T
            Ljj = np.linalg.cholesky(L[j0:j1, j0:j1])  # raises LinAlgError like the oracle's cho_factor
            L[j0:j1, j0:j1] = Ljj
            W = sla.solve_triangular(Ljj, np.eye(j1 - j0), lower=True)
            if "x" in panel + sweep:  # the correctly rounded inverse (long double substitution): what remains is the
                Ll = Ljj.astype(np.longdouble)  # instability of multiplying by ANY stored inverse
                Wl = np.eye(j1 - j0, dtype=np.longdouble)
                for i in range(j1 - j0):
                    Wl[i, :] = (Wl[i, :] - Ll[i, :i] @ Wl[:i, :]) / Ll[i, i]
                W = Wl.astype(float)
            if "ns" in panel + sweep:  # one Newton-Schulz step in fp64: W <- W + W (I - L W)
                W = W + W @ (np.eye(j1 - j0) - Ljj @ W)
                W = np.tril(W)
            self.W.append(W)
            if j1 < n:
                C = L[j1:, j0:j1]
                L[j1:, j0:j1] = self._rsolve(C, Ljj, W, panel)
        self.L = np.tril(L)

    @staticmethod
    def _rsolve(C, Ljj, W, how):  # X Ljj' = C
        if how == "sub":
            return sla.solve_triangular(Ljj, C.T, lower=True).T
        if how.startswith("sub") and how[3:].isdigit():  # block substitution with explicit inverses of g x g diagonal sub-blocks
            g, w = int(how[3:]), Ljj.shape[0]
            X = np.zeros_like(C)
            for j0 in range(0, w, g):
                j1 = min(w, j0 + g)
                Wj = sla.solve_triangular(Ljj[j0:j1, j0:j1], np.eye(j1 - j0), lower=True)
                X[:, j0:j1] = (C[:, j0:j1] - X[:, :j0] @ Ljj[j0:j1, :j0].T) @ Wj.T
            return X
        X = C @ W.T
        if how.startswith("ref"):
            X = X + (C - X @ Ljj.T) @ W.T
        return X

    def _dsolve(self, b, v, trans):
        j0, j1 = self.blocks[b]
        Ljj, W = self.L[j0:j1, j0:j1], self.W[b]
        if self.sweep == "sub":
            return sla.solve_triangular(Ljj, v, lower=True, trans=1 if trans else 0)
        if self.sweep.startswith("sub") and self.sweep[3:].isdigit():
            g, w = int(self.sweep[3:]), Ljj.shape[0]
            x = np.zeros(w)
            blocks = [(a, min(w, a + g)) for a in range(0, w, g)]
            for a, b_ in (reversed(blocks) if trans else blocks):
                Wj = sla.solve_triangular(Ljj[a:b_, a:b_], np.eye(b_ - a), lower=True)
                if trans:
                    x[a:b_] = Wj.T @ (v[a:b_] - Ljj[b_:, a:b_].T @ x[b_:])
                else:
                    x[a:b_] = Wj @ (v[a:b_] - Ljj[a:b_, :a] @ x[:a])
            return x
        if self.sweep in ("nrm16", "dst16"):
            return self._nsolve(Ljj, v, trans, self.sweep == "nrm16")
        A, Wm = (Ljj.T, W.T) if trans else (Ljj, W)
        x = Wm @ v
        if self.sweep.startswith("ref"):
            x = x + Wm @ (v - A @ x)
        return x

    @staticmethod
    def _nsolve(Ljj, v, trans, proper, g=16):
        """UNIT block substitution with images normalised by the g x g diagonal inverses, formed once in fp64 (what a
        device kernel would store per 128-block): forward image Lt_IJ = L_IJ W_JJ (column-normalised: u_I = v_I - sum
        Lt_IJ u_J, z_I = W_II u_I), backward image Lh_IJ = W_II L_IJ (row-normalised: w_J = v_J - sum Lh_IJ' w_I,
        x_J = W_JJ' w_J).  proper=False ("dst16"): the forward sweep with the ROW-normalised image (z_I = W_II v_I - sum
        Lh_IJ z_J) and the backward sweep with the COLUMN-normalised one -- W distributed over the sum."""
        w = Ljj.shape[0]
        blocks = [(a, min(w, a + g)) for a in range(0, w, g)]
        Wd = [sla.solve_triangular(Ljj[a:b, a:b], np.eye(b - a), lower=True) for a, b in blocks]
        x = np.zeros(w)
        col = (not trans) == proper  # which normalisation this sweep uses
        nb_ = len(blocks)
        img = {}
        for I, (a, b) in enumerate(blocks):
            for J, (c, d) in enumerate(blocks[:I]):
                img[I, J] = (Ljj[a:b, c:d] @ Wd[J]) if col else (Wd[I] @ Ljj[a:b, c:d])
        if not trans:
            for I, (a, b) in enumerate(blocks):
                if col:   # u_I = v_I - sum Lt_IJ u_J ; x holds u, scaled at the end of its step
                    x[a:b] = v[a:b] - sum((img[I, J] @ x[blocks[J][0]:blocks[J][1]] for J in range(I)), np.zeros(b - a))
                else:     # z_I = W_II v_I - sum Lh_IJ z_J
                    x[a:b] = Wd[I] @ v[a:b] - sum((img[I, J] @ x[blocks[J][0]:blocks[J][1]] for J in range(I)), np.zeros(b - a))
            if col:
                u = x.copy()
                for I, (a, b) in enumerate(blocks):
                    x[a:b] = Wd[I] @ u[a:b]
                # (the updates above must use u, not z: redo with u kept separately)
                u = np.zeros(w)
                for I, (a, b) in enumerate(blocks):
                    u[a:b] = v[a:b] - sum((img[I, J] @ u[blocks[J][0]:blocks[J][1]] for J in range(I)), np.zeros(b - a))
                    x[a:b] = Wd[I] @ u[a:b]
            return x
        wv = np.zeros(w)
        for J in range(nb_ - 1, -1, -1):
            c, d = blocks[J]
            if not col:   # row-normalised image, proper backward: w_J = v_J - sum Lh_IJ' w_I ; x_J = W_JJ' w_J
                wv[c:d] = v[c:d] - sum((img[I, J].T @ wv[blocks[I][0]:blocks[I][1]] for I in range(J + 1, nb_)), np.zeros(d - c))
                x[c:d] = Wd[J].T @ wv[c:d]
            else:         # column-normalised image: x_J = W_JJ' v_J - sum Lt_IJ' x_I
                x[c:d] = Wd[J].T @ v[c:d] - sum((img[I, J].T @ x[blocks[I][0]:blocks[I][1]] for I in range(J + 1, nb_)), np.zeros(d - c))
        return x

    def solve(self, rhs):
        L, y = self.L, rhs.astype(float).copy()
        for b, (j0, j1) in enumerate(self.blocks):
            v = y[j0:j1] - L[j0:j1, :j0] @ y[:j0]
            y[j0:j1] = self._dsolve(b, v, False)
        for b in range(len(self.blocks) - 1, -1, -1):
            j0, j1 = self.blocks[b]
            v = y[j0:j1] - L[j1:, j0:j1].T @ y[j1:]
            y[j0:j1] = self._dsolve(b, v, True)
        return y


class SlaProxy:
    """scipy.linalg for oracle/mpc.py with cho_factor / cho_solve swapped for the emulation."""

    def __init__(self, panel, sweep):
        self.panel, self.sweep = panel, sweep

    def __getattr__(self, name):
        return getattr(sla, name)

    def cho_factor(self, K, lower=True):
        if self.panel == "rev":
            return ReversedLapack(K)
        if self.panel.startswith("nb"):  # "nb48": substitution everywhere, blocks of 48
            return BlockChol(K, "sub", "sub", int(self.panel[2:]))
        return BlockChol(K, self.panel, self.sweep)

    def cho_solve(self, c, rhs):
        return c.solve(rhs)


def run(qp, variant=None, **opts):
    old = mpc.sla
    try:
        if variant is not None:
            mpc.sla = SlaProxy(*variant)
        return mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8), **opts)
    finally:
        mpc.sla = old


def floor_ensemble(qp, ref, opts):
    """per-iteration noise floor: the largest distance from the LAPACK run over an ensemble of equally valid CPU
    executions -- LAPACK + one refinement step per solve, LAPACK on the reversed variable order, a blocked Cholesky with
    substitution in blocks of 48"""
    runs = [run(qp, None, refine_steps=1, **opts), run(qp, ("rev", "-"), **opts), run(qp, ("nb48", "-"), **opts)]
    if os.environ.get("EMUL_ENSEMBLE5"):
        runs += [run(qp, ("nb80", "-"), **opts), run(qp, ("rev", "-"), refine_steps=1, **opts)]
    rel = lambda a, b: abs(a - b) / max(1.0, abs(a), abs(b))
    fl = [0.0] * len(ref["trace"])
    fx = 0.0
    bad = 0
    for e in runs:
        if e["iter"] != ref["iter"]:
            bad += 1
            continue
        for i, (a, b) in enumerate(zip(ref["trace"], e["trace"])):
            fl[i] = max(fl[i], max(rel(a[k], b[k]) for k in KEYS))
        fx = max(fx, float(np.max(np.abs(e["solution"] - ref["solution"]))))
    return fl, fx, bad


def ratio_to_floor(r, ref, fl, fx):
    if r["iter"] != ref["iter"]:
        return None
    rel = lambda a, b: abs(a - b) / max(1.0, abs(a), abs(b))
    worst = 0.0
    for t, a, f in zip(r["trace"], ref["trace"], fl):
        bar = 1e-9 if min(a["mu"], t["mu"]) >= 1e-4 else 1e-6
        d = max(rel(t[k], a[k]) for k in KEYS)
        if d > bar:
            worst = max(worst, d / max(f, 1e-300))
    dx = float(np.max(np.abs(r["solution"] - ref["solution"])))
    if dx > 1e-7:
        worst = max(worst, dx / max(fx, 1e-300))
    return worst


def distances(r, ref, ref2):
    """largest per-iteration trace distance and |dx| from `ref`, in units of the distance between ref and ref2"""
    if r["iter"] != ref["iter"] or ref2["iter"] != ref["iter"]:
        return dict(iters=(r["iter"], ref["iter"], ref2["iter"]))
    rel = lambda a, b: abs(a - b) / max(1.0, abs(a), abs(b))
    worst, where = 0.0, None
    for t, a, b in zip(r["trace"], ref["trace"], ref2["trace"]):  # tests/parity.py: per iteration, the largest over the keys
        bar = 1e-9 if min(a["mu"], b["mu"]) >= 1e-4 else 1e-6
        d = max(rel(t[k], a[k]) for k in KEYS)
        s = max(rel(a[k], b[k]) for k in KEYS)
        if d > bar and d / max(s, 1e-300) > worst:  # only where the stated bar is exceeded (else it is a pass anyway)
            worst, where = d / max(s, 1e-300), (t["k"], d, s)
    dx = float(np.max(np.abs(r["solution"] - ref["solution"])))
    sx = float(np.max(np.abs(ref2["solution"] - ref["solution"])))
    return dict(trace_ratio=worst, where=where, dx=dx, sens_dx=sx, dx_ratio=(dx / sx if dx > 1e-7 else 0.0))


def cases(which):
    if which in ("dist", "all"):
        n, m = 900, 350
        yield "qp_900_350", Q.synthetic_qp(20250614, n, m), {}
        yield "qp_gondzio", Q.synthetic_qp(77, n, m), dict(max_ncorr=3)
        yield "lp", Q.synthetic_qp(5, n, m, "lp"), {}
        eq = Q.synthetic_qp(9, n, m)
        eq.lcon[[3, 10, 200]] = eq.ucon[[3, 10, 200]] = 0.25
        yield "qp_eq", eq, {}
        big = Q.synthetic_qp(31, 700, 130)
        big.A[::3] *= 40.0
        big.lcon[::3] *= 40.0
        big.ucon[::3] *= 40.0
        yield "qp_scaled_rows", big, {}
    if which in ("soak", "all"):
        for seed0, count in ((31000, 400), (9000, 200)):
            rng = np.random.default_rng(seed0)
            for t in range(count):
                n = int(rng.integers(1, 260))
                m = int(rng.integers(0, max(1, n)))
                lp = bool(rng.integers(0, 4) == 0)
                if lp and n > NB:  # one block: the panel solve never runs
                    yield f"soak{seed0 + t}", Q.random_qp(seed0 + t, n, m, lp), {}


def small_cases(nmax=64):
    """the soak streams' problems of order <= nmax, LPs and QPs: the whole matrix is one or a few 16 x 16 sub-blocks, so
    the diagonal step's own arithmetic (inverse product against substitution INSIDE a sub-block) is what shows"""
    for seed0, count in ((9000, 200), (1000, 150)):
        rng = np.random.default_rng(seed0)
        for t in range(count):
            n = int(rng.integers(1, 260))
            m = int(rng.integers(0, max(1, n)))
            lp = bool(rng.integers(0, 4) == 0)
            if n <= nmax:
                yield f"soak{seed0 + t}", Q.random_qp(seed0 + t, n, m, lp), {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="dist", choices=("dist", "soak", "all", "small"))
    ap.add_argument("--ensemble", action="store_true", help="ratios against the three-execution noise floor")
    ap.add_argument("--variants", default="inv/inv,sub/inv,inv/sub,sub/sub,ref/inv,inv/ref,ref/ref,sub/ref")
    a = ap.parse_args()
    variants = [tuple(v.split("/")) for v in a.variants.split(",")]
    print("case".ljust(16) + "".join(f"{p + '/' + s:>22}" for p, s in variants) + "    (trace ratio | dx ratio | iterations if they differ)")
    worst = {v: 0.0 for v in variants}
    mism = {v: 0 for v in variants}
    for name, qp, opts in (small_cases() if a.cases == "small" else cases(a.cases)):
        ref, ref2 = run(qp, None, **opts), run(qp, None, refine_steps=1, **opts)
        row = name.ljust(16)
        if a.ensemble:
            fl, fx, bad = floor_ensemble(qp, ref, opts)
            for v in variants:
                q = ratio_to_floor(run(qp, v, **opts), ref, fl, fx)
                if q is None:
                    mism[v] += 1
                    row += f"{'it':>22}"
                else:
                    worst[v] = max(worst[v], q)
                    row += f"{q:>22.2f}"
            print(row + (f"   ({bad} ensemble runs stop elsewhere)" if bad else ""), flush=True)
            continue
        for v in variants:
            d = distances(run(qp, v, **opts), ref, ref2)
            if "iters" in d:
                row += f"{'it ' + str(d['iters']):>22}"
                mism[v] += d["iters"][0] != d["iters"][1]
            else:
                worst[v] = max(worst[v], d["trace_ratio"], d["dx_ratio"])
                row += f"{d['trace_ratio']:>12.1f} |{d['dx_ratio']:>7.1f}"
                if os.environ.get("EMUL_WHERE") and d["where"]:
                    print("   ", name, v, d["where"])
        print(row, flush=True)
    print("worst ratio".ljust(16) + "".join(f"{worst[v]:>22.1f}" for v in variants))
    print("iter mismatches".ljust(16) + "".join(f"{mism[v]:>22d}" for v in variants))


if __name__ == "__main__":
    main()
