#!/usr/bin/env python3
"""One-off soak: many random bound patterns / shapes through the per-problem drivers, the sparse front end and
the batched engine, each against the CPU oracle.  python tests/soak_random.py --count 150"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--count", type=int, default=100)
    p.add_argument("--seed0", type=int, default=1000)
    p.add_argument("--nmin", type=int, default=1)
    p.add_argument("--nmax", type=int, default=260, help="orders are drawn from [nmin, nmax): beyond 128 the mid-size "
                   "factorisation with its sweep on U = L' runs, one more 128-block per 128 variables")
    p.add_argument("--mode", choices=("drivers", "sparse", "augmented"), default="drivers",
                   help="drivers: python / native / batched on dense data; sparse: CSR front end, condensed and (LPs) "
                        "normal equations, against the dense path's oracle; augmented: the K2 system (dense and CSR Jacobian) with "
                        "the reference's default regularization against the oracle's K2 path")
    a = p.parse_args()
    import madqp_jl_amd as M
    from oracle import mpc
    from oracle import qp as Q

    be = M.HipBackend(0)
    REG, OREG = M.FixedRegularization(1e-8, -1e-8), mpc.FixedRegularization(1e-8, -1e-8)
    rng = np.random.default_rng(a.seed0)
    bad = 0
    for t in range(a.count):
        seed = a.seed0 + t
        n = int(rng.integers(a.nmin, a.nmax))
        m = int(rng.integers(0, max(1, n)))
        lp = bool(rng.integers(0, 4) == 0)
        qp = Q.random_qp(seed, n, m, lp)
        ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG)
        dq = M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
        outs = {}
        if a.mode == "sparse":
            if m == 0 or n < 2:
                continue
            keep = rng.random(qp.A.shape) < min(1.0, 6.0 / n)  # ~6 entries per row, feasibility kept by re-centring
            keep[np.arange(m), rng.integers(0, n, m)] = True
            xf = rng.uniform(-0.5, 0.5, n)
            xf = np.clip(xf, np.where(np.isfinite(qp.lvar), qp.lvar + 0.1, -np.inf), np.where(np.isfinite(qp.uvar), qp.uvar - 0.1, np.inf))
            old = qp.A @ xf
            qp.A = np.where(keep, qp.A, 0.0)
            shift = qp.A @ xf - old
            qp.lcon, qp.ucon = qp.lcon + shift, qp.ucon + shift
            ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG)
            ds = M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0, sparse=True)
            cands = [("sparse-condensed", lambda: M.MPCSolver(ds, be, regularization=REG, driver="native"))]
        elif a.mode == "augmented":
            ref = mpc.solve(qp, kkt_system="K2")
            cands = [("augmented-python", lambda: M.MPCSolver(dq, be, kkt_system="augmented")),
                     ("augmented-native", lambda: M.MPCSolver(dq, be, kkt_system="augmented", driver="native"))]
            if m > 0:
                ds = M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0,
                                           qp.c0, sparse=True)
                cands.append(("augmented-csr", lambda: M.MPCSolver(ds, be, kkt_system="augmented", driver="native")))
        else:
            cands = [("python", lambda: M.MPCSolver(dq, be, regularization=REG)),
                     ("native", lambda: M.MPCSolver(dq, be, regularization=REG, driver="native")),
                     ("batched", lambda: M.BatchedMPCSolver([dq], be, regularization=REG))]
        for name, mk in cands:
            s = mk()
            r = s.solve()
            outs[name] = r[0] if name == "batched" else r
            s.close()
        for name, r in outs.items():
            ok = (r["status"] == ref["status"] and (r["status"] != 1 or (
                r["iter"] == ref["iter"] and abs(r["objective"] - ref["objective"]) <= 1e-7 * max(1, abs(ref["objective"]))
                and np.max(np.abs(r["solution"] - ref["solution"]), initial=0) <= 1e-5)))
            if not ok:
                bad += 1
                print("MISMATCH", seed, n, m, lp, name, r["status"], ref["status"], r["iter"], ref["iter"],
                      r["objective"], ref["objective"], flush=True)
        if (t + 1) % 25 == 0:
            print(f"{t + 1} problems, {bad} mismatches", flush=True)
    print(f"done: {a.count} problems, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
