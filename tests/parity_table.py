#!/usr/bin/env python3
"""Measured device / CPU-distance ratios, case by case ("commit a table").  TEST TOOLING (imports
oracle/): every whole-solve parity case of the GPU suite -- the five problems of tests/dist2d_kkt_worker.py on one GPU and
the soak streams of tests/test_gpu_soak.py -- through the native driver, against the LAPACK oracle, in units of
  (a) the distance between the oracle with and without one refinement step (round 3's sensitivity), and
  (b) the ensemble floor of tests/parity.py (five valid CPU executions).
    python tests/parity_table.py --out profiles/r04_parity_ratios.json [--soak-count 120] [--refine 0,1]
Ratios are only formed where the stated bar (1e-9 / 1e-6 per iteration, 1e-7 in x, 1e-9 in the objective) is exceeded."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity  # noqa: E402
from oracle import mpc  # noqa: E402
from oracle import qp as Q  # noqa: E402

OREG = mpc.FixedRegularization(1e-8, -1e-8)


def make_cases(soak_count):
    n, m = 900, 350
    yield "qp_900_350", ("synthetic", 20250614, n, m, None), {}
    yield "qp_gondzio", ("synthetic", 77, n, m, None), dict(max_ncorr=3)
    yield "lp", ("synthetic", 5, n, m, "lp"), {}
    yield "qp_eq", ("eq", 9, n, m, None), {}
    yield "qp_scaled_rows", ("scaled", 31, 700, 130, None), {}
    for seed0, count, only_lp in ((31000, 400, True), (9000, 200, False), (1000, 150, False)):
        rng = np.random.default_rng(seed0)
        k = 0
        for t in range(count):
            nn = int(rng.integers(1, 260))
            mm = int(rng.integers(0, max(1, nn)))
            lp = bool(rng.integers(0, 4) == 0)
            if (lp or not only_lp) and k < soak_count:
                k += 1
                yield f"soak{seed0 + t}{'_lp' if lp else ''}", ("random", seed0 + t, nn, mm, lp), {}


def build(spec):
    kind, seed, n, m, extra = spec
    if kind == "synthetic":
        return Q.synthetic_qp(seed, n, m, *((extra,) if extra else ()))
    if kind == "eq":
        qp = Q.synthetic_qp(seed, n, m)
        qp.lcon[[3, 10, 200]] = qp.ucon[[3, 10, 200]] = 0.25
        return qp
    if kind == "scaled":
        qp = Q.synthetic_qp(seed, n, m)
        qp.A[::3] *= 40.0
        qp.lcon[::3] *= 40.0
        qp.ucon[::3] *= 40.0
        return qp
    return Q.random_qp(seed, n, m, extra)


def slim(r):
    keys = ("k",) + parity.TRACE_KEYS
    return dict(status=int(r["status"]), iter=int(r["iter"]), objective=float(r["objective"]),
                solution=np.asarray(r["solution"]), multipliers=np.asarray(r["multipliers"]),
                trace=[{k: float(t[k]) for k in keys} for t in r["trace"]])


def cpu_side(item):
    """(worker process) the oracle's reference run, the two-run sensitivity and the ensemble floor of one case"""
    name, spec, opts = item
    qp = build(spec)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG, **opts)
    if ref["status"] != 1:
        return name, None
    fl = parity.ensemble_floor(qp, ref, regularization=OREG, **opts)
    fl2 = parity.ensemble_floor(qp, ref, members=("refine",), regularization=OREG, **opts)
    return name, dict(ref=slim(ref), floor=fl, floor2=fl2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--soak-count", type=int, default=1000, help="cases per soak stream")
    ap.add_argument("--refine", default="auto", help="device refine_steps settings to run, comma separated (auto = the default rule)")
    ap.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    a = ap.parse_args()
    import madqp_jl_amd as M

    cases = list(make_cases(a.soak_count))
    t0 = time.time()
    cpu = {}
    with ProcessPoolExecutor(a.workers) as ex:  # (started before this process touches the GPU)
        from concurrent.futures import as_completed

        futs = [ex.submit(cpu_side, c) for c in cases]  # (the five large cases come first in the list)
        for k, f in enumerate(as_completed(futs)):
            name, rec = f.result()
            cpu[name] = rec
            if k % 20 == 0:
                print(f"[parity_table] CPU side {k + 1}/{len(cases)} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    print(f"[parity_table] {len(cases)} cases, CPU side {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    be = M.HipBackend(0)
    REG = M.FixedRegularization(1e-8, -1e-8)
    rows = []
    for kk, (name, spec, opts) in enumerate(cases):
        c = cpu[name]
        if c is None:
            continue
        if kk % 50 == 0:
            print(f"[parity_table] device side {kk}/{len(cases)} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
        qp = build(spec)
        dq = M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
        row = dict(case=name, n=int(qp.nvar), m=int(qp.ncon), lp=qp.H is None or not np.any(qp.H), iter_ref=c["ref"]["iter"],
                   floor_dx=c["floor"]["dx"], floor2_dx=c["floor2"]["dx"],
                   ensemble_stopped_elsewhere=c["floor"]["stopped_elsewhere"])
        for rs in a.refine.split(","):  # "auto": the library's default rule (options.py: one step up to order 1024)
            s = M.MPCSolver(dq, be, regularization=REG, driver="native", refine_steps=None if rs == "auto" else int(rs), **opts)
            r = s.solve()
            s.close()
            key = f"refine{rs}"
            if r["status"] != c["ref"]["status"]:
                row[key] = dict(status=int(r["status"]))
                continue
            q5, q2 = parity.ratios_to_floor(r, c["ref"], c["floor"]), parity.ratios_to_floor(r, c["ref"], c["floor2"])
            row[key] = dict(iter=int(r["iter"]), vs_ensemble=q5, vs_two_runs=q2)
        rows.append(row)
    be.close()

    def summary(key, which):
        v = [max(r[key][which].values()) for r in rows if key in r and r[key].get(which)]
        mism = sum(1 for r in rows if key in r and "iter" in r[key] and r[key]["iter"] != r["iter_ref"])
        v = np.array(v) if v else np.zeros(1)
        return dict(cases=int(len(v)), iteration_mismatches=mism, worst=float(v.max()),
                    **{f"over_{t}": int((v > t).sum()) for t in (1, 2, 4, 8, 16)})

    out = dict(what="device (native driver) distance from the LAPACK oracle in units of the CPU noise floor; 0 = within the "
                    "stated bar", library_panel=os.environ.get("MADQP_CHOL_PANEL", "sub16"),
               library_sweep_diag=os.environ.get("MADQP_SWEEP_DIAG", "nrm16"),
               summary={f"refine{rs}": dict(vs_ensemble=summary(f"refine{rs}", "vs_ensemble"),
                                            vs_two_runs=summary(f"refine{rs}", "vs_two_runs"))
                        for rs in a.refine.split(",")},
               rows=rows)
    txt = json.dumps(out, indent=1, default=float)
    if a.out:
        open(a.out, "w").write(txt)
    print(json.dumps(out["summary"], indent=1))


if __name__ == "__main__":
    main()
