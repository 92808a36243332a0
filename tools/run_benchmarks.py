#!/usr/bin/env python3
"""The reference's benchmark loop (scripts/benchmarks_cpu.jl:10-62, scripts/benchmarks_gpu.jl:11-67) over a
directory of MPS / QPS / SIF instances with the HIP solver in the middle:

    import_mps -> presolve_qp -> scale_qp -> [standard_form_qp] -> MPCSolver(...) -> solve! -> 9 numbers per instance

Options of the solver call as in the scripts (max_iter 300, max_ncorr 3, scaling, AdaptiveStep(0.995),
FixedRegularization(1e-8, -1e-8), mu_min 1e-12).  The KKT system: the scripts' NormalKKTSystem for LPs and
diagonal-Hessian QPs, the condensed system otherwise (``--kkt-system`` overrides).  Writes the table the scripts write
(nvar ncon nnzj nnzh status iter objective total_time linear_solver_time) next to the instance names.

    python tools/run_benchmarks.py DIR [--reformulate] [--out results.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXT = (".mps", ".qps", ".sif", ".mps.gz", ".qps.gz", ".sif.gz")


def run(directory, reformulate=False, kkt_system=None, out=None, backend=None, verbose=True):
    import madqp_jl_amd as M
    from madqp_jl_amd import preprocess as P

    be = backend or M.HipBackend(0)
    names = sorted(f for f in os.listdir(directory) if f.lower().endswith(EXT))
    results = np.zeros((len(names), 9))
    for k, fname in enumerate(names):
        try:
            qp = P.read_qps(os.path.join(directory, fname))  # import_mps, scripts/common.jl:21-36
        except Exception as e:  # benchmarks_cpu.jl:19-24
            print(f"Failed to import {fname}: {e}", file=sys.stderr)
            continue
        ps = P.presolve(qp)  # :27
        if not ps.flag:  # :28  problem already solved, unbounded or infeasible
            if verbose:
                print(f"{fname}: presolve -> {ps.status}")
            continue
        scaled, Dr, Dc = P.ruiz_scale(ps.qp)  # :29
        model = P.standard_form(scaled) if reformulate else scaled  # :30
        diag_or_lp = model.H.nnz == 0 or np.all(model.H.tocoo().row == model.H.tocoo().col)
        ksys = kkt_system or ("normal" if diag_or_lp and np.all(model.lvar < model.uvar) else "condensed")
        try:
            lin_classes = ("potrf_gemm", "potrf_trsm", "potrf_diag", "trsv")  # factorize! + solve! of the linear solver
            be.prof_enable(lin_classes)
            be.prof_reset()
            t0 = time.perf_counter()
            s = M.MPCSolver(P.to_device(model, be), be, max_iter=300, max_ncorr=3, scaling=True,
                            step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8),
                            kkt_system=ksys, rethrow_error=True, mu_min=1e-12, driver="native")  # :33-45
            r = s.solve()
            total = time.perf_counter() - t0
            prof = be.prof_get()
            be.prof_enable(())
            lin = sum(prof[c][0] for c in lin_classes) * 1e-3  # counters.linear_solver_time, device seconds
            s.close()
            results[k] = P.benchmark_row(model, r, r.get("total_time", total), lin)  # :47-55
            if verbose:
                print(f"{fname}: n={model.nvar} m={model.ncon} {ksys} status={int(r['status'])} iter={r['iter']} "
                      f"obj={r['objective']:.10e} time={r.get('total_time', total):.3f}s", flush=True)
        except Exception as e:  # :56-60
            results[k, 7] = -1
            print(f"Failed to solve {fname}: {type(e).__name__}: {e}", file=sys.stderr)
    if out:
        with open(out, "w") as fh:  # writedlm of [names results], scripts/benchmarks_cpu.jl:91-93
            for name, row in zip(names, results):
                fh.write(name + "\t" + "\t".join(repr(float(v)) for v in row) + "\n")
    return names, results


def main():
    p = argparse.ArgumentParser()
    p.add_argument("directory")
    p.add_argument("--reformulate", action="store_true", help="standard_form_qp before the solve")
    p.add_argument("--kkt-system", choices=("condensed", "normal", "augmented"), default=None)
    p.add_argument("--out", default=None)
    a = p.parse_args()
    run(a.directory, a.reformulate, a.kkt_system, a.out)


if __name__ == "__main__":
    main()
