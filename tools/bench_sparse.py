#!/usr/bin/env python3
"""Sparse-A front end on a synthetic sparse LP (normal equations) or QP (condensed): time per IPM iteration
and the per-class split.  python tools/bench_sparse.py --nx 60000 --ncon 20000 --per-row 8 --kkt normal"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--nx", type=int, default=60000)
    p.add_argument("--ncon", type=int, default=20000)
    p.add_argument("--per-row", type=int, default=8)
    p.add_argument("--kkt", choices=("normal", "condensed"), default="normal")
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--cont", type=int, default=0, help="N > 0: the CONT-type boundary-control QP on an N x N grid "
                   "(preprocess.boundary_control_qp; N = 300 has the shape of Maros-Meszaros CONT-300), solved to the end")
    a = p.parse_args()
    import torch

    import madqp_jl_amd as M

    be = M.HipBackend(0)
    if a.cont:
        from madqp_jl_amd import preprocess as P

        t0 = time.perf_counter()
        h = P.boundary_control_qp(a.cont)
        qp = P.to_device(h, be)
        t_gen = time.perf_counter() - t0
        s = M.MPCSolver(qp, be, kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0), driver="native",
                        max_iter=100)
        be.prof_enable(M._lib.PROF_CLASSES)
        be.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = s.solve()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = be.prof_get()
        nf = max(r["n_factorizations"], 1)
        print(json.dumps({"workload": f"boundary-control QP N={a.cont}: nx={h.nvar} m={h.ncon} nnz={h.nnzj}, diagonal H, "
                                      "normal equations (m x m dense)", "status": r["status"], "iterations": r["iter"],
                          "objective": r["objective"], "solve_seconds": dt, "generate_seconds": t_gen,
                          "factorizations": r["n_factorizations"],
                          "ms_per_factorization": {c: round(v[0] / nf, 2) for c, v in prof.items() if v[1]},
                          "last": r["trace"][-1]}))
        s.close()
        return
    rng = np.random.default_rng(7)
    m, n, k = a.ncon, a.nx, a.per_row
    rows = np.repeat(np.arange(m), k + 1)
    cols = np.concatenate([rng.integers(0, n, size=(m, k)), (np.arange(m) % n)[:, None]], axis=1).ravel()
    key = np.unique(rows * n + cols)  # drop duplicates
    rows, cols = key // n, key % n
    csr = M.DeviceCSR(be.device, m, n, rows, cols, rng.standard_normal(len(rows)))
    f64 = dict(dtype=torch.float64, device=be.device)
    z = lambda c, v: torch.full((c,), v, **f64)
    q = torch.as_tensor(rng.standard_normal(n), **f64)
    H = None
    if a.kkt == "condensed":
        H = torch.empty((n, n), **f64)
        be.gen_wigner(M.stream_key(11, 2), n, 1.0 / np.sqrt(n), H)
    qp = M.DeviceQP(H, q, csr, z(n, 0.0), z(n, 1.0), z(m, 0.0), z(m, 1.0), z(n, 0.0))
    reg = M.FixedRegularization(1e-8, 0.0 if a.kkt == "normal" else -1e-8)
    s = M.MPCSolver(qp, be, kkt_system=a.kkt, regularization=reg, driver="native", max_iter=300)
    s.initialize()
    s.iteration_head()
    s.iteration_body()
    be.prof_enable(M._lib.PROF_CLASSES)
    be.prof_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for _ in range(a.steps):
        if s.iteration_head() is not None:
            break
        s.iteration_body()
        done += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = be.prof_get()
    print(json.dumps({"workload": f"sparse {a.kkt} nx={n} m={m} nnz={csr.nnz}", "ms_per_iteration": dt / max(done, 1) * 1e3,
                      "iterations": done, "split_ms_per_iteration": {c: round(v[0] / max(done, 1), 3) for c, v in prof.items() if v[1]},
                      "last": s.trace[-1] if s.trace else None}))
    s.close()


if __name__ == "__main__":
    main()
