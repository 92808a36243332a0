#!/usr/bin/env python3
"""Runs every per-variable kernel of the path (src/kernels.jl) REPS times on a
state of BASELINE configs[4] size -- n = nx + m = 140 000 primal entries, m = 40 000 rows, nlb = nub = n (the synthetic
family: every variable and slack two-sided) -- so that rocprofv3 can count their HBM bytes (tools/runs/r3_vec_pmc.sh:
one --kernel-trace --stats pass, one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass; summarised by
tools/summarize_vec_pmc.py).  No KKT matrix is needed: the kernels see the iterate vectors only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import madqp_jl_amd as M  # noqa: E402

REPS = int(os.environ.get("VEC_PMC_REPS", "20"))


def main():
    nx, m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    n = nx + m
    be = M.HipBackend(0)
    rng = np.random.default_rng(7)
    st = be.new_state(n, m, np.arange(n), np.arange(n))
    xl = rng.uniform(-2, 0, n)
    xu = xl + rng.uniform(0.5, 3, n)
    vals = dict(xl=xl, xu=xu, x=xl + (xu - xl) * rng.uniform(0.01, 0.99, n), zl=rng.uniform(0.01, 2, n),
                zu=rng.uniform(0.01, 2, n), f=rng.standard_normal(n), jacl=rng.standard_normal(n),
                y=rng.standard_normal(m), c=rng.standard_normal(m), rhs=rng.standard_normal(m),
                d=1e-3 * rng.standard_normal(st.ntot), p=rng.standard_normal(st.ntot), w1=rng.standard_normal(st.ntot),
                w2=rng.standard_normal(st.ntot))
    for k, v in vals.items():
        getattr(st, k).copy_(torch.as_tensor(v))
    for _ in range(REPS):
        be.set_aug_diagonal_reg(st, 1e-8, -1e-8)  # aug_diag_fill / _lb / _ub            (row 1)
        be.set_predictive_rhs(st)                   # rhs_kernel mode 0                     (row 4)
        be.get_correction(st)                       # correction_kernel                     (row 10)
        be.set_correction_rhs(st, 0.37)             # rhs_kernel mode 1                     (row 5)
        be.reduce_rhs(st, st.w1)                    # reduce_rhs_kernel x 2                 (row 6-ii)
        be.finish_aug_solve(st, st.w1)              # finish_aug_solve_kernel               (row 6-vi)
        be.kktmul(st, st.w1, st.d, -1.0, 1.0)       # kktmul_diag / _lb / _ub               (row 7)
        be.get_alpha_max(st, 0.995)                 # alpha_max_kernel (+ final)            (row 8)
        be.get_complementarity_measure(st)          # compl_kernel                          (row 9)
        be.get_affine_complementarity_measure(st, 0.9, 0.8)
        be.norm_inf3(st.w1, st.p, st.d)             # norm_inf3_kernel                      (row 6-vii)
        be.get_inf(st)                              # inf_kernel                            (row 14)
        be.update_iterates(st, 1e-6, 1e-6)          # update_iterates_kernel                (row 13)
        be.adjust_boundary(st, 1e-3)                # adjust_boundary_kernel                (row 16)
    torch.cuda.synchronize()
    print(f"vec_pmc: n={n} m={m} reps={REPS} done")
    be.close()


if __name__ == "__main__":
    main()
