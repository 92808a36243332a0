"""Worker of tests/test_gpu_dist.py: the multi-GPU KKT path (madqp_jl_amd/dist.py) rehearsed on ONE GPU:
`world` processes share device 0 and exchange panels over gloo (RCCL refuses two ranks on one device);
the rank-local kernels and the orchestration are exactly those of the multi-GPU run.  Each rank checks
itself against the single-GPU path of the same library and against the CPU oracle."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import madqp_jl_amd as M  # noqa: E402
from oracle import mpc  # noqa: E402
from oracle import qp as Q  # noqa: E402


def to_device(qp, be):
    return M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)


def lower_of(be, kkt, n):
    ptr, ld = be.kkt_matrix(kkt._h, n)
    K = be.read_doubles(ptr, ld * n).reshape(n, ld)[:, :n].T  # K[i, j]
    return np.tril(K)


def main():
    out_path = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    be = M.HipBackend(0)
    rec = dict(rank=rank, world=world)
    reg = M.FixedRegularization(1e-8, -1e-8)

    # (1) factor of the start-point KKT: distributed (panels of 128) == single GPU, panel by panel
    qp = Q.synthetic_qp(41, 700, 260)
    single = M.MPCSolver(to_device(qp, be), be, regularization=reg)
    single.initialize()
    multi = M.MPCSolver(to_device(qp, be), be, regularization=reg, distributed=True, panel_width=128)
    multi.initialize()
    for s in (single, multi):
        be.set_aug_diagonal_reg(s.st, 1e-8, -1e-8)
        s.kkt.factorize_wrapper()
    L1, L2 = lower_of(be, single.kkt, 700), lower_of(be, multi.kkt, 700)
    rec["factor_err"] = float(np.max(np.abs(L1 - L2)) / np.max(np.abs(L1)))
    rec["factor_info"] = [single.kkt.linear_solver.info, multi.kkt.linear_solver.info]
    rec["panels"] = len(multi.kkt.dchol.panels)
    rec["own"] = multi.kkt.dchol.own_panels()
    # not positive definite: every rank reports the first failing column of the one-GPU factorisation
    for s in (single, multi):
        s.H.sub_(1e6 * torch.eye(700, dtype=torch.float64, device=be.device))
        s.kkt.factorize_wrapper()
    rec["notpd_info"] = [single.kkt.linear_solver.info, multi.kkt.linear_solver.info]
    single.close()
    multi.close()

    # (2) whole solves: traces of the distributed run vs the CPU oracle and vs the one-GPU run
    cases = {"qp_900_350": (Q.synthetic_qp(42, 900, 350), "condensed", reg, 256),
             "qp_gondzio": (Q.synthetic_qp(43, 300, 120), "condensed", reg, 128),
             "lp_normal": (Q.synthetic_qp(5, 400, 150, "lp"), "normal", M.FixedRegularization(1e-8, 0.0), 128)}
    for name, (qp, ksys, r, nb) in cases.items():
        ncorr = 3 if name == "qp_gondzio" else 0
        s1 = M.MPCSolver(to_device(qp, be), be, regularization=r, kkt_system=ksys, max_ncorr=ncorr)
        r1 = s1.solve()
        s1.close()
        s2 = M.MPCSolver(to_device(qp, be), be, regularization=r, kkt_system=ksys, max_ncorr=ncorr,
                         distributed=True, panel_width=nb)
        r2 = s2.solve()
        s2.close()
        ref = mpc.solve(qp, kkt_system=ksys, max_ncorr=ncorr,
                        regularization=mpc.FixedRegularization(r.delta_p, r.delta_d))
        rec[name] = dict(
            status=[r1["status"], r2["status"], ref["status"]], iters=[r1["iter"], r2["iter"], ref["iter"]],
            trace=r2["trace"], ref_trace=ref["trace"],
            dx_single=float(np.max(np.abs(r2["solution"] - r1["solution"]))),
            dx_oracle=float(np.max(np.abs(r2["solution"] - ref["solution"]))),
            obj=[r2["objective"], ref["objective"]],
            xsum=float(np.sum(r2["solution"])))  # must be bitwise equal on all ranks
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(rec, f)
    be.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
