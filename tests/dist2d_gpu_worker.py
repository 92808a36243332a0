"""Rank program of tests/test_gpu_dist2d.py: madqp_dist_* (HIP kernels, csrc/dist.hip) on a P x Q grid whose ranks
share the one MI355X of the test box; collectives host-staged over gloo (RCCL refuses two ranks per device).  Same
checks as the CPU rehearsal (tests/dist2d_worker.py), NaN-poisoned tiles included."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import madqp_jl_amd as M  # noqa: E402
from madqp_jl_amd.dist2d import DistCholesky2D, HostStagedComm  # noqa: E402


def run(be, P, Q, n, nb, comm):
    dc = DistCholesky2D(be, n, nb, (P, Q), comm)
    p, q, mloc, nloc, ld, ncp = dc.p, dc.q, dc.mloc, dc.nloc, dc.ld, dc.ncp
    Kptr, _ = dc.matrix()
    rng = np.random.default_rng(42)
    G = rng.standard_normal((n, n))
    K = G @ G.T / n + 2.0 * np.eye(n)
    tiles = dc.local_tiles()

    def load(Mat):
        loc = np.zeros((ncp, ld))
        loc[:nloc, :mloc] = np.nan
        for I, J, li, lj in tiles:
            blk = Mat[I * nb:(I + 1) * nb, J * nb:(J + 1) * nb]
            if I == J:
                blk = np.where(np.tri(*blk.shape, dtype=bool), blk, np.nan)
            loc[lj * nb:lj * nb + blk.shape[1], li * nb:li * nb + blk.shape[0]] = blk.T
        be._ck(be.lib.madqp_memcpy_h2d(be.ctx, Kptr, loc.ctypes.data, loc.nbytes))

    rec = dict(p=p, q=q, tiles=len(tiles))
    load(K)
    rec["spd_info"] = dc.factor()
    loc = be.read_doubles(Kptr, ncp * ld).reshape(ncp, ld)
    L = np.linalg.cholesky(K)
    err = 0.0
    for I, J, li, lj in tiles:
        ref = L[I * nb:(I + 1) * nb, J * nb:(J + 1) * nb]
        got = loc[lj * nb:lj * nb + ref.shape[1], li * nb:li * nb + ref.shape[0]].T
        if I == J:
            got, ref = np.tril(got), np.tril(ref)
        err = max(err, float(np.max(np.abs(got - ref))))
    rec["factor_err"] = err
    rec["pad_clean"] = bool(np.all(loc[:, mloc:] == 0.0) and np.all(loc[nloc:, :] == 0.0))
    b = rng.standard_normal(n)
    x = torch.as_tensor(b, device=be.device).clone()
    dc.solve(x)
    ref = np.linalg.solve(K, b)
    rec["solve_err"] = float(np.max(np.abs(x.cpu().numpy() - ref)) / np.max(np.abs(ref)))
    rec["bytes_sent"] = dc.bytes_sent()
    bad = min(n - 1, nb + nb // 2 + 3)
    K2 = K.copy()
    K2[bad, bad] = -5.0
    load(K2)
    rec["notpd_info"], rec["notpd_expected"] = dc.factor(), bad + 1
    dc.close()
    return rec


def main():
    out, P, Q, n, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    be = M.HipBackend(0)  # every rank on the one GPU of the box
    comm = HostStagedComm(P, Q)
    rec = run(be, P, Q, n, nb, comm)
    assert comm.error is None, comm.error
    rec["calls"] = dict(comm.calls)
    json.dump(rec, open(f"{out}.{rank}", "w"))
    be.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
