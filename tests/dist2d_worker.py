"""Rank program of tests/test_dist2d.py (CPU, gloo): the distributed schedule of csrc/dist_core.inc, compiled with
CPU loops for the rank-local kernels (tests/csrc/dist_cpu.cpp), on a P x Q grid.  Tiles this rank does not own -- and
the upper triangle -- are poisoned with NaN, so any read of data that was not sent shows up in the result."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madqp_jl_amd.dist2d import HostStagedComm  # noqa: E402


def main():
    out, P, Q, n, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == P * Q
    lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmadqp_dist_cpuref.so"))
    comm = HostStagedComm(P, Q)
    h = C.c_void_p()
    rc = lib.madqp_distcpu_create(rank, world, P, Q, C.c_int64(n), C.c_int64(nb), C.byref(comm.ops), C.byref(h))
    assert rc == 0, rc
    lay = (C.c_int64 * 8)()
    lib.madqp_distcpu_layout(h, lay)
    p, q, mt, nt, mloc, nloc, ld, ncp = list(lay)
    kp, kld = C.c_void_p(), C.c_int64()
    lib.madqp_distcpu_matrix(h, C.byref(kp), C.byref(kld))
    Kloc = np.ctypeslib.as_array(C.cast(kp, C.POINTER(C.c_double)), shape=(ncp, ld))  # Kloc[col, row]: column-major
    T = (n + nb - 1) // nb
    rng = np.random.default_rng(42)  # the same matrix on every rank
    G = rng.standard_normal((n, n))
    K = G @ G.T / n + 2.0 * np.eye(n)
    tiles = [(I, J) for J in range(q, T, Q) for I in range(p, T, P) if I >= J]

    def load(M):
        Kloc[:, :] = 0.0
        Kloc[:nloc, :mloc] = np.nan  # poison: tiles that are not mine to fill (upper triangle) must never be read
        for I, J in tiles:
            blk = M[I * nb:(I + 1) * nb, J * nb:(J + 1) * nb]
            if I == J:
                blk = np.where(np.tri(*blk.shape, dtype=bool), blk, np.nan)  # strictly upper part: poison too
            li, lj = I // P, J // Q
            Kloc[lj * nb:lj * nb + blk.shape[1], li * nb:li * nb + blk.shape[0]] = blk.T

    rec = dict(rank=rank, p=p, q=q, tiles=len(tiles), mloc=mloc, nloc=nloc)
    mem = (C.c_int64 * 8)()
    lib.madqp_distcpu_memory(h, mem)  # (total, K, XW, YW, bands, staging, levels, 0)
    rec["memory"] = dict(total=mem[0], K=mem[1], xw=mem[2], yw=mem[3], levels=mem[6], ld=ld, ncp=ncp, T=T)
    load(K)
    info = C.c_int32(-1)
    rc = lib.madqp_distcpu_factor(h, C.byref(info))
    assert rc == 0 and comm.error is None, (rc, comm.error)
    rec["spd_info"] = info.value
    lib.madqp_distcpu_trace.restype = C.c_int64
    cnt = lib.madqp_distcpu_trace(h, None, C.c_int64(0))
    buf = (C.c_int64 * max(cnt, 1))()
    lib.madqp_distcpu_trace(h, buf, C.c_int64(cnt))
    rec["schedule"] = [list(buf[i:i + 3]) for i in range(0, cnt, 3)]  # (mark, k, stream) | (-1, from, to)
    L = np.linalg.cholesky(K)
    err = 0.0
    for I, J in tiles:
        ref = L[I * nb:(I + 1) * nb, J * nb:(J + 1) * nb]
        li, lj = I // P, J // Q
        got = Kloc[lj * nb:lj * nb + ref.shape[1], li * nb:li * nb + ref.shape[0]].T
        if I == J:
            got, ref = np.tril(got), np.tril(ref)
        err = max(err, float(np.max(np.abs(got - ref))))
    rec["factor_err"] = err
    rec["pad_clean"] = bool(np.all(Kloc[:, mloc:] == 0.0) and np.all(Kloc[nloc:, :] == 0.0))
    # solves: replicated right-hand side in, replicated solution out
    b = rng.standard_normal(n)
    x = b.copy()
    before = dict(comm.calls)
    rc = lib.madqp_distcpu_solve(h, x.ctypes.data_as(C.c_void_p))
    assert rc == 0 and comm.error is None, (rc, comm.error)
    rec["solve_calls"] = {k: comm.calls[k] - before[k] for k in ("reduce", "bcast")}
    ref = np.linalg.solve(K, b)
    rec["solve_err"] = float(np.max(np.abs(x - ref)) / np.max(np.abs(ref)))
    rec["nan"] = bool(np.isnan(x).any())
    sent = C.c_int64()
    lib.madqp_distcpu_bytes_sent(h, C.byref(sent))
    rec["bytes_sent"] = sent.value
    rec["calls"] = dict(comm.calls)
    # not positive definite: LAPACK's info (first failing column, 1-based) on every rank
    bad = min(n - 1, nb + nb // 2 + 3)
    K2 = K.copy()
    K2[bad, bad] = -5.0
    load(K2)
    rc = lib.madqp_distcpu_factor(h, C.byref(info))
    assert rc == 0, rc
    rec["notpd_info"] = info.value
    rec["notpd_expected"] = bad + 1
    lib.madqp_distcpu_destroy(h)
    json.dump(rec, open(f"{out}.{rank}", "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
