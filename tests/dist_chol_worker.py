"""Worker of tests/test_dist.py::test_distributed_cholesky_*: the panel-cyclic right-looking Cholesky
driver of madqp_jl_amd/dist.py over gloo on the CPU, with numpy stand-ins for the rank-local HIP
primitives (same contracts as include/madqp.h "multi-GPU factorisation pieces").  Panels a rank does
not own start as NaN, so any read of data that has not been received yet poisons the result."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from madqp_jl_amd.dist import DistributedCholesky, default_panel_width, panel_ranges  # noqa: E402


class NumpyPanelOps:
    """Rank-local pieces on a dense numpy matrix (A[i, j], lower triangle significant)."""

    def __init__(self, A):
        self.A, self.n, self.info = A, A.shape[0], 0
        self.log = []

    def chol_factor_begin(self, ch, A_ptr, lda):
        self.info = 0

    def chol_factor_panel(self, ch, j, w):
        A = self.A
        self.log.append(("factor", j))
        D = np.tril(A[j:j + w, j:j + w]) + np.tril(A[j:j + w, j:j + w], -1).T
        try:
            L = np.linalg.cholesky(D)
        except np.linalg.LinAlgError:
            if self.info == 0:
                k = next(k for k in range(1, w + 1) if np.any(np.linalg.eigvalsh(D[:k, :k]) <= 0))
                self.info = j + k
            L = np.full_like(D, np.nan)
        A[j:j + w, j:j + w] = np.tril(L) + np.triu(A[j:j + w, j:j + w], 1)
        if j + w < self.n:
            A[j + w:, j:j + w] = np.linalg.solve(L, A[j + w:, j:j + w].T).T if self.info == 0 else np.nan

    def chol_update_cols(self, ch, c0, cw, p0, pw):
        A = self.A
        assert p0 + pw <= c0
        self.log.append(("update", c0, p0))
        upd = A[c0:, p0:p0 + pw] @ A[c0:c0 + cw, p0:p0 + pw].T
        blk = A[c0:, c0:c0 + cw]
        low = np.tril(np.ones((self.n - c0, cw), dtype=bool))
        blk[low] -= upd[low]

    def chol_update_multi(self, ch, cols, p0, pw):
        for c0, cw in cols:
            self.chol_update_cols(ch, c0, cw, p0, pw)

    def chol_panel_doubles(self, ch, j, w):
        return 2 + w * (self.n - j)

    def chol_panel_pack(self, ch, j, w, buf):
        buf[0] = float(self.info)
        buf[2:2 + w * (self.n - j)] = torch.from_numpy(np.ascontiguousarray(self.A[j:, j:j + w].T).ravel())

    def chol_panel_unpack(self, ch, j, w, buf):
        if self.info == 0 and int(buf[0]) != 0:
            self.info = int(buf[0])
        self.A[j:, j:j + w] = buf[2:2 + w * (self.n - j)].numpy().reshape(w, self.n - j).T

    def chol_factor_end(self, ch):
        return self.info


def main():
    out_path, n, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    world, rank, _ = bench.dist_setup("gloo")
    rng = np.random.default_rng(5)
    R = rng.standard_normal((n, n))
    K = R @ R.T + n * np.eye(n)
    rec = dict(rank=rank, world=world)
    for case in ("spd", "not_pd"):
        Kc = K.copy()
        if case == "not_pd":
            Kc[300, 300] = -1.0
        A = np.tril(Kc)
        A[np.triu_indices(n, 1)] = 0.0
        panels = panel_ranges(n, nb)
        for p, (j, w) in enumerate(panels):
            if p % world != rank:
                A[j:, j:j + w] = np.nan  # not mine: must arrive through a broadcast before any use
        ops = NumpyPanelOps(A)
        dc = DistributedCholesky(ops, None, n, nb, "cpu")
        assert dc.own_ranges() == [(j, j + w) for p, (j, w) in enumerate(panels) if p % world == rank]
        info = dc.factor(0, n)
        rec[case + "_info"] = info
        if case == "spd":
            L = np.linalg.cholesky(K)
            rec["err"] = float(np.max(np.abs(np.tril(A) - L)) / np.max(np.abs(L)))
            rec["nan"] = bool(np.isnan(np.tril(A)).any())
            rec["factored"] = [e[1] for e in ops.log if e[0] == "factor"]
            rec["updates"] = sum(e[0] == "update" for e in ops.log)
            rec["bytes_sent"] = dc.bytes_sent
    rec["default_nb"] = [default_panel_width(50000, 8), default_panel_width(5000, 8), default_panel_width(600, 2)]
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(rec, f)
    import torch.distributed as dist

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
