"""GPU: madqp_dist_* (2-D block-cyclic distributed Cholesky + distributed sweeps, csrc/dist.hip).
(a) one rank: the same entry points against LAPACK, in process -- the MFMA / sweep kernels under the schedule;
(b) P x Q ranks sharing the one MI355X of the test box, collectives host-staged over gloo (RCCL itself needs one GPU
per rank: the driver's 8-GPU run).  The CPU suite runs the same schedule with CPU loops (tests/test_dist2d.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,nb", [(1500, 256), (1024, 1024), (700, 1024), (2500, 512), (130, 128)])
def test_single_rank_against_lapack(hip, n, nb):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist2d_gpu_worker import run

    rec = run(hip, 1, 1, n, nb, None)
    assert rec["spd_info"] == 0 and rec["factor_err"] < 1e-11 and rec["solve_err"] < 1e-10
    assert rec["pad_clean"] and rec["notpd_info"] == rec["notpd_expected"] and rec["bytes_sent"] == 0


# at most 4 ranks: the box allows 6 processes on its GPU, and the test runner and the launcher count (2 x 3 and larger
# grids run in the CPU rehearsal)
@pytest.mark.parametrize("P,Q,n,nb,port", [(1, 2, 1500, 256, 29551), (2, 2, 2100, 256, 29553), (2, 1, 3000, 384, 29555),
                                           (2, 2, 700, 512, 29557)])
def test_grid_on_one_gpu(tmp_path, P, Q, n, nb, port):
    world = P * Q
    out = str(tmp_path / "rec")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist2d_gpu_worker.py"), out, str(P), str(Q), str(n), str(nb)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    recs = [json.load(open(f"{out}.{k}")) for k in range(world)]
    T = (n + nb - 1) // nb
    assert sum(rec["tiles"] for rec in recs) == T * (T + 1) // 2
    for k, rec in enumerate(recs):
        assert (rec["p"], rec["q"]) == (k // Q, k % Q)
        assert rec["spd_info"] == 0 and rec["factor_err"] < 1e-11 and rec["solve_err"] < 1e-10, rec
        assert rec["pad_clean"] and rec["notpd_info"] == rec["notpd_expected"]
