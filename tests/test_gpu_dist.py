"""GPU: one dense KKT system shared by several ranks, rehearsed on one MI355X with 2 and 3
processes on the same device exchanging panels over gloo (tests/dist_gpu_worker.py).  Parity bar as in
test_gpu_solver.py: traces vs the CPU oracle within the stated tolerance, identical iteration counts;
plus: the distributed factor equals the one-GPU factor to rounding, and all ranks agree bitwise."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, tol):
    return abs(a - b) <= tol * max(1.0, abs(a), abs(b))


@pytest.mark.parametrize("world,port", [(2, 29531), (3, 29533), (4, 29535)])
def test_distributed_kkt_on_one_gpu(tmp_path, world, port):
    out = str(tmp_path / "rec")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    recs = [json.load(open(f"{out}.{r}")) for r in range(world)]
    for r, rec in enumerate(recs):
        assert rec["panels"] == 6 and rec["own"] == [p for p in range(6) if p % world == r]
        assert rec["factor_info"] == [0, 0] and rec["factor_err"] < 1e-11
        assert rec["notpd_info"][0] > 0 and rec["notpd_info"][1] == rec["notpd_info"][0]
        for name in ("qp_900_350", "qp_gondzio", "lp_normal"):
            c = rec[name]
            assert c["status"] == [1, 1, 1] and c["iters"][0] == c["iters"][1] == c["iters"][2], (name, c["iters"])
            assert len(c["trace"]) == len(c["ref_trace"])
            for t, g in zip(c["trace"], c["ref_trace"]):
                tol = 1e-9 if min(t["mu"], g["mu"]) >= 1e-4 else 1e-6
                for key in ("alpha_p", "alpha_d", "inf_pr", "inf_du", "inf_compl", "mu"):
                    assert close(t[key], g[key], tol), (name, t["k"], key, t[key], g[key])
            assert c["dx_oracle"] <= 1e-7 and c["dx_single"] <= 1e-7
            assert close(c["obj"][0], c["obj"][1], 1e-9)
    for name in ("qp_900_350", "qp_gondzio", "lp_normal"):  # replicated scalars: bitwise equal across ranks
        assert all(rec[name]["xsum"] == recs[0][name]["xsum"] for rec in recs)
        assert all(rec[name]["trace"] == recs[0][name]["trace"] for rec in recs)
