"""CPU: the N > 1 path of bench.py (one process per GPU, independent QPs per rank, barrier + MAX-over-
ranks timing, whole-job value) exercised with world_size 2 over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_gloo(tmp_path):
    out = str(tmp_path / "rec")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "tests", "dist_worker.py"), out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    recs = [json.load(open(f"{out}.{r}")) for r in (0, 1)]
    assert [r["rank"] for r in recs] == [0, 1] and all(r["world"] == 2 for r in recs)
    assert recs[0]["seed"] != recs[1]["seed"]  # independent instances per rank
    # both ranks agree on the job time = the slow rank's time, bracketed by barriers
    assert abs(recs[0]["tmax"] - recs[1]["tmax"]) < 1e-9
    assert recs[0]["tmax"] >= 0.3 - 1e-3 and recs[0]["tmax"] >= max(r["elapsed"] for r in recs) - 1e-9
    # whole-job aggregate: all ranks' iterations over the max time
    assert abs(recs[0]["value"] - 2 * 3 / recs[0]["tmax"]) < 1e-9
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "only rank 0 prints the JSON line"
    # the set-up collectives of a shared QP (dist2d.DistributedQP) over the same process group
    assert all(r["rowmax_err"] == 0.0 and r["hx_err"] < 1e-13 for r in recs)


def test_setup_collectives_follow_the_group_backend():
    """`bench.py --gpus N` initialises nccl only, and RCCL has no backend for CPU tensors -- the
    replicated set-up reductions must stay on the device there and go through the host only under gloo."""
    sys.path.insert(0, ROOT)
    from madqp_jl_amd.dist2d import reduce_where

    assert reduce_where("nccl", True) == "as_is"
    assert reduce_where("cpu:gloo,cuda:nccl", True) == "as_is" and reduce_where("cpu:gloo,cuda:nccl", False) == "as_is"
    assert reduce_where("gloo", True) == "to_cpu" and reduce_where("gloo", False) == "as_is"
    with pytest.raises(RuntimeError):
        reduce_where("nccl", False)  # what round 2 did: a CPU tensor on an nccl-only group


def test_single_process_helpers():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.job_value(1, 5, 2.0) == 2.5 and bench.job_value(8, 5, 2.0) == 20.0
    assert bench.max_over_ranks(1.25, 1, "cpu") == 1.25
    assert bench.rank_seed(7, 3) == 10
