"""GPU: bench.py's N > 1 path on the one MI355X of the test box -- plain `python bench.py --gpus 2`, no launcher around
it, the two ranks it starts share device 0 and exchange through host-staged collectives over gloo
(MADQP_DIST_BACKEND=gloo, MADQP_DIST_SHARE_DEVICE=1: RCCL refuses two ranks per device).  Everything else is the
product: the HIP library, the 1 x 2 grid, both legs of the protocol, the one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, extra_env, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MADQP_BENCH_TEST_DOUBLE")}
    env.update(OMP_NUM_THREADS="2", **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_2_without_a_launcher_runs_two_ranks_of_the_hip_library():
    p = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--nx", "2000", "--m", "800", "--extra-timeout", "300"],
                  {"MADQP_DIST_BACKEND": "gloo", "MADQP_DIST_SHARE_DEVICE": "1"})
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert "test_double" not in out and out["data"] == "synthetic"
    assert out["n_gpus"] == 2 and out["n_gpus_requested"] == 2 and out["scaling"] == "strong"
    assert out["ranks"]["started_by"].startswith("bench.py")
    assert out["comm"]["world_size"] == 2 and out["comm"]["row_comm_size"] == 2 and out["comm"]["col_comm_size"] == 1
    assert out["comm"]["backend"].startswith("host-staged")
    d = out["distributed"]
    assert d["grid"] == [1, 2] and d["bytes_broadcast_by_rank0"] > 0
    parts = d["rank0_bytes_by_part"]
    assert parts["stored_operands_XW"] > 0 and parts["stored_operands_YW"] > 0  # 1 x 2: both operand stores exist
    assert d["rank0_matrix_bytes"] == sum(parts.values()) and d["rank0_library_bytes"] >= parts["K"]
    assert out["value"] > 0 and out["independent_qps"]["value"] > 0 and out["last_trace"]["k"] >= 3


@pytest.mark.parametrize("bcast", ["collective", "p2p"])
def test_rehearsal_of_the_drivers_multi_gpu_command_on_a_2x2_grid(bcast):
    """The driver's N > 1 form with its own step counts (`--gpus N --steps 20 --warmup 5`), un-wrapped, at a reduced size,
    FOUR ranks of the HIP library sharing the test GPU on a 2 x 2 grid (the pool allows six processes on a card, so the
    eight-rank form is rehearsed on the CPU double, tests/test_bench.py): ONE line, both process-grid directions have two
    ranks -- diagonal-tile broadcast down a column, row and transposed column broadcasts, band collection, grouped
    solves all run -- `roofline` and the independent-QPs leg present, finished well inside the driver's limit.  Both
    broadcast forms (MADQP_DIST_BCAST)."""
    import time

    env = {"MADQP_DIST_BACKEND": "gloo", "MADQP_DIST_SHARE_DEVICE": "1"}
    if bcast == "p2p":
        env["MADQP_DIST_BCAST"] = "p2p"
    t0 = time.time()
    p = run_bench(["--gpus", "4", "--steps", "20", "--warmup", "5", "--nx", "3000", "--m", "1200", "--extra-timeout", "300",
                   "--no-cpu-baseline"], env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["n_gpus_requested"] == 4 and out["scaling"] == "strong"
    assert out["steps"] == 20 and out["warmup"] == 5
    assert out["comm"]["world_size"] == 4 and out["comm"]["row_comm_size"] == 2 and out["comm"]["col_comm_size"] == 2
    assert out["distributed"]["grid"] == [2, 2] and out["distributed"]["bytes_broadcast_by_rank0"] > 0
    assert out["roofline"]["frac"] > 0 and "job_fraction_of_peak" in out["roofline"]
    assert out["value"] > 0 and out["independent_qps"]["value"] > 0 and out["last_trace"]["k"] >= 3
    assert time.time() - t0 < 500


def test_the_test_double_is_refused_on_a_box_with_a_gpu():
    p = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--nx", "40", "--m", "16"],
                  {"MADQP_BENCH_TEST_DOUBLE": "bench_double:Double",
                   "PYTHONPATH": os.pathsep.join([os.path.join(ROOT, "tests"), ROOT])}, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "refused" in p.stderr
