"""CPU (gloo, world_size 2..6): the 2-D block-cyclic distributed Cholesky and its distributed triangular solves
(BASELINE configs[4]).  What runs is the product's own schedule -- madqp_jl_amd/csrc/dist_core.inc,
the file dist.hip compiles for the GPU -- built with CPU loops for the rank-local kernels (tests/csrc/dist_cpu.cpp)
and host-staged collectives over gloo; tiles a rank does not own are NaN-poisoned (tests/dist2d_worker.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cpuref():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "csrc")], stdout=subprocess.DEVNULL)


def test_grid_and_tile_defaults():
    sys.path.insert(0, ROOT)
    from madqp_jl_amd.dist2d import default_grid, default_tile

    assert [default_grid(w) for w in (1, 2, 3, 4, 6, 8)] == [(1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (2, 4)]
    assert default_tile(50000, 8) == 1024 and default_tile(100000, 8) == 1024 and default_tile(5000, 2) == 384
    assert default_tile(300, 4) == 128


@pytest.mark.parametrize("P,Q,n,nb,port", [
    (1, 2, 700, 128, 29541),   # one process row: no transposed broadcast needed beyond the row itself
    (2, 1, 700, 128, 29543),   # one process column
    (2, 2, 1000, 128, 29545),  # partial last tile (1000 = 7*128 + 104)
    (2, 3, 900, 256, 29547),   # P and Q coprime, nb = 2 blocks
    (2, 2, 300, 384, 29549),   # fewer tiles than ranks in one direction: some ranks own nothing
])
def test_distributed_cholesky_2d(cpuref, tmp_path, P, Q, n, nb, port):
    world = P * Q
    out = str(tmp_path / "rec")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist2d_worker.py"), out, str(P), str(Q), str(n), str(nb)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.load(open(f"{out}.{k}")) for k in range(world)]
    T = (n + nb - 1) // nb
    assert sum(rec["tiles"] for rec in recs) == T * (T + 1) // 2  # every lower tile has exactly one owner
    for k, rec in enumerate(recs):
        assert (rec["p"], rec["q"]) == (k // Q, k % Q)
        assert rec["spd_info"] == 0 and not rec["nan"]
        assert rec["factor_err"] < 1e-12 and rec["solve_err"] < 1e-11
        assert rec["pad_clean"]  # the zero padding of the local matrix survives (the MFMA kernels read it)
        assert rec["notpd_info"] == rec["notpd_expected"]  # LAPACK's info, identical on every rank
    # volume: a rank receives each panel tile at most once per operand role; as roots the ranks send
    # (P > 1) T diagonal images + (Q > 1) the row operands + (P > 1) the transposed operands -- never the matrix twice
    total = sum(rec["bytes_sent"] for rec in recs)
    assert total <= 8 * (3 * n * n // 2 + 4 * T * (nb * nb + 2 * 128 * 128 * (nb // 128)) + 64 * T * nb * max(P, Q))
