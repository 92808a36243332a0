"""Worker of tests/test_dist.py: runs bench.py's multi-rank scaffolding over gloo on the CPU with a
stand-in step (the HIP step needs a GPU).  Launched by torch.distributed.run with 2 ranks."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def shared_qp_setup_products(world, rank):
    """DistributedQP.row_absmax / hess_times (the two collectives of MPCSolver.initialize on a shared QP, dist2d.py)
    on a 1 x world grid over the group bench.py set up, against the dense products."""
    import types

    import numpy as np
    import torch

    from madqp_jl_amd import dist2d

    n, m, nb, P, Q = 300, 40, 128, 1, world
    T = (n + nb - 1) // nb
    rng = np.random.default_rng(5)
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((n, n))
    H = G + G.T
    x = rng.standard_normal(n)
    p, q = rank // Q, rank % Q
    rows = [i for I in range(p, T, P) for i in range(I * nb, min(n, (I + 1) * nb))]
    cols = [j for J in range(q, T, Q) for j in range(J * nb, min(n, (J + 1) * nb))]
    mloc, nloc = len(rows), len(cols)
    pad = lambda v: max(128, (v + 127) // 128 * 128)
    grid = types.SimpleNamespace(n=n, nb=nb, P=P, Q=Q, p=p, q=q, mloc=mloc, nloc=nloc, ld=pad(mloc), ncp=pad(nloc))

    class Be:
        device = torch.device("cpu")

    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))
    z = np.zeros
    dq = dist2d.DistributedQP.from_dense(Be(), grid, H, z(n), A, z(n), z(n) + 1, z(m), z(m) + 1, z(n))
    rmax = dq.row_absmax().numpy()
    hx = dq.hess_times(t(x)).numpy()
    return dict(rowmax_err=float(np.max(np.abs(rmax - np.abs(A).max(axis=1)))),
                hx_err=float(np.max(np.abs(hx - H @ x)) / np.max(np.abs(H @ x))))


def main():
    out_path = sys.argv[1]
    world, rank, local_rank = bench.dist_setup("gloo")
    steps = 3
    bench.dist_barrier(world, cuda=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        time.sleep(0.05 * (1 + rank))  # rank 1 is the slow one
    bench.dist_barrier(world, cuda=False)
    elapsed = time.perf_counter() - t0
    tmax = bench.max_over_ranks(elapsed, world, "cpu")
    rec = dict(world=world, rank=rank, local_rank=local_rank, seed=bench.rank_seed(100, rank),
               elapsed=elapsed, tmax=tmax, value=bench.job_value(world, steps, tmax))
    rec.update(shared_qp_setup_products(world, rank))
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(rec, f)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    import torch.distributed as dist

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
