"""Worker of tests/test_dist.py: runs bench.py's multi-rank scaffolding over gloo on the CPU with a
stand-in step (the HIP step needs a GPU).  Launched by torch.distributed.run with 2 ranks."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


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
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(rec, f)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    import torch.distributed as dist

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
