#!/usr/bin/env python3
"""Batch of independent QPs (BASELINE configs[3]: 1024 x (n=512, m=256)) on one GPU / one rank.

    python tools/bench_batch.py [--batch 128] [--nx 512] [--m 256] [--streams 16]

Under torch.distributed.run each rank takes problems rank, rank+N, ... (no communication) and rank 0
reports the aggregate.  Prints one JSON line: QPs/s and IPM iterations/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def batch_roofline(nx, m, iters, seconds, max_ncorr=0):
    """Where a batch of small QPs stands against the chip's two bounds (whole-run averages).
    MFMA flops per problem-iteration: assembly m nx^2 + Cholesky nx^3/3.
    Algorithmic HBM bytes per problem-iteration: every matrix pass the iteration needs at 8 B per entry, a pass over a
    triangle counted as half a matrix -- per solve_system (two at max_ncorr = 0): A' u, A dx, A' v_y (m nx each; the
    residual check reuses the solve's A dx), H v from the lower triangle (nx^2/2), two sweeps over L (nx^2/2 each); per
    iteration besides: jtprod A' y (m nx), the model evaluation H x + A x (nx^2/2 + m nx); assembly: H lower (nx^2/2) in, the scaled operand
    sqrt(Theta) A written and read (2 m nx), K lower out (nx^2/2); Cholesky: K in, L out (nx^2/2 each)."""
    flops = m * nx * nx + nx ** 3 / 3.0
    solves = 2 + max_ncorr
    doubles = solves * (3 * m * nx + nx * nx // 2 + nx * nx) + (m * nx) + (nx * nx // 2 + m * nx) + (nx * nx // 2 + 2 * m * nx + nx * nx // 2) + nx * nx
    it_per_s = iters / seconds
    tf, gbs = it_per_s * flops * 1e-12, it_per_s * 8.0 * doubles * 1e-9
    return {"flops_per_problem_iteration": flops, "algorithmic_bytes_per_problem_iteration": 8.0 * doubles,
            "achieved_TFLOPs": tf, "frac_of_fp64_mfma_peak": tf / bench.PEAK_F64_MFMA_TFLOPS,
            "achieved_GBps_algorithmic": gbs, "frac_of_hbm_peak": gbs / bench.PEAK_HBM_GBS}


def run_batched(M, be, batch, nx, m, seed, repeats=3, check_every=2, profile=False):
    """The lock-step batched engine on `batch` synthetic QPs of this rank: (median seconds, results, all times)."""
    import torch

    opts = dict(max_iter=300, step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8),
                mu_min=1e-12)
    qps = [M.DeviceQP.synthetic(be, seed + i, nx, m) for i in range(batch)]
    warm = M.BatchedMPCSolver(qps, be, **opts)
    warm.solve(check_every=check_every)
    warm.close()
    times, res = [], None
    for _ in range(max(1, repeats)):
        solver = M.BatchedMPCSolver(qps, be, **opts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = solver.solve(check_every=check_every)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        solver.close()
    return sorted(times)[len(times) // 2], res, times


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=128)
    p.add_argument("--nx", type=int, default=512)
    p.add_argument("--m", type=int, default=256)
    p.add_argument("--streams", type=int, default=16)
    p.add_argument("--seed", type=int, default=20250614 + 3)
    p.add_argument("--driver", choices=("native", "python"), default="native")
    p.add_argument("--engine", choices=("batched", "streams"), default="batched",
                   help="batched: lock-step engine of csrc/batch.hip; streams: one MPCSolver per problem, "
                        "several HIP streams in flight (madqp_jl_amd/batch.py)")
    p.add_argument("--check-every", type=int, default=2)
    p.add_argument("--repeats", type=int, default=3, help="timed solves (fresh solver each); the median is reported")
    p.add_argument("--profile", action="store_true", help="print ms / launches per kernel class (perturbs timing)")
    a = p.parse_args()
    import torch

    world, rank, local_rank = bench.dist_setup("nccl")
    import madqp_jl_amd as M

    mine = M.shard(range(a.batch), rank, world)
    make = lambda be, i: M.DeviceQP.synthetic(be, a.seed + i, a.nx, a.m)
    opts = dict(max_iter=300, step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8),
                mu_min=1e-12, driver=a.driver)
    if a.engine == "streams":
        M.solve_batch(make, mine[: min(len(mine), a.streams)], local_rank, a.streams, **opts)  # warm-up
        bench.dist_barrier(world)
        t0 = time.perf_counter()
        res = list(M.solve_batch(make, mine, local_rank, a.streams, **opts).values())
        bench.dist_barrier(world)
        dt = bench.max_over_ranks(time.perf_counter() - t0, world, torch.device("cuda", local_rank))
        lockstep = None
    else:
        opts.pop("driver")
        be = M.HipBackend(local_rank)
        qps = [make(be, i) for i in mine]  # data generation is not part of the timed solve
        # warm-up with the full batch (a first solver of a given size pays one-time costs -- kernel load, graph
        # instantiation, allocator growth -- that later ones do not: 187 vs 37 ms at 128 problems), then the median
        # of --repeats timed solves, each with a fresh solver: set-up (scaling, start point) + all iterations +
        # read-back are inside the timed region
        warm = M.BatchedMPCSolver(qps, be, **opts)
        warm.solve(check_every=a.check_every)
        warm.close()
        times = []
        for _ in range(max(1, a.repeats)):
            solver = M.BatchedMPCSolver(qps, be, **opts)
            if a.profile:
                be.prof_enable(M._lib.PROF_CLASSES)
                be.prof_reset()
            bench.dist_barrier(world)
            t0 = time.perf_counter()
            res = solver.solve(check_every=a.check_every)
            bench.dist_barrier(world)
            times.append(bench.max_over_ranks(time.perf_counter() - t0, world, torch.device("cuda", local_rank)))
            if a.profile and rank == 0:
                print({k: (round(v[0], 2), v[1]) for k, v in be.prof_get().items() if v[1]}, flush=True)
            solver.close()
        dt = sorted(times)[len(times) // 2]
        lockstep = int(max(r["iter"] for r in res))
    iters = sum(r["iter"] for r in res)
    ok = sum(r["status"] == M.SOLVE_SUCCEEDED for r in res)
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([iters, ok], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t)
        iters, ok = int(t[0].item()), int(t[1].item())
    if rank == 0:
        print(json.dumps({"metric": "independent QPs solved per second", "value": a.batch / dt, "unit": "QP/s",
                          "ipm_iterations_per_s": iters / dt, "n_gpus": world, "batch": a.batch,
                          "roofline": batch_roofline(a.nx, a.m, iters, dt),
                          "solved": ok, "config": {"workload": f"{a.batch} x synthetic dense QP nx={a.nx} m={a.m}",
                                                   "engine": a.engine, "streams_per_gpu": a.streams if a.engine == "streams" else None,
                                                   "lock_step_iterations": lockstep}, "seconds": dt,
                          "all_seconds": times if a.engine == "batched" else [dt]}), flush=True)


if __name__ == "__main__":
    main()
