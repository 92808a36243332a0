#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec of the Mehrotra predictor-corrector KKT path on MI355X.

Metric (BASELINE.json): IPM iterations/sec + KKT factor+solve ms, dense QP n=50k m=20k, fp64.
A "step" is one pass of the hot path = one predictor-corrector iteration (src/solver.jl:259-343:
residuals, Sigma update, condensed-KKT assembly (SYRK), Cholesky, two solves with residual check,
step lengths, iterate update, model evaluation) on a synthetic dense QP generated in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nx 50000] [--m 20000]

N > 1, one rank per GPU: ONE QP of that size is shared by all GPUs on a P x Q grid -- 2-D block-cyclic distributed
assembly, Cholesky and triangular solves over RCCL (csrc/dist.hip): strong scaling, value = iterations /
max-over-ranks time.  The independent-QPs rate (every rank its own instance, no data-path collective: BASELINE
configs[3] style weak scaling) is measured first and reported beside it (`independent_qps`); `--kkt local` makes it the
headline.  Rank 0 prints ONE JSON line.

How the N ranks come to be (`launch_plan`): under a launcher (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: WORLD_SIZE is set) this process IS one of them; a plain `python bench.py --gpus N` with N > 1
and no WORLD_SIZE starts that launcher itself as a child process -- before torch is imported or anything touches the
GPU -- and hands its stdout line and exit code through; a WORLD_SIZE that differs from --gpus is refused (exit code 2,
no line): a run labelled N GPUs is a run on N ranks, or no run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6  # 256 CU x 2.4 GHz x 128 flop/clk/CU (MI355X datasheet fp64 matrix)
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--nx", type=int, default=50000)
    p.add_argument("--m", "--ncon", dest="m", type=int, default=20000)
    p.add_argument("--max-ncorr", type=int, default=3,
                   help="Gondzio corrections; 3 = scripts/benchmarks_cpu.jl:38 (the headline), 0 = the solver default "
                        "src/utils.jl:94 (reported beside it as `max_ncorr_0`, SURVEY.md 8d)")
    p.add_argument("--no-second-ncorr", action="store_true",
                   help="skip the additional leg with the other max_ncorr setting (0 <-> 3)")
    p.add_argument("--seed", type=int, default=20250614 + 1)
    p.add_argument("--driver", choices=("native", "python"), default="native",
                   help="host driver of the loop body: csrc/mpc.hip (one C call per iteration) or solver.py")
    p.add_argument("--kkt", choices=("local", "distributed"), default=None,
                   help="local: every GPU solves its own QP (weak scaling); distributed: all GPUs share ONE QP on a "
                        "P x Q grid -- 2-D block-cyclic distributed Cholesky over RCCL (strong scaling).  Default: "
                        "local on one GPU, distributed on several")
    p.add_argument("--kkt-system", choices=("condensed", "augmented"), default="condensed",
                   help="condensed (headline): K = H + Sigma_x + A' Theta A, Cholesky; augmented: the K2 form "
                        "[H + Sigma_x, A'; A, -D], L diag(I,-I) L' (reported with its own flop count)")
    p.add_argument("--panel-width", type=int, default=None, help="tile size nb of --kkt distributed")
    p.add_argument("--no-independent-leg", action="store_true",
                   help="N > 1: skip the additional measurement of one independent QP per GPU")
    p.add_argument("--extra-timeout", type=float, default=240.0,
                   help="N > 1: seconds after which the shared-QP leg is given up (watchdog, exit code 3)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-batch-extra", action="store_true",
                   help="skip the BASELINE configs[3] measurement (1024 x (512, 256) batch, a few seconds) in `extras`")
    p.add_argument("--no-whole-solve", action="store_true",
                   help="skip `whole_solve`: one complete solve! at the workload size with the reference's own rate "
                        "definitions (iter / total_time incl. initialize!; ~30 s at the metric size)")
    p.add_argument("--no-kernel-timers", action="store_true",
                   help="do not time the MFMA kernel classes with event pairs (roofline.achieved is then 0): the "
                        "pure throughput of small problems, whose factorisation is replayed as a hipGraph only when "
                        "no events sit between its launches")
    p.add_argument("--cpu-sample-nx", type=int, default=8000)
    p.add_argument("--cpu-budget-s", type=float, default=240.0,
                   help="host seconds the cpu_baseline leg may spend on ONE iteration of the CPU port at the full "
                        "workload size, timed inside this run (predicted from the bounded sample; when it does not "
                        "fit, the committed full-size record or the flop-scaled sample is reported and `kind` says so)")
    p.add_argument("--cpu-full", action="store_true",
                   help="only the CPU port at the FULL workload size: initialize + one iteration, timed (minutes); "
                        "prints its own JSON line -- kept under profiles/ and quoted by the default run")
    p.add_argument("--profile-all", action="store_true", help="time every kernel class (perturbs the "
                   "launch-bound ones); default times only the MFMA classes")
    # ranks started by spawn_ranks() find their arguments in the environment: torch.distributed.run's own parser
    # would otherwise claim what it can abbreviate (`--m` is ambiguous between --max-restarts, --master-addr, ...)
    if os.environ.get("MADQP_BENCH_SPAWNED_BY") and os.environ.get("MADQP_BENCH_ARGV") and len(sys.argv) == 1:
        return p.parse_args(json.loads(os.environ["MADQP_BENCH_ARGV"]))
    return p.parse_args()


def launch_plan(gpus, env):
    """What `python bench.py --gpus N` has to do to BE a run on N ranks (VERDICT r3 missing #1: round 3 parsed --gpus and
    never read it; world came from WORLD_SIZE alone, so the un-wrapped form ran ONE rank and printed n_gpus: 1).
      ("run", world)    this process is rank RANK of `world` == gpus ranks (launcher present), or the 1-GPU run;
      ("spawn", gpus)   gpus > 1 and no launcher: start `python -m torch.distributed.run` as a child, pass its line and
                        exit code through (before torch is imported: a process that has touched the GPU starts nothing);
      ("refuse", text)  WORLD_SIZE is set and differs from --gpus: no line, exit code 2."""
    if gpus < 1:
        return "refuse", f"--gpus {gpus}: at least one GPU"
    ws = env.get("WORLD_SIZE")
    if ws is None or ws == "":
        return ("spawn", gpus) if gpus > 1 else ("run", 1)
    try:
        world = int(ws)
    except ValueError:
        return "refuse", f"WORLD_SIZE={ws!r} is not a number"
    if world != gpus:
        return "refuse", (f"--gpus {gpus} but WORLD_SIZE={world}: the launcher started {world} rank(s); a line labelled "
                          f"n_gpus={gpus} would mislabel the run (start `python -m torch.distributed.run --nproc-per-node "
                          f"{gpus} bench.py --gpus {gpus} ...`, or plain `python bench.py --gpus {gpus} ...`)")
    return "run", world


def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(gpus, argv):
    """`python bench.py --gpus N` without a launcher: N ranks of this same script under torch.distributed.run (one per
    GPU, rendezvous on 127.0.0.1), as a CHILD process of this one -- which has imported neither torch nor the library
    and never will.  The child's rank 0 writes the JSON line to the stdout it inherits; the exit code is the child's.
    SIGTERM / SIGINT are handed on to exactly that child."""
    import signal
    import subprocess

    port = os.environ.get("MADQP_BENCH_MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)]
    env = dict(os.environ, MADQP_BENCH_SPAWNED_BY=str(os.getpid()), MADQP_BENCH_ARGV=json.dumps(list(argv)))
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "1")  # what the launcher would set (and say) itself
    print(f"[bench] --gpus {gpus} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env)

    def hand_on(signum, _frame):
        try:
            child.send_signal(signum)
        except OSError:
            pass

    old = {s: signal.signal(s, hand_on) for s in (signal.SIGTERM, signal.SIGINT)}
    try:
        return child.wait()
    finally:
        for s, h in old.items():
            signal.signal(s, h)


def host_cores():
    """Threads the CPU legs use = the cores this process may run on: the affinity mask, cut to the cgroup's CPU quota
    when one is set (a GPU box hands one GPU's job a share of the host).  MADQP_CPU_THREADS overrides."""
    if os.environ.get("MADQP_CPU_THREADS"):
        return max(1, int(os.environ["MADQP_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_full_size(args, M, be):
    """SURVEY.md 8d / BASELINE.md section 3: the CPU port timed AT the metric size -- set-up with its start-point
    factorisation and two solves (src/solver.jl:127-182) and one full iteration (:259-343) of oracle/mpc.py on the
    same inputs bit for bit (generated on the device, copied to the host).  Takes minutes: run on its own
    (`python bench.py --cpu-full`), the result is kept under profiles/ and quoted by the default run."""
    import numpy as np
    import torch

    from oracle import mpc
    from oracle import qp as Q

    nx, m = args.nx, args.m
    cores = host_cores()
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=cores)
    except Exception:
        pass
    note = lambda msg: print(f"[cpu-full {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)
    dq = M.DeviceQP.synthetic(be, args.seed, nx, m)
    h = lambda t: t.cpu().numpy()
    qp = Q.DenseQP(H=h(dq.H), q=h(dq.q), A=h(dq.A), lvar=h(dq.lvar), uvar=h(dq.uvar), lcon=h(dq.lcon),
                   ucon=h(dq.ucon), x0=h(dq.x0))
    del dq
    torch.cuda.empty_cache()
    note(f"inputs on the host (nx={nx}, m={m}), {cores} BLAS threads")
    s = mpc.MPCSolver(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8),
                      step_rule=mpc.AdaptiveStep(0.995), mu_min=1e-12, max_iter=300, max_ncorr=args.max_ncorr)
    t0 = time.perf_counter()
    s.initialize()
    t_init = time.perf_counter() - t0
    note(f"initialize (start-point factorisation + 2 solves): {t_init:.1f} s")
    s.iteration_head()
    t0 = time.perf_counter()
    s.iteration_body()
    s.iteration_head()
    t_iter = time.perf_counter() - t0
    note(f"one iteration: {t_iter:.1f} s")
    flops = m * nx * nx + nx ** 3 / 3.0
    return dict(value=1.0 / t_iter, unit="IPM iterations/s", cores=cores, kind="port", nx=nx, m=m,
                max_ncorr=args.max_ncorr, seconds_per_iteration=t_iter, seconds_initialize=t_init,
                algorithmic_tflops=flops / t_iter * 1e-12, n_factorizations=s.kkt.n_factorizations,
                sample=f"oracle/mpc.py (numpy + scipy LAPACK, {cores} BLAS threads) at the metric size nx={nx}, m={m}: "
                       f"initialize {t_init:.1f} s, one full iteration {t_iter:.1f} s; no extrapolation",
                trace=[{k: float(t[k]) for k in ("k", "inf_pr", "inf_du", "inf_compl", "mu", "alpha_p", "alpha_d")}
                       for t in s.trace])


def cpu_full_size_record(nx, m):
    """The committed result of `bench.py --cpu-full` for this workload (profiles/*cpu_fullsize*.json), if any."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*cpu_fullsize*.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("nx") == nx and d.get("m") == m:
                d = {k: d[k] for k in d if k != "trace"}
                d["source"] = os.path.basename(f)
                return d
        except Exception:
            continue
    return None


def cpu_one_iteration_at_size(args, M, be, cores, note=lambda msg: None):
    """ONE predictor-corrector iteration of the CPU port (oracle/mpc.py) at the FULL workload size, timed inside this
    run: inputs bit for bit those of the device (generated there, copied to the host), start point = the iterate the
    device's init_starting_point computed (oracle.initialize(start=...): equal to the port's own to 1e-9,
    tests/test_gpu_solver.py), so the 90 s start-point factorisation is not paid a second time.  The iteration itself
    -- Sigma update, assembly (5.0e13 flop at the metric size), Cholesky (4.2e13), all solves with their residual
    checks, step lengths, update, model evaluation -- is the port's, untouched."""
    import torch

    from oracle import mpc
    from oracle import qp as Q

    nx, m = args.nx, args.m
    dq = M.DeviceQP.synthetic(be, args.seed, nx, m)
    ds = M.MPCSolver(dq, be, max_iter=300, step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8),
                     mu_min=1e-12, max_ncorr=args.max_ncorr, scaling=True, driver="python")
    ds.initialize()
    h = lambda t: t.detach().cpu().numpy().copy()
    start = dict(x=h(ds.st.x), y=h(ds.st.y), zl=h(ds.st.zl), zu=h(ds.st.zu))
    qp = Q.DenseQP(H=h(dq.H), q=h(dq.q), A=h(dq.A), lvar=h(dq.lvar), uvar=h(dq.uvar), lcon=h(dq.lcon),
                   ucon=h(dq.ucon), x0=h(dq.x0))
    ds.close()
    del ds, dq
    torch.cuda.empty_cache()
    note(f"inputs and the device's start point on the host (nx={nx}, m={m}), {cores} BLAS threads")
    s = mpc.MPCSolver(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8),
                      step_rule=mpc.AdaptiveStep(0.995), mu_min=1e-12, max_iter=300, max_ncorr=args.max_ncorr)
    s.initialize(start=start)
    s.iteration_head()
    t0 = time.perf_counter()
    s.iteration_body()
    s.iteration_head()
    dt = time.perf_counter() - t0
    note(f"one iteration at full size: {dt:.1f} s")
    return dt, s.kkt.n_factorizations


def cpu_baseline(args, nx, m, M=None, be=None):
    """The oracle (numpy/scipy LAPACK port of the same loop) timed on this box's host cores.  `value` is MEASURED at
    the workload size inside this run when one iteration fits --cpu-budget-s (predicted from a bounded sample of the
    same family at a smaller nx); otherwise the committed full-size record (profiles/*cpu_fullsize*.json) or, failing
    that, the flop-scaled sample -- `kind` / `value_source` say which."""
    import numpy as np  # noqa: F401

    from oracle import mpc
    from oracle import qp as Q

    cores = host_cores()
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=cores)
    except Exception:
        pass
    def sample(sn, budget_s, max_iters):
        sm = max(1, int(round(sn * m / nx)))
        qp = Q.synthetic_qp(args.seed, sn, sm)
        s = mpc.MPCSolver(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8),
                          step_rule=mpc.AdaptiveStep(0.995), mu_min=1e-12, max_iter=300, max_ncorr=args.max_ncorr)
        s.initialize()
        s.iteration_head()
        s.iteration_body()  # warm-up
        iters, t0 = 0, time.perf_counter()
        while iters < 3 or (time.perf_counter() - t0 < budget_s and iters < max_iters):
            if s.iteration_head() is not None:
                break
            s.iteration_body()
            iters += 1
        dt = time.perf_counter() - t0
        flops = lambda a, b: b * a * a + a ** 3 / 3.0
        return sn, sm, iters, dt, flops(sn, sm) / flops(nx, m)

    sn, sm, iters, dt0, scale = sample(min(args.cpu_sample_nx, nx), 8.0, 50)
    out = dict(
        value=(iters / dt0) * scale, unit="IPM iterations/s", cores=cores, kind="port",
        sample=(f"oracle/mpc.py (numpy+scipy LAPACK, {cores} BLAS threads) on the same synthetic "
                f"family at nx={sn}, m={sm}: {iters} iterations in {dt0:.2f} s = {iters / dt0:.3f} it/s, "
                f"scaled by the flop ratio (m nx^2 + nx^3/3) {scale:.3e} to nx={nx}, m={m}"),
        measured_it_per_s_at_sample=iters / dt0, sample_nx=sn, sample_m=sm)
    try:  # per-core normalisation (the reference's CPU solver is single threaded)
        threadpool_limits(limits=1)
        sn1, sm1, it1, dt1, sc1 = sample(min(2500, nx), 4.0, 20)
        out["single_thread"] = dict(value=(it1 / dt1) * sc1, cores=1,
                                    sample=f"nx={sn1}, m={sm1}: {it1} iterations in {dt1:.2f} s, flop-scaled {sc1:.3e}")
        threadpool_limits(limits=cores)
    except Exception as e:
        out["single_thread"] = {"error": str(e)[:200]}
    out["value_source"] = "flop-scaled bounded sample"
    out["flop_scaled_from_sample"] = out["value"]
    full = cpu_full_size_record(nx, m)
    if full is not None:  # measured once at the metric size on a GPU box's host (minutes of CPU time): no extrapolation
        out["committed_record_at_metric_size"] = full
    # large DGEMM / DPOTRF run ~2.3x closer to peak than the sample: predict the full-size iteration from the record when
    # there is one, else from the sample
    predicted = full["seconds_per_iteration"] if full and "seconds_per_iteration" in full else 1.0 / (2.3 * out["value"])
    if M is not None and be is not None and sn < nx and predicted <= args.cpu_budget_s:
        try:
            note = lambda msg: print(f"[cpu_baseline {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)
            dt, nf = cpu_one_iteration_at_size(args, M, be, cores, note)
            out.update(value=1.0 / dt, value_source="measured in this run at the workload size",
                       seconds_per_iteration=dt, n_factorizations=nf,
                       algorithmic_tflops=(m * nx * nx + nx ** 3 / 3.0) / dt * 1e-12,
                       sample=(f"oracle/mpc.py (numpy + scipy LAPACK, {cores} BLAS threads) at the workload size "
                               f"nx={nx}, m={m}, timed in this run: ONE full iteration {dt:.1f} s from the start point "
                               f"the device computed (no extrapolation); beside it the bounded sample at nx={sn}: "
                               f"{iters} iterations in {dt0:.2f} s, flop-scaled {out['flop_scaled_from_sample']:.5f} it/s"))
        except Exception as e:  # e.g. host memory: keep what is there
            out["in_run_error"] = f"{type(e).__name__}: {e}"[:300]
    if out["value_source"].startswith("flop") and full is not None:
        out.update(value=full["value"], value_source=f"committed record {full.get('source')} (not re-timed in this run)",
                   kind="port (committed full-size record)")
    return out


def pmc_traffic(nx, m):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_summary.json, FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), if a
    summary for this workload exists; PMC counters cannot be collected inside the timed run."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("config", {}).get("nx") != nx or d.get("config", {}).get("m") != m:
                continue
            g = d["gemm_tn_f64_kernel"]
            return ((g["hbm_read_GB_corrected"] + g["hbm_write_GB"]) * 1e9 / g["dispatches"], os.path.basename(f),
                    d.get("mfma_busy"))
        except Exception:
            continue
    return None, None, None


def hbm_traffic(which, nx, m):
    """Counter bytes per launch of the sweeps from the committed PMC summary (hbm_bound_kernels), if any."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("config", {}).get("nx") != nx or d.get("config", {}).get("m") != m:
                continue
            ks = [v for k, v in d.get("hbm_bound_kernels", {}).items() if k.startswith(which)]
            if ks:
                return sum(k["hbm_GB_per_launch"] for k in ks) / len(ks) * 1e9
        except Exception:
            continue
    return None


def batch_extra(M, be, seed):
    """BASELINE configs[3] on this GPU, beside the headline: 1024 independent QPs (n_x = 512, m = 256) through the
    lock-step batched engine (csrc/batch.hip) -- QP/s (set-up, all iterations and the read-back inside the timed region,
    median of three solves after a full-size warm-up) and where that stands against the two bounds of the chip
    (tools/bench_batch.py: batch_roofline)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_batch

        dt, res, times = bench_batch.run_batched(M, be, 1024, 512, 256, seed + 2)
        iters = sum(r["iter"] for r in res)
        return {"metric": "independent QPs solved per second", "value": 1024 / dt, "unit": "QP/s", "seconds": dt,
                "ipm_iterations_per_s": iters / dt, "solved": sum(r["status"] == M.SOLVE_SUCCEEDED for r in res),
                "lock_step_iterations": int(max(r["iter"] for r in res)), "all_seconds": times,
                "roofline": bench_batch.batch_roofline(512, 256, iters, dt),
                "profile": "profiles/r03_batch1024_* (kernel stats, FETCH_SIZE / WRITE_SIZE)"}
    except Exception as e:  # the headline must not depend on it
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def c2_extra(args, M, be):
    """BASELINE configs[1] on this GPU, beside the headline (the number existed only in the builder's
    own records): synthetic dense QP n_x = 5 000, m = 2 000 with SURVEY.md 8d's "dummy" Hessian family (H = G'G + 100 I,
    MadNLPTests.DenseDummyQP of test/runtests.jl:9), the scripts' options, the same step definition as the headline --
    (a) 60 timed iterations after 10 warm-up iterations with no kernel timers (an event pair around each of the ~80
    short launches of a mid-size factorisation perturbs it); (b) a few more with the MFMA classes timed: the
    factorisation's and the assembly's TFLOP/s against the fp64 MFMA peak; (c) one complete solve! by the reference's
    definition (iter / total_time, src/solver.jl:353,392)."""
    try:
        nx, m = 5000, 2000
        dq = M.DeviceQP.synthetic(be, args.seed, nx, m, "dummy")
        opts = dict(max_iter=300, step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8),
                    mu_min=1e-12, max_ncorr=args.max_ncorr, scaling=True, driver=args.driver)
        import torch

        def timed(steps, warmup, classes):
            solver = M.MPCSolver(dq, be, **opts)
            solver.initialize()
            loop = StepLoop(solver, torch.cuda.synchronize)
            for _ in range(warmup):
                loop.step()
            be.prof_enable(classes)
            be.prof_reset()
            loop.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loop.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0 - loop.excluded
            prof = be.prof_get()
            be.prof_enable(())
            nf = loop.factorizations()
            solver.close()
            return dt, prof, nf

        dt, _, nf = timed(60, 10, ())
        dt2, prof, nf2 = timed(20, 5, ("syrk", "potrf_gemm", "potrf_trsm", "potrf_diag"))
        factor_ms = (prof["potrf_gemm"][0] + prof["potrf_trsm"][0] + prof["potrf_diag"][0]) / max(nf2, 1)
        syrk_ms = prof["syrk"][0] / max(nf2, 1)
        solver = M.MPCSolver(dq, be, **opts)
        torch.cuda.synchronize()
        r = solver.solve()
        torch.cuda.synchronize()
        solver.close()
        tf = lambda flop, ms: (flop / (ms * 1e-3) * 1e-12) if ms > 0 else None
        fr = lambda x: (x / PEAK_F64_MFMA_TFLOPS) if x else None
        return {"metric": "IPM iterations/sec, value = steps / time of the timed iterations (initialize! excluded)",
                "workload": f"synthetic dense QP nx={nx} m={m} (0<=x<=1, 0<=Ax<=1, dummy H = G'G + 100 I, Gaussian A), "
                            f"max_ncorr={args.max_ncorr}",
                "value": 60 / dt, "unit": "iterations/s", "ms_per_step": dt / 60 * 1e3, "steps": 60, "warmup": 10,
                "factorizations": nf, "kernel_timers": "off in the timed region of `value`",
                "with_mfma_timers": {"ms_per_step": dt2 / 20 * 1e3, "factor_potrf_ms": factor_ms, "assemble_syrk_ms": syrk_ms,
                                     "factor_tflops": tf(nx ** 3 / 3.0, factor_ms),
                                     "factor_frac_of_fp64_mfma_peak": fr(tf(nx ** 3 / 3.0, factor_ms)),
                                     "assemble_tflops": tf(m * nx * nx, syrk_ms),
                                     "assemble_frac_of_fp64_mfma_peak": fr(tf(m * nx * nx, syrk_ms))},
                "whole_solve": {"status": int(r["status"]), "iter": int(r["iter"]), "total_time_s": r["total_time"],
                                "iterations_per_s": r["iter"] / r["total_time"] if r["total_time"] > 0 else None,
                                "n_factorizations": int(r["n_factorizations"]), "objective": float(r["objective"])}}
    except Exception as e:  # the headline must not depend on it
        return {"error": f"{type(e).__name__}: {e}"[:300]}


class Product:
    """What is measured: the HIP library behind madqp_jl_amd (no fallback: the backend raises without a GPU)."""

    name = None  # not a test double
    cuda = True

    def module(self):
        import madqp_jl_amd as M

        return M

    def backend(self, local_rank):
        return self.module().HipBackend(local_rank)

    def local_qp(self, be, seed, nx, m):
        return self.module().DeviceQP.synthetic(be, seed, nx, m)

    def shared_qp(self, be, world, seed, nx, m, nb):
        """ONE QP over all ranks: (grid handle, the pieces this rank holds)."""
        from madqp_jl_amd import dist2d

        comm = None
        if world > 1 and os.environ.get("MADQP_DIST_BACKEND", "nccl") != "nccl":
            comm = dist2d.HostStagedComm(*dist2d.default_grid(world))  # rehearsal: several ranks on one GPU over gloo
        grid = dist2d.DistCholesky2D(be, nx, nb, None, comm)
        return grid, dist2d.DistributedQP.synthetic(be, grid, seed, nx, m)

    def sync(self):
        import torch

        torch.cuda.synchronize()


def load_factory():
    """The product -- unless MADQP_BENCH_TEST_DOUBLE="module:attr" names a TEST DOUBLE with the same five methods
    (tests/bench_double.py: numpy stand-ins on the CPU).  That exists for ONE purpose: tests/test_bench.py runs this
    file's launcher and multi-rank scaffolding end to end (`python bench.py --gpus 2`, no launcher, no GPU) and reads
    the line.  The line then says `test_double` and `data: "TEST DOUBLE ..."`, and on a box that HAS a GPU the variable
    is refused (exit code 2): nothing the driver measures can be a double."""
    spec = os.environ.get("MADQP_BENCH_TEST_DOUBLE")
    if not spec:
        return Product()
    import importlib

    import torch

    if torch.cuda.is_available():
        print("[bench] MADQP_BENCH_TEST_DOUBLE is set on a box with a GPU: refused (measure the product)", file=sys.stderr)
        sys.exit(2)
    mod, _, attr = spec.partition(":")
    return getattr(importlib.import_module(mod), attr or "Double")()


def dist_setup(backend="nccl", share_device=False):
    """One process per GPU (torch.distributed.run): returns (world, rank, local_rank).
    ``share_device``: rehearsal with all ranks on device 0 (gloo only)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl" and not share_device:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return world, rank, local_rank


def dist_barrier(world, cuda=True):
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


def max_over_ranks(elapsed, world, device):
    """The job's time is the slowest rank's (contract: MAX over ranks)."""
    import torch
    import torch.distributed as dist

    if world <= 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_value(world, steps, tmax):
    """Whole-job throughput: every rank runs `steps` iterations of its own QP (weak scaling)."""
    return world * steps / tmax


def rank_seed(seed, rank):
    """Independent instances: rank r solves the QP of seed + r."""
    return seed + rank


class StepLoop:
    """The timed unit and its bookkeeping.  A step = one predictor-corrector iteration (iteration_head +
    iteration_body).  When the solve converges inside the loop the same instance is started again; the time of
    that re-initialisation is excluded from the iteration time, but its start-point factorisation (src/solver.jl:21)
    runs the same kernels under the same timers, so it COUNTS as a factorisation -- `factorizations()` reads the
    solver's monotone counter (MPCSolver.n_factorizations_total), never a field of the KKT object that
    initialize() replaces."""

    def __init__(self, solver, sync=lambda: None):
        self.solver, self.sync = solver, sync
        self.reset()

    def reset(self):
        self.excluded, self.reinits, self.steps = 0.0, 0, 0
        self._f0 = self.solver.n_factorizations_total

    def factorizations(self):
        return self.solver.n_factorizations_total - self._f0

    def step(self):
        s = self.solver
        if s.iteration_head() is not None:  # converged: start the same instance again
            self.sync()
            t = time.perf_counter()
            s.initialize()
            s.iteration_head()
            self.sync()
            self.excluded += time.perf_counter() - t
            self.reinits += 1
        s.iteration_body()
        self.steps += 1


def grid_layout(grid, dq):
    """What rank 0 holds of the shared QP, in bytes, by part (csrc/dist_core.inc::create, csrc/dist.hip::madqp_dkkt_create).
    The lazy left-looking updates keep the operands of ALL steps: XW (this rank's tile ROWS of L) and YW (its tile COLUMNS
    of L) -- the factor is replicated Q-fold along process rows and P-fold along process columns, stored as staircases
    (rows above a block's diagonal tile are never read and are not stored)."""
    info = grid.memory() if hasattr(grid, "memory") else {}
    parts = dict(K=8 * grid.ld * grid.ncp, H=8 * grid.ld * grid.ncp if dq.H is not None else 0,
                 A_I=8 * dq.A_I.numel(), A_J=8 * dq.A_J.numel(), theta_A_J=8 * dq.A_J.numel(),
                 stored_operands_XW=int(info.get("xw_bytes", 0)), stored_operands_YW=int(info.get("yw_bytes", 0)))
    return dict(kind="P x Q block-cyclic: K/L and H 1/(PQ) each, A (1/P + 1/Q), stored operands of the lazy updates "
                     "(1/P + 1/Q) n^2/2 as staircases -- the factor IS replicated Q-fold along rows, P-fold along columns",
                grid=[grid.P, grid.Q], tile=grid.nb, local_matrix=[grid.mloc, grid.nloc],
                rank0_bytes_by_part=parts, rank0_matrix_bytes=int(sum(parts.values())),
                rank0_library_bytes=int(info.get("total_bytes", 0)),
                bytes_broadcast_by_rank0=grid.bytes_sent(),
                comm=grid.comm_info() if hasattr(grid, "comm_info") else None)


def solver_options(M, args, max_ncorr, mode):
    """options of scripts/benchmarks_cpu.jl:35-44 (kkt_system -> condensed, linear_solver -> HIP Cholesky)"""
    return dict(max_iter=300, step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8),
                mu_min=1e-12, max_ncorr=max_ncorr, scaling=True,
                driver=args.driver if mode == "local" else "python", kkt_system=args.kkt_system)


def measure(args, F, be, world, seed, mode, max_ncorr=None, steps=None, warmup=None, progress=lambda: None):
    """Warm-up + the timed region (barrier / sync on both sides, MAX over ranks) for one solver set-up.
    mode "local": this rank's own QP; "grid": ONE QP over all ranks on a P x Q grid (madqp_dist_* / madqp_dkkt_*,
    SURVEY.md 8e)."""
    M = F.module()
    nx, m = args.nx, args.m
    max_ncorr = args.max_ncorr if max_ncorr is None else max_ncorr
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    grid = None
    if mode == "grid":
        grid, dq = F.shared_qp(be, world, seed, nx, m, args.panel_width)
    else:
        dq = F.local_qp(be, seed, nx, m)
    solver = M.MPCSolver(dq, be, **solver_options(M, args, max_ncorr, mode))
    progress()
    solver.initialize()
    progress()
    loop = StepLoop(solver, F.sync)
    for _ in range(warmup):
        loop.step()
        progress()
    mfma_classes = ("syrk", "potrf_gemm", "potrf_trsm", "potrf_diag")
    if nx >= 20000:  # millisecond-scale sweeps: an event pair around them perturbs nothing
        mfma_classes += ("trsv",)
    be.prof_enable(() if args.no_kernel_timers else (M._lib.PROF_CLASSES if args.profile_all else mfma_classes))
    be.prof_reset()
    loop.reset()
    dist_barrier(world, cuda=F.cuda)
    t0 = time.perf_counter()
    for _ in range(steps):
        loop.step()
        progress()  # (a host-side timer reset: no device work, no synchronisation)
    dist_barrier(world, cuda=F.cuda)
    elapsed = time.perf_counter() - t0 - loop.excluded
    prof = be.prof_get()
    be.prof_enable(())
    res = dict(tmax=max_over_ranks(elapsed, world, be.device), prof=prof, mode=mode, max_ncorr=max_ncorr,
               nfact=loop.factorizations(), reinits=loop.reinits, steps=steps, warmup=warmup, k=solver.k,
               last_trace={k: solver.trace[-1][k] for k in ("k", "inf_pr", "inf_du", "inf_compl", "mu")}
               if solver.trace else None)
    if mode == "grid":
        res["layout"] = grid_layout(grid, dq)
    solver.close()
    if grid is not None:
        grid.close()
    del solver, dq
    if F.cuda:
        import torch

        torch.cuda.empty_cache()
    return res


def whole_solve(args, F, be):
    """The reference's OWN rate definitions (VERDICT r3 missing #4): ONE complete `solve!` at the workload
    size -- `initialize!` (scaling, KKT object, the start point's factorisation and two solves, src/solver.jl:127-182)
    plus every iteration up to the termination test -- timed as src/solver.jl:353,392 does (`counters.total_time`), with
        iterations/s       = iter / total_time                                  (scripts/benchmarks_cpu.jl:52-55)
        linear-solver ms   = linear_solver_time / #factorizations, #factorizations = iter + 1 + retries
    where linear_solver_time is what MadNLP's `factorize_wrapper!` accumulates: the `factorize!` calls alone (build_kkt!
    and the triangular solves are outside that counter; they are reported beside it).  The timed region of the headline
    (`value`) excludes initialisation; this entry is the whole thing."""
    M = F.module()
    nx, m = args.nx, args.m
    dq = F.local_qp(be, args.seed, nx, m)
    solver = M.MPCSolver(dq, be, **solver_options(M, args, args.max_ncorr, "local"))
    classes = ("syrk", "potrf_gemm", "potrf_trsm", "potrf_diag") + (("trsv",) if nx >= 20000 else ())
    be.prof_enable(() if args.no_kernel_timers else classes)
    be.prof_reset()
    F.sync()
    t0 = time.perf_counter()
    r = solver.solve()
    F.sync()
    wall = time.perf_counter() - t0
    prof = be.prof_get()
    be.prof_enable(())
    nf = max(int(r["n_factorizations"]), 1)
    timed = not args.no_kernel_timers  # (classes that were not timed are reported as null, never as 0.0)
    trsv_timed = timed and "trsv" in classes
    factor_ms = prof["potrf_gemm"][0] + prof["potrf_trsm"][0] + prof["potrf_diag"][0]
    out = {"what": "ONE complete solve! (initialize! + all iterations), the reference's definitions: iter / "
                   "counters.total_time, counters.linear_solver_time / #factorizations (src/solver.jl:353,392; "
                   "scripts/benchmarks_cpu.jl:52-55; MadNLP.factorize_wrapper! times factorize! only)",
           "status": int(r["status"]), "solved": bool(r["status"] == M.SOLVE_SUCCEEDED), "iter": int(r["iter"]),
           "total_time_s": r["total_time"], "wall_s": wall,
           "iterations_per_s": r["iter"] / r["total_time"] if r["total_time"] > 0 else None,
           "n_factorizations": int(r["n_factorizations"]),
           "linear_solver_time_s": factor_ms * 1e-3 if timed else None,
           "linear_solver_ms_per_factorization": factor_ms / nf if timed else None,
           "build_kkt_ms_per_factorization": prof["syrk"][0] / nf if timed else None,
           "solve_sweeps_ms_per_factorization": (prof["trsv"][0] / nf) if trsv_timed else None,
           # assembly + factorisation + the sweeps of every solve that follows it; null when the sweeps were not timed
           # (below nx = 20 000 an event pair around a 100 us sweep perturbs the iteration)
           "kkt_factor_solve_ms_per_factorization": ((factor_ms + prof["syrk"][0] + prof["trsv"][0]) / nf) if trsv_timed else None,
           "kkt_build_factor_ms_per_factorization": ((factor_ms + prof["syrk"][0]) / nf) if timed else None,
           "objective": float(r["objective"]), "primal_feas": float(r["primal_feas"]), "dual_feas": float(r["dual_feas"])}
    solver.close()
    del solver, dq
    if F.cuda:
        import torch

        torch.cuda.empty_cache()
    return out


def bench_line(args, res, world, F=None):
    """The JSON body of one measurement (rank 0)."""
    nx, m = args.nx, args.m
    tmax, prof, nfact, mode, steps = res["tmax"], res["prof"], res["nfact"], res["mode"], res["steps"]
    shared = mode != "local"
    # dominant kernel: gemm_tn_f64_kernel (assembly + panel updates + panel x inverse block)
    gemm_ms = prof["syrk"][0] + prof["potrf_gemm"][0] + prof["potrf_trsm"][0]
    gemm_launches = prof["syrk"][1] + prof["potrf_gemm"][1] + prof["potrf_trsm"][1]
    alg_flops = nfact * (m * nx * nx + nx ** 3 / 3.0)  # SYRK m nx^2 + POTRF nx^3/3
    if args.kkt_system == "augmented":  # no SYRK; L diag(I,-I) L' of order nx + m
        alg_flops = nfact * (nx + m) ** 3 / 3.0
    if shared:
        alg_flops /= world  # rank 0's share of the MFMA work (cyclic deal of the tiles)
    achieved = alg_flops / (gemm_ms * 1e-3) * 1e-12 if gemm_ms > 0 else 0.0
    traffic, traffic_src, mfma_busy = pmc_traffic(nx, m)
    if shared:
        traffic, traffic_src, mfma_busy = None, None, None
    what = {"local": "one independent QP per GPU",
            "grid": "ONE QP shared by all GPUs: 2-D block-cyclic distributed assembly + Cholesky + solves over RCCL "
                    "(csrc/dist.hip)"}[mode]
    par = "independent" if mode == "local" else "grid %dx%d nb=%d" % (*res["layout"]["grid"], res["layout"]["tile"])
    out = {
        # `value` = timed iterations / their time (a step is one iteration; initialize! is outside the timed region).  The
        # reference's own rate, iter / counters.total_time of ONE complete solve! with initialize! inside
        # (src/solver.jl:353,392), is `value_reference_definition` = whole_solve.iterations_per_s below.
        "metric": ("IPM iterations/sec (Mehrotra predictor-corrector, condensed KKT + Cholesky), dense QP fp64; value = "
                   "steps / time of the timed iterations, initialize! excluded (value_reference_definition: iter / total_time)"
                   if args.kkt_system == "condensed" else
                   "IPM iterations/sec (Mehrotra predictor-corrector, augmented K2 KKT, L diag(I,-I) L'), dense QP fp64; value "
                   "= steps / time of the timed iterations, initialize! excluded"),
        "value": (steps / tmax) if shared else job_value(world, steps, tmax),
        "unit": "iterations/s",
        "n_gpus": world,
        "n_gpus_requested": args.gpus,
        "ranks": {"world_size": world, "started_by": ("bench.py (python -m torch.distributed.run as a child process)"
                                                      if os.environ.get("MADQP_BENCH_SPAWNED_BY") else
                                                      "an external launcher" if world > 1 else "this process alone")},
        "steps": steps,
        "warmup": res["warmup"],
        "ms_per_step": tmax / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if shared else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"synthetic dense QP nx={nx} m={m} (0<=x<=1, 0<=Ax<=1, Wigner H, Gaussian A), "
                               f"{what}, max_ncorr={res['max_ncorr']}",
                   "nx": nx, "m": m, "n_slack": m, "max_ncorr": res["max_ncorr"], "parallelism": par,
                   "driver": args.driver if mode == "local" else "python", "kkt": mode,
                   "kkt_system": args.kkt_system,
                   "options": f"max_ncorr={res['max_ncorr']}, FixedRegularization(1e-8,-1e-8), AdaptiveStep(0.995), "
                              "mu_min=1e-12, max_iter=300 (scripts/benchmarks_cpu.jl:35-44, whose max_ncorr is 3) "
                              "with kkt_system="
                              + ("HIPAugmentedKKTSystem" if args.kkt_system == "augmented" else "HIPCondensedKKTSystem")
                              + ", linear_solver=HIPCholeskySolver"},
        # per factorisation (linear_solver_time / #factorizations, #factorizations = iterations +
        # start points + x100 retries); solve = the triangular sweeps of all solves that follow one factorisation
        "kkt_factor_solve_ms": {
            "assemble_syrk": prof["syrk"][0] / max(nfact, 1),
            "factor_potrf": (prof["potrf_gemm"][0] + prof["potrf_trsm"][0] + prof["potrf_diag"][0]) / max(nfact, 1),
            "solve_trsv": (prof["trsv"][0] / max(nfact, 1)) if prof["trsv"][1] else None,
            "solves": prof["trsv"][1],
            "factorizations": nfact,
            "reinitializations_in_timed_region": res["reinits"],
            "iteration_total": tmax / steps * 1e3,
        },
        "roofline": {
            "bound": "mfma", "kernel": "gemm_tn_f64_kernel",
            "achieved": achieved, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            # MFMA-busy counter fractions of the committed PMC passes (same command under rocprofv3 --pmc): the assembly
            # SYRK and the Cholesky PANEL GEMM (north_star: "MFMA utilisation on the panel GEMM against chip peak")
            "mfma_busy_counter": mfma_busy,
            "launches": gemm_launches, "avg_launch_ms": gemm_ms / max(gemm_launches, 1),
            "algorithmic_flops_per_launch": alg_flops / max(gemm_launches, 1),
            "split": {k: {"ms": prof[k][0], "launches": prof[k][1]} for k in prof if prof[k][1]},
        },
        "iterations_done": res["k"],
        "last_trace": res["last_trace"],
    }
    if prof["trsv"][1] and mode == "local" and args.kkt_system == "condensed":
        # second bound: the triangular sweeps stream the factor once each, 4 nx^2 bytes per sweep, two
        # sweeps per solve; timed with event pairs on the launch stream like the MFMA classes
        sweeps = 2 * prof["trsv"][1]
        gbs = sweeps * 4.0 * nx * nx / (prof["trsv"][0] * 1e-3) * 1e-9
        out["roofline_hbm"] = {"bound": "hbm", "kernel": "trsv_fwd_sweep_kernel + trsv_bwd_sweep_kernel",
                               "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                               "algorithmic_bytes_per_launch": 4.0 * nx * nx, "launches": sweeps,
                               "avg_launch_ms": prof["trsv"][0] / sweeps, "traffic": hbm_traffic("trsv", nx, m)}
    if F is not None and F.name:
        out["test_double"] = F.name
        out["data"] = "TEST DOUBLE on the CPU (tests only): not a measurement of the product"
    if shared:
        out["distributed"] = res["layout"]
        # the communicators the LIBRARY built (madqp_dist_comm_info): a line labelled N GPUs ran its collectives on N ranks
        out["comm"] = res["layout"].get("comm")
        # whole-job rate against the chip peaks of all GPUs: what the scaling curve is judged by
        out["roofline"]["job_fraction_of_peak"] = (
            (nfact * (m * nx * nx + nx ** 3 / 3.0)) / tmax * 1e-12 / (PEAK_F64_MFMA_TFLOPS * world))
    return out


_REPORTER_SRC = r"""
import sys
held = None
for line in sys.stdin:
    line = line.rstrip("\n")
    if line == "DONE":
        sys.exit(0)
    if line.startswith("HOLD "):
        held = line[5:]
if held is not None:  # the pipe closed without DONE: the bench process is gone (crash, or killed with its group)
    sys.stdout.write(held + "\n")
    sys.stdout.flush()
"""


class Watchdog:
    """Calls `on_silence` when nothing has kicked it for `seconds`."""

    def __init__(self, seconds, on_silence):
        import threading

        self.seconds, self.on_silence, self._t, self._threading = seconds, on_silence, None, threading
        self.kick()

    def kick(self):
        self.stop()
        self._t = self._threading.Timer(self.seconds, self.on_silence)
        self._t.daemon = True
        self._t.start()

    def stop(self):
        if self._t is not None:
            self._t.cancel()
            self._t = None


class LastResortReporter:
    """A child process of rank 0, started BEFORE anything initialises the GPU (a process that has touched the GPU
    must not start programs on this pool).  It holds the JSON line of what is already measured; if the bench process
    disappears before saying DONE -- a fault inside the never-before-run multi-GPU leg, or the launcher tearing the
    group down because another rank died -- the child prints that line, so that the run still leaves its one line."""

    def __init__(self, enabled):
        import subprocess
        self.p = None
        if enabled:
            try:
                self.p = subprocess.Popen([sys.executable, "-c", _REPORTER_SRC], stdin=subprocess.PIPE, text=True)
            except OSError:
                self.p = None

    def _send(self, text):
        if self.p is not None and self.p.stdin is not None:
            try:
                self.p.stdin.write(text + "\n")
                self.p.stdin.flush()
            except (BrokenPipeError, OSError):
                self.p = None

    def hold(self, obj):
        self._send("HOLD " + json.dumps(obj))

    def done(self):
        self._send("DONE")
        if self.p is not None:
            try:
                self.p.stdin.close()
                self.p.wait(timeout=5)
            except Exception:
                pass
            self.p = None


_REAL_STDOUT = None


def protect_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner when
    its first communicator comes up): from here on file descriptor 1 is stderr, and only `emit_json` reaches the
    stdout the launcher gave this process."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_json(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def main():
    args = parse()
    plan, detail = launch_plan(args.gpus, os.environ)
    if plan == "refuse":
        print(f"[bench] refused: {detail}", file=sys.stderr, flush=True)
        sys.exit(2)
    if plan == "spawn":  # nothing below has run: no torch, no library, no GPU in this process
        sys.exit(spawn_ranks(detail, sys.argv[1:]))
    reporter = LastResortReporter(int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0)
    protect_stdout()  # (after the reporter child exists: it keeps the real stdout for the held line)
    F = load_factory()
    import torch.distributed as dist

    world, rank, local_rank = dist_setup(os.environ.get("MADQP_DIST_BACKEND", "nccl" if F.cuda else "gloo"),
                                         share_device=bool(os.environ.get("MADQP_DIST_SHARE_DEVICE")) or not F.cuda)
    assert world == args.gpus, (world, args.gpus)  # launch_plan saw to it
    if os.environ.get("MADQP_DIST_SHARE_DEVICE"):
        local_rank = 0  # rehearsal: several ranks on one GPU (gloo)

    M = F.module()
    be = F.backend(local_rank)
    nx, m = args.nx, args.m
    if args.cpu_full:
        emit_json(cpu_full_size(args, M, be))
        be.close()
        return
    mode = {"local": "local", "distributed": "grid", None: "local" if world == 1 else "grid"}[args.kkt]
    if args.kkt_system != "condensed" and mode != "local":
        mode = "local"  # the augmented system is factorised on one GPU
    out = None

    def second_ncorr():
        # both max_ncorr settings are reported.  Same barrier / MAX-over-ranks protocol, fewer steps.
        other = 0 if args.max_ncorr else 3
        s2 = max(1, min(args.steps, 5))
        r2 = measure(args, F, be, world, rank_seed(args.seed, rank), "local", max_ncorr=other, steps=s2,
                     warmup=min(args.warmup, 1))
        if rank == 0:
            out[f"max_ncorr_{other}"] = {"value": job_value(world, s2, r2["tmax"]), "unit": "iterations/s",
                                         "ms_per_step": r2["tmax"] / s2 * 1e3, "steps": s2,
                                         "factorizations": r2["nfact"], "solves": r2["prof"]["trsv"][1]}

    if world == 1 or mode == "local":
        res = measure(args, F, be, world, args.seed if mode != "local" else rank_seed(args.seed, rank), mode)
        if rank == 0:
            out = bench_line(args, res, world, F)
        if mode == "local" and not args.no_second_ncorr:
            second_ncorr()
        if rank == 0 and world == 1 and mode == "local" and not args.no_whole_solve:
            try:
                out["whole_solve"] = whole_solve(args, F, be)
            except Exception as e:  # the headline must not depend on it
                out["whole_solve"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            out["value_reference_definition"] = out["whole_solve"].get("iterations_per_s")
        if rank == 0 and world == 1 and mode == "local" and not args.no_batch_extra and F.cuda:
            out["extras"] = {"c2_nx5000_m2000": c2_extra(args, M, be), "batch_1024x512x256": batch_extra(M, be, args.seed)}
        if rank == 0 and world == 1 and not args.no_cpu_baseline and F.cuda:
            out["cpu_baseline"] = cpu_baseline(args, nx, m, M, be)
        if rank == 0:
            emit_json(out)
    else:
        # N > 1: the headline is ONE QP over all GPUs (strong scaling, north_star's 2-D block-cyclic distributed
        # Cholesky).  The independent-QPs number (one QP per GPU, no collective in the data path) is measured first and
        # reported beside it.  A collective that never returns raises nothing, so a watchdog guards the shared leg: it
        # prints the line with what is complete (headline = the independent-QPs measurement, an `error` entry for the
        # shared leg) and ends EVERY rank with exit code 3; nothing is retried in this process.
        import threading

        weak = None
        if not args.no_independent_leg:
            weak = measure(args, F, be, world, rank_seed(args.seed, rank), "local", steps=max(1, min(args.steps, 5)),
                           warmup=min(args.warmup, 1))
        state = {"printed": False}
        lock = threading.Lock()

        def emit(obj):
            with lock:
                if state["printed"]:
                    return
                state["printed"] = True
                if rank == 0:
                    emit_json(obj)
                reporter.done()

        def fallback(msg):
            # NOT a scaling result: the shared-QP (strong scaling) leg failed; what is printed is the independent-QPs
            # measurement, flagged, with a non-zero exit code
            if weak is None:
                return {"error": msg, "n_gpus": world, "n_gpus_requested": args.gpus}
            o = bench_line(args, weak, world, F) if rank == 0 else {}
            o["distributed_kkt"] = {"error": msg}
            o["error"] = "shared-QP leg failed: this line is the independent-QPs (weak scaling) fallback, not the headline"
            return o

        def bail():
            emit(fallback(f"the shared-QP leg gave no result after {args.extra_timeout} s; exit code 3"))
            os._exit(3)

        if rank == 0 and weak is not None:  # what a crash of the shared leg must not take with it
            reporter.hold(fallback("the process ended inside the shared-QP leg (fault or killed with its group)"))
        # a hung collective raises nothing: the watchdog is re-armed at every sign of progress (set-up done, every step),
        # so it measures silence, not the length of the run
        dog = Watchdog(args.extra_timeout, bail)
        try:
            res = measure(args, F, be, world, args.seed, mode, progress=dog.kick)
        except Exception as e:
            dog.stop()
            emit(fallback(f"{type(e).__name__}: {e}"[:400] + "; exit code 3"))
            os._exit(3)
        dog.stop()
        if rank == 0:
            out = bench_line(args, res, world, F)
            comm = (out.get("comm") or {})
            if mode == "grid" and comm.get("world_size") not in (None, world):  # cannot happen; never mislabel if it does
                emit(fallback(f"the library's world communicator has {comm.get('world_size')} ranks, not {world}"))
                os._exit(3)
            if weak is not None:
                out["independent_qps"] = {"what": "one independent QP per GPU, no collective in the data path (weak scaling)",
                                          "value": job_value(world, weak["steps"], weak["tmax"]), "unit": "iterations/s",
                                          "ms_per_step": weak["tmax"] / weak["steps"] * 1e3, "steps": weak["steps"]}
        emit(out)

    reporter.done()
    be.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
