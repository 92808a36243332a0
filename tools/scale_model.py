#!/usr/bin/env python3
"""The 1 -> 8 GPU model of DESIGN.md section 7 ("What to expect"), written out so that a measured SCALE_rNN.json can be held
against it.  NOT a measurement: every input below is either measured on ONE MI355X (profiles/r04_*,
profiles/r05_*) or a stated assumption about links nobody in the build loop has seen (xGMI: 153 GB/s per link and direction,
SURVEY.md 5/8e).  python tools/scale_model.py > profiles/r05_scale_model.json"""
import json

N, M, NB = 50000, 20000, 1024
ONE_GPU = dict(assembly_ms=664.0, factor_ms=607.0, solves_vec_ms=43.0)   # BENCH_r04 / r5 bench: per iteration, max_ncorr = 3
GEMM_TFLOPS_WIDE = 72.0         # wide-K updates of the one-GPU factorisation (profiles/r04_gemm_launch_breakdown.txt)
GEMM_TFLOPS_GRID = 66.0         # the grid's 1024-wide lazy updates (1 x 1 grid: 608 against 585 ms)
DIAG_TILE_MS = 0.62             # blocked Cholesky of one 1024 x 1024 diagonal tile (49 tiles: ~30 ms per factorisation on one GPU)
PANEL_SOLVE_MS_FULL = 12.7      # all panel solves of a factorisation on one GPU (1 x 1 grid, round 3)
LINK_GBS, LINK_EFF, COLL_LAT_MS = 153.0, 0.7, 0.03   # ASSUMPTIONS: per-link rate, achieved fraction, latency of one collective
FREE_SLOTS_COST = 0.035         # capped persistent GEMM that leaves 16 workgroup slots to RCCL (measured on one GPU)
SOLVE_GROUPS, VEC_ALLREDUCE_PER_ITER = 13, 30

def model(P, Q):
    n_gpu, T = P * Q, (N + NB - 1) // NB
    if n_gpu == 1:
        t = sum(ONE_GPU.values())
        return dict(grid=[1, 1], ms_per_iteration=t, parts=ONE_GPU, iterations_per_s=1e3 / t)
    assembly = ONE_GPU["assembly_ms"] / n_gpu * 1.01  # no communication: K_loc = H_loc + A_I' (Theta A_J), masked tile table
    flops = N ** 3 / 3.0
    bw = LINK_GBS * LINK_EFF * 1e9
    chain = exposed = update = 0.0
    for k in range(T):
        rows_below = N - k * NB
        upd = (rows_below ** 2 * NB) / n_gpu / (GEMM_TFLOPS_GRID * 1e12) * 1e3 * (1 + FREE_SLOTS_COST)
        diag_bc = (10e6 / bw * 1e3 + COLL_LAT_MS) if P > 1 else 0.0
        panel = PANEL_SOLVE_MS_FULL * (rows_below / N) / (T / 2.0) / P
        row_bc = ((rows_below / P) * NB * 8 / bw * 1e3 + COLL_LAT_MS) if Q > 1 else 0.0
        col_bc = ((rows_below / Q) * NB * 8 / bw * 1e3 + COLL_LAT_MS * P) if P > 1 else 0.0
        c = DIAG_TILE_MS + diag_bc + panel + row_bc + col_bc
        chain += c
        update += upd
        exposed += max(c, upd)      # look-ahead 1: the panel phase of step k+1 and its broadcasts travel beside update k
    solves = 4 * 2 * (SOLVE_GROUPS * (2 * COLL_LAT_MS + 0.05)) + ONE_GPU["solves_vec_ms"] / n_gpu * 0.5  # grouped sweeps: 2 collectives per group
    vec = 20.0 / n_gpu + VEC_ALLREDUCE_PER_ITER * COLL_LAT_MS + ONE_GPU["solves_vec_ms"] * 0.3
    t = assembly + exposed + solves + vec
    return dict(grid=[P, Q], ms_per_iteration=t, iterations_per_s=1e3 / t,
                parts=dict(assembly_ms=assembly, factor_ms=exposed, factor_chain_only_ms=chain, factor_update_only_ms=update,
                           solves_ms=solves, vector_and_matvec_ms=vec),
                wire_GB_per_rank_and_factorisation=N * N / 2 * (1.0 / P + 1.0 / Q) * 8 / 1e9,
                flops_per_rank=(M * N * N + flops) / n_gpu)

out = dict(what="MODEL, not a measurement: predicted ms per IPM iteration of ONE dense QP n_x=50000 m=20000 shared by all GPUs "
                "(2-D block-cyclic distributed assembly + Cholesky + solves, csrc/dist_core.inc), max_ncorr=3",
           inputs=dict(one_gpu_measured_ms=ONE_GPU, gemm_tflops_grid=GEMM_TFLOPS_GRID, diag_tile_ms=DIAG_TILE_MS,
                       assumed=dict(link_GBs=LINK_GBS, link_efficiency=LINK_EFF, collective_latency_ms=COLL_LAT_MS,
                                    broadcast="each peer of a process row / column receives at one link's rate (ring of <= 4, or "
                                              "MADQP_DIST_BCAST=p2p on separate links)")),
           curve=[dict(n_gpus=p * q, **model(p, q)) for p, q in ((1, 1), (1, 2), (2, 2), (2, 4))])
base = out["curve"][0]["ms_per_iteration"]
for c in out["curve"]:
    c["speedup_vs_1"] = base / c["ms_per_iteration"]
    c["scaling_efficiency"] = c["speedup_vs_1"] / c["n_gpus"]
out["reading"] = ("strong scaling of a fixed 9.17e13-flop iteration: the chain (diagonal tile -> panel solve -> three broadcast stages) "
                  "is ~2 ms per step whatever the grid, the update shrinks with 1/N; from 4 GPUs on the steps of the second half are "
                  "chain bound.  If the measured curve is far below: look at comm (bytes_broadcast_by_rank0, MADQP_DIST_BCAST=p2p, "
                  "MADQP_DIST_FREE_SLOTS) first.")
print(json.dumps(out, indent=1))
