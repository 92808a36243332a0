#!/usr/bin/env python3
"""HBM bytes of the per-variable kernels against their algorithmic bytes (tools/vec_pmc.py under rocprofv3).

    python tools/summarize_vec_pmc.py <tag> <nx> <m>   ->  profiles/<tag>_pmc_table.{json,txt}

Counter bytes: FETCH_SIZE (KiB) x 2 (the gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md, HBM section)
+ WRITE_SIZE (KiB); 8-byte-per-lane accesses are outside the guide's calibration, so the table also gives the
uncorrected reading.  Durations: the un-instrumented --kernel-trace --stats pass.  Algorithmic bytes: 8 B x every
distinct vector entry a launch reads or writes (index lists included, they are int64), SURVEY.md 8a."""
import collections
import csv
import glob
import json
import sys

tag, nx, m = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = nx + m
nl = nu = n
ntot = n + m + nl + nu
ALG = {  # doubles (or int64 indices) touched per launch
    "aug_diag_fill_kernel": 2 * n + m,
    "aug_diag_lb_kernel": 8 * nl,
    "aug_diag_ub_kernel": 8 * nu,
    "rhs_kernel": (8 * n + 2 * m + 2 * nl + 2 * nu) + (nl + nu) / 2,  # modes 0 and 1 alternate: mode 1 reads the corrections
    "correction_kernel": n + 3 * nl + 3 * nu,
    "reduce_rhs_kernel": 5 * nl,
    "finish_aug_solve_kernel": n + 5 * nl + 5 * nu,
    "kktmul_diag_kernel": n + (n + m) + m + 2 * (n + m),
    "kktmul_lb_kernel": 9 * nl,
    "kktmul_ub_kernel": 9 * nu,
    "alpha_max_kernel": 6 * n + 2 * (nl + nu),
    "compl_kernel": (5 * n + nl + nu) + (n + nl + nu) / 2,  # plain and affine alternate: the affine one reads dx, dzl, dzu
    "norm_inf3_kernel": 3 * ntot,
    "inf_kernel": m + 7 * n + nl + nu,
    "update_iterates_kernel": 3 * (n + m + nl + nu) + nl + nu,
    "adjust_boundary_kernel": 3 * n + nl + nu,
}


def short(name):
    for k in ALG:
        if k in name:
            return k
    return None


def rows(kind):
    f = glob.glob(f"gpurun_out/prof_{tag}_{kind}/*/*_counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []


cnt = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for kind in ("fetch", "write"):
    for r in rows(kind):
        k = short(r["Kernel_Name"])
        if k:
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if kind == "fetch":
                disp[k].add(r["Dispatch_Id"])
stats = {}
for f in glob.glob(f"gpurun_out/prof_{tag}_stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            stats[k] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) * 1e-3)
out = {"config": dict(nx=nx, m=m, n=n, nlb=nl, nub=nu,
                      command="rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE --kernel-trace | --pmc WRITE_SIZE "
                              "--kernel-trace} -- python3 tools/vec_pmc.py 100000 40000 (tools/runs/r3_vec_pmc.sh)"),
       "kernels": {}}
lines = [f"# per-variable kernels at BASELINE configs[4] vector sizes (n = {n}, m = {m}, nlb = nub = {n}); peak 8000 GB/s",
         f"# {'kernel':26s} {'alg MB':>8s} {'fetch MB':>9s} {'(x2) MB':>8s} {'write MB':>9s} {'avg us':>8s} {'alg GB/s':>9s} {'frac':>6s}"]
for k in ALG:
    if k not in stats or k not in disp:
        continue
    nd = max(len(disp[k]), 1)
    alg = 8.0 * ALG[k]
    fetch = cnt[k].get("FETCH_SIZE", 0.0) * 1024 / nd
    write = cnt[k].get("WRITE_SIZE", 0.0) * 1024 / nd
    us = stats[k]["avg_us"]
    gbs = alg / (us * 1e-6) * 1e-9
    out["kernels"][k] = dict(algorithmic_bytes=alg, fetch_bytes_raw=fetch, fetch_bytes_corrected=2 * fetch, write_bytes=write,
                             avg_us=us, calls=stats[k]["calls"], achieved_GBps_algorithmic=gbs, frac_of_8TBps=gbs / 8000.0,
                             counter_over_algorithmic=(2 * fetch + write) / alg)
    lines.append(f"  {k:26s} {alg / 1e6:8.2f} {fetch / 1e6:9.2f} {2 * fetch / 1e6:8.2f} {write / 1e6:9.2f} {us:8.1f} {gbs:9.0f} {gbs / 8000:6.3f}")
json.dump(out, open(f"profiles/{tag}_pmc_table.json", "w"), indent=1)
open(f"profiles/{tag}_pmc_table.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
