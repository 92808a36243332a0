#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_bench.sh into profiles/<tag>_*.

    python tools/summarize_pmc.py r01

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request of a wide coalesced stream, i.e. exactly half the bytes (MI355X_MICROARCH.md, HBM
section), so it is doubled here; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob
import json
import shutil
import os
import sys

PROF = os.environ.get("PROF_DIR", "gpurun_out")  # where tools/profile_bench.sh wrote its passes
tag = sys.argv[1]
out = {}


def rows(kind):
    f = glob.glob(f"{PROF}/prof_{tag}_{kind}/*/*_counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []


def short(name):
    for k in ("gemm_tn_f64_kernel", "potf2_inv_kernel", "gemv_n_wave", "gemv_n_block", "gemv_t_kernel",
              "trsv_fwd_sweep_kernel", "trsv_bwd_sweep_kernel", "scale_rows_kernel"):
        if k in name:
            return k
    return None


per = collections.defaultdict(lambda: dict(dispatches=set(), ms=0.0, counters=collections.defaultdict(float)))
# every dispatch of every pass, in dispatch order: which launches of the GEMM kernel belong to the assembly (between the
# operand scaling and the first diagonal block of a factorisation) and which to the factorisation (after it: the wide
# outer-panel updates, the in-panel updates)
disp = collections.defaultdict(dict)  # kind -> dispatch id -> dict(name, ms, counters)
for kind in ("fetch", "write", "sq", "tcc"):
    seen = set()
    for r in rows(kind):
        full = r["Kernel_Name"]
        d = disp[kind].setdefault(int(r["Dispatch_Id"]), dict(
            name=full, ms=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6, counters={}))
        d["counters"][r["Counter_Name"]] = d["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        k = short(full)
        if not k:
            continue
        p = per[k]
        p["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (kind, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            if kind == "fetch":
                p["dispatches"].add(r["Dispatch_Id"])
                p["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6


def largest_gemm(kind):
    """(largest assembly dispatch, largest dispatch inside a factorisation) of the GEMM kernel in one pass"""
    best = {"assembly": None, "panel_update": None}
    in_fact = False
    for i in sorted(disp[kind]):
        d = disp[kind][i]
        if "scale_rows_kernel" in d["name"]:
            in_fact = False
        elif "potf2_inv_kernel" in d["name"] or "chol_mid_step_kernel" in d["name"]:
            in_fact = True
        elif "gemm_tn_f64_kernel" in d["name"]:
            which = "panel_update" if in_fact else "assembly"
            if best[which] is None or d["ms"] > best[which]["ms"]:
                best[which] = d
    return best


for k, p in per.items():
    c = dict(p["counters"])
    e = dict(dispatches=len(p["dispatches"]), total_ms_under_pmc=round(p["ms"], 3), counters=c)
    if "FETCH_SIZE" in c:
        e["hbm_read_GB_corrected"] = c["FETCH_SIZE"] * 1024 * 2 / 1e9
    if "WRITE_SIZE" in c:
        e["hbm_write_GB"] = c["WRITE_SIZE"] * 1024 / 1e9
    if "TCC_HIT_sum" in c:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    out[k] = e
CLOCK_HZ = 2.385e9  # measured (tools/mfma_probe.hip); SQ_VALU_MFMA_BUSY_CYCLES sums over 256 CUs x 4 SIMDs
big = {kind: largest_gemm(kind) for kind in disp}
for which, label in (("assembly", "largest_gemm_dispatch (the assembly SYRK)"),
                     ("panel_update", "largest_outer_panel_update_dispatch (Cholesky panel GEMM)")):
    pp = {kind: dict(ms=v[which]["ms"], **v[which]["counters"]) for kind, v in big.items() if v.get(which)}
    if not pp:
        continue
    rd = pp.get("fetch", {}).get("FETCH_SIZE", 0) * 1024 * 2
    wr = pp.get("write", {}).get("WRITE_SIZE", 0) * 1024
    sq = pp.get("sq", {})
    out[label] = dict(per_pass=pp, hbm_read_bytes_corrected=rd, hbm_write_bytes=wr, traffic_bytes=rd + wr,
                      mfma_busy_fraction=(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / max(sq.get("ms", 1) * 1e-3 * CLOCK_HZ, 1))
out["mfma_busy"] = {
    "assembly_syrk": out.get("largest_gemm_dispatch (the assembly SYRK)", {}).get("mfma_busy_fraction"),
    "cholesky_panel_gemm": out.get("largest_outer_panel_update_dispatch (Cholesky panel GEMM)", {}).get("mfma_busy_fraction"),
    "how": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch duration x 2.385 GHz), one dispatch each: the longest GEMM "
           "launch of the assembly and the longest one inside a factorisation (a wide outer-panel update)"}
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
# HBM-bound kernels: achieved rate = PMC bytes (read corrected + written) / duration of the same dispatches
# in the un-instrumented --kernel-trace --stats pass (average duration x dispatch count)
stats = {}
for f in glob.glob(f"{PROF}/prof_{tag}_stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            stats[k] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) * 1e-3)
rates = {}
for k in ("trsv_fwd_sweep_kernel", "trsv_bwd_sweep_kernel", "gemv_n_wave", "gemv_t_kernel", "scale_rows_kernel"):
    if k in out and k in stats and "hbm_read_GB_corrected" in out[k]:
        gb = out[k]["hbm_read_GB_corrected"] + out[k].get("hbm_write_GB", 0.0)
        per_launch = gb / max(out[k]["dispatches"], 1)
        e = dict(avg_us=round(stats[k]["avg_us"], 1), hbm_GB_per_launch=round(per_launch, 3),
                 achieved_TB_per_s=round(per_launch / (stats[k]["avg_us"] * 1e-6) / 1e3, 2),
                 frac_of_8_TB_per_s=round(per_launch / (stats[k]["avg_us"] * 1e-6) / 8e3, 3))
        if k.startswith("trsv"):
            e["algorithmic_GB_per_launch"] = round(4.0 * nx * nx / 1e9, 3)  # half of L, 8 B per entry
        rates[k] = e
out["hbm_bound_kernels"] = rates
bench_args = sys.argv[4] if len(sys.argv) > 4 else "(defaults: --steps 3 --warmup 1)"
out["config"] = dict(nx=nx, m=m, command=f"rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --no-cpu-baseline {bench_args} "
                                         f"[nx={nx}, m={m}]; one pass per counter group (tools/profile_bench.sh)")
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
for f in glob.glob(f"{PROF}/prof_{tag}_stats/*/*_kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_bench50k_kernel_stats.csv")
print(json.dumps(out, indent=1)[:3000])
