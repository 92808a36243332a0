#!/usr/bin/env python3
"""Per-kernel time and HBM bytes of `tools/bench_batch.py --batch 1024` under rocprofv3 (tools/runs/r3_batch_prof.sh):

    python tools/summarize_batch_prof.py r03batch  ->  profiles/<tag>_summary.{txt,json}

time: the --kernel-trace --stats pass (warm-up solve + one timed solve: two solves in all); bytes: FETCH_SIZE (KiB) x 2
(gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE (KiB), separate passes."""
import collections
import csv
import glob
import json
import re
import sys

tag = sys.argv[1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"wg(256|512)::", "", n)
    return n.split("(")[0][:44]


stats = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"gpurun_out/prof_{tag}_stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        stats[k][0] += int(r["Calls"])
        stats[k][1] += float(r["TotalDurationNs"]) * 1e-6
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for kind in ("fetch", "write"):
    for f in glob.glob(f"gpurun_out/prof_{tag}_{kind}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
total = sum(v[1] for v in stats.values())
out, lines = {}, [f"# tools/bench_batch.py --batch 1024 --repeats 1 under rocprofv3: warm-up solve + timed solve (2 x 18 lock-step iterations)",
                  f"# kernel time in all {total:.1f} ms; bytes = FETCH_SIZE x 2 + WRITE_SIZE (separate --pmc passes)",
                  f"# {'kernel':44s} {'calls':>6s} {'ms':>8s} {'share':>6s} {'avg us':>8s} {'read GB':>8s} {'write GB':>8s} {'GB/s':>7s}"]
for k, (calls, ms) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if ms < 0.05:
        continue
    rd = cnt[k].get("FETCH_SIZE", 0.0) * 1024 * 2 / 1e9
    wr = cnt[k].get("WRITE_SIZE", 0.0) * 1024 / 1e9
    gbs = (rd + wr) / (ms * 1e-3) if ms > 0 else 0.0
    out[k] = dict(calls=calls, ms=ms, share=ms / total, avg_us=ms / calls * 1e3, hbm_read_GB_corrected=rd, hbm_write_GB=wr,
                  GBps=gbs)
    lines.append(f"  {k:44s} {calls:6d} {ms:8.2f} {ms / total:6.3f} {ms / calls * 1e3:8.1f} {rd:8.2f} {wr:8.2f} {gbs:7.0f}")
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
open(f"profiles/{tag}_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
