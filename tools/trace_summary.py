#!/usr/bin/env python3
"""Per-iteration summary of a rocprofv3 --kernel-trace run of bench.py (rocpd sqlite output): time and launches per
kernel, device-busy fraction, gaps.  python tools/trace_summary.py results.db [blocks_per_factorization]"""
import collections
import re
import sqlite3
import statistics
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[int(len(rows) * 0.4):]  # steady state: drop set-up and warm-up
    tot = collections.defaultdict(lambda: [0, 0])
    for name, s, e in rows:
        k = re.sub(r"\(anonymous namespace\)::", "", name)
        k = re.sub(r"^void ", "", k).split("(")[0].split("<")[0][-48:]
        tot[k][0] += e - s
        tot[k][1] += 1
    span = rows[-1][2] - rows[0][1]
    busy = sum(v[0] for v in tot.values())
    syrk = [k for k in tot if "scale_rows" in k]
    iters = tot[syrk[0]][1] if syrk else 1  # one scaled operand per assembly = per factorisation
    print(f"window {span / 1e6:.1f} ms, {iters} factorisations, {span / 1e6 / iters:.3f} ms each, device busy "
          f"{busy / span:.3f}, {len(rows) / iters:.0f} launches each")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:22]:
        print(f"{k:48s} {v[0] / 1e6 / iters:8.3f} ms  {v[1] / iters:7.1f} launches  avg {v[0] / v[1] / 1e3:8.1f} us")
    gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
    print(f"median gap {statistics.median(gaps) / 1e3:.1f} us, idle {sum(g for g in gaps if g > 0) / 1e6 / iters:.3f} ms per factorisation")


if __name__ == "__main__":
    main()
