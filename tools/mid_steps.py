#!/usr/bin/env python3
"""Per block step of the mid-size factorisation, from a rocprofv3 --kernel-trace run of bench.py: duration of the step
kernel, of the panel kernel behind it, and the time from one step kernel's start to the next one's (median over the
factorisations in the trace).  python tools/mid_steps.py <kernel_trace.csv>"""
import csv
import statistics
import sys


def main():
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))),
                  key=lambda r: r[1])
    facts, cur = [], None
    for i, (n, s, e) in enumerate(rows):
        if "chol_mid_step_kernel" in n:
            if cur is None or (cur and s - cur[-1][1] > 400_000):
                cur = []
                facts.append(cur)
            nxt = rows[i + 1] if i + 1 < len(rows) else None
            pan = (nxt[2] - nxt[1]) if nxt and "panel" in nxt[0] else 0
            cur.append((s, e, pan))
    nst = max(len(f) for f in facts)
    facts = [f for f in facts if len(f) == nst][2:]
    print(f"{len(facts)} factorisations of {nst} steps")
    tot = 0.0
    for k in range(nst):
        d = statistics.median((f[k][1] - f[k][0]) / 1e3 for f in facts)
        p = statistics.median(f[k][2] / 1e3 for f in facts)
        per = statistics.median((f[k + 1][0] - f[k][0]) / 1e3 for f in facts) if k + 1 < nst else d + p
        tot += per
        print(f"k={k:2d}  step {d:6.1f} us  panel {p:5.1f} us  start-to-start {per:6.1f} us")
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()
