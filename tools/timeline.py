#!/usr/bin/env python3
"""One steady-state iteration of a rocprofv3 --kernel-trace run of bench.py as a timeline: kernel, start offset,
duration, gap to the previous kernel.  python tools/timeline.py <kernel_trace.csv | results.db> [iteration index]"""
import csv
import re
import sqlite3
import sys


def load(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        return [(n, int(s), int(e)) for n, s, e in c.execute("select name, start, end from kernels order by start")]
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path))]
    return sorted(rows, key=lambda r: r[1])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0].split("<")[0][-40:]


def main():
    rows = load(sys.argv[1])
    marks = [i for i, r in enumerate(rows) if "scale_rows" in r[0]]  # one per assembly = per factorisation
    it = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) * 3 // 4
    a, b = marks[it], marks[it + 1]
    t0 = rows[a][1]
    print(f"iteration {it}: {b - a} launches, {(rows[b][1] - t0) / 1e3:.1f} us")
    busy = 0
    for i in range(a, b):
        n, s, e = rows[i]
        gap = s - rows[i - 1][2]
        busy += e - s
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {short(n)}")
    print(f"busy {busy / 1e3:.1f} us of {(rows[b][1] - t0) / 1e3:.1f}")


if __name__ == "__main__":
    main()
