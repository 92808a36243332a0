#!/usr/bin/env python3
"""Average duration of the sweep kernels in a rocprofv3 --kernel-trace db.  python tools/sweep_times.py results.db"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tot = collections.defaultdict(lambda: [0, 0])
for name, s, e in c.execute("select name, start, end from kernels"):
    for key in ("trsv_fwd_sweep", "trsv_bwd_sweep", "trsv_fwd_chain", "trsv_bwd_chain", "chain_images"):
        if key in name:
            tot[key][0] += e - s
            tot[key][1] += 1
for k, v in tot.items():
    print(f"{k:18s} {v[1]:5d} launches  avg {v[0] / v[1] / 1e3:9.1f} us")
