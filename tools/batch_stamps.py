#!/usr/bin/env python3
"""Diagnostic: phase times of problem 0 inside the batched engine's workgroup programs.  Needs the library built
with -DMADQP_BATCH_STAMPS (see batch_wg.inc) copied over madqp_jl_amd/libmadqp_hip.so."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import madqp_jl_amd as M  # noqa: E402
from madqp_jl_amd import _lib  # noqa: E402

NAMES = {0: "other", 1: "solve: copy+reduce_rhs", 2: "solve: condense + A'u", 3: "solve: chol_solve", 4: "solve: A dx",
         5: "solve: decondense+finish", 6: "mul: A'vy", 7: "mul: H vx", 8: "mul: A vx", 9: "mul: rows+diag+bounds",
         10: "eval: H x", 11: "eval: grad + A x + cons", 12: "gap post->pre", 13: "pre: jtprod", 14: "gap pre->post (assembly+chol)",
         15: "post: everything between the listed phases", 16: "pre: norms, reg, sigma", 17: "pre: build operands"}
be = M.HipBackend(0)
B, nx, m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 512, 256
qps = [M.DeviceQP.synthetic(be, 20250614 + b, nx, m) for b in range(B)]
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 32)()
s = M.BatchedMPCSolver(qps, be, regularization=M.FixedRegularization(1e-8, -1e-8))
lib.madqp_batch_read_stamps(buf, 1)
r = s.solve()
lib.madqp_batch_read_stamps(buf, 0)
s.close()
tot = sum(buf[i] for i in range(31)) * 1e-5
print(f"B={B}: total {tot:.1f} ms over {max(x['iter'] for x in r)} lock-step iterations")
for i in range(31):
    if buf[i]:
        print(f"  {NAMES.get(i, i):45s} {buf[i] * 1e-5:8.2f} ms  {100 * buf[i] * 1e-5 / tot:5.1f} %")
