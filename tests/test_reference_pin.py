"""CPU: oracle/mpc.py against traces of the REFERENCE itself -- when they exist.

SURVEY.md 8c: the reference is pure Julia and neither Julia nor MadNLP is in the build container or on the GPU box, so
`oracle/` is pinned only by known answers and self-consistency ("parity unpinned").  tools/julia/dump_reference_traces.jl
is the executable form of the survey's "first action when a Julia box exists": it runs simple_lp() (test/runtests.jl:24-55)
and MadNLPTests.DenseDummyQP(zeros(10); m=5) through MadIPM.MPCSolver (K2 + LapackCPUSolver) and writes the problem data
and the per-iteration tuple of src/structure.jl:178-195 to tests/golden/reference_*.json.  This test replays every such
file through the oracle's K2 path on the dumped data and compares iteration count, per-iteration quantities, objective,
solution and multipliers.  No file, no claim: it is SKIPPED (not passed) until someone commits the dumps."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import mpc
from oracle import qp as Q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "reference_*.json")))


def _vec(v):
    return np.array([np.inf if x == "inf" else -np.inf if x == "-inf" else float(x) for x in v], dtype=float)


def load_reference_case(path):
    """(DenseQP, record) from a file written by tools/julia/dump_reference_traces.jl"""
    rec = json.load(open(path))
    d = rec["data"]
    n, m = d["n"], d["m"]
    H = np.array(d["H"], dtype=float).reshape(n, n)
    A = np.array(d["A"], dtype=float).reshape(m, n) if m else np.zeros((0, n))
    qp = Q.DenseQP(H=H, q=_vec(d["q"]), A=A, lvar=_vec(d["lvar"]), uvar=_vec(d["uvar"]), lcon=_vec(d["lcon"]),
                   ucon=_vec(d["ucon"]), x0=_vec(d["x0"]), c0=float(d["c0"]), name=rec["case"])
    return qp, rec


def test_loader_round_trip(tmp_path):
    """The loader itself, on a file in the dump's format written from the oracle (so that the day real dumps arrive the
    only unknown is the reference's numbers, not this code)."""
    qp = Q.simple_lp()
    r = mpc.solve(qp, kkt_system="K2")
    inf = lambda v: ["inf" if x == np.inf else "-inf" if x == -np.inf else float(x) for x in v]
    rec = dict(case="simple_lp", iter=r["iter"], status=int(r["status"]), objective=r["objective"],
               solution=list(map(float, r["solution"])), multipliers=list(map(float, r["multipliers"])),
               options={}, trace=[{k: float(t[k]) for k in ("k", "obj", "inf_pr", "inf_du", "inf_compl", "mu", "alpha_p", "alpha_d")}
                                  for t in r["trace"]],
               data=dict(n=2, m=1, H=qp.H.tolist(), A=qp.A.tolist(), q=qp.q.tolist(), c0=qp.c0, lvar=inf(qp.lvar),
                         uvar=inf(qp.uvar), lcon=inf(qp.lcon), ucon=inf(qp.ucon), x0=qp.x0.tolist()))
    p = tmp_path / "reference_fake.json"
    p.write_text(json.dumps(rec))
    qp2, rec2 = load_reference_case(str(p))
    r2 = mpc.solve(qp2, kkt_system="K2")
    assert r2["iter"] == rec2["iter"] and abs(r2["objective"] - 1.0) < 1e-8
    assert np.array_equal(qp2.uvar, qp.uvar) and np.array_equal(qp2.A, qp.A)


@pytest.mark.skipif(not FILES, reason="no tests/golden/reference_*.json: the reference (Julia, MadNLP 0.8.x) cannot run in this "
                                      "container or on the GPU box (SURVEY.md 8c); run tools/julia/dump_reference_traces.jl on a "
                                      "Julia box and commit its output to pin the oracle")
@pytest.mark.parametrize("path", FILES or ["<none>"])
def test_oracle_follows_the_reference_trace(path):
    qp, rec = load_reference_case(path)
    opts = {}
    if "max_ncorr" in rec.get("options", {}):
        opts["max_ncorr"] = int(rec["options"]["max_ncorr"])
    r = mpc.solve(qp, kkt_system="K2", **opts)
    assert int(r["status"]) == int(rec["status"])
    assert r["iter"] == rec["iter"], (r["iter"], rec["iter"])
    for t, a in zip(r["trace"], rec["trace"]):
        tol = 1e-9 if min(t["mu"], a["mu"]) >= 1e-4 else 1e-6
        for key in ("obj", "inf_pr", "inf_du", "inf_compl", "mu", "alpha_p", "alpha_d"):
            assert abs(t[key] - a[key]) <= tol * max(1.0, abs(a[key])), (path, t["k"], key, t[key], a[key])
    assert abs(r["objective"] - rec["objective"]) <= 1e-9 * max(1.0, abs(rec["objective"]))
    assert np.max(np.abs(np.asarray(r["solution"]) - np.asarray(rec["solution"], dtype=float))) <= 1e-7
    assert np.max(np.abs(np.asarray(r["multipliers"]) - np.asarray(rec["multipliers"], dtype=float)), initial=0.0) <= 1e-6
