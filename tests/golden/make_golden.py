"""Regenerates tests/golden/traces.json from the CPU oracle (oracle/mpc.py).

The reference cannot run in this environment (pure Julia, no Julia toolchain) and
holds no golden vectors, so these traces pin GPU-vs-oracle parity and guard the oracle against
regressions; they do NOT pin parity with the reference ("parity unpinned").

    python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import mpc, qp as Q  # noqa: E402

CASES = {
    "hs21": (lambda: Q.hs21(), dict(max_ncorr=0)),
    "dummy_10_5": (lambda: Q.dummy_qp(10, 5), dict(max_ncorr=0)),
    "dummy_50_10": (lambda: Q.dummy_qp(50, 10), dict(max_ncorr=0)),
    "dummy_20_15_eq": (lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)), dict(max_ncorr=0)),
    "dummy_20_15_eq_gondzio": (lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)), dict(max_ncorr=5)),
    "synthetic_40_16": (lambda: Q.synthetic_qp(20250614, 40, 16), dict(max_ncorr=0)),
    "synthetic_40_16_gondzio": (lambda: Q.synthetic_qp(20250614, 40, 16), dict(max_ncorr=3)),
    "synthetic_lp_30_12": (lambda: Q.synthetic_qp(20250615, 30, 12, "lp"), dict(max_ncorr=0)),
    # the reference's default formulation and options: K2 system, FixedRegularization(1e-8, 0.0) (src/utils.jl:69-108)
    "k2_simple_lp": (lambda: Q.simple_lp(), dict(kkt_system="K2", default_reg=True)),
    "k2_hs21": (lambda: Q.hs21(), dict(kkt_system="K2", default_reg=True)),
    "k2_dummy_20_15_eq": (lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)), dict(kkt_system="K2", default_reg=True)),
    "k2_dummy_20_15_eq_gondzio": (lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
                                  dict(kkt_system="K2", default_reg=True, max_ncorr=3)),
    "k2_random_40_22": (lambda: Q.random_qp(23, 40, 22, False), dict(kkt_system="K2", default_reg=True)),
}
KEYS = ("k", "obj", "inf_pr", "inf_du", "inf_compl", "mu", "alpha_p", "alpha_d")


def run(name):
    make, opts = CASES[name]
    opts = dict(opts)
    kw = dict(kkt_system=opts.pop("kkt_system", "condensed"))
    if not opts.pop("default_reg", False):
        kw["regularization"] = mpc.FixedRegularization(1e-8, -1e-8)
    r = mpc.solve(make(), **kw, **opts)
    return dict(status=r["status"], iter=r["iter"], objective=r["objective"],
                solution=[float(v) for v in r["solution"]],
                trace=[{k: float(t[k]) for k in KEYS} for t in r["trace"]])


if __name__ == "__main__":
    out = {name: run(name) for name in CASES}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traces.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(k, v["status"], v["iter"], v["objective"])
