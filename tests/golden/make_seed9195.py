#!/usr/bin/env python3
"""Writes the CPU oracle's trace of soak seed 9195 (LP, n_x = 186, m = 78, condensed form) on THIS host:

    python tests/golden/make_seed9195.py tests/golden/seed9195_oracle_<host>.json

DESIGN.md section 4 / tests/parity.py claim that the oracle does not agree with itself on this problem -- LAPACK through
scipy stops after 12 iterations on the GPU box's host CPU and after 13 in the build container -- because the deciding
residual sits within rounding of the termination threshold.  The two committed files (one per host) make that claim
checkable without either machine: tests/test_oracle.py::test_seed_9195_two_hosts_is_a_threshold_tie."""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy  # noqa: E402

from oracle import mpc  # noqa: E402
from oracle import qp as Q  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    qp = Q.random_qp(9195, 186, 78, True)
    r = mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8))
    keys = ("k", "inf_pr", "inf_du", "inf_compl", "mu", "alpha_p", "alpha_d")
    rec = dict(problem="oracle.qp.random_qp(9195, 186, 78, lp=True), condensed KKT, FixedRegularization(1e-8, -1e-8)",
               host=dict(cpu=cpu_model(), numpy=np.__version__, scipy=scipy.__version__, machine=platform.machine()),
               status=int(r["status"]), iter=int(r["iter"]), objective=float(r["objective"]),
               trace=[{k: float(t[k]) for k in keys} for t in r["trace"]])
    json.dump(rec, open(sys.argv[1], "w"), indent=1)
    print(rec["host"], rec["iter"])


if __name__ == "__main__":
    main()
