#!/bin/bash
cat > /tmp/dbg.py <<'PY'
import sys, numpy as np
sys.path.insert(0,'.')
import madqp_jl_amd as M
from oracle import mpc, qp as Q
be=M.HipBackend(0)
REG = M.FixedRegularization(1e-8, -1e-8)
for n, mm in ((130,70),(256,70),(512,256),(300,100)):
    qp=Q.synthetic_qp(5,n,mm)
    dq=M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
    b=M.BatchedMPCSolver([dq], be, regularization=REG, max_iter=2); r=b.solve()[0]; b.close()
    s=M.MPCSolver(dq, be, regularization=REG, driver="native", max_iter=2); r1=s.solve(); s.close()
    print(n, mm, "inf_du", r["inf_du"], r1["trace"][-1]["inf_du"], "OK" if abs(r["inf_du"]-r1["trace"][-1]["inf_du"])<1e-3*abs(r1["trace"][-1]["inf_du"]) else "BAD", flush=True)
PY
python /tmp/dbg.py 2>/dev/null
python -m pytest tests/test_gpu_random.py tests/test_gpu_batched.py -x -q -m gpu 2>&1 | tail -3
