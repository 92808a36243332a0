import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import madqp_jl_amd as M
be = M.HipBackend(0)
B, nx, m = 1024, 512, 256
qps = [M.DeviceQP.synthetic(be, 20250617 + i, nx, m) for i in range(B)]
opts = dict(max_iter=300, step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8), mu_min=1e-12)
w = M.BatchedMPCSolver(qps, be, **opts); w.solve(check_every=2); w.close()
for rep in range(3):
    s = M.BatchedMPCSolver(qps, be, **opts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.initialize(); torch.cuda.synchronize(); t1 = time.perf_counter()
    s.iterate(check_every=2); torch.cuda.synchronize(); t2 = time.perf_counter()
    r = s.results(); torch.cuda.synchronize(); t3 = time.perf_counter()
    s.close(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"initialize {1e3*(t1-t0):.1f} ms  iterate {1e3*(t2-t1):.1f}  results {1e3*(t3-t2):.1f}  close {1e3*(t4-t3):.1f}  total solve {1e3*(t3-t0):.1f}")
