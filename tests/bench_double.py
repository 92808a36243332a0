"""TEST INFRASTRUCTURE ONLY -- stand-ins with which tests/test_bench.py runs bench.py END TO END on a box without a GPU:
the launcher (`python bench.py --gpus 2` with no launcher around it), the rank scaffolding over gloo, both legs of the
N > 1 run, the one JSON line.  Selected by MADQP_BENCH_TEST_DOUBLE="bench_double:Double" (bench.load_factory), which
bench.py refuses on a box that has a GPU; the line it prints says `test_double` and `data: "TEST DOUBLE ..."`.

What is doubled: the C ABI (tests/fake_backend.py: numpy on CPU tensors, the same test double the host-logic tests use)
and the distributed KKT object -- every rank solves the SAME small QP on its own (the schedule itself is covered by
tests/test_dist2d.py on the CPU build of csrc/dist_core.inc and by tests/test_gpu_dist2d.py on the GPU).  What is NOT
doubled, and is what these tests are about: bench.py's own control flow and the product's host loop (solver.py)."""
import torch
import torch.distributed as dist

import madqp_jl_amd as M
from fake_backend import FakeBackend
from oracle import qp as Q


class DoubleBackend(FakeBackend):
    name = "bench-double-cpu"

    def prof_enable(self, classes=()):
        self._classes = tuple(classes or ())

    def prof_reset(self):
        pass

    def prof_get(self):
        return {c: (0.0, 0) for c in M._lib.PROF_CLASSES}


class FakeGrid:
    def __init__(self, world, n, nb):
        from madqp_jl_amd.dist2d import default_grid, default_tile

        self.P, self.Q = default_grid(world)
        self.n, self.nb = n, int(nb or default_tile(n, world))
        self.mloc = self.nloc = self.ld = self.ncp = n

    def bytes_sent(self):
        return 0

    def comm_info(self):
        ws = dist.get_world_size() if dist.is_initialized() else 1
        return dict(backend="TEST DOUBLE (torch.distributed %s group)" % (dist.get_backend() if ws > 1 else "no"),
                    world_size=ws, row_comm_size=self.Q, col_comm_size=self.P,
                    world_rank=dist.get_rank() if ws > 1 else 0, free_slots=0, internal_streams=0)

    def memory(self):
        return dict(total_bytes=0, matrix_bytes=0, xw_bytes=0, yw_bytes=0, band_bytes=0, staging_bytes=0, levels=1)

    def close(self):
        pass


class Double:
    name = "tests/bench_double.py: numpy test double of the C ABI on the CPU, replicated QP instead of a distributed one"
    cuda = False

    def module(self):
        return M

    def backend(self, local_rank):
        return DoubleBackend()

    def _qp(self, seed, nx, m):
        qp = Q.synthetic_qp(seed, nx, m)
        return M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)

    def local_qp(self, be, seed, nx, m):
        return self._qp(seed, nx, m)

    def shared_qp(self, be, world, seed, nx, m, nb):
        dq = self._qp(seed, nx, m)
        dq.A_I = dq.A_J = dq.A
        return FakeGrid(world, nx, nb), dq

    def sync(self):
        pass
