"""One dense KKT matrix on a P x Q grid of GPUs (BASELINE configs[4]): host side of ``madqp_dist_*``.

The schedule lives in the library (``csrc/dist_core.inc``: 2-D block-cyclic right-looking Cholesky with look-ahead,
distributed triangular sweeps) and calls RCCL itself; this module only

* picks the grid (:func:`default_grid`: 8 GPUs -> 2 x 4) and the tile size (:func:`default_tile`),
* ships RCCL's unique id from rank 0 to the other ranks over the process group the launcher already set up
  (``torch.distributed.broadcast_object_list`` -- 128 bytes, once), and
* offers :class:`HostStagedComm`, the ``madqp_comm_ops`` callbacks over a ``gloo`` process group for rehearsals with
  several ranks on ONE GPU (RCCL refuses two ranks per device) and for the CPU suite's run of the same schedule.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

GRP_WORLD, GRP_ROW, GRP_COL = 0, 1, 2

_BCAST = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)
_REDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)
_ALLREDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class CCommOps(C.Structure):
    """``madqp_comm_ops`` of include/madqp.h."""

    _fields_ = [("user", C.c_void_p), ("bcast", _BCAST), ("reduce_sum", _REDUCE), ("allreduce_sum", _ALLREDUCE)]


def default_grid(world: int):
    """(P, Q) with P <= Q, P*Q = world, as square as possible: 1x1, 1x2, 2x2, 2x3, 2x4."""
    P = int(np.floor(np.sqrt(world)))
    while world % P:
        P -= 1
    return P, world // P


def default_tile(n: int, world: int) -> int:
    """Tile size nb (multiple of 128): wide enough for the per-step GEMM (K = nb) to run near the MFMA rate, narrow
    enough for >= ~6 tile columns per process column (load balance of the cyclic deal): 1024 at C-main / C5."""
    P, Q = default_grid(world)
    return int(min(1024, max(128, n // (6 * Q) // 128 * 128)))


class HostStagedComm:
    """Collectives for ``madqp_comm_ops`` over a CPU (gloo) process group: every rank calls the constructor (it creates
    one group per process row and per process column, in the same order everywhere)."""

    def __init__(self, P: int, Q: int):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert P * Q == self.world
        self.P, self.Q, self.p, self.q = P, Q, self.rank // Q, self.rank % Q
        rows = [dist.new_group([pp * Q + qq for qq in range(Q)], backend="gloo") for pp in range(P)]
        cols = [dist.new_group([pp * Q + qq for pp in range(P)], backend="gloo") for qq in range(Q)]
        world = dist.new_group(list(range(self.world)), backend="gloo")
        self.groups = {GRP_WORLD: world, GRP_ROW: rows[self.p], GRP_COL: cols[self.q]}
        self.calls = {"bcast": 0, "reduce": 0, "allreduce": 0}
        self.error = None
        self._cb = (_BCAST(self._bcast), _REDUCE(self._reduce), _ALLREDUCE(self._allreduce))  # keep alive
        self.ops = CCommOps(None, *self._cb)

    def _global(self, root, g):
        return root if g == GRP_WORLD else (self.p * self.Q + root if g == GRP_ROW else root * self.Q + self.q)

    @staticmethod
    def _tensor(ptr, count, dtype):
        n = count * (8 if dtype == torch.float64 else 1)
        return torch.frombuffer((C.c_char * n).from_address(ptr), dtype=dtype)

    def _bcast(self, user, buf, nbytes, root, g):
        try:
            self.calls["bcast"] += 1
            dist.broadcast(self._tensor(buf, nbytes, torch.uint8), src=self._global(root, g), group=self.groups[g])
            return 0
        except Exception as e:  # never raise through the C frame
            self.error = e
            return 1

    def _reduce(self, user, buf, count, root, g):
        try:
            self.calls["reduce"] += 1
            dist.reduce(self._tensor(buf, count, torch.float64), dst=self._global(root, g), op=dist.ReduceOp.SUM,
                        group=self.groups[g])
            return 0
        except Exception as e:
            self.error = e
            return 1

    def _allreduce(self, user, buf, count, g):
        try:
            self.calls["allreduce"] += 1
            dist.all_reduce(self._tensor(buf, count, torch.float64), op=dist.ReduceOp.SUM, group=self.groups[g])
            return 0
        except Exception as e:
            self.error = e
            return 1


def rccl_unique_id(backend, rank: int):
    """128 bytes of ``madqp_dist_unique_id`` drawn on rank 0 and shipped to every rank of the default process group."""
    buf = (C.c_char * 128)()
    if rank == 0:
        backend._ck(backend.lib.madqp_dist_unique_id(backend.ctx, buf))
    box = [bytes(buf)]
    dist.broadcast_object_list(box, src=0)
    return (C.c_char * 128).from_buffer_copy(box[0])


class DistCholesky2D:
    """``madqp_dist`` handle: create / layout / factor / solve (the AbstractLinearSolver of the distributed KKT system).

    ``comm``: None = RCCL (or a single rank); a :class:`HostStagedComm` = host-staged collectives (rehearsal)."""

    def __init__(self, backend, n: int, nb: int = None, grid=None, comm: HostStagedComm = None):
        self.be = backend
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.P, self.Q = grid or default_grid(self.world)
        self.n, self.nb = int(n), int(nb or default_tile(n, self.world))
        self.comm = comm
        uid = None
        if self.world > 1 and comm is None:
            uid = rccl_unique_id(backend, self.rank)
        h = C.c_void_p()
        backend._ck(backend.lib.madqp_dist_create(backend.ctx, self.rank, self.world, self.P, self.Q, self.n, self.nb,
                                                  uid, C.byref(comm.ops) if comm is not None else None, C.byref(h)))
        self._h = h
        out = (C.c_int64 * 8)()
        backend._ck(backend.lib.madqp_dist_layout(h, out))
        self.p, self.q, self.mt, self.nt, self.mloc, self.nloc, self.ld, self.ncp = list(out)
        self.info = 0

    def matrix(self):
        """(device address, leading dimension) of the local matrix."""
        p, ld = C.c_void_p(), C.c_int64()
        self.be._ck(self.be.lib.madqp_dist_matrix(self._h, C.byref(p), C.byref(ld)))
        return p.value, ld.value

    def local_tiles(self):
        """[(I, J, li, lj)] of the lower-triangle tiles this rank holds."""
        T = (self.n + self.nb - 1) // self.nb
        return [(I, J, I // self.P, J // self.Q) for J in range(self.q, T, self.Q) for I in range(self.p, T, self.P)
                if I >= J]

    def factor(self) -> int:
        info = C.c_int32()
        self.be._ck(self.be.lib.madqp_dist_factor(self._h, C.byref(info)))
        self.info = info.value
        return self.info

    def solve(self, rhs):
        self.be._ck(self.be.lib.madqp_dist_solve(self._h, rhs.data_ptr()))
        return rhs

    def bytes_sent(self) -> int:
        b = C.c_int64()
        self.be._ck(self.be.lib.madqp_dist_bytes_sent(self._h, C.byref(b)))
        return b.value

    def close(self):
        if self._h is not None:
            self.be.lib.madqp_dist_destroy(self._h)
            self._h = None
