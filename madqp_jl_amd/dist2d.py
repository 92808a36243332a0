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
_P2P = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class CCommOps(C.Structure):
    """``madqp_comm_ops`` of include/madqp.h."""

    _fields_ = [("user", C.c_void_p), ("bcast", _BCAST), ("reduce_sum", _REDUCE), ("allreduce_sum", _ALLREDUCE),
                ("send", _P2P), ("recv", _P2P)]


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
        self.calls = {"bcast": 0, "reduce": 0, "allreduce": 0, "send": 0, "recv": 0}
        self.error = None
        self._cb = (_BCAST(self._bcast), _REDUCE(self._reduce), _ALLREDUCE(self._allreduce), _P2P(self._send),
                    _P2P(self._recv))  # keep alive
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


    def _send(self, user, buf, nbytes, dst):
        try:
            self.calls["send"] += 1
            dist.send(self._tensor(buf, nbytes, torch.uint8), dst=dst, group=self.groups[GRP_WORLD])
            return 0
        except Exception as e:
            self.error = e
            return 1

    def _recv(self, user, buf, nbytes, src):
        try:
            self.calls["recv"] += 1
            dist.recv(self._tensor(buf, nbytes, torch.uint8), src=src, group=self.groups[GRP_WORLD])
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

    def comm_info(self) -> dict:
        """Who carries the collectives, as the library sees it (``madqp_dist_comm_info``)."""
        out = (C.c_int64 * 8)()
        self.be._ck(self.be.lib.madqp_dist_comm_info(self._h, out))
        return dict(backend={0: "none (one rank)", 1: "rccl", 2: "host-staged callbacks (madqp_comm_ops)"}[out[0]],
                    world_size=int(out[1]), row_comm_size=int(out[2]), col_comm_size=int(out[3]), world_rank=int(out[4]),
                    free_slots=int(out[5]), internal_streams=int(out[6]))

    def memory(self) -> dict:
        """Device bytes of the handle by part (``madqp_dist_memory``)."""
        out = (C.c_int64 * 8)()
        self.be._ck(self.be.lib.madqp_dist_memory(self._h, out))
        return dict(total_bytes=int(out[0]), matrix_bytes=int(out[1]), xw_bytes=int(out[2]), yw_bytes=int(out[3]),
                    band_bytes=int(out[4]), staging_bytes=int(out[5]), levels=int(out[6]))

    def close(self):
        if self._h is not None:
            self.be.lib.madqp_dist_destroy(self._h)
            self._h = None


# =====================================================================================================
# One QP shared by all ranks: what a rank holds of it, and the KKT system above madqp_dkkt_*
def reduce_where(backend_name: str, tensor_is_cuda: bool) -> str:
    """Where a tensor must live for a collective on the default process group: RCCL ("nccl") has no backend for CPU
    tensors and gloo none (worth relying on) for device tensors, so the tensor goes where the group's backend is --
    "as_is" when it already is there, else "to_cpu" (rehearsals: gloo with the data on a GPU).  A device tensor on
    an nccl group is reduced in place on the device: `bench.py --gpus N` initialises nccl only."""
    name = backend_name.lower()
    has_cuda = "nccl" in name
    has_cpu = "gloo" in name or "mpi" in name
    if tensor_is_cuda:
        if has_cuda:
            return "as_is"
        if has_cpu:
            return "to_cpu"
        raise RuntimeError(f"process group backend {backend_name!r} can reduce neither device nor host tensors")
    if has_cpu:
        return "as_is"
    raise RuntimeError(f"process group backend {backend_name!r} has no CPU backend: keep the tensor on the device")


def all_reduce_replicated(t, op):
    """all_reduce of a replicated vector over the default process group, on the device the group's backend serves."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    if reduce_where(str(dist.get_backend()), t.is_cuda) == "as_is":
        dist.all_reduce(t, op=op)
        return t
    c = t.cpu()
    dist.all_reduce(c, op=op)
    return c.to(t.device)


def _cyclic_index(count_local, nb, R, r, device):
    """Global index of every local row / column of a direction with modulus R and residue r."""
    c = torch.arange(count_local, device=device)
    return (c // nb * R + r) * nb + c % nb


class DistributedQP:
    """A dense QP ``min x'Hx/2 + q'x  s.t. lcon <= Ax <= ucon, lvar <= x <= uvar`` whose matrices are spread over the
    grid of a :class:`DistCholesky2D` (the vectors are replicated): ``Hloc`` -- the tiles of H in the local layout of K
    (tensor ``(ncp, ld)``, entry (i, j) of the local matrix at ``[j, i]``: column-major), ``A_I`` / ``A_J`` -- the
    columns of A of this rank's tile rows / tile columns (``(ceil16(m), ld)`` / ``(ceil16(m), ncp)``, zero padded)."""

    def __init__(self, grid, Hloc, q, A_I, A_J, lvar, uvar, lcon, ucon, x0, c0=0.0, y0=None, name="distributed-qp"):
        self.grid, self.H, self.q, self.A_I, self.A_J = grid, Hloc, q, A_I, A_J
        self.lvar, self.uvar, self.lcon, self.ucon, self.x0, self.c0, self.name = lvar, uvar, lcon, ucon, x0, float(c0), name
        self.y0 = y0 if y0 is not None else torch.zeros_like(lcon)
        self.A = None  # there is no full A anywhere

    nvar = property(lambda s: s.q.numel())
    ncon = property(lambda s: s.lcon.numel())

    def eliminate_fixed(self):
        return None

    @classmethod
    def synthetic(cls, backend, grid, seed: int, n: int, m: int, family: str = "wigner"):
        """The benchmark family of :meth:`DeviceQP.synthetic`, every rank generating only its own pieces (the
        generator is position addressable: bit for bit the entries of the one-GPU problem)."""
        import math

        from .qp import STREAM_A, STREAM_H, STREAM_Q, stream_key

        assert n == grid.n
        dev, f64 = backend.device, torch.float64
        m16 = (m + 15) // 16 * 16
        A_I = torch.zeros((max(m16, 1), grid.ld), dtype=f64, device=dev)
        A_J = torch.zeros((max(m16, 1), grid.ncp), dtype=f64, device=dev)
        kA = stream_key(seed, STREAM_A)
        backend._ck(backend.lib.madqp_gen_normal_cyclic(backend.ctx, kA, m, n, grid.nb, grid.P, grid.p, grid.mloc,
                                                        A_I.data_ptr(), grid.ld))
        backend._ck(backend.lib.madqp_gen_normal_cyclic(backend.ctx, kA, m, n, grid.nb, grid.Q, grid.q, grid.nloc,
                                                        A_J.data_ptr(), grid.ncp))
        q = torch.empty(n, dtype=f64, device=dev)
        backend.gen_normal(stream_key(seed, STREAM_Q), 0, q)
        H = None
        if family == "wigner":
            H = torch.zeros((grid.ncp, grid.ld), dtype=f64, device=dev)
            backend._ck(backend.lib.madqp_gen_wigner_cyclic(backend.ctx, stream_key(seed, STREAM_H), n, 1.0 / math.sqrt(n),
                                                            grid.nb, grid.P, grid.p, grid.Q, grid.q, grid.mloc, grid.nloc,
                                                            H.data_ptr(), grid.ld))
        elif family != "lp":
            raise ValueError(family)
        z = lambda k, v: torch.full((k,), v, dtype=f64, device=dev)
        return cls(grid, H, q, A_I, A_J, z(n, 0.0), z(n, 1.0), z(m, 0.0), z(m, 1.0), z(n, 0.0),
                   name=f"synthetic-{family}-n{n}-m{m}-s{seed}-grid{grid.P}x{grid.Q}")

    @classmethod
    def from_dense(cls, backend, grid, H, q, A, lvar, uvar, lcon, ucon, x0, c0=0.0):
        """Pieces of a QP given in full on the host (numpy; tests and small problems)."""
        dev, f64 = backend.device, torch.float64
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
        n, m = len(q), A.shape[0]
        assert n == grid.n
        m16 = (m + 15) // 16 * 16
        gi = _cyclic_index(grid.mloc, grid.nb, grid.P, grid.p, "cpu").numpy()
        gj = _cyclic_index(grid.nloc, grid.nb, grid.Q, grid.q, "cpu").numpy()
        A_I = torch.zeros((max(m16, 1), grid.ld), dtype=f64, device=dev)
        A_J = torch.zeros((max(m16, 1), grid.ncp), dtype=f64, device=dev)
        if m:
            A_I[:m, :grid.mloc] = t(A[:, gi])
            A_J[:m, :grid.nloc] = t(A[:, gj])
        Hloc = None
        if H is not None and np.any(H):
            Hloc = torch.zeros((grid.ncp, grid.ld), dtype=f64, device=dev)
            Hloc[:grid.nloc, :grid.mloc] = t(H[np.ix_(gi, gj)].T)
        return cls(grid, Hloc, t(q), A_I, A_J, t(lvar), t(uvar), t(lcon), t(ucon), t(x0), c0)

    # ---- the two places the set-up needs a product with the matrices before a KKT object exists (src/solver.jl:148-159)
    def row_absmax(self):
        """max_j |A[k, j]| per row, over all ranks."""
        r = self.A_I[: self.ncon].abs().amax(dim=1) if self.ncon and self.grid.mloc else torch.zeros(self.ncon, dtype=torch.float64, device=self.q.device)
        return all_reduce_replicated(r, dist.ReduceOp.MAX)

    def hess_times(self, x):
        """H x from the local tiles (each lower tile once, its mirror image too), summed over the ranks."""
        g, n = self.grid, self.nvar
        y = torch.zeros(n, dtype=torch.float64, device=x.device)
        if self.H is not None and g.mloc and g.nloc:
            gi = _cyclic_index(g.mloc, g.nb, g.P, g.p, x.device)
            gj = _cyclic_index(g.nloc, g.nb, g.Q, g.q, x.device)
            Hl = self.H[: g.nloc, : g.mloc]  # [j, i]
            lower = (gi[None, :] // g.nb) >= (gj[:, None] // g.nb)  # tile row >= tile column
            strict = (gi[None, :] // g.nb) > (gj[:, None] // g.nb)
            y.index_add_(0, gi, (Hl * lower).t() @ x[gj])
            y.index_add_(0, gj, (Hl * strict) @ x[gi])
        return all_reduce_replicated(y, dist.ReduceOp.SUM)


class HIPDistributedCondensedKKTSystem2D:
    """:class:`HIPCondensedKKTSystem` with ``K`` on a P x Q grid (``madqp_dkkt_*``): same methods, same fields."""

    def __init__(self, backend, st, nx, ind_ineq, grid, Hloc, A_I, A_J):
        from .kkt import HIPCholeskySolver

        self.be, self.st, self.grid = backend, st, grid
        self.nx, self.m = int(nx), st.m
        self.ind_ineq = [int(i) for i in ind_ineq]
        self.ns = len(self.ind_ineq)
        assert st.n == self.nx + self.ns and grid.n == self.nx
        self.H, self.A_I, self.A_J = Hloc, A_I, A_J  # keep the borrowed tensors alive
        ineq = (C.c_int64 * max(self.ns, 1))(*self.ind_ineq)
        h = C.c_void_p()
        backend._ck(backend.lib.madqp_dkkt_create(grid._h, self.nx, self.m, self.ns, ineq,
                                                  None if Hloc is None else Hloc.data_ptr(), grid.ld,
                                                  A_I.data_ptr(), A_I.stride(0), A_J.data_ptr(), A_J.stride(0), C.byref(h)))
        self._h = h
        self.linear_solver = HIPCholeskySolver(backend, None)
        self.linear_solver.factorize = self._factorize
        self.n_factorizations = 0
        self.panel_width = grid.nb

    reg = property(lambda s: s.st.reg)
    pr_diag = property(lambda s: s.st.pr_diag)
    du_diag = property(lambda s: s.st.du_diag)

    def close(self):
        if self._h is not None:
            self.be.lib.madqp_dkkt_destroy(self._h)
            self._h = None

    def initialize(self):
        st, be = self.st, self.be
        for t, v in ((st.reg, 1.0), (st.pr_diag, 1.0), (st.du_diag, 0.0), (st.l_lower, 0.0), (st.u_lower, 0.0),
                     (st.l_diag, 1.0), (st.u_diag, 1.0)):
            be.fill(v, t)

    def set_aug_diagonal_reg(self, del_w, del_c):
        self.be.set_aug_diagonal_reg(self.st, del_w, del_c)

    def _factorize(self):
        info = C.c_int32()
        self.be._ck(self.be.lib.madqp_dkkt_factorize(self._h, C.byref(info)))
        self.linear_solver.info = info.value
        return self.linear_solver

    def build_kkt(self):
        self.be._ck(self.be.lib.madqp_dkkt_build(self._h, C.byref(self.st.cstruct)))

    def factorize_wrapper(self):
        self.build_kkt()
        self._factorize()
        self.n_factorizations += 1

    def solve(self, w):
        self.be._ck(self.be.lib.madqp_dkkt_solve(self._h, C.byref(self.st.cstruct), w.data_ptr()))
        return w

    def mul(self, w, v, alpha=1.0, beta=0.0):
        self.be._ck(self.be.lib.madqp_dkkt_mul(self._h, C.byref(self.st.cstruct), w.data_ptr(), v.data_ptr(), alpha, beta))
        return w

    def jtprod(self, out, y):
        self.be._ck(self.be.lib.madqp_dkkt_jtprod(self._h, out.data_ptr(), y.data_ptr()))

    def eval_model(self, q, rhs, c0) -> float:
        obj = C.c_double()
        self.be._ck(self.be.lib.madqp_dkkt_eval(self._h, C.byref(self.st.cstruct), q.data_ptr(), rhs.data_ptr(), c0,
                                                C.byref(obj)))
        return obj.value
