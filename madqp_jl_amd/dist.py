"""One large dense KKT system over several GPUs (BASELINE configs[4]).

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the node, ``gloo`` in
the CPU/1-GPU rehearsals).  Every rank runs the same Mehrotra loop on the same QP; what is divided is
the MFMA work, i.e. > 95 % of an iteration:

* **Layout**: the block columns ("panels", width ``nb`` = a multiple of 128) of the lower triangle of
  ``K`` / ``L`` are dealt round-robin: panel p belongs to rank ``p mod N`` (a 1 x N block-cyclic grid).
* **Assembly** (``build_kkt!``): ``A`` (and ``H``) are replicated -- 8-32 GB next to 288 GB of HBM --
  so a rank forms ``(H + Sigma + A' Theta A)[:, own panels]`` with no communication at all.
* **Factorisation**: right-looking over panels with look-ahead 1.  The owner factors panel p (diagonal
  kernel + inverse-block TRSM over the full height, as on one GPU), packs it, and the packed image is
  **broadcast** once; every rank unpacks it into its copy of L and applies it to the panels it owns
  (one MFMA GEMM per owned panel, K = nb).  The owner of panel p+1 applies panel p to it first,
  factors it and starts its broadcast *before* touching its other panels, so the transfer of p+1 runs
  under the rank-local updates by p (the collective is issued on every rank before those updates are
  enqueued, hence it only depends on work that precedes it).
* **Solves and vector kernels**: after the last broadcast every rank holds the whole factor, so the
  two triangular sweeps per direction, GEMVs and all reductions run replicated: no collective on the
  latency-critical part of the iteration, bitwise identical scalars on all ranks, identical control
  flow without exchanging a single flag.  (A 2-D P x Q grid would cut the panel volume a rank
  receives to 1/P + 1/Q of it only if L stayed distributed, which puts a reduction per 128-block into
  each of the >= 4 triangular sweeps of an iteration; with 8 fully connected GPUs and room for the
  whole factor on each, the 1 x N grid with a gathered factor is the cheaper design.)

The first failing column travels inside the packed image, so ``info`` agrees on all ranks without
an extra collective.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .kkt import HIPCondensedKKTSystem, HIPNormalKKTSystem

NB = 128


def panel_ranges(n: int, nb: int):
    """[(start, width)] of the block columns of an order-n matrix."""
    if nb <= 0 or nb % NB:
        raise ValueError("panel width must be a positive multiple of 128")
    return [(j, min(nb, n - j)) for j in range(0, n, nb)]


def default_panel_width(n: int, world: int) -> int:
    """About 6 panels per rank (load balance of the cyclic deal), a multiple of 256 in [256, 2048].  Wider panels
    mean fewer passes over the trailing matrix (C-main on one GPU: 1520 / 1416 / 1392 / 1377 ms per iteration at
    nb = 512 / 1024 / 1536 / 2048), narrower ones a better deal: 1024 at C-main on 8 GPUs, 2048 on 1-4 and at C5."""
    return int(min(2048, max(256, n // (6 * max(world, 1)) // 256 * 256)))


class DistributedCholesky:
    """Panel-cyclic right-looking Cholesky driver above ``madqp_chol_*`` (see the module docstring).

    ``ops`` provides the rank-local primitives (``HipBackend`` on the GPU, a numpy double in the CPU
    tests): chol_factor_begin / chol_factor_panel / chol_update_cols / chol_update_multi / chol_panel_doubles /
    chol_panel_pack / chol_panel_unpack / chol_factor_end.  ``group``: the process group (None = world).
    """

    def __init__(self, ops, chol, n, nb, device, group=None):
        self.ops, self.chol, self.n, self.group = ops, chol, int(n), group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.panels = panel_ranges(self.n, nb)
        self.sizes = [ops.chol_panel_doubles(chol, j, w) for j, w in self.panels]
        cap = max(self.sizes, default=0)
        self.bufs = [torch.empty(cap, dtype=torch.float64, device=device) for _ in range(2)]
        self.bytes_sent = 0

    def owner(self, p: int) -> int:
        return p % self.world

    def own_panels(self):
        return [p for p in range(len(self.panels)) if self.owner(p) == self.rank]

    def own_ranges(self):
        return [(j, j + w) for p, (j, w) in enumerate(self.panels) if self.owner(p) == self.rank]

    def _src(self, p):
        r = self.owner(p)
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _bcast(self, p):
        buf = self.bufs[p % 2][: self.sizes[p]]
        if self.world == 1:
            return None
        if self.owner(p) == self.rank:
            self.bytes_sent += 8 * self.sizes[p]
        return dist.broadcast(buf, src=self._src(p), group=self.group, async_op=True)

    def factor(self, A_ptr: int, lda: int) -> int:
        """Factor the matrix whose OWN panels hold K (lower part); returns LAPACK-style info."""
        ops, ch, P = self.ops, self.chol, len(self.panels)
        ops.chol_factor_begin(ch, A_ptr, lda)
        if P == 0:
            return ops.chol_factor_end(ch)
        mine = self.own_panels()
        if self.owner(0) == self.rank:
            ops.chol_factor_panel(ch, *self.panels[0])
            ops.chol_panel_pack(ch, *self.panels[0], self.bufs[0])
        work = self._bcast(0)
        for p in range(P):
            j, w = self.panels[p]
            if work is not None:
                work.wait()  # the compute stream now waits for the transfer of panel p
            if self.owner(p) != self.rank:
                ops.chol_panel_unpack(ch, j, w, self.bufs[p % 2])
            nxt = p + 1
            work = None
            if nxt < P:
                if self.owner(nxt) == self.rank:  # look-ahead: finish and ship panel p+1 first
                    ops.chol_update_cols(ch, *self.panels[nxt], j, w)
                    ops.chol_factor_panel(ch, *self.panels[nxt])
                    ops.chol_panel_pack(ch, *self.panels[nxt], self.bufs[nxt % 2])
                work = self._bcast(nxt)
            # remaining rank-local updates by panel p, one launch (they run under the transfer of p+1)
            ops.chol_update_multi(ch, [self.panels[q] for q in mine if q > nxt], j, w)
        return ops.chol_factor_end(ch)


class _DistributedMixin:
    """``factorize_wrapper`` of a KKT system whose matrix is assembled and factored by all ranks."""

    def _init_dist(self, nb=None, group=None):
        be = self.be
        self._chol, order = be.kkt_chol(self._h)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.panel_width = int(nb) if nb else default_panel_width(order, world)
        self.dchol = DistributedCholesky(be, self._chol, order, self.panel_width, be.device, group)
        self._Kptr, self._ldk = be.kkt_matrix(self._h, order)

    def build_kkt(self):  # MadNLP.build_kkt! on the panels this rank owns
        self.be.kkt_build_cols(self._h, self.st, self.dchol.own_ranges())

    def factorize_wrapper(self):  # MadNLP.factorize_wrapper! (src/linear_solver.jl:10)
        self.build_kkt()
        self.linear_solver.info = self.dchol.factor(self._Kptr, self._ldk)
        self.n_factorizations += 1


class HIPDistributedCondensedKKTSystem(_DistributedMixin, HIPCondensedKKTSystem):
    """:class:`HIPCondensedKKTSystem` with the assembly and the Cholesky factorisation shared by the
    ranks of ``group``; everything else (solve!, mul!, jtprod!, the kernels) is replicated."""

    def __init__(self, backend, st, nx, ind_ineq, H, A, panel_width=None, group=None):
        super().__init__(backend, st, nx, ind_ineq, H, A)
        self._init_dist(panel_width, group)


class HIPDistributedNormalKKTSystem(_DistributedMixin, HIPNormalKKTSystem):
    def __init__(self, backend, st, nx, ind_ineq, H, At, panel_width=None, group=None):
        super().__init__(backend, st, nx, ind_ineq, H, At)
        self._init_dist(panel_width, group)
