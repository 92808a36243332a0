"""Dense QP held in HBM and the on-device synthetic instance generator.

    min c0 + q'x + x'Hx/2   s.t.  lcon <= A x <= ucon,  lvar <= x <= uvar

``H`` is a full symmetric (nx, nx) tensor (or None for an LP) and ``A`` an
(m, nx) row-major tensor: the layouts the MFMA kernels consume directly.
The synthetic family is the one of BASELINE.md section 3; entries are produced
in place on the device by ``madqp_gen_*`` and are bit-identical to the CPU
generator the tests use.
"""
from __future__ import annotations

import math

import numpy as np
import torch

_MASK = 0xFFFFFFFFFFFFFFFF
STREAM_A, STREAM_H, STREAM_Q = 1, 2, 3


def _mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def stream_key(seed: int, stream: int) -> int:
    """Per-(seed, stream) generator key (host side of ``madqp_gen_normal``)."""
    return _mix64((seed ^ ((stream * 0xD1B54A32D192ED03) & _MASK)) & _MASK)


class DeviceCSR:
    """A sparse Jacobian on the device: CSR of A (``ptr, col, val``; m rows, column indices ascending within
    a row) plus what the transposed products need -- the CSR of A' (``t_ptr, t_col``) and the permutation
    ``t_perm`` with ``val_of_At = val[t_perm]`` -- and ``row`` (row index of every stored entry).  The
    pattern is built once on the host (A is constant for a QP), the reference's ``coo_to_csr``
    (src/utils.jl:148-197) in spirit; values can be rescaled on the device without touching it."""

    def __init__(self, device, m, n, rows, cols, vals):
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float64)
        order = np.lexsort((cols, rows))  # by row, then column
        rows, cols, vals = rows[order], cols[order], vals[order]
        if len(rows) > 1 and np.any((rows[1:] == rows[:-1]) & (cols[1:] == cols[:-1])):
            raise ValueError("duplicate entries in the sparse Jacobian")
        t_perm = np.lexsort((rows, cols))  # by column, then row: the order of the entries of A'
        dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)
        count = lambda idx, k: np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=k))])
        self.m, self.n, self.nnz = int(m), int(n), len(vals)
        self.ptr, self.col, self.val = dev(count(rows, m), torch.int64), dev(cols, torch.int64), dev(vals, torch.float64)
        self.row = dev(rows, torch.int64)
        self.t_ptr, self.t_col = dev(count(cols, n), torch.int64), dev(rows[t_perm], torch.int64)
        self.t_perm = dev(t_perm, torch.int64)

    @classmethod
    def from_dense(cls, device, A):
        A = np.asarray(A, dtype=np.float64)
        r, c = np.nonzero(A)
        return cls(device, A.shape[0], A.shape[1], r, c, A[r, c])

    def scaled(self, row_scale):
        """A copy that shares the pattern, with row i multiplied by ``row_scale[i]``."""
        out = object.__new__(DeviceCSR)
        out.__dict__.update(self.__dict__)
        out.val = (self.val * row_scale[self.row]).contiguous()
        return out

    @property
    def t_val(self):
        return self.val[self.t_perm].contiguous()

    def row_absmax(self):
        out = torch.zeros(self.m, dtype=torch.float64, device=self.val.device)
        return out.scatter_reduce(0, self.row, self.val.abs(), reduce="amax", include_self=True)

    def to_dense(self):
        A = torch.zeros((self.m, self.n), dtype=torch.float64, device=self.val.device)
        A[self.row, self.col] = self.val
        return A


class DeviceQP:
    def __init__(self, H, q, A, lvar, uvar, lcon, ucon, x0, c0=0.0, y0=None, name="qp"):
        """``A``: (m, nx) row-major tensor, or a :class:`DeviceCSR` (sparse front end)."""
        self.H, self.q, self.A = H, q, A
        self.lvar, self.uvar, self.lcon, self.ucon, self.x0 = lvar, uvar, lcon, ucon, x0
        self.c0 = float(c0)
        self.y0 = torch.zeros_like(lcon) if y0 is None else y0
        self.name = name
        self.nvar, self.ncon = q.numel(), lcon.numel()

    @classmethod
    def from_numpy(cls, device, H, q, A, lvar, uvar, lcon, ucon, x0, c0=0.0, y0=None, name="qp", sparse=False):
        f = lambda a: None if a is None else torch.as_tensor(
            np.ascontiguousarray(a, dtype=np.float64), device=device)
        n = len(q)
        Hn = None if (H is None or not np.any(H)) else f(H)
        A2 = np.asarray(A, dtype=np.float64).reshape(len(lcon), n)
        An = DeviceCSR.from_dense(device, A2) if sparse else f(A2)
        return cls(Hn, f(q), An, f(lvar), f(uvar), f(lcon), f(ucon), f(x0), c0, f(y0), name)

    def eliminate_fixed(self):
        """``MadNLP.MakeParameter`` (the fixed-variable treatment src/utils.jl:81 selects for every KKT system that
        is not condensed): variables with ``lvar == uvar`` leave the problem as parameters.  Returns None when there
        are none, else ``(reduced DeviceQP, free, fixed, xfix, shift)`` with index tensors ``free`` / ``fixed``, the
        fixed values and ``shift = A[:, fixed] xfix`` (the rows of the reduced model are ``A_f x_f`` in
        ``[lcon - shift, ucon - shift]``).  All on the device."""
        mask = self.lvar == self.uvar
        if not bool(mask.any()):
            return None
        fixed, free = torch.nonzero(mask).flatten(), torch.nonzero(~mask).flatten()
        xf = self.lvar[fixed]
        c0 = self.c0 + float(self.q[fixed] @ xf)
        q = self.q[free].clone()
        H = self.H
        if H is not None and H.dim() == 1:
            c0 += 0.5 * float((H[fixed] * xf) @ xf)
            H = H[free].contiguous()
        elif H is not None:
            Hx = H.index_select(1, fixed) @ xf  # H[:, fixed] xfix
            c0 += 0.5 * float(Hx[fixed] @ xf)
            q += Hx[free]
            H = H.index_select(0, free).index_select(1, free).contiguous()
        if isinstance(self.A, DeviceCSR):
            a = self.A
            isfix = mask[a.col]
            shift = torch.zeros(self.ncon, dtype=torch.float64, device=self.q.device)
            shift.index_add_(0, a.row[isfix], a.val[isfix] * self.lvar[a.col[isfix]])
            newcol = torch.cumsum((~mask).to(torch.int64), 0) - 1
            keep = ~isfix
            A = DeviceCSR(self.q.device, self.ncon, free.numel(), a.row[keep].cpu().numpy(),
                          newcol[a.col[keep]].cpu().numpy(), a.val[keep].cpu().numpy())
        else:
            shift = self.A.index_select(1, fixed) @ xf
            A = self.A.index_select(1, free).contiguous()
        red = DeviceQP(H, q, A, self.lvar[free].clone(), self.uvar[free].clone(), self.lcon - shift, self.ucon - shift,
                       self.x0[free].clone(), c0, self.y0, self.name + "-free")
        return red, free, fixed, xf, shift

    @classmethod
    def synthetic(cls, backend, seed: int, n: int, m: int, family: str = "wigner"):
        """0 <= x <= 1, 0 <= Ax <= 1, x0 = 0; A ~ N(0,1); H by family: "wigner" (Wigner + 3 I, O(n^2) to
        generate: the large configs), "dummy" (H = G'G + 100 I with G n x n N(0,1) from the H stream -- R R' + 100 I of
        MadNLPTests.DenseDummyQP, test/runtests.jl:9, with R = G': n_x <= ~10 000, one call of the library's own SYRK),
        "lp" (no Hessian)."""
        dev = backend.device
        A = torch.empty((m, n), dtype=torch.float64, device=dev)
        backend.gen_normal(stream_key(seed, STREAM_A), 0, A)
        q = torch.empty(n, dtype=torch.float64, device=dev)
        backend.gen_normal(stream_key(seed, STREAM_Q), 0, q)
        H = None
        if family == "wigner":
            H = torch.empty((n, n), dtype=torch.float64, device=dev)
            backend.gen_wigner(stream_key(seed, STREAM_H), n, 1.0 / math.sqrt(n), H)
        elif family == "dummy":
            G = torch.empty((n, n), dtype=torch.float64, device=dev)
            backend.gen_normal(stream_key(seed, STREAM_H), 0, G)
            H = torch.zeros((n, n), dtype=torch.float64, device=dev)
            d = torch.full((n,), 100.0, dtype=torch.float64, device=dev)
            # lower triangle of G'G + 100 I (column-major lower = row-major upper of the same symmetric matrix)
            backend.syrk_assemble(n, n, G, n, None, None, n, d, H, n)
            H = torch.triu(H) + torch.triu(H, 1).T
            del G
        elif family != "lp":
            raise ValueError(family)
        z = lambda k, v: torch.full((k,), v, dtype=torch.float64, device=dev)
        return cls(H, q, A, z(n, 0.0), z(n, 1.0), z(m, 0.0), z(m, 1.0), z(n, 0.0),
                   name=f"synthetic-{family}-n{n}-m{m}-s{seed}")
