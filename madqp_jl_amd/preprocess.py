"""The step in front of the path: instance files -> a QP the solver accepts.

Host-side numpy / scipy.sparse restatement of what the reference's benchmark scripts do before they
call ``MPCSolver`` (scripts/benchmarks_cpu.jl:17-31, scripts/common.jl):

* :func:`read_qps` -- MPS / QPS (SIF) reader with the conventions of QPSReader as the scripts use it
  (``import_mps``, scripts/common.jl:21-36);
* :func:`ruiz_scale` -- ``scale_qp`` (scripts/common.jl:57-100): row / column equilibration
  ``As = Dr^-1 A Dc^-1`` and the matching transformation of H, c, bounds and starting points.  The
  reference takes ``Dr, Dc`` from HSL ``mc77`` (Ruiz' algorithm, infinity norm); that code is licensed
  and absent, so the published iteration is restated: agreement is in the property (unit row / column
  norms), not digit for digit;
* :func:`standard_form` -- ``standard_form_qp`` (scripts/common.jl:109-288) line for line;
* :func:`to_device` -- hands the result to the sparse front end (``DeviceCSR``);
* :func:`benchmark_row` -- the nine columns the scripts record (scripts/benchmarks_cpu.jl:47-55).

Not covered: ``presolve_qp`` (it delegates to ``QuadraticModels.presolve``, an external package).
"""
from __future__ import annotations

import gzip
import io
import os
from dataclasses import dataclass, field, replace

import numpy as np
import scipy.sparse as sp


@dataclass
class HostQP:
    """min c0 + c'x + x'Hx/2  s.t.  lcon <= A x <= ucon, lvar <= x <= uvar  (host, sparse)."""

    c0: float
    c: np.ndarray
    H: sp.csr_matrix  # symmetric, both triangles stored
    A: sp.csr_matrix
    lvar: np.ndarray
    uvar: np.ndarray
    lcon: np.ndarray
    ucon: np.ndarray
    x0: np.ndarray = None
    y0: np.ndarray = None
    name: str = "qp"
    varnames: list = field(default_factory=list)
    connames: list = field(default_factory=list)

    def __post_init__(self):
        n, m = len(self.c), len(self.lcon)
        if self.x0 is None:
            self.x0 = np.zeros(n)
        if self.y0 is None:
            self.y0 = np.zeros(m)
        self.H = sp.csr_matrix(self.H, shape=(n, n), dtype=np.float64)
        self.A = sp.csr_matrix(self.A, shape=(m, n), dtype=np.float64)

    nvar = property(lambda s: len(s.c))
    ncon = property(lambda s: len(s.lcon))
    nnzj = property(lambda s: int(s.A.nnz))
    nnzh = property(lambda s: int(sp.tril(s.H).nnz))  # lower triangle, as QuadraticModels counts it


# --------------------------------------------------------------------------------------------- reader
_SECTIONS = {"NAME", "OBJSENSE", "OBJSENSE", "ROWS", "COLUMNS", "RHS", "RANGES", "BOUNDS", "QUADOBJ", "QMATRIX",
             "QSECTION", "ENDATA"}


def _open_text(source):
    if isinstance(source, str) and "\n" not in source and os.path.exists(source):
        if source.endswith(".gz"):  # import_mps, scripts/common.jl:26-29
            return gzip.open(source, "rt")
        return open(source, "r")
    return io.StringIO(source)


def read_qps(source) -> HostQP:
    """Parse an MPS / QPS model (free or fixed format; names without blanks).  ``source``: path
    (``.mps/.qps/.sif/.SIF``, optionally ``.gz``) or the text itself.

    Conventions (those of QPSReader / the MPS standard): the first N row is the objective, further N rows
    are dropped; an RHS entry on the objective row is minus the constant term; RANGES: E rows use the sign
    of R, L / G rows its magnitude; default bounds 0 <= x < inf; an UP bound with a negative value on a
    variable whose lower bound is still 0 makes the lower bound -inf; QUADOBJ / QSECTION hold one
    triangle of H, QMATRIX both; integrality markers are ignored; OBJSENSE MAX negates the objective."""
    name, sense_max = "qp", False
    row_kind, row_idx, obj_row = {}, {}, None
    connames, varnames, var_idx = [], [], {}
    a_r, a_c, a_v = [], [], []
    cost = {}
    rhs, rng_val, obj_rhs = {}, {}, 0.0
    bnd_lo, bnd_up, has_lo, has_up = {}, {}, set(), set()
    h_r, h_c, h_v, h_full = [], [], [], False
    section = None
    pending_sense = False
    with _open_text(source) as fh:
        for raw in fh:
            if not raw.strip() or raw.lstrip().startswith("*"):
                continue
            tok = raw.split()
            if not raw[0].isspace():  # section header
                key = tok[0].upper()
                if key not in _SECTIONS:
                    raise ValueError(f"unknown MPS section {tok[0]!r}")
                section = key
                if key == "NAME" and len(tok) > 1:
                    name = tok[1]
                if key == "OBJSENSE":
                    if len(tok) > 1:
                        sense_max = tok[1].upper().startswith("MAX")
                    else:
                        pending_sense = True
                if key == "QMATRIX":
                    h_full = True
                if key == "ENDATA":
                    break
                continue
            if section == "OBJSENSE" and pending_sense:
                sense_max, pending_sense = tok[0].upper().startswith("MAX"), False
            elif section == "ROWS":
                kind, rname = tok[0].upper(), tok[1]
                if kind == "N":
                    if obj_row is None:
                        obj_row = rname
                    row_kind[rname] = "N"
                else:
                    row_kind[rname] = kind
                    row_idx[rname] = len(connames)
                    connames.append(rname)
            elif section == "COLUMNS":
                if len(tok) >= 3 and tok[1].upper() == "'MARKER'":
                    continue
                v = tok[0]
                if v not in var_idx:
                    var_idx[v] = len(varnames)
                    varnames.append(v)
                for rname, val in zip(tok[1::2], tok[2::2]):
                    if rname == obj_row:
                        cost[var_idx[v]] = cost.get(var_idx[v], 0.0) + float(val)
                    elif row_kind.get(rname) == "N":
                        continue
                    else:
                        a_r.append(row_idx[rname])
                        a_c.append(var_idx[v])
                        a_v.append(float(val))
            elif section in ("RHS", "RANGES"):
                pairs = tok[1:] if len(tok) % 2 == 1 else tok  # the set name is optional
                for rname, val in zip(pairs[0::2], pairs[1::2]):
                    if section == "RHS":
                        if rname == obj_row:
                            obj_rhs = float(val)
                        elif row_kind.get(rname) != "N":
                            rhs[row_idx[rname]] = float(val)
                    elif row_kind.get(rname) != "N":
                        rng_val[row_idx[rname]] = float(val)
            elif section == "BOUNDS":
                kind = tok[0].upper()
                if kind in ("FR", "MI", "PL", "BV"):
                    v = tok[2] if len(tok) >= 3 else tok[1]
                    val = 0.0
                else:
                    v, val = (tok[2], float(tok[3])) if len(tok) >= 4 else (tok[1], float(tok[2]))
                j = var_idx[v]
                if kind == "UP":
                    bnd_up[j] = val
                    has_up.add(j)
                    if val < 0.0 and j not in has_lo:
                        bnd_lo[j] = -np.inf
                elif kind == "LO":
                    bnd_lo[j] = val
                    has_lo.add(j)
                elif kind == "FX":
                    bnd_lo[j] = bnd_up[j] = val
                    has_lo.add(j)
                    has_up.add(j)
                elif kind == "FR":
                    bnd_lo[j], bnd_up[j] = -np.inf, np.inf
                elif kind == "MI":
                    bnd_lo[j] = -np.inf
                elif kind == "PL":
                    bnd_up[j] = np.inf
                elif kind == "BV":
                    bnd_lo[j], bnd_up[j] = 0.0, 1.0
                else:
                    raise ValueError(f"unknown bound type {kind!r}")
            elif section in ("QUADOBJ", "QSECTION", "QMATRIX"):
                if section == "QSECTION" and len(tok) == 1:
                    continue
                h_r.append(var_idx[tok[0]])
                h_c.append(var_idx[tok[1]])
                h_v.append(float(tok[2]))
    n, m = len(varnames), len(connames)
    c = np.zeros(n)
    for j, v in cost.items():
        c[j] = v
    lvar, uvar = np.zeros(n), np.full(n, np.inf)
    for j, v in bnd_lo.items():
        lvar[j] = v
    for j, v in bnd_up.items():
        uvar[j] = v
    lcon, ucon = np.full(m, -np.inf), np.full(m, np.inf)
    for rname, i in row_idx.items():
        b, kind = rhs.get(i, 0.0), row_kind[rname]
        if kind == "E":
            lcon[i] = ucon[i] = b
        elif kind == "L":
            ucon[i] = b
        else:
            lcon[i] = b
        if i in rng_val:
            r = rng_val[i]
            if kind == "E":
                lcon[i], ucon[i] = (b, b + abs(r)) if r >= 0 else (b - abs(r), b)
            elif kind == "L":
                lcon[i] = b - abs(r)
            else:
                ucon[i] = b + abs(r)
    A = sp.csr_matrix((a_v, (a_r, a_c)), shape=(m, n)) if m else sp.csr_matrix((0, n))
    H = sp.csr_matrix((h_v, (h_r, h_c)), shape=(n, n))
    if not h_full:  # one triangle given: mirror it
        H = H + H.T - sp.diags(H.diagonal())
    c0 = -obj_rhs
    if sense_max:
        c, H, c0 = -c, -H, -c0
    return HostQP(c0, c, H, A, lvar, uvar, lcon, ucon, name=name, varnames=varnames, connames=connames)


# --------------------------------------------------------------------------------------------- scaling
def ruiz_factors(A: sp.csr_matrix, max_iter: int = 100, tol: float = 1e-8):
    """Ruiz' simultaneous row / column equilibration in the infinity norm (the algorithm behind HSL mc77 as
    called at scripts/common.jl:66): returns (Dr, Dc) such that Dr^-1 A Dc^-1 has unit row and column
    maxima up to ``tol``.  Empty rows / columns keep the factor 1."""
    A = sp.csr_matrix(abs(A), dtype=np.float64)
    m, n = A.shape
    Dr, Dc = np.ones(m), np.ones(n)
    for _ in range(max_iter):
        rmax = np.asarray(A.max(axis=1).todense()).ravel() if A.nnz else np.zeros(m)
        cmax = np.asarray(A.max(axis=0).todense()).ravel() if A.nnz else np.zeros(n)
        if max(np.max(np.abs(1.0 - rmax[rmax > 0]), initial=0.0), np.max(np.abs(1.0 - cmax[cmax > 0]), initial=0.0)) <= tol:
            break
        r = np.where(rmax > 0, np.sqrt(rmax), 1.0)
        cc = np.where(cmax > 0, np.sqrt(cmax), 1.0)
        A = sp.diags(1.0 / r) @ A @ sp.diags(1.0 / cc)
        Dr *= r
        Dc *= cc
    return Dr, Dc


def ruiz_scale(qp: HostQP, **kw):
    """``scale_qp`` (scripts/common.jl:57-100).  Returns (scaled qp, Dr, Dc); a solution xs of the scaled
    problem is x = xs / Dc in the original variables, multipliers y = ys / Dr."""
    Dr, Dc = ruiz_factors(qp.A, **kw)
    iDr, iDc = sp.diags(1.0 / Dr), sp.diags(1.0 / Dc)
    scaled = replace(qp, c=qp.c / Dc, H=sp.csr_matrix(iDc @ qp.H @ iDc), A=sp.csr_matrix(iDr @ qp.A @ iDc),
                     lvar=qp.lvar * Dc, uvar=qp.uvar * Dc, lcon=qp.lcon / Dr, ucon=qp.ucon / Dr,
                     x0=qp.x0 * Dc, y0=qp.y0 / Dr)
    return scaled, Dr, Dc


# ---------------------------------------------------------------------------------------- standard form
def standard_form(qp: HostQP) -> HostQP:
    """``standard_form_qp`` (scripts/common.jl:109-288): slack s = A x on inequality rows, upper bounds of
    range-bounded x / s moved into equality rows x + w = xu with w >= 0; equality rows and fixed variables
    kept as they are.  Variables [x; s; w], constraints [original rows; range rows]."""
    n, m = qp.nvar, qp.ncon
    lvar, uvar, lcon, ucon = qp.lvar, qp.uvar, qp.lcon, qp.ucon
    ind_ineq = [i for i in range(m) if lcon[i] < ucon[i]]  # :140-144
    ns = len(ind_ineq)
    ind_rng, ind_fixed, xu = [], [], []
    for i in range(n):  # :147-164
        if lvar[i] == uvar[i]:
            ind_fixed.append(i)
        elif -np.inf < lvar[i] < uvar[i] < np.inf:
            ind_rng.append(i)
            xu.append(uvar[i])
    for k, i in enumerate(ind_ineq):  # :167-176
        if -np.inf < lcon[i] < ucon[i] < np.inf:
            ind_rng.append(n + k)
            xu.append(ucon[i])
    nw = len(ind_rng)
    nvar, ncon = n + ns + nw, m + nw
    H = sp.csr_matrix((nvar, nvar))
    Hc = qp.H.tocoo()
    H = sp.csr_matrix((Hc.data, (Hc.row, Hc.col)), shape=(nvar, nvar))  # :185
    Ac = qp.A.tocoo()
    Bi, Bj, Bx = [], [], []
    for k, i in enumerate(ind_ineq):  # slack contribution A x - s = 0, :192-197
        Bi.append(i)
        Bj.append(n + k)
        Bx.append(-1.0)
    for k, i in enumerate(ind_rng):  # x + w = xu, :199-208
        Bi += [m + k, m + k]
        Bj += [i, k + n + ns]
        Bx += [1.0, 1.0]
    A = sp.csr_matrix((np.concatenate([Ac.data, Bx]), (np.concatenate([Ac.row, Bi]).astype(np.int64),
                                                       np.concatenate([Ac.col, Bj]).astype(np.int64))),
                      shape=(ncon, nvar))
    lcon_, ucon_ = np.zeros(ncon), np.zeros(ncon)
    for i in range(m):  # :216-226
        if not lcon[i] < ucon[i]:
            lcon_[i], ucon_[i] = lcon[i], ucon[i]
    for k in range(nw):  # :227-230
        lcon_[m + k] = ucon_[m + k] = xu[k]
    lvar_ = np.concatenate([lvar, lcon[ind_ineq], np.zeros(nw)])  # :232
    uvar_ = np.concatenate([uvar, ucon[ind_ineq], np.full(nw, np.inf)])
    uvar_[ind_rng] = np.inf  # :235
    uvar_[ind_fixed] = uvar[ind_fixed]  # :237
    return HostQP(qp.c0, np.concatenate([qp.c, np.zeros(ns + nw)]), H, A, lvar_, uvar_, lcon_, ucon_,
                  x0=np.concatenate([qp.x0, np.zeros(ns + nw)]), y0=np.concatenate([qp.y0, np.zeros(nw)]),
                  name=qp.name + "-std")


# ------------------------------------------------------------------------------------------- hand-over
def to_device(qp: HostQP, backend, sparse: bool = True):
    """A :class:`DeviceQP` for the HIP path: the Jacobian as ``DeviceCSR`` (or dense), H dense or None."""
    import torch

    from .qp import DeviceCSR, DeviceQP

    dev = backend.device
    f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    Hc = qp.H.tocoo()
    if qp.H.nnz == 0:
        H = None
    elif sparse and np.all(Hc.row == Hc.col):
        H = f(qp.H.diagonal())  # diagonal Hessian: kept as a vector (madqp_kkt_set_hdiag)
    else:
        H = f(qp.H.toarray())
    Ac = qp.A.tocoo()
    A = DeviceCSR(dev, qp.ncon, qp.nvar, Ac.row, Ac.col, Ac.data) if sparse else f(qp.A.toarray())
    return DeviceQP(H, f(qp.c), A, f(qp.lvar), f(qp.uvar), f(qp.lcon), f(qp.ucon), f(qp.x0), qp.c0, f(qp.y0), qp.name)


def benchmark_row(qp: HostQP, result: dict, total_time: float, linear_solver_time: float):
    """The nine numbers the reference's scripts record per instance (scripts/benchmarks_cpu.jl:47-55)."""
    return (qp.nvar, qp.ncon, qp.nnzj, qp.nnzh, int(result["status"]), int(result["iter"]),
            float(result["objective"]), float(total_time), float(linear_solver_time))


# ------------------------------------------------------------------------ a CONT-type instance generator
def boundary_control_qp(N: int, alpha: float = 0.01, ymax: float = 0.8) -> HostQP:
    """Elliptic boundary-control QP on an N x N grid, the problem class of the Maros-Meszaros CONT-xxx
    instances (BASELINE configs[2] names CONT-300; its data file is not available offline, so this is a
    stand-in of the same shape, not that instance):

        min  h^2/2 sum (y_ij - yd_ij)^2 + alpha h/2 sum u_k^2
        s.t. 4 y_ij - y_(i-1)j - y_(i+1)j - y_i(j-1) - y_i(j+1) = 0   (5-point Laplacian; a neighbour outside
             0 <= y <= ymax,  0 <= u <= 1                              the grid is a boundary control u_k)

    n = N^2 + 4N variables [y; u_left; u_right; u_bottom; u_top], m = N^2 equality rows with <= 5 entries,
    diagonal Hessian.  N = 300: n = 91 200, m = 90 000 (CONT-300: 90 597 x 90 298)."""
    h = 1.0 / (N + 1)
    idx = lambda i, j: i * N + j
    ul, ur, ub, ut = N * N, N * N + N, N * N + 2 * N, N * N + 3 * N
    rows, cols, vals = [], [], []
    for i in range(N):
        for j in range(N):
            r = idx(i, j)
            rows.append(r); cols.append(r); vals.append(4.0)
            for (ii, jj, ctrl) in ((i - 1, j, ul + j), (i + 1, j, ur + j), (i, j - 1, ub + i), (i, j + 1, ut + i)):
                inside = 0 <= ii < N and 0 <= jj < N
                rows.append(r)
                cols.append(idx(ii, jj) if inside else ctrl)
                vals.append(-1.0)
    n, m = N * N + 4 * N, N * N
    g = (np.arange(N) + 1) * h
    X, Y = np.meshgrid(g, g, indexing="ij")
    yd = (1.0 + 2.0 * (X * (X - 1.0) + Y * (Y - 1.0))).ravel()
    hd = np.concatenate([np.full(N * N, h * h), np.full(4 * N, alpha * h)])
    c = np.concatenate([-h * h * yd, np.zeros(4 * N)])
    lvar = np.zeros(n)
    uvar = np.concatenate([np.full(N * N, ymax), np.ones(4 * N)])
    return HostQP(0.5 * h * h * float(yd @ yd), c, sp.diags(hd).tocsr(), sp.csr_matrix((vals, (rows, cols)), shape=(m, n)),
                  lvar, uvar, np.zeros(m), np.zeros(m), name=f"boundary-control-{N}")
