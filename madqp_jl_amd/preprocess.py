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
* :func:`standard_form` -- ``standard_form_qp`` (scripts/common.jl:156-288) line for line;
* :func:`to_device` -- hands the result to the sparse front end (``DeviceCSR``);
* :func:`benchmark_row` -- the nine columns the scripts record (scripts/benchmarks_cpu.jl:47-55).

* :func:`presolve` -- ``presolve_qp`` (scripts/common.jl:109-126) delegates to ``QuadraticModels.presolve``, an
  un-vendored package (compat ``QuadraticModels`` in scripts/Project.toml); its documented basic reductions are
  restated here -- fixed variables, empty rows, singleton rows, unconstrained linear variables, rows made redundant
  by the variable bounds -- with the postsolve that maps a primal-dual solution back.  Same contract as the script
  uses: a reduced model and a flag, ``False`` when presolve alone settles the instance (solved, infeasible or
  unbounded).
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


def write_qps(qp: HostQP, path) -> None:
    """Write the model as a free-format QPS file that :func:`read_qps` reads back exactly (values with ``repr``):
    the inverse of the reader's conventions -- objective row first, ``RHS`` of the objective = -c0, ranged rows as
    ``L`` + ``RANGES``, bounds relative to the MPS default 0 <= x < inf, ``QUADOBJ`` = upper triangle of H.
    Rows without any finite bound are written as ``N`` rows (the reader drops those)."""
    n, m = qp.nvar, qp.ncon
    vn = qp.varnames if len(qp.varnames) == n else [f"X{j + 1}" for j in range(n)]
    cn = qp.connames if len(qp.connames) == m else [f"R{i + 1}" for i in range(m)]
    At = qp.A.tocsc()
    f = lambda v: repr(float(v))
    out = [f"NAME {qp.name.replace(' ', '_')}", "ROWS", " N OBJ"]
    kind = []
    for i in range(m):
        lo, hi = qp.lcon[i], qp.ucon[i]
        k = "E" if lo == hi else "N" if (lo == -np.inf and hi == np.inf) else "G" if hi == np.inf else "L"
        kind.append(k)
        out.append(f" {k} {cn[i]}")
    out.append("COLUMNS")
    for j in range(n):
        out.append(f" {vn[j]} OBJ {f(qp.c[j])}")
        col = At.getcol(j)
        for i, v in zip(col.indices, col.data):
            if v != 0.0:
                out.append(f" {vn[j]} {cn[i]} {f(v)}")
    out.append("RHS")
    if qp.c0 != 0.0:
        out.append(f" RHS OBJ {f(-qp.c0)}")
    for i in range(m):
        b = qp.lcon[i] if kind[i] in ("E", "G") else qp.ucon[i]
        if kind[i] != "N" and b != 0.0:
            out.append(f" RHS {cn[i]} {f(b)}")
    rng = [i for i in range(m) if kind[i] == "L" and qp.lcon[i] > -np.inf]
    if rng:
        out.append("RANGES")
        out += [f" RNG {cn[i]} {f(qp.ucon[i] - qp.lcon[i])}" for i in rng]
    out.append("BOUNDS")
    for j in range(n):
        lo, hi = qp.lvar[j], qp.uvar[j]
        if lo == -np.inf and hi == np.inf:
            out.append(f" FR BND {vn[j]}")
        elif lo == hi:
            out.append(f" FX BND {vn[j]} {f(lo)}")
        else:
            if lo == -np.inf:
                out.append(f" MI BND {vn[j]}")
            elif lo != 0.0 or hi < 0.0:
                out.append(f" LO BND {vn[j]} {f(lo)}")
            if hi != np.inf:
                out.append(f" UP BND {vn[j]} {f(hi)}")
    Hu = sp.triu(qp.H).tocoo()
    if Hu.nnz:
        out.append("QUADOBJ")
        out += [f" {vn[i]} {vn[j]} {f(v)}" for i, j, v in zip(Hu.row, Hu.col, Hu.data) if v != 0.0]
    out.append("ENDATA")
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wt") as fh:
        fh.write("\n".join(out) + "\n")


# -------------------------------------------------------------------------------------------- presolve
@dataclass
class Presolved:
    """Outcome of :func:`presolve`.  ``flag`` as in ``presolve_qp`` (scripts/common.jl:109-126): True = ``qp`` is the
    reduced model to solve; False = nothing left to solve (``status``: "solved", "infeasible" or "unbounded")."""

    qp: HostQP
    flag: bool
    status: str  # "reduced" | "unchanged" | "solved" | "infeasible" | "unbounded"
    original: HostQP
    keep_var: np.ndarray
    keep_con: np.ndarray
    x_removed: np.ndarray  # full length; values of the eliminated variables
    ops: list  # the reductions in the order they were applied; the postsolve undoes them in reverse

    def postsolve(self, x=None, y=None, zl=None, zu=None):
        """Solution of the reduced model -> primal-dual point of the ORIGINAL model (stationarity convention of the
        solver: H x + c + A'y - zl + zu = 0).  With status "solved" call it without arguments."""
        o = self.original
        n, m = o.nvar, o.ncon
        xf = self.x_removed.copy()
        yf, zlf, zuf = np.zeros(m), np.zeros(n), np.zeros(n)
        if len(self.keep_var):
            xf[self.keep_var] = x
        if y is not None and len(self.keep_con):
            yf[self.keep_con] = y
        if zl is not None and zu is not None:
            zlf[self.keep_var], zuf[self.keep_var] = zl, zu
            At = o.A.tocsc()
            g = o.H @ xf + o.c  # gradient of the objective at the (complete) primal point
            for op in reversed(self.ops):
                if op[0] == "var":  # eliminated variable: the reduced cost goes to whichever bound holds it
                    jv = op[1]
                    col = At.getcol(jv)
                    r = g[jv] + col.data @ yf[col.indices]
                    zlf[jv], zuf[jv] = max(r, 0.0), max(-r, 0.0)
                elif op[0] == "singleton":  # a bound that came from this row: its multiplier belongs to the row
                    _, iv, jv, a, set_lo, set_hi = op  # (a y_i = -zl_j for the lower, a y_i = zu_j for the upper bound)
                    if set_lo:
                        yf[iv] -= zlf[jv] / a
                        zlf[jv] = 0.0
                    if set_hi:
                        yf[iv] += zuf[jv] / a
                        zuf[jv] = 0.0
        obj = o.c0 + o.c @ xf + 0.5 * xf @ (o.H @ xf)
        return dict(x=xf, y=yf, zl=zlf, zu=zuf, objective=float(obj))


def presolve(qp: HostQP, feas_tol: float = 1e-9, max_pass: int = 50) -> Presolved:
    """Basic presolve (the reductions ``QuadraticModels.presolve`` documents), repeated until nothing changes:

    1. fixed variables (lvar == uvar) are substituted out;
    2. empty rows are checked (lcon <= 0 <= ucon) and dropped;
    3. singleton rows ``l <= a x_j <= u`` become bounds on x_j;
    4. variables that appear in no row and no quadratic term go to the bound their cost points at;
    5. rows whose activity range implied by the variable bounds lies inside [lcon, ucon] are dropped.
    """
    n, m = qp.nvar, qp.ncon
    A, At, H = qp.A.tocsr(), qp.A.tocsc(), qp.H.tocsr()
    lvar, uvar, lcon, ucon = (np.array(v, dtype=np.float64) for v in (qp.lvar, qp.uvar, qp.lcon, qp.ucon))
    c, c0 = np.array(qp.c, dtype=np.float64), float(qp.c0)
    av, ac = np.ones(n, dtype=bool), np.ones(m, dtype=bool)
    xr = np.zeros(n)
    ops = []
    pat, patH = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape), H.copy()
    patH.data = np.ones(H.nnz)
    status = None

    def remove_var(j, v):
        nonlocal c0
        xr[j] = v
        hj = H.getrow(j)
        c0 += c[j] * v + 0.5 * H[j, j] * v * v
        c[hj.indices] += hj.data * v  # H symmetric: column j = row j (c[j] itself is no longer used)
        col = At.getcol(j)
        lcon[col.indices] -= col.data * v
        ucon[col.indices] -= col.data * v
        av[j] = False
        ops.append(("var", int(j)))

    for _ in range(max_pass):
        changed = False
        if np.any(av & (lvar > uvar + feas_tol * np.maximum(1.0, np.abs(lvar)))):
            status = "infeasible"
            break
        for j in np.flatnonzero(av & (lvar >= uvar)):  # 1. fixed (also after a singleton row closed the interval)
            remove_var(j, lvar[j])
            changed = True
        rcount = np.asarray(pat @ av.astype(np.float64)).ravel()
        for i in np.flatnonzero(ac & (rcount == 0)):  # 2. empty rows
            if lcon[i] > feas_tol * max(1.0, abs(lcon[i])) or ucon[i] < -feas_tol * max(1.0, abs(ucon[i])):
                status = "infeasible"
                break
            ac[i] = False
            changed = True
        if status:
            break
        for i in np.flatnonzero(ac & (rcount == 1)):  # 3. singleton rows
            row = A.getrow(i)
            k = np.flatnonzero(av[row.indices])
            if len(k) != 1 or row.data[k[0]] == 0.0:
                continue
            j, a = int(row.indices[k[0]]), float(row.data[k[0]])
            lo, hi = (lcon[i] / a, ucon[i] / a) if a > 0 else (ucon[i] / a, lcon[i] / a)
            set_lo, set_hi = bool(lo > lvar[j]), bool(hi < uvar[j])
            if set_lo:
                lvar[j] = lo
            if set_hi:
                uvar[j] = hi
            ops.append(("singleton", int(i), j, a, set_lo, set_hi))
            ac[i] = False
            changed = True
        ccount = np.asarray(pat.T @ ac.astype(np.float64)).ravel()
        hcount = np.asarray(patH @ av.astype(np.float64)).ravel()
        for j in np.flatnonzero(av & (ccount == 0) & (hcount == 0)):  # 4. unconstrained linear variables
            if lvar[j] > uvar[j]:
                continue  # caught as infeasible at the top of the next pass
            if c[j] > 0.0:
                v = lvar[j]
            elif c[j] < 0.0:
                v = uvar[j]
            else:
                v = lvar[j] if np.isfinite(lvar[j]) else (uvar[j] if np.isfinite(uvar[j]) else 0.0)
            if not np.isfinite(v):
                status = "unbounded"
                break
            remove_var(j, v)
            changed = True
        if status:
            break
        # 5. redundant rows: activity range over the box of the active variables
        if np.any(ac) and np.any(av):
            Aa = A.multiply(av.astype(np.float64)[None, :]).tocsr()
            pos, neg = Aa.maximum(0), Aa.minimum(0)
            with np.errstate(invalid="ignore"):
                lo_b, hi_b = np.where(av, lvar, 0.0), np.where(av, uvar, 0.0)
                amin = _bound_dot(pos, lo_b) + _bound_dot(neg, hi_b)
                amax = _bound_dot(pos, hi_b) + _bound_dot(neg, lo_b)
            tol = feas_tol * np.maximum(1.0, np.maximum(np.abs(lcon), np.abs(ucon)))
            tol[~np.isfinite(tol)] = feas_tol
            if np.any(ac & ((amin > ucon + tol) | (amax < lcon - tol))):
                status = "infeasible"
                break
            red = ac & (rcount > 1) & (amin >= lcon) & (amax <= ucon) & (lcon < ucon)
            if np.any(red):
                ac[red] = False
                changed = True
        if not changed:
            break
    kv, kc = np.flatnonzero(av), np.flatnonzero(ac)
    if status in ("infeasible", "unbounded"):
        return Presolved(qp, False, status, qp, np.arange(n), np.arange(m), xr, [])
    if len(kv) == 0:
        return Presolved(qp, False, "solved", qp, kv, kc[:0], xr, ops)
    if len(kv) == n and len(kc) == m:
        return Presolved(qp, True, "unchanged", qp, kv, kc, xr, [])
    names = lambda lst, idx: [lst[i] for i in idx] if lst else []
    red = HostQP(c0, c[kv], H[kv][:, kv], A[kc][:, kv], lvar[kv], uvar[kv], lcon[kc], ucon[kc], x0=qp.x0[kv],
                 y0=qp.y0[kc], name=qp.name + "-ps", varnames=names(qp.varnames, kv), connames=names(qp.connames, kc))
    return Presolved(red, True, "reduced", qp, kv, kc, xr, ops)


def _bound_dot(M, b):
    """M @ b for a sign-definite sparse M and bounds b that may be infinite (0 * inf = 0)."""
    fin = np.where(np.isfinite(b), b, 0.0)
    out = np.asarray(M @ fin).ravel()
    inf_hit = np.asarray(abs(M) @ (~np.isfinite(b)).astype(np.float64)).ravel() > 0
    if np.any(inf_hit):
        sgn = np.asarray(M @ np.where(np.isfinite(b), 0.0, np.sign(b))).ravel()
        out = np.where(inf_hit, np.where(sgn > 0, np.inf, -np.inf), out)
    return out


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
    """``standard_form_qp`` (scripts/common.jl:156-288): slack s = A x on inequality rows, upper bounds of
    range-bounded x / s moved into equality rows x + w = xu with w >= 0; equality rows and fixed variables
    kept as they are.  Variables [x; s; w], constraints [original rows; range rows]."""
    n, m = qp.nvar, qp.ncon
    lvar, uvar, lcon, ucon = qp.lvar, qp.uvar, qp.lcon, qp.ucon
    ind_ineq = [i for i in range(m) if lcon[i] < ucon[i]]  # :163-168
    ns = len(ind_ineq)
    ind_rng, ind_fixed, xu = [], [], []
    for i in range(n):  # :170-187
        if lvar[i] == uvar[i]:
            ind_fixed.append(i)
        elif -np.inf < lvar[i] < uvar[i] < np.inf:
            ind_rng.append(i)
            xu.append(uvar[i])
    for k, i in enumerate(ind_ineq):  # :189-199
        if -np.inf < lcon[i] < ucon[i] < np.inf:
            ind_rng.append(n + k)
            xu.append(ucon[i])
    nw = len(ind_rng)
    nvar, ncon = n + ns + nw, m + nw
    H = sp.csr_matrix((nvar, nvar))
    Hc = qp.H.tocoo()
    H = sp.csr_matrix((Hc.data, (Hc.row, Hc.col)), shape=(nvar, nvar))  # :207
    Ac = qp.A.tocoo()
    Bi, Bj, Bx = [], [], []
    for k, i in enumerate(ind_ineq):  # slack contribution A x - s = 0, :213-219
        Bi.append(i)
        Bj.append(n + k)
        Bx.append(-1.0)
    for k, i in enumerate(ind_rng):  # x + w = xu, :220-230
        Bi += [m + k, m + k]
        Bj += [i, k + n + ns]
        Bx += [1.0, 1.0]
    A = sp.csr_matrix((np.concatenate([Ac.data, Bx]), (np.concatenate([Ac.row, Bi]).astype(np.int64),
                                                       np.concatenate([Ac.col, Bj]).astype(np.int64))),
                      shape=(ncon, nvar))
    lcon_, ucon_ = np.zeros(ncon), np.zeros(ncon)
    for i in range(m):  # :238-248
        if not lcon[i] < ucon[i]:
            lcon_[i], ucon_[i] = lcon[i], ucon[i]
    for k in range(nw):  # :249-252
        lcon_[m + k] = ucon_[m + k] = xu[k]
    lvar_ = np.concatenate([lvar, lcon[ind_ineq], np.zeros(nw)])  # :254
    uvar_ = np.concatenate([uvar, ucon[ind_ineq], np.full(nw, np.inf)])
    uvar_[ind_rng] = np.inf  # :257
    uvar_[ind_fixed] = uvar[ind_fixed]  # :259
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
