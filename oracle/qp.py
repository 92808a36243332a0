"""Dense QP container, the synthetic-instance generator and the known-answer problems.

Oracle (test infrastructure) -- see ``oracle/__init__.py``.

Problem form (QuadraticModels convention used by the reference, objective and
gradient as in ``scripts/qp_gpu.jl:29-40``)::

    min  c0 + q'x + 1/2 x'Hx    s.t.  lcon <= A x <= ucon,  lvar <= x <= uvar

The generator is counter based and position addressable so that the HIP
library (``madqp_gen_*`` in ``include/madqp.h``) produces bit-identical
entries on the device: ``h = mix64(key + idx)``, the four 16-bit fields of
``h`` are summed (Irwin-Hall, exact integer arithmetic), centred and scaled
to unit variance with one multiplication.  No transcendental function is
involved, hence CPU == GPU bit for bit (SURVEY.md 8d asks for a
position-addressable splitmix64 generator; Box-Muller was replaced by
Irwin-Hall-4 to make the device copy exact).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

U64 = np.uint64
_GAMMA = U64(0x9E3779B97F4A7C15)
_M1 = U64(0xBF58476D1CE4E5B9)
_M2 = U64(0x94D049BB133111EB)
_STREAM_MUL = 0xD1B54A32D192ED03
# 1/sqrt((65536^2-1)/3): std of the sum of four uniform 16-bit integers
GEN_SCALE = float.fromhex("0x1.bb67ae86627e7p-16")
STREAM_A, STREAM_H, STREAM_Q = 1, 2, 3


def mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 output function applied to ``z + GAMMA`` (uint64, wrapping)."""
    with np.errstate(over="ignore"):
        z = z + _GAMMA
        z = (z ^ (z >> U64(30))) * _M1
        z = (z ^ (z >> U64(27))) * _M2
        return z ^ (z >> U64(31))


def stream_key(seed: int, stream: int) -> int:
    """Per-(seed, stream) key; computed on the host and handed to the device."""
    k = (seed ^ ((stream * _STREAM_MUL) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    return int(mix64(np.array([k], dtype=U64))[0])


def gen_normal(key: int, idx: np.ndarray) -> np.ndarray:
    """Unit-variance Irwin-Hall-4 variate for every counter in ``idx`` (uint64)."""
    with np.errstate(over="ignore"):
        h = mix64(U64(key) + idx.astype(U64))
    g = (
        (h & U64(0xFFFF)).astype(np.int64)
        + ((h >> U64(16)) & U64(0xFFFF)).astype(np.int64)
        + ((h >> U64(32)) & U64(0xFFFF)).astype(np.int64)
        + (h >> U64(48)).astype(np.int64)
        - 131070
    )
    return g.astype(np.float64) * GEN_SCALE


def gen_A(seed: int, m: int, n: int) -> np.ndarray:
    """Dense constraint matrix, entry (k, i) from counter k*n + i (row-major)."""
    idx = np.arange(m * n, dtype=U64)
    return gen_normal(stream_key(seed, STREAM_A), idx).reshape(m, n)


def gen_q(seed: int, n: int) -> np.ndarray:
    return gen_normal(stream_key(seed, STREAM_Q), np.arange(n, dtype=U64))


def gen_H_wigner(seed: int, n: int) -> np.ndarray:
    """Dense symmetric H with spectrum inside roughly [1, 5].

    ``H[i, j] = g(min, max) / sqrt(n)`` off the diagonal and
    ``H[i, i] = 3 + g(i, i) / sqrt(n)`` (Wigner semicircle of radius 2 shifted by 3).
    """
    i = np.arange(n, dtype=U64)[:, None]
    j = np.arange(n, dtype=U64)[None, :]
    lo = np.minimum(i, j)
    hi = np.maximum(i, j)
    g = gen_normal(stream_key(seed, STREAM_H), (lo * U64(n) + hi).ravel()).reshape(n, n)
    inv_sqrt_n = 1.0 / math.sqrt(n)
    H = g * inv_sqrt_n
    H[np.arange(n), np.arange(n)] = 3.0 + H[np.arange(n), np.arange(n)]
    return H


@dataclasses.dataclass
class DenseQP:
    H: np.ndarray  # (n, n) symmetric, may be all zero (LP)
    q: np.ndarray  # (n,)
    A: np.ndarray  # (m, n) row-major
    lvar: np.ndarray
    uvar: np.ndarray
    lcon: np.ndarray
    ucon: np.ndarray
    x0: np.ndarray
    c0: float = 0.0
    y0: np.ndarray | None = None
    name: str = "qp"

    def __post_init__(self):
        f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        self.H, self.q, self.A = f(self.H), f(self.q), f(self.A)
        self.lvar, self.uvar, self.lcon, self.ucon, self.x0 = map(
            f, (self.lvar, self.uvar, self.lcon, self.ucon, self.x0)
        )
        self.y0 = np.zeros(self.ncon) if self.y0 is None else f(self.y0)
        if self.A.size == 0:
            self.A = self.A.reshape(0, self.nvar)

    @property
    def nvar(self) -> int:
        return self.q.shape[0]

    @property
    def ncon(self) -> int:
        return self.lcon.shape[0]

    @property
    def is_lp(self) -> bool:
        return not np.any(self.H)

    def obj(self, x):  # scripts/qp_gpu.jl:29-34
        return self.c0 + self.q @ x + 0.5 * (x @ (self.H @ x))

    def grad(self, x):  # scripts/qp_gpu.jl:36-40
        return self.H @ x + self.q


def synthetic_qp(seed: int, n: int, m: int, family: str = "wigner") -> DenseQP:
    """Synthetic dense QP of BASELINE.md section 3: 0<=x<=1, 0<=Ax<=1, x0=0."""
    if family == "wigner":
        H = gen_H_wigner(seed, n)
    elif family == "lp":
        H = np.zeros((n, n))
    elif family == "dummy":  # R R' + 100 I of MadNLPTests.DenseDummyQP (test/runtests.jl:9), R = G'
        G = gen_normal(stream_key(seed, STREAM_H), np.arange(n * n, dtype=U64)).reshape(n, n)
        H = G.T @ G + 100.0 * np.eye(n)
    else:
        raise ValueError(family)
    return DenseQP(
        H=H,
        q=gen_q(seed, n),
        A=gen_A(seed, m, n),
        lvar=np.zeros(n),
        uvar=np.ones(n),
        lcon=np.zeros(m),
        ucon=np.ones(m),
        x0=np.zeros(n),
        name=f"synthetic-{family}-n{n}-m{m}-s{seed}",
    )


def sparse_qp(seed: int, n: int, m: int, per_row: int = 4, family: str = "wigner", equality_cons=()) -> DenseQP:
    """The synthetic family with a SPARSE Jacobian (held dense here): row i keeps the Gaussian entries of
    ``gen_A`` at ``per_row`` pseudo-random columns plus the band entry (i mod n), so that every row is
    non-empty and columns overlap between rows (the Gram matrix A' Theta A is genuinely dense-ish).
    ``equality_cons``: rows turned into equalities at their lower bound (Netlib-style LPs)."""
    qp = synthetic_qp(seed, n, m, family)
    mask = np.zeros((m, n), dtype=bool)
    key = stream_key(seed, 7)
    with np.errstate(over="ignore"):
        idx = np.arange(m * per_row, dtype=np.uint64)
        cols = (mix64(np.uint64(key) ^ idx) % np.uint64(n)).astype(np.int64).reshape(m, per_row)
    for i in range(m):
        mask[i, cols[i]] = True
        mask[i, i % n] = True
    qp.A = np.where(mask, qp.A, 0.0)
    for j in equality_cons:
        qp.ucon[j] = qp.lcon[j]
    qp.name = f"sparse-{family}-n{n}-m{m}-s{seed}"
    return qp


def random_qp(seed: int, n: int, m: int, lp: bool = False, pattern_seed=None) -> DenseQP:
    """Feasible dense QP / LP with a RANDOM pattern of bounds: variables free / lower / upper / boxed,
    rows equality / >= / <= / ranged (test generator: exercises ind_lb != ind_ub, free variables, one-sided
    rows -- the synthetic family has every bound finite).  ``pattern_seed`` fixes which bounds are finite
    independently of the data (batches need one pattern).  numpy Generator, CPU only."""
    rng = np.random.default_rng(seed)
    prng = np.random.default_rng(seed if pattern_seed is None else pattern_seed)
    R = rng.standard_normal((n, n))
    H = np.zeros((n, n)) if lp else R @ R.T / n + np.eye(n)
    A = rng.standard_normal((m, n))
    xf = rng.uniform(-1.0, 1.0, n)  # a strictly feasible point
    kind = prng.integers(0, 4, n)
    lvar = np.where((kind == 1) | (kind == 3), xf - rng.uniform(0.5, 2.0, n), -np.inf)
    uvar = np.where((kind == 2) | (kind == 3), xf + rng.uniform(0.5, 2.0, n), np.inf)
    if lp:  # keep the LP bounded
        lvar = np.where(np.isfinite(lvar), lvar, xf - 3.0)
        uvar = np.where(np.isfinite(uvar), uvar, xf + 3.0)
    ax = A @ xf
    ck = prng.integers(0, 4, m)
    lcon = np.where(ck == 0, ax, np.where((ck == 1) | (ck == 3), ax - rng.uniform(0.1, 1.0, m), -np.inf))
    ucon = np.where(ck == 0, ax, np.where((ck == 2) | (ck == 3), ax + rng.uniform(0.1, 1.0, m), np.inf))
    return DenseQP(H=H, q=rng.standard_normal(n), A=A, lvar=lvar, uvar=uvar, lcon=lcon, ucon=ucon,
                   x0=np.zeros(n), name=f"random-n{n}-m{m}-s{seed}")


def dummy_qp(n: int, m: int, seed: int = 1, equality_cons=(), fixed_variables=()) -> DenseQP:
    """Small QP in the spirit of ``MadNLPTests.DenseDummyQP`` (``test/runtests.jl:9``).

    The Julia fixture draws from Julia's RNG and cannot be reproduced here
    (SURVEY.md 8c); the structure is kept: ``P = G + G' + 100 I``, bidiagonal
    ``A`` (``A[j, j] = 1, A[j, j+1] = -1``), ``0 <= x <= 1``, ``0 <= Ax <= 1``,
    rows in ``equality_cons`` become equalities at their lower bound (x_j = x_{j+1}), variables in
    ``fixed_variables`` are fixed at their lower bound (the cases named at ``test/runtests.jl:63-75``).
    """
    G = gen_normal(stream_key(seed, STREAM_H), np.arange(n * n, dtype=U64)).reshape(n, n)
    H = G + G.T + 100.0 * np.eye(n)
    A = np.zeros((m, n))
    for j in range(m):
        A[j, j] = 1.0
        A[j, j + 1] = -1.0
    lcon, ucon = np.zeros(m), np.ones(m)
    for j in equality_cons:
        ucon[j] = lcon[j]
    uvar = np.ones(n)
    uvar[list(fixed_variables)] = 0.0
    return DenseQP(H, gen_q(seed, n), A, np.zeros(n), uvar, lcon, ucon, np.zeros(n),
                   name=f"dummy-n{n}-m{m}")


def simple_lp() -> DenseQP:
    """``simple_lp()`` of ``test/runtests.jl:24-55``: min x1+x2, x1+x2=1, x>=0, x0=(1,1)."""
    return DenseQP(
        H=np.zeros((2, 2)), q=np.ones(2), A=np.array([[1.0, 1.0]]),
        lvar=np.zeros(2), uvar=np.full(2, np.inf), lcon=np.array([1.0]), ucon=np.array([1.0]),
        x0=np.ones(2), name="simpleLP",
    )


def hs21() -> DenseQP:
    """Hock-Schittkowski 21 / Maros-Meszaros HS21 (BASELINE.json configs[0]).

    min 0.01 x1^2 + x2^2 - 100, 10 x1 - x2 >= 10, 2<=x1<=50, -50<=x2<=50;
    x* = (2, 0), f* = -99.96.  Data hand-encoded from the published problem
    statement (no .SIF on disk); start (-1, -1).
    """
    return DenseQP(
        H=np.diag([0.02, 2.0]), q=np.zeros(2), A=np.array([[10.0, -1.0]]),
        lvar=np.array([2.0, -50.0]), uvar=np.array([50.0, 50.0]),
        lcon=np.array([10.0]), ucon=np.array([np.inf]), x0=np.array([-1.0, -1.0]),
        c0=-100.0, name="HS21",
    )
