"""GPU parity: dense kernels (generator, GEMV, MFMA SYRK assembly, Cholesky, solves) through the C ABI
against numpy / scipy (LAPACK) on the same inputs."""
import math

import numpy as np
import pytest
import scipy.linalg as sla
import torch

from oracle import qp as Q

pytestmark = pytest.mark.gpu


def dev(a, be):
    return torch.as_tensor(np.ascontiguousarray(a), device=be.device)


def test_generator_bit_exact(hip):
    import madqp_jl_amd as M

    seed, n, m = 20250614, 67, 29
    A = torch.empty((m, n), dtype=torch.float64, device=hip.device)
    hip.gen_normal(M.stream_key(seed, 1), 0, A)
    assert M.stream_key(seed, 1) == Q.stream_key(seed, 1)
    np.testing.assert_array_equal(A.cpu().numpy(), Q.gen_A(seed, m, n))
    q = torch.empty(n, dtype=torch.float64, device=hip.device)
    hip.gen_normal(M.stream_key(seed, 3), 0, q)
    np.testing.assert_array_equal(q.cpu().numpy(), Q.gen_q(seed, n))
    H = torch.empty((n, n), dtype=torch.float64, device=hip.device)
    hip.gen_wigner(M.stream_key(seed, 2), n, 1.0 / math.sqrt(n), H)
    np.testing.assert_array_equal(H.cpu().numpy(), Q.gen_H_wigner(seed, n))
    # offset addressing: any tile can be generated independently
    part = torch.empty(50, dtype=torch.float64, device=hip.device)
    hip.gen_normal(M.stream_key(seed, 1), 1000, part)
    np.testing.assert_array_equal(part.cpu().numpy(), Q.gen_A(seed, m, n).ravel()[1000:1050])


@pytest.mark.parametrize("rows,cols", [(1, 1), (5, 3), (64, 130), (257, 1000), (3000, 77), (128, 5000),
                                       (2500, 2049), (71, 2061), (2000, 5000)])  # (the last four: one-pass strips for A' x)
@pytest.mark.parametrize("trans", [0, 1])
def test_gemv(hip, rows, cols, trans):
    rng = np.random.default_rng(rows * 1000 + cols + trans)
    lda = cols + (cols % 2)  # even and odd-free paths are both hit across the cases
    if rows == 257:
        lda = cols + 3  # odd leading dimension -> scalar path
    A = rng.standard_normal((rows, lda))
    x = rng.standard_normal(rows if trans else cols)
    y = rng.standard_normal(cols if trans else rows)
    for alpha, beta in [(1.0, 0.0), (-1.0, 1.0), (0.5, -2.0)]:
        yd = dev(y, hip)
        hip.gemv(trans, rows, cols, alpha, dev(A, hip), lda, dev(x, hip), beta, yd)
        Am = A[:, :cols]
        ref = alpha * ((Am.T @ x) if trans else (Am @ x)) + beta * y
        scale = np.abs(Am).T @ np.abs(x) if trans else np.abs(Am) @ np.abs(x)
        err = np.max(np.abs(yd.cpu().numpy() - ref) / (scale + np.abs(y) + 1e-300))
        assert err < 1e-14, (rows, cols, trans, alpha, beta, err)


def test_gemv_beta_zero_ignores_nan(hip):
    A = dev(np.ones((4, 4)), hip)
    y = dev(np.full(4, np.nan), hip)
    hip.gemv(0, 4, 4, 1.0, A, 4, dev(np.ones(4), hip), 0.0, y)
    np.testing.assert_array_equal(y.cpu().numpy(), np.full(4, 4.0))


def _assemble_ref(B, w, base, dvec):
    K = (B.T * w) @ B
    if base is not None:
        K = K + base
    if dvec is not None:
        K = K + np.diag(dvec)
    return K


@pytest.mark.parametrize("n,k", [(1, 1), (16, 4), (37, 13), (128, 16), (130, 33), (256, 64), (300, 1),
                                 (385, 200), (1000, 333)])
def test_syrk_assemble(hip, n, k):
    """K = H + diag + B' W B, lower triangle only; asymmetric data catches a swapped C/D lane map."""
    rng = np.random.default_rng(n * 7 + k)
    B = rng.standard_normal((k, n))
    w = rng.uniform(0.5, 2.0, k)
    G = rng.standard_normal((n, n))
    base = G + G.T
    dvec = rng.uniform(1.0, 2.0, n)
    ldc = n + 5
    C = torch.full((n, ldc), np.nan, dtype=torch.float64, device=hip.device)  # column j = row j of the tensor
    hip.syrk_assemble(n, k, dev(B, hip), n, dev(w, hip), dev(base, hip), n, dev(dvec, hip), C, ldc)
    out = C.cpu().numpy()[:, :n].T  # out[i, j] = C[i + j*ldc]
    ref = _assemble_ref(B, w, base, dvec)
    low = np.tril_indices(n)
    scale = (np.abs(B).T * w) @ np.abs(B) + np.abs(base) + np.diag(dvec)
    err = np.max(np.abs(out[low] - ref[low]) / scale[low])
    assert err < 1e-14, err
    up = np.triu_indices(n, 1)
    assert np.all(np.isnan(out[up])), "strict upper triangle must not be written"
    # optional arguments
    C2 = torch.zeros((n, n), dtype=torch.float64, device=hip.device)
    hip.syrk_assemble(n, k, dev(B, hip), n, None, None, n, None, C2, n)
    ref2 = B.T @ B
    out2 = C2.cpu().numpy().T
    assert np.max(np.abs(out2[low] - ref2[low]) / (np.abs(B).T @ np.abs(B))[low]) < 1e-14


def _spd(n, rng, cond=1e3):
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0, math.log10(cond), n)
    return (Qm * ev) @ Qm.T


@pytest.mark.parametrize("n,padded", [(n, False) for n in [1, 2, 17, 128, 129, 255, 256, 400, 1024, 1100, 2300]]
                         + [(n, True) for n in [129, 255, 300, 1024, 1100, 2300, 4100]])
def test_cholesky_factor_and_solve(hip, n, padded):
    """padded: the leading dimension covers the order rounded up to 128, as the KKT objects allocate K -- the
    layout that takes the right-looking mid-size schedule (chol.hip, chol_mid_step_kernel) up to n = 13 312."""
    rng = np.random.default_rng(n)
    K = _spd(n, rng)
    K = 0.5 * (K + K.T)
    lda = (n + 127) // 128 * 128 + 2 if padded else n + (3 if n % 2 else 2)
    host = np.full((n, lda), np.nan)  # column-major lower: tensor row j = column j of K
    host[:, :n] = np.where(np.triu(np.ones((n, n))) > 0, K.T, np.nan)
    Kd = dev(host, hip)
    h = hip.chol_create(n)
    try:
        info = hip.chol_factor(h, Kd, lda)
        assert info == 0
        L = np.tril(Kd.cpu().numpy()[:, :n].T)
        Lref = sla.cholesky(K, lower=True)
        rel = np.linalg.norm(L @ L.T - K) / np.linalg.norm(K)
        assert rel < 1e-14 * max(1, math.sqrt(n)), rel
        assert np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)) < 1e-11
        up = np.triu_indices(n, 1)
        assert np.all(np.isnan(Kd.cpu().numpy()[:, :n].T[up])), "upper triangle must stay untouched"
        b = rng.standard_normal(n)
        bd = dev(b, hip)
        hip.chol_solve(h, bd)
        x = bd.cpu().numpy()
        xref = sla.cho_solve((Lref, True), b)
        assert np.linalg.norm(K @ x - b) / (np.linalg.norm(K) * np.linalg.norm(x)) < 1e-14
        assert np.linalg.norm(x - xref) / np.linalg.norm(xref) < 1e-10
    finally:
        hip.chol_destroy(h)


@pytest.mark.parametrize("n,chunk,padded", [(1100, 2, False), (1100, 3, False), (2300, 5, False), (700, 1, False),
                                            (1100, 2, True), (2300, 5, True), (700, 1, True), (1153, 3, True)])
def test_sweeps_with_split_block_rows(hip, n, chunk, padded):
    """Block rows longer than MADQP_SWEEP_CHUNK tiles are streamed by several workgroups (chol.hip, SweepPlan; 64 tiles
    by default, i.e. only beyond n = 8 320).  A small chunk puts the same code under a case the CPU can check: the
    solution must agree with the one-job-per-row sweep to rounding (the partial sums are added in another order).
    padded: the layout of the mid-size factorisation, whose backward sweep is the forward kernel on U = L' with the
    blocks -- and the jobs of a split block row -- taken from the far end (trsv_fwd_sweep_kernel<1, true>)."""
    import os
    rng = np.random.default_rng(n + chunk)
    K = _spd(n, rng)
    K = 0.5 * (K + K.T)
    lda = (n + 127) // 128 * 128 + 2 if padded else n + 2
    host = np.zeros((n, lda))
    host[:, :n] = K.T
    b = rng.standard_normal(n)
    xs = []
    for env in (str(chunk), None):
        if env is None:
            os.environ.pop("MADQP_SWEEP_CHUNK", None)
        else:
            os.environ["MADQP_SWEEP_CHUNK"] = env
        try:
            Kd = dev(host, hip)
            h = hip.chol_create(n)  # the job list is built here
        finally:
            os.environ.pop("MADQP_SWEEP_CHUNK", None)
        try:
            assert hip.chol_factor(h, Kd, lda) == 0
            bd = dev(b, hip)
            hip.chol_solve(h, bd)
            xs.append(bd.cpu().numpy())
        finally:
            hip.chol_destroy(h)
    xref = sla.cho_solve((sla.cholesky(K, lower=True), True), b)
    for x in xs:
        assert np.linalg.norm(K @ x - b) / (np.linalg.norm(K) * np.linalg.norm(x)) < 1e-14
        assert np.linalg.norm(x - xref) / np.linalg.norm(xref) < 1e-10
    assert np.linalg.norm(xs[0] - xs[1]) / np.linalg.norm(xs[1]) < 1e-12


@pytest.mark.parametrize("padded", [False, True])
def test_cholesky_not_positive_definite_reports_column(hip, padded):
    """LAPACK info convention: first failing leading minor, never aborts (SURVEY 8b errors); both schedules of
    madqp_chol_factor (padded leading dimension: the right-looking mid-size one)."""
    n = 300
    rng = np.random.default_rng(5)
    K = _spd(n, rng)
    K[200, 200] = -1.0
    lda = 384 if padded else n
    host = np.zeros((n, lda))
    host[:, :n] = np.tril(K).T
    Kd = dev(host, hip)
    h = hip.chol_create(n)
    try:
        info = hip.chol_factor(h, Kd, lda)
        ref_info = sla.lapack.dpotrf(K, lower=1)[1]
        assert info == ref_info == 201
    finally:
        hip.chol_destroy(h)


@pytest.mark.parametrize("padded", [False, True])
def test_cholesky_graded_ipm_like(hip, padded):
    """Diagonal scaling over 16 orders of magnitude (late-IPM Sigma): backward error stays at eps."""
    n = 700
    rng = np.random.default_rng(11)
    A = rng.standard_normal((300, n))
    sig = 10.0 ** rng.uniform(-8, 8, n)
    th = 10.0 ** rng.uniform(-6, 6, 300)
    K = (A.T * th) @ A + np.diag(sig)
    K = 0.5 * (K + K.T)
    lda = 768 if padded else n
    host = np.zeros((n, lda))
    host[:, :n] = np.tril(K).T
    Kd = dev(host, hip)
    h = hip.chol_create(n)
    try:
        assert hip.chol_factor(h, Kd, lda) == 0
        L = np.tril(Kd.cpu().numpy()[:, :n].T)
        d = np.sqrt(np.diag(K))
        E = (L @ L.T - K) / np.outer(d, d)  # scaled backward error
        assert np.max(np.abs(E)) < 1e-13, np.max(np.abs(E))
        b = rng.standard_normal(n)
        bd = dev(b, hip)
        hip.chol_solve(h, bd)
        x = bd.cpu().numpy()
        xref = sla.cho_solve(sla.cho_factor(K, lower=True), b)
        res = np.linalg.norm((K @ x - b) / d) / np.linalg.norm(b / d)
        res_ref = np.linalg.norm((K @ xref - b) / d) / np.linalg.norm(b / d)
        assert res < 50 * max(res_ref, 1e-15), (res, res_ref)
    finally:
        hip.chol_destroy(h)


def test_pointers_aligned_to_8_bytes_only(hip):
    """Every kernel has a 16-byte fast path (double2 / LDS-DMA loads); operands that start 8 bytes off must
    take the guarded path and give the same results."""
    rng = np.random.default_rng(3)
    n, k = 300, 140

    def off(a):  # device copy whose first element sits at an odd multiple of 8 bytes
        flat = torch.empty(a.size + 1, dtype=torch.float64, device=hip.device)
        flat[1:] = torch.as_tensor(np.ascontiguousarray(a).ravel(), device=hip.device)
        v = flat[1:].view(*a.shape)
        assert v.data_ptr() % 16 == 8
        return v

    B, w = rng.standard_normal((k, n)), rng.uniform(0.5, 2.0, k)
    G = rng.standard_normal((n, n))
    base, dvec = G + G.T + 2 * n * np.eye(n), rng.uniform(1.0, 2.0, n)
    C = off(np.zeros((n, n)))
    hip.syrk_assemble(n, k, off(B), n, off(w), off(base), n, off(dvec), C, n)
    K = _assemble_ref(B, w, base, dvec)
    low = np.tril_indices(n)
    assert np.max(np.abs(C.cpu().numpy().T[low] - K[low])) <= 1e-12 * np.max(np.abs(K))
    x, y = rng.standard_normal(n), rng.standard_normal(k)
    for trans, vec_in, vec_out in ((0, x, np.zeros(k)), (1, y, np.zeros(n))):
        out = off(vec_out)
        hip.gemv(trans, k, n, 1.0, off(B), n, off(vec_in), 0.0, out)
        ref = B.T @ vec_in if trans else B @ vec_in
        assert np.max(np.abs(out.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
    h = hip.chol_create(n)
    try:
        assert hip.chol_factor(h, C, n) == 0  # C holds the lower triangle of the SPD matrix K
        b = rng.standard_normal(n)
        bd = off(b)
        hip.chol_solve(h, bd)
        xs = bd.cpu().numpy()
        assert np.linalg.norm(K @ xs - b) / (np.linalg.norm(K) * np.linalg.norm(xs)) < 1e-14
    finally:
        hip.chol_destroy(h)


def test_sweep_timeout_is_reported_as_a_device_fault(hip):
    """A triangular sweep whose producer block never publishes sets the context's fault word (chol.hip,
    sweep_poll_block); the next scalar read-back must return MADQP_ERR_HIP with its own message instead of letting
    the NaNs pass for MadNLP.SolveException (src/linear_solver.jl:41-43).  The word is injected here."""
    import madqp_jl_amd as M

    v = torch.full((9,), -3.0, dtype=torch.float64, device=hip.device)
    assert hip.norm_inf(v) == 3.0
    hip._ck(hip.lib.madqp_debug_inject_fault(hip.ctx))
    with pytest.raises(M.MadQPError, match="hand-off timed out"):
        hip.norm_inf(v)
    assert hip.norm_inf(v) == 3.0  # reported once, then cleared


@pytest.mark.parametrize("n,ld", [(12288, 12288), (12500, 12504), (13001, 13002), (300, 300)])
def test_symmetric_matvec_from_the_lower_triangle(hip, n, ld):
    """H x for a symmetric H from its lower triangle only (gemv.hip: madqp_symv_lower, behind mul! and the model
    evaluation of a dense QP): the UPPER triangle is poisoned with NaN -- it must never be read -- and the product
    equals the full one to rounding; orders that are not multiples of the 256 x 512 tiles, a leading dimension larger
    than the order, alpha / beta, and (n = 300) the general kernel that small problems still take."""
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n))
    Hs = G + G.T
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(n)
    Hd = torch.zeros((n, ld), dtype=torch.float64, device=hip.device)
    Hd[:, :n] = torch.as_tensor(Hs, device=hip.device)
    if n >= 12288:  # (the library's threshold for the triangular kernel, gemv.hip)
        iu = torch.triu_indices(n, n, offset=1, device=hip.device)
        Hd[iu[0], iu[1]] = float("nan")
    st = hip.new_state(n, 0, np.arange(0), np.arange(0))
    z = lambda k: torch.zeros(max(k, 1), dtype=torch.float64, device=hip.device)[:k]
    h = hip.kkt_create(n, 0, [], Hd, ld, z(0), n)  # no constraints: the KKT object is H (+ diagonal) alone
    st.x.copy_(torch.as_tensor(x))
    v = torch.as_tensor(x.copy(), device=hip.device)
    w = torch.as_tensor(y0.copy(), device=hip.device)
    hip.fill(0.0, st.reg)
    hip.kkt_mul(h, st, w, v, -0.5, 2.0)  # w = 2 w - 0.5 (H + reg) v
    ref = 2.0 * y0 - 0.5 * (Hs @ x)
    assert np.max(np.abs(w.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
    obj = hip.kkt_eval(h, st, z(n), z(0), 0.0)
    assert abs(obj - 0.5 * x @ Hs @ x) <= 1e-12 * abs(x @ Hs @ x)
    assert np.max(np.abs(st.f.cpu().numpy() - Hs @ x)) <= 1e-12 * np.max(np.abs(Hs @ x))
    hip.kkt_destroy(h)


def test_sweeps_leave_the_residual_of_substitution():
    """Round 5: the diagonal step of both sweeps is UNIT BLOCK SUBSTITUTION over the 16 x 16 sub-blocks with images
    normalised once per factorisation (chol.hip: sweep_image_kernel, sweep_diag_fwd / _bwd) -- the default.  On a matrix
    whose diagonal blocks are ill conditioned the product with the stored 128 x 128 inverse (rounds 1-4, kept as
    MADQP_SWEEP_DIAG=inv) leaves a residual of cond(L_rr) eps; the default, and plain block substitution
    (MADQP_SWEEP_DIAG=sub16, the A/B form), must leave the residual of a substitution -- backward stable.  The panel
    solve is block substitution either way (MADQP_CHOL_PANEL=inv: the inverse products of rounds 1-3, for comparison;
    its images are full inverses, so its sweeps multiply too)."""
    import json
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import json, sys
        import numpy as np, torch
        sys.path.insert(0, %r)
        import madqp_jl_amd as M
        be = M.HipBackend(0)
        out = {}
        for n in (300, 1000):
            rng = np.random.default_rng(n)
            Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
            K = (Qm * np.logspace(0, 11, n)) @ Qm.T        # cond 1e11, every diagonal block ill conditioned
            K = (K + K.T) / 2
            x = rng.standard_normal(n)
            b = K @ x
            ld = (n + 127) // 128 * 128
            Kd = torch.zeros((ld, ld), dtype=torch.float64, device=be.device)
            Kd[:n, :n] = torch.as_tensor(K, device=be.device)
            h = be.chol_create(n)
            assert be.chol_factor(h, Kd, ld) == 0
            bd = torch.as_tensor(b.copy(), device=be.device)
            be.chol_solve(h, bd)
            xs = bd.cpu().numpy()
            be.chol_destroy(h)
            out[str(n)] = float(np.linalg.norm(K @ xs - b) / (np.linalg.norm(K, 2) * np.linalg.norm(xs)))
        be.close()
        print(json.dumps(out))
    ''') % root
    res = {}
    for name, env in (("default", {}), ("sub16", {"MADQP_SWEEP_DIAG": "sub16"}), ("inv_sweeps", {"MADQP_SWEEP_DIAG": "inv"}),
                      ("inv_panel", {"MADQP_CHOL_PANEL": "inv"})):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    print("scaled residuals:", res)
    for n in ("300", "1000"):
        assert res["inv_sweeps"][n] < 1e-11 and res["inv_panel"][n] < 1e-11  # (all of them solve the system)
        for name in ("default", "sub16"):
            assert res[name][n] <= 4e-16 * int(n), (name, n, res[name][n])  # backward stable: the residual of substitution
            assert res[name][n] <= res["inv_sweeps"][n] * 1.5, (name, n)


def test_mid_size_schedules_factor_the_same_matrix():
    """chol_factor_enqueue's mid-size path (n <= 13 312) has three schedules of the trailing update: planned visits
    (default, mid_plan.inc), two panels every second step (MADQP_CHOL_MID_LAZY=0) and every panel at once
    (+ MADQP_CHOL_MID_TWO=0; + MADQP_CHOL_MID_DSYRK=0: the diagonal tile by the GEMM path).  All must give LAPACK's
    factor up to rounding -- orders with a ragged last block and a single trailing block -- and, since every tile
    receives its panels in ascending order whatever the schedule, the SAME bits: the factor does not depend on the plan
    (hence not on the number of CUs the plan was made for)."""
    import json
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import hashlib, json, sys
        import numpy as np, torch
        sys.path.insert(0, %r)
        import madqp_jl_amd as M
        be = M.HipBackend(0)
        out = {}
        for n in (129, 257, 1000, 2500, 5001, 12900):
            rng = np.random.default_rng(n)
            G = torch.as_tensor(rng.standard_normal((n, n + 8)), device=be.device)
            K = G @ G.T + n * torch.eye(n, dtype=torch.float64, device=be.device)
            ld = (n + 127) // 128 * 128
            Kd = torch.zeros((ld, ld), dtype=torch.float64, device=be.device)
            Kd[:n, :n] = K
            h = be.chol_create(n)
            assert be.chol_factor(h, Kd, ld) == 0
            Lc = torch.tril(Kd.T[:n, :n])                     # the library's layout is column-major: read it transposed
            ref = torch.linalg.cholesky(K)
            err = float((Lc - ref).abs().max() / ref.abs().max())
            x = torch.as_tensor(rng.standard_normal(n), device=be.device)
            b = K @ x
            be.chol_solve(h, b)
            out[str(n)] = [err, float((b - x).abs().max()), hashlib.sha256(Lc.cpu().numpy().tobytes()).hexdigest()]
            be.chol_destroy(h)
        be.close()
        print(json.dumps(out))
    ''') % root
    variants = (("planned", {}), ("two_panels", {"MADQP_CHOL_MID_LAZY": "0"}),
                ("every_panel", {"MADQP_CHOL_MID_LAZY": "0", "MADQP_CHOL_MID_TWO": "0"}),
                ("gemm_diagonal", {"MADQP_CHOL_MID_LAZY": "0", "MADQP_CHOL_MID_TWO": "0", "MADQP_CHOL_MID_DSYRK": "0"}))
    first = None
    for name, env in variants:
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (name, p.stderr[-2000:])
        res = json.loads(p.stdout.strip().splitlines()[-1])
        print(name, {n: v[:2] for n, v in res.items()})
        for n, (err, dx, digest) in res.items():
            assert err < 1e-12 and dx < 1e-9, (name, n, err, dx)
        first = first or res
        if name != "gemm_diagonal":  # (the GEMM path sums the diagonal tile's products in its own order)
            assert {n: v[2] for n, v in res.items()} == {n: v[2] for n, v in first.items()}, name
