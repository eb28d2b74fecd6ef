"""Chebyshev iteration and MINRES (src/chebyshev.jl, src/minres.jl): the
reference's property tests against the oracle (CPU) and bit-level parity of the device compositions (GPU)."""
import numpy as np
import pytest


def rand_spd(rng, n, dtype):
    B = rng.random((n, n)) + n * np.eye(n)                       # test/chebyshev.jl:8-11
    return (B.T @ B).astype(dtype)


def eig_bounds(Ad):
    lam = np.linalg.eigvalsh(Ad.astype(np.float64))              # test/chebyshev.jl:13-18
    d = (lam[-1] - lam[0]) / 100
    return lam[0] - d, lam[-1] + d


# ---- oracle ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_chebyshev(orc, dtype):
    rng = np.random.default_rng(1234321)
    n = 10
    Ad, b = rand_spd(rng, n, dtype), rng.random(n).astype(dtype)
    lo, hi = eig_bounds(Ad)
    reltol = float(np.sqrt(np.finfo(dtype).eps))
    A = orc.CSC.from_dense(Ad)
    x, h = orc.chebyshev(A, b, lo, hi, reltol=reltol, maxiter=10 * n)
    assert h["isconverged"] and np.linalg.norm(Ad @ x - b) / np.linalg.norm(b) <= reltol                 # test/chebyshev.jl:37-38
    x0 = rng.random(n).astype(dtype)
    r0 = np.linalg.norm(Ad @ x0 - b)
    x, h = orc.chebyshev(A, b, lo, hi, x0, reltol=reltol, maxiter=10 * n)
    assert h["isconverged"] and np.linalg.norm(Ad @ x - b) <= reltol * r0 * 1.01                         # :49
    x, h = orc.chebyshev(A, b, lo, hi, x0, abstol=reltol, reltol=0.0, maxiter=10 * n)
    assert h["isconverged"] and np.linalg.norm(Ad @ x - b) <= 2 * reltol                                  # :55
    x, h = orc.chebyshev(A, b, lo, hi, reltol=reltol, maxiter=10 * n, pl_diag=np.ones(n, dtype))
    assert h["isconverged"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_minres(orc, dtype):
    rng = np.random.default_rng(123)
    n = 15
    B = rng.random((n, n)) + n * np.eye(n)
    reltol = float(np.sqrt(np.finfo(dtype).eps))
    As, bs = (B + B.T).astype(dtype), (B @ np.ones(n)).astype(dtype)                                       # test/minres.jl:12-18
    x1, h1 = orc.minres(orc.CSC.from_dense(As), bs, maxiter=10 * n, reltol=reltol)
    assert h1["isconverged"] and np.linalg.norm(bs - As @ x1) / np.linalg.norm(bs) <= reltol               # :43-45
    x2, h2 = orc.minres(orc.CSC.from_dense(As), bs, rng.random(n).astype(dtype), maxiter=10 * n, reltol=reltol)
    assert np.linalg.norm(bs - As @ x2) / np.linalg.norm(bs) <= reltol                                     # :46
    Ak = (B - B.T).astype(dtype)
    bk = (Ak @ np.ones(n, dtype)).astype(dtype)
    xk, hk = orc.minres(orc.CSC.from_dense(Ak), bk, skew_hermitian=True, maxiter=10 * n, reltol=reltol)
    assert hk["isconverged"] and np.linalg.norm(bk - Ak @ xk) / np.linalg.norm(bk) <= reltol * 4           # :55-56
    T3 = np.array([[2, -1, 0], [-1, 2, -1], [0, -1, 2]], dtype)                                            # :76-99
    b3 = np.ones(3, dtype)
    x0 = np.linalg.solve(T3.astype(np.float64), b3.astype(np.float64)).astype(dtype)
    pert = (10 * np.sqrt(np.finfo(dtype).eps) * np.array([-1.0, 1.0, -1.0])).astype(dtype)
    x, ch = orc.minres(orc.CSC.from_dense(T3), b3, x0 + pert)
    assert 2 <= ch["iters"] <= 3
    r0 = float(np.linalg.norm(T3 @ (x0 + pert) - b3))
    x, ch = orc.minres(orc.CSC.from_dense(T3), b3, x0 + pert, abstol=2 * r0, reltol=0.0)
    assert ch["iters"] == 0


def test_givens_host_entry_matches_oracle(pkg, orc):
    for f, g in [(1.0, 0.0), (0.0, 2.0), (3.0, 4.0), (-3.0, 4.0), (1e-200, 1e-200), (-5.0, 1.0)]:
        assert pkg.givens_algorithm(f, g) == orc.givens(f, g)
    assert pkg.givens_algorithm(3.0, -4.0, np.float32) == orc.givens(3.0, -4.0, np.float32)


# ---- device ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("precond", [False, True])
def test_chebyshev_device_bit_exact(pkg, orc, ctx, dtype, precond):
    import scipy.sparse as sp
    # shifted Laplacian: spectrum in (20.2, 31.8).  The v0.9.4 iteration as written (u = c + beta*c) is only
    # stable for well-conditioned operators like the reference's own randSPD test matrices.
    L0 = orc.laplace(10, 3)
    A = orc.CSC.from_scipy((L0.to_scipy() + 20 * sp.eye(L0.n)).tocsc()).astype(dtype)
    b = orc.hashed_rhs(A.n).astype(dtype)
    lo, hi = 20.0, 32.0
    d = (6 + 0.1 * np.arange(A.n) / A.n).astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    okw = {"pl_diag": d / 6} if precond else {}
    dkw = {"Pl": pkg.JacobiPrec(pkg.HipVector.from_numpy(d / 6))} if precond else {}
    x0 = np.random.default_rng(0).standard_normal(A.n).astype(dtype)
    for start in (None, x0):
        if start is None:
            x, ch = pkg.chebyshev(dA, pkg.HipVector.from_numpy(b), lo, hi, log=True, maxiter=400, **dkw)
        else:
            x, ch = pkg.chebyshev_(pkg.HipVector.from_numpy(start), dA, pkg.HipVector.from_numpy(b), lo, hi, log=True, maxiter=400, **dkw)
        xo, ho = orc.chebyshev(A, b, lo, hi, start, maxiter=400, mode="tree", shape=ctx.reduce_shape(dtype), **okw)
        assert ch.iters == ho["iters"] and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"]
        assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    assert ch.isconverged


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_minres_device_bit_exact(pkg, orc, ctx, dtype):
    A = orc.laplace(10, 3).astype(dtype)
    b = orc.hashed_rhs(A.n).astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    x0 = np.random.default_rng(0).standard_normal(A.n).astype(dtype)
    # proj = dot(v_curr, v_next) is formed inside the SpMV launch where the operator's kernel takes the Lanczos step as its epilogue
    # (round 4): the oracle is told which tree that dot has (mik_minres_proj_shape)
    ps = pkg.minres_iterable_(pkg.HipVector.from_numpy(np.zeros(A.n, dtype)), dA, pkg.HipVector.from_numpy(b), initially_zero=True, maxiter=1).proj_shape()
    assert ps == ctx.spmv_dot_shape()                    # this operator (even n, 7-point pattern) runs on k_spmv_sdiab2
    for start in (None, x0):
        if start is None:
            x, ch = pkg.minres(dA, pkg.HipVector.from_numpy(b), log=True)
        else:
            x, ch = pkg.minres_(pkg.HipVector.from_numpy(start), dA, pkg.HipVector.from_numpy(b), log=True)
        xo, ho = orc.minres(A, b, start, mode="tree", shape=ctx.reduce_shape(dtype), proj_shape=ps)
        assert ch.iters == ho["iters"] and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"]
        assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    S = A.to_scipy()
    assert np.linalg.norm(S @ x.to_numpy() - b) / np.linalg.norm(b) <= (1e-6 if dtype == np.float64 else 2e-2)   # recurrence residual drifts


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["csr", "fe"])
def test_minres_lanczos_epilogue_in_the_csr_and_jagged_kernels(pkg, orc, ctx, dtype, kind):
    """the Lanczos step as the SpMV's epilogue (y = A x - H[2] v_prev stored once, proj formed in the launch) also in k_spmv_rowgather
    (the operator on its plain CSR arrays) and k_spmv_jds (a finite-element operator): bit-exact against the oracle with the
    projection in the SpMV-dot tree"""
    import scipy.sparse as sp
    if kind == "csr":
        A = orc.laplace(10, 3).astype(dtype)
        dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
        dA.set_layout("csr")
        want_kernel = "k_spmv_rowgather"
    else:
        n, rowptr, colidx, val = pkg.fixtures.fe_matrix((12, 12), 3, dtype)
        M = sp.csr_matrix((val, colidx, rowptr), shape=(n, n)).tocsc()
        M.sort_indices()
        A = orc.CSC(n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(dtype), 0)
        dA = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
        want_kernel = "k_spmv_jds"
    assert dA.spmv_kernel() == want_kernel
    b = orc.hashed_rhs(A.n).astype(dtype)
    x = pkg.HipVector.from_numpy(np.zeros(A.n, dtype))
    it = pkg.minres_iterable_(x, dA, pkg.HipVector.from_numpy(b), initially_zero=True, maxiter=30, reltol=0.0)
    assert it.proj_shape() == ctx.spmv_dot_shape()
    hist = np.array(list(it))
    xo, ho = orc.minres(A, b, maxiter=30, reltol=0.0, mode="tree", shape=ctx.reduce_shape(dtype), proj_shape=it.proj_shape())
    assert hist.size == 30 and np.array_equal(hist, ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)


@pytest.mark.gpu
def test_minres_survives_a_knob_change_between_steps(pkg, orc, ctx):
    """the Lanczos epilogue is decided per STEP (like BiCGStab's): switching the epilogues off after the iterable was created
    (development knob MIK_KNOB_SOLVER_FORM = 2; a layout change does the same) makes the next steps form the projection in a sweep of their own instead of
    failing with MIK_ERR_NOTIMPL, mik_minres_proj_shape follows, and the solve goes on -- same iteration to rounding (the projection's
    tree changed shape in mid-solve, so only the first steps are bit-equal to the oracle run with the epilogue shape)."""
    A = orc.laplace(10, 3)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    b = orc.hashed_rhs(A.n)
    x = pkg.HipVector.from_numpy(np.zeros(A.n))
    it = pkg.minres_iterable_(x, dA, pkg.HipVector.from_numpy(b), initially_zero=True, maxiter=40, reltol=0.0)
    assert it.proj_shape() == ctx.spmv_dot_shape()
    _, ho = orc.minres(A, b, maxiter=40, reltol=0.0, mode="tree", shape=ctx.reduce_shape(np.float64), proj_shape=it.proj_shape())
    hist, k = [], it.start()
    L = pkg.lib()
    try:
        for j in range(40):
            if j == 5:
                L.mik_set_tuning(8, 2)                    # MIK_KNOB_SOLVER_FORM
                assert it.proj_shape() == ctx.reduce_shape(np.float64)
            res, k = it.iterate(k)
            hist.append(res)
    finally:
        L.mik_set_tuning(8, 0)
    hist = np.array(hist)
    assert np.array_equal(hist[:5], ho["resnorm"][:5])
    np.testing.assert_allclose(hist, ho["resnorm"], rtol=1e-9)


@pytest.mark.gpu
def test_minres_skew_symmetric_device(pkg, orc, ctx):
    rng = np.random.default_rng(123)
    n = 15
    B = rng.random((n, n)) + n * np.eye(n)
    Ak = B - B.T
    bk = Ak @ np.ones(n)
    A = orc.CSC.from_dense(Ak)
    dA = pkg.HipCSR(n, n, A.colptr, A.rowval, A.nzval)
    ps = pkg.minres_iterable_(pkg.HipVector.from_numpy(np.zeros(n)), dA, pkg.HipVector.from_numpy(bk), initially_zero=True, maxiter=1).proj_shape()
    x, ch = pkg.minres(dA, pkg.HipVector.from_numpy(bk), skew_hermitian=True, maxiter=10 * n, log=True)
    xo, ho = orc.minres(A, bk, skew_hermitian=True, maxiter=10 * n, mode="tree", shape=ctx.reduce_shape(np.float64), proj_shape=ps)
    assert ch.isconverged and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    assert np.linalg.norm(bk - Ak @ x.to_numpy()) / np.linalg.norm(bk) <= 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fused_sweeps_equal_the_statement_by_statement_path(pkg, orc, ctx, dtype):
    """mik_axpy_dot / mik_minres_update / mik_cheb_direction / mik_axpy2_nrm2 vs one L1 call per reference statement"""
    import scipy.sparse as sp
    A = orc.laplace(9, 3).astype(dtype)
    b = orc.hashed_rhs(A.n).astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    for Am in (A, orc.laplace(8, 3).astype(dtype)):     # 9^3 rows: odd, no Lanczos epilogue; 8^3: even, the SpMV forms proj
        bm = orc.hashed_rhs(Am.n).astype(dtype)
        dAm = pkg.HipCSR(Am.n, Am.n, Am.colptr, Am.rowval, Am.nzval)
        shapes = []
        for fused in (True, False):
            x = pkg.HipVector.from_numpy(np.zeros(Am.n, dtype))
            it = pkg.minres_iterable_(x, dAm, pkg.HipVector.from_numpy(bm), initially_zero=True, maxiter=40, reltol=0.0, fused=fused)
            hist = np.array(list(it))
            shapes.append(it.proj_shape())
            xo, ho = orc.minres(Am, bm, maxiter=40, reltol=0.0, mode="tree", shape=ctx.reduce_shape(dtype), proj_shape=it.proj_shape())
            assert hist.size == 40 and np.array_equal(hist, ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
        assert shapes[1] == ctx.reduce_shape(dtype) and (shapes[0] == ctx.spmv_dot_shape()) == (Am.n % 2 == 0)
    As = orc.CSC.from_scipy((A.to_scipy() + 20 * sp.eye(A.n)).tocsc()).astype(dtype)
    dAs = pkg.HipCSR(As.n, As.n, As.colptr, As.rowval, As.nzval)
    d = pkg.HipVector.from_numpy((1 + 0.1 * np.arange(A.n) / A.n).astype(dtype))
    for Pl in (None, pkg.JacobiPrec(d)):
        runs = []
        for fused in (True, False):
            x = pkg.HipVector.from_numpy(np.zeros(A.n, dtype))
            it = pkg.chebyshev_iterable_(x, dAs, pkg.HipVector.from_numpy(b), 20.0, 32.0, initially_zero=True, maxiter=25, reltol=0.0, Pl=Pl, fused=fused)
            runs.append((np.array(list(it)), x.to_numpy()))
        assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][0].size == 25


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,shift", [(1, 0), (1000, 0), (5000, 1), (300001, 0)])
def test_fused_entry_points_against_numpy(pkg, orc, ctx, dtype, n, shift):
    """element-wise results exact (one rounded op per step, as numpy does); reductions equal the tree oracle;
    `shift` = 1 offsets every pointer by one element (unaligned scalar path)"""
    rng = np.random.default_rng(n + shift)
    W, L = ctx.reduce_shape(dtype)
    T = np.dtype(dtype).type

    def dev(a):
        full = pkg.HipVector.from_numpy(np.concatenate([np.zeros(shift, dtype), a]))
        return full, pkg.HipVector.wrap(full.ptr + shift * np.dtype(dtype).itemsize, a.size, dtype, ctx, owner=full)

    x, y, z = (rng.standard_normal(n).astype(dtype) for _ in range(3))
    alpha = T(-0.7321)
    (_, dx), (fy, dy), (_, dz) = dev(x), dev(y), dev(z)
    got = pkg.axpy_dot_(alpha, dx, dy, dz)
    y1 = y + alpha * x
    assert np.array_equal(dy.to_numpy(), y1) and got == orc.dot(z, y1, "tree", W, L)
    got = pkg.axpy_dot_(alpha, None, dy, dz)                                    # no update, just the projection
    assert np.array_equal(dy.to_numpy(), y1) and got == orc.dot(z, y1, "tree", W, L)
    got = pkg.axpy_dot_(alpha, dz, dy, None, hints=1)                           # update + norm, x streamed
    y2 = y1 + alpha * z
    assert np.array_equal(dy.to_numpy(), y2) and got == orc.nrm2(y2, "tree", W, L)
    # x += a u; r -= a c; norm(r)
    u, xs, c, r = (rng.standard_normal(n).astype(dtype) for _ in range(4))
    (_, du), (_, dxs), (_, dc), (_, dr) = dev(u), dev(xs), dev(c), dev(r)
    got = pkg.axpy2_nrm2_(alpha, du, dxs, dc, dr, hints=7 if n > 1000 else 0)
    r1 = r - alpha * c
    assert np.array_equal(dxs.to_numpy(), xs + alpha * u) and np.array_equal(dr.to_numpy(), r1) and got == orc.nrm2(r1, "tree", W, L)
    # Chebyshev direction
    dd = (1 + rng.random(n)).astype(dtype)
    (_, ddd), (_, dout) = dev(dd), dev(np.zeros(n, dtype))
    L_ = pkg.lib()
    import ctypes as C
    beta = np.array([0.37], dtype)
    for first in (1, 0):
        for diag in (None, ddd):
            assert L_.mik_cheb_direction(ctx.handle, dr.code, n, C.c_void_p(dr.ptr), C.c_void_p(diag.ptr if diag else None),
                                         beta.ctypes.data_as(C.c_void_p), first, C.c_void_p(dout.ptr)) == 0
            cc = r1 / dd if diag else r1
            assert np.array_equal(dout.to_numpy(), cc if first else cc + beta[0] * cc)
    # MINRES tail
    vn, vc, wc, wp, xx = (rng.standard_normal(n).astype(dtype) for _ in range(5))
    sc = np.array([1.37, -0.41, 0.93, 0.77, -1.9], dtype)
    for use_c, use_p in ((False, False), (True, False), (True, True)):
        (_, dvn), (_, dvc), (_, dwc), (_, dwp), (_, dwn), (_, dxx) = dev(vn), dev(vc), dev(wc), dev(wp), dev(np.zeros(n, dtype)), dev(xx)
        p = [sc[i:i + 1].ctypes.data_as(C.c_void_p) for i in range(5)]
        assert L_.mik_minres_update(ctx.handle, dvn.code, n, p[0], C.c_void_p(dvn.ptr), C.c_void_p(dvc.ptr), p[1],
                                    C.c_void_p(dwc.ptr if use_c else None), p[2], C.c_void_p(dwp.ptr if use_p else None), p[3],
                                    C.c_void_p(dwn.ptr), p[4], C.c_void_p(dxx.ptr), 3 if use_p else 0) == 0
        w = vc.copy()
        if use_c:
            w = w + sc[1] * wc
        if use_p:
            w = w + sc[2] * wp
        w = w * sc[3]
        assert np.array_equal(dvn.to_numpy(), vn * sc[0]) and np.array_equal(dwn.to_numpy(), w) and np.array_equal(dxx.to_numpy(), xx + sc[4] * w)


@pytest.mark.gpu
def test_minres_whole_iteration_call_rescales_the_lanczos_norm(pkg, orc, ctx):
    """fp32 operator with entries ~1e22: |v_next|^2 overflows a plain sum of squares, so norm(v_next) (src/minres.jl:112) takes
    the scaled path -- inside mik_minres_step the tail sweep is held back, the host rescales, the scalar kernel and the tail
    follow.  The whole-iteration call and the statement-by-statement path both equal the oracle (each with the tree its projection has), and stay finite."""
    dtype = np.float32
    A = orc.laplace(6, 3)
    A = orc.CSC(A.n, A.colptr, A.rowval, (A.nzval * 1e22).astype(dtype), A.index_base)
    b = (orc.hashed_rhs(A.n) * 1e22).astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    for fused in (True, False):
        x = pkg.HipVector.from_numpy(np.zeros(A.n, dtype))
        it = pkg.minres_iterable_(x, dA, pkg.HipVector.from_numpy(b), reltol=0.0, initially_zero=True, maxiter=12, fused=fused)
        hist = np.array(list(it))
        assert hist.size == 12 and np.all(np.isfinite(hist)) and np.all(np.isfinite(x.to_numpy()))
        xo, ho = orc.minres(A, b, maxiter=12, reltol=0.0, mode="tree", shape=ctx.reduce_shape(dtype), proj_shape=it.proj_shape())
        assert np.array_equal(hist, ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
