"""BiCGStab(l) (src/bicgstabl.jl): the reference's property tests against
the oracle (CPU) and bit-level parity of the device composition against the oracle (GPU)."""
import numpy as np
import pytest


# ---- oracle: test/bicgstabl.jl ---------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("l", [2, 4])
def test_oracle_dense(orc, dtype, l):
    rng = np.random.default_rng(123)
    n = 20
    Ad = (rng.random((n, n)) + 15 * np.eye(n)).astype(dtype)                 # test/bicgstabl.jl:17
    b = (Ad @ np.ones(n, dtype)).astype(dtype)
    reltol = float(np.sqrt(np.finfo(dtype).eps))
    A = orc.CSC.from_dense(Ad)
    sh = rng.random(n).astype(dtype)
    x1, h1 = orc.bicgstabl(A, b, l, r_shadow=sh, max_mv_products=100, reltol=reltol)
    assert np.linalg.norm(Ad @ x1 - b) / np.linalg.norm(b) <= reltol          # :27
    x2, h2 = orc.bicgstabl(A, b, l, rng.random(n).astype(dtype), r_shadow=sh, max_mv_products=100, reltol=reltol)
    assert np.linalg.norm(Ad @ x2 - b) / np.linalg.norm(b) <= reltol          # :34


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_termination(orc, dtype):
    T3 = np.array([[2, -1, 0], [-1, 2, -1], [0, -1, 2]], dtype)               # test/bicgstabl.jl:50-72
    A = orc.CSC.from_dense(T3)
    b = np.ones(3, dtype)
    x0 = np.linalg.solve(T3.astype(np.float64), b.astype(np.float64)).astype(dtype)
    pert = (10 * np.sqrt(np.finfo(dtype).eps) * np.array([-1.0, 1.0, -1.0])).astype(dtype)
    sh = np.array([0.3, 0.7, 0.5], dtype)
    x, ch = orc.bicgstabl(A, b, 2, x0 + pert, r_shadow=sh)
    assert 1 <= ch["iters"] <= 3 // 2                                          # :63
    r0 = float(np.linalg.norm(T3 @ (x0 + pert) - b))
    x, ch = orc.bicgstabl(A, b, 2, x0 + pert, r_shadow=sh, abstol=2 * r0, reltol=0.0)
    assert ch["iters"] == 0                                                    # :70


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("l", [1, 2, 4])
@pytest.mark.parametrize("with_x0", [False, True])
def test_oracle_pl_equals_the_explicitly_preconditioned_system(orc, dtype, l, with_x0):
    """Pins the oracle's `Pl` path (ldiv! at src/bicgstabl.jl:55, :98, :108) without Julia: with a diagonal of POWERS OF TWO,
    scaling the rows of A and the entries of b by 1 / d_i is exact and commutes with every rounding of the row sums, so
    bicgstabl(A, b; Pl = D) must give bit for bit the history and x of bicgstabl(D^-1 A, D^-1 b; Pl = Identity) -- in every
    summation mode.  A generic (non power of two) diagonal then differs from that only by the roundings of the divisions."""
    A, b = orc.advdiff(8, 150.0)
    A, b = A.astype(dtype), b.astype(dtype)
    n = A.n
    d = (2.0 ** ((np.arange(n) * 7) % 9 - 4)).astype(dtype)
    S = A.to_scipy().tocsc()
    import scipy.sparse as sp
    SA = (sp.diags(1.0 / d.astype(np.float64)) @ S.astype(np.float64)).astype(dtype).tocsc()
    SA.sort_indices()
    As = orc.CSC(n, SA.indptr.astype(np.int64), SA.indices.astype(np.int64), SA.data.astype(dtype), 0)
    sh = (orc.hashed_rhs(n) + 0.5).astype(dtype)
    x0 = np.random.default_rng(4).standard_normal(n).astype(dtype) if with_x0 else None
    for mode, shape in (("seq", (1, 1)), ("tree", (2, 2))):
        x1, h1 = orc.bicgstabl(A, b, l, x0, r_shadow=sh, max_mv_products=40 * l, reltol=0.0, mode=mode, shape=shape, pl_diag=d)
        if with_x0:   # r0 = D^-1 (b - A x0) on the left, D^-1 b - (D^-1 A) x0 on the right: equal because the scaling is exact
            x2, h2 = orc.bicgstabl(As, (b / d).astype(dtype), l, x0, r_shadow=sh, max_mv_products=40 * l, reltol=0.0, mode=mode, shape=shape)
        else:
            x2, h2 = orc.bicgstabl(As, (b / d).astype(dtype), l, None, r_shadow=sh, max_mv_products=40 * l, reltol=0.0, mode=mode, shape=shape)
        assert h1["iters"] == h2["iters"] >= 10 and h1["mvps"] == h2["mvps"]
        assert np.array_equal(h1["resnorm"], h2["resnorm"], equal_nan=True) and np.array_equal(x1, x2, equal_nan=True)
    # and Pl really acts: not the unpreconditioned history
    _, h0 = orc.bicgstabl(A, b, l, x0, r_shadow=sh, max_mv_products=40 * l, reltol=0.0)
    assert not np.array_equal(h0["resnorm"], h1["resnorm"])


def test_lu_solve_matches_numpy(pkg, orc):
    rng = np.random.default_rng(0)
    for n in (1, 2, 4, 7):
        A = rng.standard_normal((n, n)) + n * np.eye(n)
        b = rng.standard_normal(n)
        x = orc.lu_solve(A, b)
        np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-12)
        Af, bf = np.asfortranarray(A.copy()), b.copy()
        assert np.array_equal(pkg.lu_solve_(Af, bf), x)                        # product (C++) == oracle (C), bit for bit
    with pytest.raises(np.linalg.LinAlgError):
        pkg.lu_solve_(np.zeros((2, 2), order="F"), np.ones(2))


# ---- device ----------------------------------------------------------------------------------------
def step_dot_shape(pkg, dA, b, l, **kw):
    """(W, L) of sigma and of rho from the second column on in the whole-iteration call for this operator (mik_bicgstab_dot_shape)"""
    x = pkg.HipVector.from_numpy(np.zeros(b.size, b.dtype))
    return pkg.bicgstabl_iterator_(x, dA, pkg.HipVector.from_numpy(b), l, max_mv_products=2 * l, initial_zero=True, **kw).dot_shape()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("l", [2, 4])
@pytest.mark.parametrize("knob25", [0, 1, 2])
def test_device_matches_oracle_bit_exact(pkg, orc, ctx, l, dtype, knob25):
    """the reference authors' own BiCGStab benchmark operator (benchmark/benchmark-linear-systems.jl:68-77), small.  Development knob MIK_KNOB_SOLVER_FORM:
    0 = the sweeps finalise the reductions in front of them, 1 = separate finaliser launches, 2 = no SpMV epilogues (sigma and rho as
    sweeps of their own, in the vector shape)"""
    A, b = orc.advdiff(12, 300.0)
    A, b = A.astype(dtype), b.astype(dtype)
    sh = (orc.hashed_rhs(A.n) + 0.5).astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    x0 = np.random.default_rng(2).standard_normal(A.n).astype(dtype)
    ctx.set_tuning(8, knob25)
    ds = step_dot_shape(pkg, dA, b, l)
    assert ds == (ctx.reduce_shape(dtype) if knob25 == 2 else ctx.spmv_dot_shape())
    for start in (None, x0):
        if start is None:
            x, ch = pkg.bicgstabl(dA, pkg.HipVector.from_numpy(b), l, log=True, max_mv_products=2000, r_shadow=pkg.HipVector.from_numpy(sh))
        else:
            x, ch = pkg.bicgstabl_(pkg.HipVector.from_numpy(start), dA, pkg.HipVector.from_numpy(b), l, log=True, max_mv_products=2000,
                                   r_shadow=pkg.HipVector.from_numpy(sh))
        xo, ho = orc.bicgstabl(A, b, l, start, r_shadow=sh, max_mv_products=2000, mode="tree", shape=ctx.reduce_shape(dtype), dot_shape=ds)
        assert ch.iters == ho["iters"] and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"]
        assert np.array_equal(ch["resnorm"], ho["resnorm"], equal_nan=True) and np.array_equal(x.to_numpy(), xo, equal_nan=True)   # fp32, l = 4 breaks down (NaN) identically on both sides
    if dtype == np.float64:
        S = A.to_scipy()
        assert ch.isconverged and np.linalg.norm(S @ x.to_numpy() - b) / np.linalg.norm(b) <= 1e-3   # recurrence residual, not the true one (src/bicgstabl.jl:161-163)


@pytest.mark.gpu
def test_device_reference_properties(pkg, orc, ctx):
    rng = np.random.default_rng(123)
    n = 20
    Ad = rng.random((n, n)) + 15 * np.eye(n)
    b = Ad @ np.ones(n)
    dA = pkg.HipCSR(n, n, *[getattr(orc.CSC.from_dense(Ad), f) for f in ("colptr", "rowval", "nzval")])
    for l in (2, 4):
        x1, his1 = pkg.bicgstabl(dA, pkg.HipVector.from_numpy(b), l, max_mv_products=100, log=True)
        assert isinstance(his1, pkg.ConvergenceHistory)
        assert np.linalg.norm(Ad @ x1.to_numpy() - b) / np.linalg.norm(b) <= np.sqrt(np.finfo(float).eps)
        xg = pkg.HipVector.from_numpy(rng.random(n))
        x2, his2 = pkg.bicgstabl_(xg, dA, pkg.HipVector.from_numpy(b), l, max_mv_products=100, log=True)
        assert x2 is xg                                                          # test/bicgstabl.jl:33
        assert np.linalg.norm(Ad @ x2.to_numpy() - b) / np.linalg.norm(b) <= np.sqrt(np.finfo(float).eps)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,k", [(1, 1), (1000, 3), (4099, 5), (300001, 2)])
def test_gram_equals_pairwise_dots(pkg, orc, ctx, dtype, n, k):
    """mik_gram: every entry bit-identical to dot(V[:, r], V[:, c]) (tree oracle); symmetric; one pass"""
    rng = np.random.default_rng(n + k)
    V = np.asfortranarray(rng.standard_normal((n, k)).astype(dtype))
    M = pkg.gram_(pkg.HipMatrix.from_numpy(V), k)
    W, L = ctx.reduce_shape(dtype)
    want = np.array([[orc.dot(np.ascontiguousarray(V[:, r]), np.ascontiguousarray(V[:, c]), "tree", W, L) for c in range(k)] for r in range(k)], dtype)
    assert np.array_equal(M, want) and np.array_equal(M, M.T)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("l", [1, 2, 4])
@pytest.mark.parametrize("N", [14, 11])
def test_fused_bicgstab_equals_statement_by_statement(pkg, orc, ctx, l, dtype, N):
    """the whole-iteration call and the statement-by-statement path against the oracle, each with the tree of its own sigma / rho (the
    call forms them in the SpMV launches: one partial per 256-row block); with the epilogues off (MIK_KNOB_SOLVER_FORM = 2) the two
    paths give the same bits"""
    A, b = orc.advdiff(N, 300.0)                             # (an even n: the operator's default kernel, two rows per lane, takes epilogues; N = 11: it does not)
    A, b = A.astype(dtype), b.astype(dtype)
    sh = (orc.hashed_rhs(A.n) + 0.5).astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    runs = {}
    for fused, knob in ((True, 0), (False, 0), (True, 2)):
        ctx.set_tuning(8, knob)
        x = pkg.HipVector.from_numpy(np.zeros(A.n, dtype))
        it = pkg.bicgstabl_iterator_(x, dA, pkg.HipVector.from_numpy(b), l, max_mv_products=60 * l, reltol=0.0, initial_zero=True,
                                     r_shadow=pkg.HipVector.from_numpy(sh), fused=fused)
        ds = it.dot_shape()
        assert ds == (ctx.spmv_dot_shape() if (fused and knob == 0 and A.n % 2 == 0) else ctx.reduce_shape(dtype))
        runs[fused, knob] = (np.array(list(it)), x.to_numpy())
        xo, ho = orc.bicgstabl(A, b, l, None, r_shadow=sh, max_mv_products=60 * l, reltol=0.0, mode="tree", shape=ctx.reduce_shape(dtype), dot_shape=ds)
        assert runs[fused, knob][0].size == 30
        assert np.array_equal(runs[fused, knob][0], ho["resnorm"], equal_nan=True) and np.array_equal(runs[fused, knob][1], xo, equal_nan=True)
    assert np.array_equal(runs[False, 0][0], runs[True, 2][0], equal_nan=True) and np.array_equal(runs[False, 0][1], runs[True, 2][1], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["csr", "fe", "irregular"])
def test_sigma_and_rho_in_the_csr_and_jagged_spmv_kernels(pkg, orc, ctx, dtype, kind):
    """dot(r_shadow, A u) as the SpMV's epilogue also in k_spmv_rowgather (the operator on its plain CSR arrays) and k_spmv_jds (a
    finite-element operator): bit-exact against the oracle with those dots in the SpMV-dot tree; an operator with split-off long rows
    has no epilogue and keeps the vector shape"""
    import scipy.sparse as sp
    if kind == "csr":
        A, b = orc.advdiff(10, 200.0)
        A, b = A.astype(dtype), b.astype(dtype)
        dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
        dA.set_layout("csr")
        want_kernel, want_shape = "k_spmv_rowgather", ctx.spmv_dot_shape()
    else:
        if kind == "fe":
            n, rowptr, colidx, val = pkg.fixtures.fe_matrix((12, 12), 3, dtype)
            want_kernel, want_shape = "k_spmv_jds", ctx.spmv_dot_shape()
        else:
            n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(20000, dtype)
            want_kernel, want_shape = "k_spmv_rowblock", ctx.reduce_shape(dtype)
        M = sp.csr_matrix((val, colidx, rowptr), shape=(n, n)).tocsc()
        M.sort_indices()
        A = orc.CSC(n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(dtype), 0)
        dA = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
        b = orc.hashed_rhs(n).astype(dtype)
    assert dA.spmv_kernel().split("+")[0] == want_kernel
    sh = (orc.hashed_rhs(A.n) + 0.5).astype(dtype)
    x = pkg.HipVector.from_numpy(np.zeros(A.n, dtype))
    it = pkg.bicgstabl_iterator_(x, dA, pkg.HipVector.from_numpy(b), 2, max_mv_products=40, reltol=0.0, initial_zero=True, r_shadow=pkg.HipVector.from_numpy(sh))
    assert it.dot_shape() == want_shape
    if kind == "irregular":
        orc.set_long_row(ctx.spmv_long_row(), ctx.spmv_long_segment(), ctx.spmv_long_group())
    try:
        hist = np.array(list(it))
        xo, ho = orc.bicgstabl(A, b, 2, None, r_shadow=sh, max_mv_products=40, reltol=0.0, mode="tree", shape=ctx.reduce_shape(dtype), dot_shape=it.dot_shape())
    finally:
        orc.set_long_row(0)
    assert hist.size == 10 and np.array_equal(hist, ho["resnorm"], equal_nan=True) and np.array_equal(x.to_numpy(), xo, equal_nan=True)


@pytest.mark.gpu
def test_sigma_and_rho_through_the_spread_finalisers(pkg, orc, ctx):
    """more than 16,384 row-blocks: the partials of sigma / rho out of the SpMV launches go through the 16-workgroup finalisers"""
    N = 164                                                  # 164^3 = 4.4 M rows = 17,231 row-blocks
    A = orc.laplace(N, 3)
    b = orc.hashed_rhs(A.n)
    sh = orc.hashed_rhs(A.n) + 0.5
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    x = pkg.HipVector.from_numpy(np.zeros(A.n))
    it = pkg.bicgstabl_iterator_(x, dA, pkg.HipVector.from_numpy(b), 2, max_mv_products=12, reltol=0.0, initial_zero=True, r_shadow=pkg.HipVector.from_numpy(sh))
    assert it.dot_shape() == ctx.spmv_dot_shape() and (A.n + 255) // 256 > 16384
    hist = np.array(list(it))
    xo, ho = orc.bicgstabl(A, b, 2, None, r_shadow=sh, max_mv_products=12, reltol=0.0, mode="tree", shape=ctx.reduce_shape(np.float64), dot_shape=it.dot_shape())
    assert hist.size == 3 and np.array_equal(hist, ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_singular_mr_system_raises_like_lu(pkg, fused):
    """A = 2 I, l = 1: the BiCG step lands on the solution (alpha = 1 / 2 exactly), rs = 0, and lu! of the 1 x 1 MR system
    [rs_2' rs_2] meets a zero pivot -- SingularException in the reference (src/bicgstabl.jl:124), LinAlgError here, from the
    whole-iteration call (the flag travels through its mirror) as from the statement-by-statement path."""
    n = 96
    dA = pkg.HipCSR(n, n, np.arange(1, n + 2), np.arange(1, n + 1), np.full(n, 2.0))
    b = pkg.HipVector.from_numpy(np.linspace(1.0, 2.0, n))
    it = pkg.bicgstabl_iterator_(pkg.zerox(dA, b), dA, b, 1, max_mv_products=100, initial_zero=True, fused=fused)
    with pytest.raises(np.linalg.LinAlgError):
        list(it)
    if fused:
        # a code of its own (MIK_ERR_SINGULAR = 8), never the invalid-argument code; the handle is latched afterwards
        import ctypes as C
        out = np.zeros(1)
        assert pkg.lib().mik_bicgstab_step(it._step, out.ctypes.data_as(C.c_void_p)) == 8
        assert pkg.lib().mik_bicgstab_step(None, out.ctypes.data_as(C.c_void_p)) == 1
        assert pkg.lib().mik_lu_solve(0, np.zeros((2, 2), order="F").ctypes.data_as(C.c_void_p), 2, 2, np.ones(2).ctypes.data_as(C.c_void_p)) == 8


@pytest.mark.gpu
def test_context_frees_step_handles_the_host_never_destroyed(pkg):
    """a finalizer that finds its context closed skips mik_*_destroy; the context owns the step handles still alive and
    frees them in mik_ctx_destroy (a destroyed handle unregisters itself: no double free either way)."""
    import ctypes as C
    L = pkg.lib()
    hctx = pkg.HipContext(0)
    n = 64
    dA = pkg.HipCSR(n, n, np.arange(1, n + 2), np.arange(1, n + 1), np.full(n, 2.0), ctx=hctx)
    blk = pkg.HipMatrix(n, 6, np.float64, hctx)
    v = [pkg.HipVector(n, np.float64, hctx) for _ in range(8)]
    P = lambda o: C.c_void_p(o.ptr)                                            # noqa: E731
    hs = [C.c_void_p() for _ in range(3)]
    for h in hs[:2]:
        assert L.mik_bicgstab_create(hctx.handle, dA.handle, 2, P(v[0]), P(blk.col(0)), blk.ld, P(blk.col(3)), blk.ld, P(v[1]), None, C.byref(h)) == 0
    assert L.mik_minres_create(hctx.handle, dA.handle, P(v[0]), P(v[1]), P(v[2]), P(v[3]), P(v[4]), P(v[5]), P(v[6]), 1.0, 0, C.byref(hs[2])) == 0
    assert L.mik_bicgstab_destroy(hs[0]) == 0                                  # one destroyed by the host, two left to the context
    del dA, blk, v
    hctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("l", [1, 2, 4])
@pytest.mark.parametrize("with_x0", [False, True])
def test_jacobi_pl_matches_oracle_bit_exact(pkg, orc, ctx, dtype, l, with_x0):
    """BiCGStab(l) with a diagonal Pl (ldiv! on the initial residual and after both mul! of the BiCG part,
    src/bicgstabl.jl:55, :98, :108) -- the whole-iteration call mik_bicgstab_step AND the statement-by-statement path against the
    ORACLE (tree mode; oracle/orc_impl.inc orc_bicgstabl with pl_diag): history, counters and x bit for bit, zero and nonzero
    start, fp64 and fp32."""
    A, b = orc.advdiff(10, 200.0)
    A, b = A.astype(dtype), b.astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    diag = A.to_scipy().diagonal().astype(dtype)
    sh = (orc.hashed_rhs(A.n) + 0.5).astype(dtype)
    x0 = np.random.default_rng(7).standard_normal(A.n).astype(dtype) if with_x0 else None
    max_mv = 40 * l
    xo, ho = orc.bicgstabl(A, b, l, x0, r_shadow=sh, max_mv_products=max_mv, reltol=0.0, mode="tree", shape=ctx.reduce_shape(dtype), pl_diag=diag)   # Pl: no SpMV epilogue, the vector shape throughout
    assert ho["iters"] == 20
    for fused in (True, False):
        x = pkg.HipVector.from_numpy(np.zeros(A.n, dtype) if x0 is None else x0.copy())
        it = pkg.bicgstabl_iterator_(x, dA, pkg.HipVector.from_numpy(b), l, Pl=pkg.JacobiPrec(pkg.HipVector.from_numpy(diag)), max_mv_products=max_mv,
                                     reltol=0.0, initial_zero=not with_x0, r_shadow=pkg.HipVector.from_numpy(sh), fused=fused)
        hist = np.array(list(it))
        assert hist.size == ho["iters"] and it.mv_products == ho["mvps"]
        assert np.array_equal(hist, ho["resnorm"], equal_nan=True) and np.array_equal(x.to_numpy(), xo, equal_nan=True)


@pytest.mark.gpu
def test_whole_iteration_calls_validate_their_arguments(pkg, ctx):
    """mik_bicgstab_create / mik_minres_create: status codes instead of crashes for what a host can get wrong"""
    import ctypes as C
    L = pkg.lib()
    n = 64
    dA = pkg.HipCSR(n, n, np.arange(1, n + 2), np.arange(1, n + 1), np.full(n, 2.0))
    v = [pkg.HipVector(n, np.float64) for _ in range(8)]
    blk = pkg.HipMatrix(n, 6, np.float64, v[0].ctx)
    P = lambda o: C.c_void_p(o.ptr)                                            # noqa: E731
    h = C.c_void_p()
    args = lambda l, A=dA.handle, x=P(v[0]): (v[0].ctx.handle, A, l, x, P(blk.col(0)), blk.ld, P(blk.col(0)), blk.ld, P(v[1]), None, C.byref(h))   # noqa: E731
    assert L.mik_bicgstab_create(*args(5)) == 5 and not h.value               # l > 4: MIK_ERR_NOTIMPL (the L1 entry points take over)
    assert L.mik_bicgstab_create(*args(0)) == 5
    assert L.mik_bicgstab_create(*args(2, A=None)) != 0 and not h.value
    assert L.mik_bicgstab_create(*args(2, x=None)) == 1 and not h.value
    assert L.mik_bicgstab_step(None, None) == 1 and L.mik_bicgstab_destroy(None) == 0
    rect = pkg.HipCSR(2, 3, np.array([1, 2, 3, 3]), np.array([1, 2]), np.array([1.0, 1.0]))
    assert L.mik_bicgstab_create(*args(2, A=rect.handle)) == 3                 # MIK_ERR_MISMATCH: not square
    m = lambda A=dA.handle, x=P(v[0]): (v[0].ctx.handle, A, x, P(v[1]), P(v[2]), P(v[3]), P(v[4]), P(v[5]), P(v[6]), 1.0, 0, C.byref(h))   # noqa: E731
    assert L.mik_minres_create(*m(A=rect.handle)) == 3 and not h.value
    assert L.mik_minres_create(*m(x=None)) == 1 and not h.value
    assert L.mik_minres_create(*m()) == 0 and h.value
    out = np.zeros(1)
    assert L.mik_minres_step(h, 0, out.ctypes.data_as(C.c_void_p)) == 1       # iteration counts from 1 (src/minres.jl:91)
    assert L.mik_minres_step(h, 1, None) == 1
    assert L.mik_minres_destroy(h) == 0 and L.mik_minres_destroy(None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_whole_iteration_calls_with_separate_finalisers(pkg, orc, ctx, dtype):
    """Beyond 1,024 reduction segments mik_bicgstab_step / mik_minres_step run their finalisers as separate launches; development
    MIK_KNOB_SOLVER_FORM = 1 selects that form at any size: same histories and x as the launch-lean form (and therefore as the oracle)."""
    A, b = orc.advdiff(12, 300.0)
    A, b = A.astype(dtype), b.astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    sh = (orc.hashed_rhs(A.n) + 0.5).astype(dtype)
    L = orc.laplace(10, 3).astype(dtype)
    dL = pkg.HipCSR(L.n, L.n, L.colptr, L.rowval, L.nzval)
    bl = orc.hashed_rhs(L.n).astype(dtype)
    runs = []
    for knob in (0, 1):
        ctx.set_tuning(8, knob)
        try:
            x = pkg.HipVector.from_numpy(np.zeros(A.n, dtype))
            it = pkg.bicgstabl_iterator_(x, dA, pkg.HipVector.from_numpy(b), 2, max_mv_products=80, reltol=0.0, initial_zero=True,
                                         r_shadow=pkg.HipVector.from_numpy(sh))
            hb = np.array(list(it))
            y = pkg.HipVector.from_numpy(np.zeros(L.n, dtype))
            im = pkg.minres_iterable_(y, dL, pkg.HipVector.from_numpy(bl), reltol=0.0, initially_zero=True, maxiter=25)
            hm = np.array(list(im))
            runs.append((hb, x.to_numpy(), hm, y.to_numpy()))
        finally:
            ctx.set_tuning(8, 0)
    assert runs[0][0].size == 20 and runs[0][2].size == 25
    for a, c in zip(runs[0], runs[1]):
        assert np.array_equal(a, c, equal_nan=True)
