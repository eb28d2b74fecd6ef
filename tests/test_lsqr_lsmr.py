"""LSQR and LSMR (src/lsqr.jl, src/lsmr.jl): rectangular operators and products with adjoint(A).  CPU: the oracle against what test/lsqr.jl and
test/lsmr.jl check (the deterministic SOL matrices are known-answer pins).  GPU: the device path -- every vector statement one L1 call,
adjoint(A) = the SparseMatrixCSC's own arrays read as CSR (HipCSR.with_adjoint) -- bit for bit against the oracle's TREE mode: the C
restatement and the Python mirror were written separately from the same reference lines."""
import numpy as np
import pytest


def sol_matrix(m, n):
    """test/lsqr.jl:25-29 / test/lsmr.jl:60-64: diagonal 1..mn, sub-diagonal 1..mn-1, m x n"""
    import scipy.sparse as sp
    mn = min(m, n)
    S = sp.lil_matrix((m, n))
    for i in range(mn):
        S[i, i] = float(i + 1)
    for i in range(mn - 1):
        S[i + 1, i] = float(i + 1)
    return S.tocsc()


# ---- oracle ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_lsqr_small_dense(orc, dtype):
    import scipy.sparse as sp
    rng = np.random.default_rng(1234321)
    A = rng.random((10, 5)).astype(dtype)                                    # test/lsqr.jl:14-22
    b = rng.random(10).astype(dtype)
    x, h = orc.lsqr(sp.csc_matrix(A), b)
    xs = np.linalg.lstsq(A.astype(np.float64), b.astype(np.float64), rcond=None)[0]
    sq = np.sqrt(np.finfo(dtype).eps)
    assert np.linalg.norm(x - xs) <= 4 * sq and h["isconverged"]
    assert abs(h["resnorm"][-1] - np.linalg.norm(b - A @ x)) <= sq
    assert h["mvps"] == h["iters"] and h["mtvps"] == h["iters"] + 1


@pytest.mark.parametrize("m,n", [(10, 10), (20, 10)])
def test_oracle_lsqr_lsmr_sol(orc, m, n):
    A = sol_matrix(m, n)
    xt = np.arange(n, 0, -1, dtype=np.float64)                               # test/lsqr.jl:38-41
    b = A @ xt
    x, h = orc.lsqr(A, b, atol=1e-6, btol=1e-6, conlim=1e10, maxiter=10 * n)
    assert np.linalg.norm(b - A @ x) <= 1e-4
    x, h = orc.lsmr(A, b, atol=1e-7, btol=1e-7, conlim=1e10, maxiter=10 * n)  # test/lsmr.jl:80-83
    assert np.linalg.norm(b - A @ x) <= 1e-4 and h["isconverged"]
    x, h = orc.lsmr(A, b, lam=0.1, atol=1e-7, btol=1e-7, conlim=1e10, maxiter=10 * n)
    xd = np.linalg.solve((A.T @ A).toarray() + 0.01 * np.eye(n), A.T @ b)   # the damped normal equations
    assert np.linalg.norm(x - xd) <= 1e-4 * np.linalg.norm(xd)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_lsmr_small_dense_and_dampened(orc, dtype):
    import scipy.sparse as sp
    rng = np.random.default_rng(1234321)
    A = rng.random((10, 5)).astype(dtype)                                    # test/lsmr.jl:69-75
    b = rng.random(10).astype(dtype)
    x, h = orc.lsmr(sp.csc_matrix(A), b)
    xs = np.linalg.lstsq(A.astype(np.float64), b.astype(np.float64), rcond=None)[0]
    assert np.linalg.norm(x - xs) <= (1 if dtype == np.float64 else 8) * np.sqrt(np.finfo(dtype).eps)   # (fp32: sensitive to the random matrix, like test/lsqr.jl:20)
    for m, n in ((10, 10), (20, 10)):                                        # test/lsmr.jl:86-96: [A; diag(v)] as one sparse operator
        bb, AA, v = rng.random(m), rng.random((m, n)), rng.random(n)
        Aaug = sp.vstack([sp.csc_matrix(AA), sp.diags(v)]).tocsc()
        x, h = orc.lsmr(Aaug, np.concatenate([bb, np.zeros(n)]))
        assert np.linalg.norm((AA.T @ AA + np.diag(v) ** 2) @ x - AA.T @ bb) <= 1e-3


def test_adjoint_needs_an_operator_uploaded_with_it(pkg):
    class Plain:
        pass
    with pytest.raises(pkg.MikError):
        pkg.extras.adjoint(Plain())


# ---- device ------------------------------------------------------------------------------------------
def _rect(rng, m, n, density, dtype):
    import scipy.sparse as sp
    S = sp.random(m, n, density=density, random_state=rng, format="csc")
    S = (S + sp.eye(m, n) * 2).tocsc().astype(dtype)
    S.sort_indices()
    return S


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,damp,start", [((300, 120), 0.0, False), ((300, 120), 0.3, True), ((150, 150), 0.0, True), ((120, 300), 0.0, False)])
def test_lsqr_device_bit_exact(pkg, orc, ctx, dtype, shape, damp, start):
    rng = np.random.default_rng(17)
    m, n = shape
    S = _rect(rng, m, n, 0.05, dtype)
    b = rng.standard_normal(m).astype(dtype)
    x0 = rng.standard_normal(n).astype(dtype) if start else None
    xo, ho = orc.lsqr(S, b, x0, damp=damp, maxiter=60, mode="tree", shape=ctx.reduce_shape(dtype))
    dA = pkg.extras.with_adjoint_from_scipy(S)
    assert (dA.size(1), dA.size(2)) == (m, n) and (pkg.extras.adjoint(dA).size(1), pkg.extras.adjoint(dA).size(2)) == (n, m)
    for fused in (True, False):                  # the fused sweeps (mik_xpby_nrm2, mik_lsqr_update) and one L1 call per statement: same bits
        if start:
            x, ch = pkg.extras.lsqr_(pkg.HipVector.from_numpy(x0), dA, pkg.HipVector.from_numpy(b), damp=damp, maxiter=60, log=True, fused=fused)
        else:
            x, ch = pkg.extras.lsqr(dA, pkg.HipVector.from_numpy(b), damp=damp, maxiter=60, log=True, fused=fused)
        assert ch.iters == ho["iters"] > 5 and ch.mvps == ho["mvps"] and ch.mtvps == ho["mtvps"] and ch.isconverged == ho["isconverged"], fused
        for key in ("resnorm", "anorm", "rnorm", "cnorm"):
            assert np.array_equal(ch[key], ho[key]), (key, fused)
        assert np.array_equal(x.to_numpy(), xo), fused


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,lam,start", [((300, 120), 0.0, False), ((300, 120), 0.2, True), ((150, 150), 0.0, True), ((120, 300), 0.0, False)])
def test_lsmr_device_bit_exact(pkg, orc, ctx, dtype, shape, lam, start):
    rng = np.random.default_rng(19)
    m, n = shape
    S = _rect(rng, m, n, 0.05, dtype)
    b = rng.standard_normal(m).astype(dtype)
    x0 = rng.standard_normal(n).astype(dtype) if start else None
    xo, ho = orc.lsmr(S, b, x0, lam=lam, maxiter=60, mode="tree", shape=ctx.reduce_shape(dtype))
    dA = pkg.extras.with_adjoint_from_scipy(S)
    for fused in (True, False):                  # mik_xpby_nrm2 + mik_lsmr_update, and one L1 call per statement: same bits
        if start:
            x, ch = pkg.extras.lsmr_(pkg.HipVector.from_numpy(x0), dA, pkg.HipVector.from_numpy(b), lam=lam, maxiter=60, log=True, fused=fused)
        else:
            x, ch = pkg.extras.lsmr(dA, pkg.HipVector.from_numpy(b), lam=lam, maxiter=60, log=True, fused=fused)
        assert ch.iters == ho["iters"] > 5 and ch.mvps == ho["mvps"] and ch.mtvps == ho["mtvps"] and ch.isconverged == ho["isconverged"], fused
        for key in ("anorm", "rnorm", "cnorm"):
            assert np.array_equal(ch[key], ho[key]), (key, fused)
        assert np.array_equal(x.to_numpy(), xo), fused


@pytest.mark.gpu
def test_lsqr_lsmr_device_sol_known_answer(pkg, orc, ctx):
    """the SOL test of test/lsqr.jl:31-42 / test/lsmr.jl:77-84 on the device, and the adjoint product itself against scipy"""
    for m, n in ((10, 10), (20, 10)):
        A = sol_matrix(m, n)
        xt = np.arange(n, 0, -1, dtype=np.float64)
        b = A @ xt
        dA = pkg.extras.with_adjoint_from_scipy(A)
        x = pkg.extras.lsqr(dA, pkg.HipVector.from_numpy(b), atol=1e-6, btol=1e-6, conlim=1e10, maxiter=10 * n)
        assert np.linalg.norm(b - A @ x.to_numpy()) <= 1e-4
        x = pkg.extras.lsmr(dA, pkg.HipVector.from_numpy(b), atol=1e-7, btol=1e-7, conlim=1e10, maxiter=10 * n)
        assert np.linalg.norm(b - A @ x.to_numpy()) <= 1e-4
        y = pkg.HipVector.from_numpy(np.zeros(n))
        pkg.mul_(y, pkg.extras.adjoint(dA), pkg.HipVector.from_numpy(b))
        assert np.allclose(y.to_numpy(), A.T @ b, rtol=1e-14, atol=0)


# ---- the Python mirrors on a numpy stand-in for the device types (CPU) --------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,damp,start", [((40, 15), 0.0, False), ((40, 15), 0.3, True), ((15, 40), 0.0, False), ((20, 20), 0.1, True)])
def test_python_mirrors_equal_the_c_oracle_on_a_host_double(pkg, orc, monkeypatch, dtype, shape, damp, start):
    """lsqr_ and lsmr_ of api.py -- the code the device path runs -- with every vector statement evaluated by the oracle's SEQ primitives
    (tests/host_double.py): histories, counters and solutions equal the C restatement bit for bit.  Two restatements written separately
    (C, Python) from the same reference lines; on the GPU the same Python code is compared in TREE mode (tests below)."""
    from importlib import import_module
    from host_double import FakeOperator, FakeVector, patch
    api = import_module(pkg.__name__ + ".extras")
    patch(monkeypatch, api, orc)
    rng = np.random.default_rng(23)
    m, n = shape
    S = _rect(rng, m, n, 0.2, dtype)
    b = rng.standard_normal(m).astype(dtype)
    x0 = rng.standard_normal(n).astype(dtype) if start else np.zeros(n, dtype)
    A = FakeOperator(orc, S)
    xo, ho = orc.lsqr(S, b, x0 if start else None, damp=damp, maxiter=50)
    x, ch = api.lsqr_(FakeVector(x0.copy()), A, FakeVector(b), damp=damp, maxiter=50, log=True)
    assert ch.iters == ho["iters"] > 3 and ch.mvps == ho["mvps"] and ch.mtvps == ho["mtvps"] and ch.isconverged == ho["isconverged"]
    for key in ("resnorm", "anorm", "rnorm", "cnorm"):
        assert np.array_equal(ch[key], ho[key]), key
    assert np.array_equal(x.to_numpy(), xo)
    xo, ho = orc.lsmr(S, b, x0 if start else None, lam=damp, maxiter=50)
    x, ch = api.lsmr_(FakeVector(x0.copy()), A, FakeVector(b), lam=damp, maxiter=50, log=True)
    assert ch.iters == ho["iters"] > 3 and ch.mvps == ho["mvps"] and ch.mtvps == ho["mtvps"] and ch.isconverged == ho["isconverged"]
    for key in ("anorm", "rnorm", "cnorm"):
        assert np.array_equal(ch[key], ho[key]), key
    assert np.array_equal(x.to_numpy(), xo)


@pytest.mark.gpu
def test_lsqr_lsmr_qmr_device_across_operator_layouts(pkg, orc, ctx):
    """the three solvers with adjoint products on an operator whose default layout is NOT the CSR arrays (the 12^3 Laplacian: one mask byte per row,
    for A and for adjoint(A) alike), then with both operators pinned to their plain CSR arrays: same bits, equal to the oracle"""
    A = orc.laplace(12, 3)
    b = orc.hashed_rhs(A.n)
    S = A.to_scipy()
    shape = ctx.reduce_shape(np.float64)
    ref = {"lsqr": orc.lsqr(S, b, maxiter=40, mode="tree", shape=shape), "lsmr": orc.lsmr(S, b, maxiter=40, mode="tree", shape=shape),
           "qmr": orc.qmr(S, b, maxiter=40, mode="tree", shape=shape)}
    dA = pkg.extras.with_adjoint(A.n, A.n, A.colptr, A.rowval, A.nzval)
    assert dA.layout() != "csr-rowblock" and pkg.extras.adjoint(dA).layout() == dA.layout()
    for layout in ("auto", "csr"):
        dA.set_layout(layout)
        pkg.extras.adjoint(dA).set_layout(layout)
        for name, fn in (("lsqr", pkg.extras.lsqr), ("lsmr", pkg.extras.lsmr), ("qmr", pkg.extras.qmr)):
            x, ch = fn(dA, pkg.HipVector.from_numpy(b), maxiter=40, log=True)
            xo, ho = ref[name]
            assert ch.iters == ho["iters"] and np.array_equal(x.to_numpy(), xo), (name, layout)
            key = "resnorm" if name != "lsmr" else "rnorm"
            assert np.array_equal(ch[key], ho[key]), (name, layout)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1e-140, 1e137])
def test_lsqr_lsmr_device_on_a_badly_scaled_operator(pkg, orc, ctx, scale):
    """A scaled by 1e-140 / 1e137 (x, of the size of 1 / scale, still has a representable square): rho, alpha and with them wrho = w / rho leave the range where a plain sum of squares is safe -- norm(wrho) of the
    fused LSQR tail (wrho is never stored) goes through its scratch-vector fallback, norm(v) through the scaled recomputation; same bits as the
    oracle and as the statement-by-statement path"""
    rng = np.random.default_rng(29)
    S = (_rect(rng, 200, 90, 0.06, np.float64) * scale).tocsc()
    b = rng.standard_normal(200)
    shape = ctx.reduce_shape(np.float64)
    dA = pkg.extras.with_adjoint_from_scipy(S)
    xo, ho = orc.lsqr(S, b, maxiter=25, atol=0.0, btol=0.0, conlim=0.0, mode="tree", shape=shape)
    assert np.all(np.isfinite(ho["resnorm"])) and ho["iters"] > (5 if scale < 1 else 0)     # (1e137: the reference's own 1 + test3 <= 1 stops it after one iteration)
    for fused in (True, False):
        x, ch = pkg.extras.lsqr(dA, pkg.HipVector.from_numpy(b), maxiter=25, atol=0.0, btol=0.0, conlim=0.0, log=True, fused=fused)
        assert ch.iters == ho["iters"] and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(ch["cnorm"], ho["cnorm"]), fused
        assert np.array_equal(x.to_numpy(), xo), fused
    xo, ho = orc.lsmr(S, b, maxiter=25, atol=0.0, btol=0.0, conlim=0.0, mode="tree", shape=shape)
    assert ho["iters"] >= 1                      # (LSMR's condA starts from rhobar = 1, src/lsmr.jl:126: it stops after one iteration on such operators)
    for fused in (True, False):
        x, ch = pkg.extras.lsmr(dA, pkg.HipVector.from_numpy(b), maxiter=25, atol=0.0, btol=0.0, conlim=0.0, log=True, fused=fused)
        assert ch.iters == ho["iters"] and np.array_equal(ch["rnorm"], ho["rnorm"]) and np.array_equal(x.to_numpy(), xo), fused
