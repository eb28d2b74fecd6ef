"""Edge cases of the device path: empty and tiny systems, maxiter = 0, sizes around the 256-row / segment boundaries."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def up(pkg, M):
    M = M.tocsc()
    M.sort_indices()
    return pkg.HipCSR(M.shape[0], M.shape[1], M.indptr, M.indices, M.data, index_base=0)


def test_empty_system(pkg, ctx):
    A = pkg.HipCSR(0, 0, np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0), index_base=0)
    b = pkg.HipVector(0)
    x, ch = pkg.cg(A, b, log=True)
    assert ch.iters == 0 and x.to_numpy().size == 0
    assert pkg.dot(b, b) == 0 and pkg.norm(b) == 0


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 511, 513, 1023, 1025, 2049])
def test_sizes_around_block_boundaries(pkg, orc, ctx, n):
    T = sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csc") if n > 1 else sp.csc_matrix([[2.5]])
    A = orc.CSC.from_scipy(T)
    b = orc.hashed_rhs(n)
    dA = up(pkg, T)
    x, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True)
    xo, ho = orc.cg(A, b, mode="tree", shape=ctx.cg_shape(np.float64))
    assert ch.iters == ho["iters"] and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    xg, cg_ = pkg.gmres(dA, pkg.HipVector.from_numpy(b), restart=min(7, n), log=True)
    xgo, hgo = orc.gmres(A, b, restart=min(7, n), mode="tree", shape=ctx.reduce_shape(np.float64))
    assert cg_.iters == hgo["iters"] and np.array_equal(cg_["resnorm"], hgo["resnorm"]) and np.array_equal(xg.to_numpy(), xgo)


def test_maxiter_zero_and_one(pkg, orc, ctx):
    A = orc.laplace(6, 3)
    b = orc.hashed_rhs(A.n)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    x, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=0)
    assert ch.iters == 0 and not ch.isconverged and np.all(x.to_numpy() == 0)
    x, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=1)
    assert ch.iters == 1 and ch.mvps == 1
    x, ch = pkg.gmres(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=0)
    assert ch.iters == 0 and np.all(x.to_numpy() == 0)
    xo, ho = orc.gmres(A, b, maxiter=1, mode="tree", shape=ctx.reduce_shape(np.float64))
    x, ch = pkg.gmres(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=1)
    assert ch.iters == 1 and ch.mvps == ho["mvps"] and np.array_equal(x.to_numpy(), xo)


def test_restart_larger_than_needed_and_restart_one(pkg, orc, ctx):
    A, b = orc.advdiff(6, 30.0)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    for restart in (1, 2, 60):
        x, ch = pkg.gmres(dA, pkg.HipVector.from_numpy(b), restart=restart, log=True, maxiter=150)
        xo, ho = orc.gmres(A, b, restart=restart, maxiter=150, mode="tree", shape=ctx.reduce_shape(np.float64))
        assert ch.iters == ho["iters"] and ch.mvps == ho["mvps"] and np.array_equal(ch["resnorm"], ho["resnorm"])
        assert np.array_equal(x.to_numpy(), xo)


def test_two_contexts_and_handles_are_independent(pkg, orc, ctx):
    """two solves interleaved on two contexts (own streams) do not disturb each other"""
    c2 = pkg.HipContext(0)
    A = orc.laplace(8, 3)
    b = orc.hashed_rhs(A.n)
    A1 = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, ctx=ctx)
    A2 = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, ctx=c2)
    b1, b2 = pkg.HipVector.from_numpy(b, ctx), pkg.HipVector.from_numpy(2 * b, c2)
    it1 = pkg.cg_iterator_(pkg.zerox(A1, b1), A1, b1, initially_zero=True)
    it2 = pkg.cg_iterator_(pkg.zerox(A2, b2), A2, b2, initially_zero=True)
    r1, r2, i = [], [], 0
    while True:
        a, c = it1.iterate(i), it2.iterate(i)
        if a is None and c is None:
            break
        if a is not None:
            r1.append(a[0])
        if c is not None:
            r2.append(c[0])
        i += 1
    _, h = orc.cg(A, b, mode="tree", shape=ctx.cg_shape(np.float64))
    _, h2 = orc.cg(A, 2 * b, mode="tree", shape=ctx.cg_shape(np.float64))
    assert np.array_equal(r1, h["resnorm"]) and np.array_equal(r2, h2["resnorm"])


@pytest.mark.parametrize("method", ["mgs", "cgs", "dgks"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_orthogonalize_against_an_empty_basis(pkg, orc, ctx, method, dtype):
    """k = 0: nothing to project against, w is only normalised (src/orthogonalize.jl with an n x 0 view)"""
    n = 5000
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[method]
    w0 = np.random.default_rng(5).standard_normal(n).astype(dtype)
    V = pkg.HipMatrix.from_numpy(np.zeros((n, 1), dtype, order="F"))
    w = pkg.HipVector.from_numpy(w0)
    h = np.zeros(1, dtype)
    nrm = pkg.orthogonalize_and_normalize_(V, 0, w, h, M)
    W, L = ctx.reduce_shape(dtype)
    want = orc.nrm2(w0, "tree", W, L)
    assert nrm == want and np.array_equal(w.to_numpy(), w0 * (dtype(1) / want))
    assert pkg.gemv_t_(V, 0, w).size == 0


@pytest.mark.parametrize("method", ["mgs", "cgs"])
def test_orthogonalize_large_basis_with_streaming_hints(pkg, orc, ctx, method):
    """basis larger than the cache budget (n k 8 B > 192 MB): the non-temporal paths of the Gram-Schmidt kernels"""
    n, k = 3_300_000, 8
    rng = np.random.default_rng(11)
    V = np.asfortranarray(rng.standard_normal((n, k)))
    w0 = rng.standard_normal(n)
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt()}[method]
    dw, h = pkg.HipVector.from_numpy(w0), np.zeros(k)
    nrm = pkg.orthogonalize_and_normalize_(pkg.HipMatrix.from_numpy(V), k, dw, h, M)
    W, L = ctx.reduce_shape(np.float64)
    wo, ho, no = orc.orthogonalize(V, w0, method=method, mode="tree", W=W, L=L)
    assert nrm == no and np.array_equal(h, ho) and np.array_equal(dw.to_numpy(), wo)


@pytest.mark.parametrize("off", [1, 3, 64])
def test_cg_with_x_a_view(pkg, orc, ctx, off):
    """test/cg.jl:89-96 "CG with a view": x = view(rand(10, 2), :, 1) -- the solution vector is a slice of a larger device buffer (here also at an
    offset that is only 8-byte aligned: the sweeps over x take their scalar-load form); history and solution equal the oracle's bit for bit and the
    neighbouring entries of the buffer stay untouched"""
    A = orc.laplace(9, 3)
    b = orc.hashed_rhs(A.n)
    x0 = np.random.default_rng(7).standard_normal(A.n)
    big = np.full(A.n + 130, 7.25)
    big[off:off + A.n] = x0
    dbig = pkg.HipVector.from_numpy(big)
    xv = dbig.view(off, A.n)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)
    x, ch = pkg.cg_(xv, dA, pkg.HipVector.from_numpy(b), log=True)
    xo, ho = orc.cg(A, b, x0=x0, mode="tree", shape=ctx.cg_shape(np.float64))
    assert ch.isconverged and ch.iters == ho["iters"] and np.array_equal(ch["resnorm"], np.asarray(ho["resnorm"]))
    out = dbig.to_numpy()
    assert np.array_equal(out[off:off + A.n], xo) and np.all(out[:off] == 7.25) and np.all(out[off + A.n:] == 7.25)
