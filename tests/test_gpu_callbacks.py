"""Any operator with mul!, any preconditioner with ldiv! on the fused iterables (mik_cg_create_op / mik_gmres_create_op):
the reference's contract for A and Pl / Pr (docs/src/getting_started.md:25-30, docs/src/preconditioning.md:5-14),
exercised like test/cg.jl:71-85 (LinearMap(A), JacobiPrec) and test/gmres.jl:28-35, :59-66 (lu(A) as Pl / Pr, cumsum!)."""
import numpy as np
import pytest
import scipy.linalg
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def upload(pkg, A):
    return pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)


class PlainLdiv:
    """a preconditioner that is NOT a JacobiPrec for the library: only ldiv_(y, x), like any user type"""

    def __init__(self, pkg, d):
        self.inner = pkg.JacobiPrec(d)
        self.calls = 0

    def ldiv_(self, y, x=None):
        self.calls += 1
        return self.inner.ldiv_(y, x)


class DenseSolve:
    """ldiv!(y, F, x) with F = lu(A) (test/gmres.jl:18,28-35): host LU through a staging copy -- slow, exact"""

    def __init__(self, A):
        self.lu = scipy.linalg.lu_factor(A)

    def ldiv_(self, y, x=None):
        x = y if x is None else x
        v = scipy.linalg.lu_solve(self.lu, x.to_numpy().astype(np.float64))
        y.copyto_(type(y).from_numpy(v.astype(y.dtype), y.ctx))
        return y


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cg_linear_operator_and_callback_precond(pkg, orc, ctx, dtype):
    """test/cg.jl:71-85: Af = LinearMap(A); cg(Af, rhs) and cg(Af, rhs; Pl = P) -- here bit for bit against the oracle
    (the dot(u, c) of a callback operator uses the vector tree shape instead of the SpMV epilogue's)"""
    A = orc.laplace(10, 3).astype(dtype)
    b = orc.hashed_rhs(A.n).astype(dtype)
    dA, db = upload(pkg, A), pkg.HipVector.from_numpy(b)
    calls = {"mul": 0}

    def mul(y, x):
        calls["mul"] += 1
        pkg.mul_(y, dA, x)
    Af = pkg.LinearOperator(A.n, dtype, mul)
    W, L = ctx.reduce_shape(dtype)
    x, ch = pkg.cg(Af, db, log=True)
    xo, ho = orc.cg(A, b, mode="tree", shape=(W, L, W, L))
    assert ch.iters == ho["iters"] == calls["mul"] and ch.isconverged
    assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    # a starting guess goes through the callback once more (src/cg.jl:136-137)
    x0 = np.random.default_rng(1).standard_normal(A.n).astype(dtype)
    x1, h1 = pkg.cg_(pkg.HipVector.from_numpy(x0), Af, db, log=True)
    xo1, ho1 = orc.cg(A, b, x0, mode="tree", shape=(W, L, W, L))
    assert h1.mvps == ho1["mvps"] and np.array_equal(h1["resnorm"], ho1["resnorm"]) and np.array_equal(x1.to_numpy(), xo1)
    # Pl through ldiv_ only: same bits as the fused diagonal path, on the CSR operator and on the callback operator
    d = A.to_scipy().diagonal().astype(dtype) * (1 + 0.1 * np.arange(A.n) % 3).astype(dtype)
    P = PlainLdiv(pkg, pkg.HipVector.from_numpy(d))
    xj, hj = pkg.cg(dA, db, Pl=pkg.JacobiPrec(pkg.HipVector.from_numpy(d)), log=True)
    xp, hp = pkg.cg(dA, db, Pl=P, log=True)
    assert P.calls == hp.iters and np.array_equal(hp["resnorm"], hj["resnorm"]) and np.array_equal(xp.to_numpy(), xj.to_numpy())
    xq, hq = pkg.cg(Af, db, Pl=P, log=True)
    xoq, hoq = orc.cg(A, b, jacobi_diag=d, mode="tree", shape=(W, L, W, L))
    assert np.array_equal(hq["resnorm"], hoq["resnorm"]) and np.array_equal(xq.to_numpy(), xoq)
    # batched stepping: callbacks are issued per enqueued step, the device-side stopping test still ends the batch
    it = pkg.cg_iterator_(pkg.zerox(Af, db), Af, db, initially_zero=True)
    got = it.iterate_many(0, ho["iters"] + 10)
    assert np.array_equal(got, ho["resnorm"])


def test_cg_callback_exception_surfaces(pkg, orc, ctx):
    A = orc.laplace(6, 3)
    dA, db = upload(pkg, A), pkg.HipVector.from_numpy(orc.hashed_rhs(A.n))

    def bad(y, x):
        raise RuntimeError("operator exploded")
    with pytest.raises(RuntimeError, match="operator exploded"):
        pkg.cg_(pkg.HipVector.from_numpy(np.ones(A.n)), pkg.LinearOperator(A.n, np.float64, bad), db)
    with pytest.raises(pkg.MikError) as e:
        pkg.cg(dA, db, Pl=object())
    assert e.value.code == 5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gmres_exact_left_and_right_preconditioner(pkg, orc, ctx, dtype):
    """test/gmres.jl:28-35: Pl = lu(A) (then Pr) with maxiter = 1, restart = 1 converges"""
    rng = np.random.default_rng(1234321)
    n = 10
    A = (rng.random((n, n)) + np.eye(n)).astype(dtype)
    b = rng.random(n).astype(dtype)
    reltol = float(np.sqrt(np.finfo(dtype).eps))
    dA, db = upload(pkg, orc.CSC.from_dense(A)), pkg.HipVector.from_numpy(b)
    F = DenseSolve(A.astype(np.float64))
    x, h = pkg.gmres(dA, db, Pl=F, maxiter=1, restart=1, reltol=reltol, log=True)
    assert h.isconverged and h.iters == 1
    r = scipy.linalg.lu_solve(F.lu, A.astype(np.float64) @ x.to_numpy() - b)
    assert np.linalg.norm(r) / np.linalg.norm(b) <= reltol
    x, h = pkg.gmres(dA, db, Pl=pkg.Identity(), Pr=F, maxiter=1, restart=1, reltol=reltol, log=True)
    assert h.isconverged
    assert np.linalg.norm(A.astype(np.float64) @ x.to_numpy() - b) / np.linalg.norm(b) <= reltol
    # residual non-increasing with restarts (test/gmres.jl:23-25) through the callback operator
    Af = pkg.LinearOperator(n, dtype, lambda y, v: pkg.mul_(y, dA, v))
    x, h = pkg.gmres(Af, db, log=True, restart=3, maxiter=10, reltol=reltol)
    assert np.all(np.diff(h["resnorm"]) <= 0.0)
    xo, ho = orc.gmres(orc.CSC.from_dense(A), b, restart=3, maxiter=10, reltol=reltol, mode="tree", shape=ctx.reduce_shape(dtype))
    assert np.array_equal(h["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo) and h.mvps == ho["mvps"]


def test_gmres_linear_operator_defined_as_a_function(pkg, ctx):
    """test/gmres.jl:59-66: A = LinearMap(cumsum!, 100); x = gmres(A, b; reltol = 1e-5, maxiter = 2000)"""
    import torch
    n = 100
    rng = np.random.default_rng(7)
    b = rng.random(n)

    def cumsum_(y, x):                                   # the operator is a torch op on the same device buffers
        ctx.synchronize()
        tx = torch.from_numpy(x.to_numpy()).cuda()
        y.copyto_(pkg.HipVector.from_numpy(torch.cumsum(tx, 0).cpu().numpy(), ctx))
    A = pkg.LinearOperator(n, np.float64, cumsum_)
    x = pkg.gmres(A, pkg.HipVector.from_numpy(b), reltol=1e-5, maxiter=2000)
    assert np.linalg.norm(np.cumsum(x.to_numpy()) - b) / np.linalg.norm(b) <= 1e-5


@pytest.mark.parametrize("orth", ["mgs", "cgs", "dgks"])
def test_gmres_callback_preconditioners_equal_the_fused_diagonal_path(pkg, orc, ctx, orth):
    A, b = orc.advdiff(9, 300.0)
    dA, db = upload(pkg, A), pkg.HipVector.from_numpy(b)
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    d = np.abs(A.to_scipy().diagonal()) ** 0.5
    dv = pkg.HipVector.from_numpy(d)
    x0, h0 = pkg.gmres(dA, db, Pl=pkg.JacobiPrec(dv), Pr=pkg.JacobiPrec(dv), restart=8, maxiter=60, log=True, orth_meth=M)
    Pl, Pr = PlainLdiv(pkg, dv), PlainLdiv(pkg, dv)
    x1, h1 = pkg.gmres(dA, db, Pl=Pl, Pr=Pr, restart=8, maxiter=60, log=True, orth_meth=M)
    assert Pl.calls > 0 and Pr.calls > 0
    assert np.array_equal(h0["resnorm"], h1["resnorm"]) and np.array_equal(x0.to_numpy(), x1.to_numpy()) and h0.mvps == h1.mvps
