"""Preconditioned paths: PCG / GMRES with diagonal Pl, Pr.

The reference's tests use exact factorizations as preconditioners (test/gmres.jl:27-35,48-56: LU as Pl or
Pr => converged after one iteration; test/cg.jl:43-47: Cholesky => niters <= 2).  On the device path a
preconditioner is a diagonal ``JacobiPrec`` (ldiv!(y, P, x) = y .= x ./ P.diagonal, test/cg.jl:14-18), so
the same properties are checked on a diagonal operator, where the Jacobi preconditioner IS exact; the
nonsymmetric advection-diffusion operator then checks bit-level parity of all three expand! methods
(src/gmres.jl:285-304) against the oracle.
"""
import numpy as np
import pytest
import scipy.sparse as sp


def diag_problem(orc, n=50, seed=4):
    rng = np.random.default_rng(seed)
    d = rng.random(n) + 0.5
    A = orc.CSC.from_scipy(sp.diags(d).tocsc())
    b = rng.random(n)
    return A, d, b


# ---- oracle (CPU) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("side", ["Pl", "Pr"])
def test_oracle_gmres_exact_diagonal_preconditioner_converges_in_one_iteration(orc, side):
    A, d, b = diag_problem(orc)
    kw = {"pl_diag": d} if side == "Pl" else {"pr_diag": d}
    x, h = orc.gmres(A, b, maxiter=1, restart=1, **kw)
    assert h["isconverged"] and h["iters"] == 1                                 # test/gmres.jl:29,34
    r = (A.to_scipy() @ x - b)
    if side == "Pl":
        assert np.linalg.norm(r / d) / np.linalg.norm(b) <= np.sqrt(np.finfo(float).eps)   # :30
    else:
        assert np.linalg.norm(r) / np.linalg.norm(b) <= np.sqrt(np.finfo(float).eps)       # :35


def test_oracle_gmres_preconditioned_solves_advdiff(orc):
    A, b = orc.advdiff(8, 200.0)
    S = A.to_scipy()
    d = S.diagonal()
    for kw in ({"pl_diag": d}, {"pr_diag": d}, {"pl_diag": d, "pr_diag": np.abs(d) ** 0.5}):
        x, h = orc.gmres(A, b, restart=15, **kw)
        assert h["isconverged"] and np.all(np.diff(h["resnorm"]) <= 0)
        pl = kw.get("pl_diag", np.ones_like(d))
        assert np.linalg.norm((S @ x - b) / pl) / np.linalg.norm(b / pl) <= 1e-7


# ---- device ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("side", ["Pl", "Pr"])
def test_gmres_exact_diagonal_preconditioner(pkg, orc, ctx, side):
    A, d, b = diag_problem(orc)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    P = pkg.JacobiPrec(pkg.HipVector.from_numpy(d))
    kw = {"Pl": P} if side == "Pl" else {"Pl": pkg.Identity(), "Pr": P}
    x, ch = pkg.gmres(dA, pkg.HipVector.from_numpy(b), maxiter=1, restart=1, log=True, **kw)
    assert ch.isconverged and ch.iters == 1
    r = A.to_scipy() @ x.to_numpy() - b
    scale = d if side == "Pl" else 1.0
    assert np.linalg.norm(r / scale) / np.linalg.norm(b) <= np.sqrt(np.finfo(float).eps)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("which", ["Pl", "Pr", "both"])
@pytest.mark.parametrize("orth", ["mgs", "dgks"])
def test_gmres_preconditioned_bit_exact_vs_oracle(pkg, orc, ctx, which, orth, dtype):
    A, b = orc.advdiff(10, 300.0)
    A, b = A.astype(dtype), b.astype(dtype)
    d = A.to_scipy().diagonal().astype(dtype)
    d2 = (np.abs(d) ** 0.5).astype(dtype)
    okw, dkw = {}, {}
    if which in ("Pl", "both"):
        okw["pl_diag"] = d
        dkw["Pl"] = pkg.JacobiPrec(pkg.HipVector.from_numpy(d))
    if which in ("Pr", "both"):
        okw["pr_diag"] = d2
        dkw["Pr"] = pkg.JacobiPrec(pkg.HipVector.from_numpy(d2))
    M = {"mgs": pkg.ModifiedGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    x0 = np.random.default_rng(1).standard_normal(A.n).astype(dtype)
    x, ch = pkg.gmres_(pkg.HipVector.from_numpy(x0), dA, pkg.HipVector.from_numpy(b), restart=12, log=True, orth_meth=M, **dkw)
    xo, ho = orc.gmres(A, b, x0, restart=12, orth_meth=orth, mode="tree", shape=ctx.reduce_shape(dtype), **okw)
    assert ch.iters == ho["iters"] and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"]
    assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)


@pytest.mark.gpu
def test_gmres_rejects_unsupported_preconditioner_type(pkg, orc, ctx):
    A, d, b = diag_problem(orc)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    with pytest.raises(pkg.MikError) as e:
        pkg.gmres(dA, pkg.HipVector.from_numpy(b), Pl=object())
    assert e.value.code == 5
