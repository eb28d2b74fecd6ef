"""Power method (src/simple.jl): the oracle against what test/simple_eigensolvers.jl checks; the Python mirror on the host double against the C oracle
bit for bit (CPU); the device path (one L1 call per statement) against the oracle's TREE mode, and inverse iteration through a LinearOperator (GPU)."""
import numpy as np
import pytest


def spd(rng, n, dtype):
    A = rng.random((n, n)) + np.eye(n)                                       # test/simple_eigensolvers.jl:16-17
    return (A.T @ A).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_power_iteration(orc, dtype):
    rng = np.random.default_rng(1234321)
    n = 10
    A = spd(rng, n, dtype)
    lams = np.linalg.eigvalsh(A.astype(np.float64))
    tol = n ** 2 * np.linalg.cond(A.astype(np.float64)) * np.finfo(dtype).eps            # :20
    x0 = rng.random(n).astype(dtype)
    x0 = x0 / np.linalg.norm(x0)
    lam, x, h = orc.powm(orc.CSC.from_dense(A), x0, tol=tol, maxiter=10 * n)
    assert abs(lam - lams[-1]) <= np.sqrt(np.finfo(dtype).eps) * lams[-1]               # λs[end] ≈ λ  :27
    assert np.linalg.norm(A @ x - lam * x) <= tol and h["isconverged"]                   # :28
    assert h["iters"] == h["mvps"] == len(h["resnorm"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_python_mirror_equals_the_c_oracle_on_a_host_double(pkg, orc, monkeypatch, dtype):
    import scipy.sparse as sp
    from importlib import import_module
    from host_double import FakeOperator, FakeVector, patch
    api = import_module(pkg.__name__ + ".extras")
    patch(monkeypatch, api, orc)
    rng = np.random.default_rng(5)
    n = 30
    A = spd(rng, n, dtype)
    x0 = rng.random(n).astype(dtype)
    x0 = (x0 / np.linalg.norm(x0)).astype(dtype)
    lo, xo, ho = orc.powm(orc.CSC.from_dense(A), x0, tol=1e-5, maxiter=25)
    lam, x, ch = api.powm_(FakeOperator(orc, sp.csc_matrix(A)), FakeVector(x0.copy()), tol=1e-5, maxiter=25, log=True)
    assert ch.iters == ho["iters"] > 5 and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"]
    assert lam == lo and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_powm_device_bit_exact_and_inverse_iteration(pkg, orc, ctx, dtype):
    import scipy.sparse as sp
    L0 = orc.laplace(12, 3)
    A = L0.astype(dtype)
    n = A.n
    rng = np.random.default_rng(6)
    x0 = rng.random(n).astype(dtype)
    x0 = (x0 / np.linalg.norm(x0)).astype(dtype)
    lo, xo, ho = orc.powm(A, x0, tol=1e-3, maxiter=80, mode="tree", shape=ctx.reduce_shape(dtype))
    dA = pkg.HipCSR(n, n, A.colptr, A.rowval, A.nzval)
    lam, x, ch = pkg.extras.powm_(dA, pkg.HipVector.from_numpy(x0), tol=1e-3, maxiter=80, log=True)
    assert ch.iters == ho["iters"] > 20 and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"]
    assert lam == lo and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    assert 11.0 < float(lam) < 12.0                                          # the largest eigenvalue of the 7-point Laplacian stays below 12
    if dtype == np.float64:
        # inverse iteration (src/simple.jl:145-185): B has the action of inv(A - σI) -- here cg on the shifted operator, as a LinearOperator
        sigma = 0.1
        S = (L0.to_scipy() - sigma * sp.identity(n)).tocsc()
        dS = pkg.HipCSR.from_scipy(S)

        def solve(y, v):
            y.fill_(0)
            pkg.cg_(y, dS, v, reltol=1e-12, maxiter=500)
        B = pkg.LinearOperator(n, np.float64, solve, dA.ctx)
        lam, x = pkg.extras.invpowm_(B, pkg.HipVector.from_numpy(x0.astype(np.float64)), shift=sigma, tol=1e-6, maxiter=60)
        smallest = 6 - 6 * np.cos(np.pi / 13)                                # eigenvalues of the 12^3 Dirichlet Laplacian: sum of 2 - 2 cos(k pi / 13)
        assert abs(float(lam) - smallest) <= 1e-6 * smallest
        xv = x.to_numpy()
        assert np.linalg.norm(L0.to_scipy() @ xv - float(lam) * xv) <= 1e-5
