"""QMR (src/qmr.jl): the two-sided Lanczos process on A and adjoint(A).  CPU: the oracle against what test/qmr.jl checks; the Python mirror on a
numpy stand-in for the device types against the C oracle bit for bit.  GPU: the device path (every vector statement one L1 call, adjoint(A) = the
SparseMatrixCSC's own arrays read as CSR) bit for bit against the oracle's TREE mode."""
import numpy as np
import pytest


def sprand_plus(rng, n, density, shift, dtype):
    import scipy.sparse as sp
    A = (sp.random(n, n, density=density, random_state=rng, format="csc") + shift * sp.identity(n)).tocsc().astype(dtype)   # test/qmr.jl:27
    A.sort_indices()
    return A


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_qmr_reference_properties(orc, dtype):
    import scipy.sparse as sp
    rng = np.random.default_rng(1234567)
    n = 10
    Ad = (rng.random((n, n)) + n * np.eye(n)).astype(dtype)                  # test/qmr.jl:16-24
    b = rng.random(n).astype(dtype)
    x, h = orc.qmr(sp.csc_matrix(Ad), b)
    assert h["isconverged"] and np.linalg.norm(Ad @ x - b) / np.linalg.norm(b) <= 10 * np.sqrt(np.finfo(dtype).eps)
    A = sprand_plus(rng, n, 0.5, n, dtype)                                   # :26-34
    reltol = float(np.sqrt(np.finfo(dtype).eps))
    x, h = orc.qmr(A, b, reltol=reltol)
    assert h["isconverged"] and (dtype == np.float32 or np.linalg.norm(A @ x - b) / np.linalg.norm(b) <= 2 * reltol)   # (the reference runs this one in Float64 only)
    x, h = orc.qmr(sp.csc_matrix(rng.random((5, 5))), rng.random(5), maxiter=2)                    # :36-40
    assert h["iters"] == 2 and len(h["resnorm"]) == 2
    A3 = np.array([[2, -1, 0], [-1, 2, -1], [0, -1, 2]], dtype)              # :42-66
    b3 = np.ones(3, dtype)
    x0 = np.linalg.solve(A3.astype(np.float64), b3.astype(np.float64)).astype(dtype)
    pert = (10 * np.sqrt(np.finfo(dtype).eps) * np.array([-1.0, 1.0, -1.0])).astype(dtype)
    x, ch = orc.qmr(sp.csc_matrix(A3), b3, x0 + pert)
    assert 2 <= ch["iters"] <= 3
    r0 = float(np.linalg.norm(A3 @ (x0 + pert) - b3))
    x, ch = orc.qmr(sp.csc_matrix(A3), b3, x0 + pert, abstol=2 * r0, reltol=0.0)
    assert ch["iters"] == 0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("start", [False, True])
def test_python_mirror_equals_the_c_oracle_on_a_host_double(pkg, orc, monkeypatch, dtype, start):
    from importlib import import_module
    from host_double import FakeOperator, FakeVector, patch
    api = import_module(pkg.__name__ + ".extras")
    patch(monkeypatch, api, orc)
    monkeypatch.setattr(api, "givens_algorithm", lambda f, g, dt=np.float64: orc.givens(f, g, dt))
    monkeypatch.setattr(api, "zerox", lambda A, b: FakeVector(np.zeros(A.size(2), b.dtype)))
    rng = np.random.default_rng(41)
    n = 40
    S = sprand_plus(rng, n, 0.2, 4, dtype)
    b = rng.standard_normal(n).astype(dtype)
    x0 = rng.standard_normal(n).astype(dtype)
    A = FakeOperator(orc, S)
    xo, ho = orc.qmr(S, b, x0 if start else None, maxiter=60)
    if start:
        x, ch = api.qmr_(FakeVector(x0.copy()), A, FakeVector(b), maxiter=60, log=True)
    else:
        x, ch = api.qmr(A, FakeVector(b), maxiter=60, log=True)
    assert ch.iters == ho["iters"] > 5 and ch.mvps == 0 and ch.isconverged == ho["isconverged"]
    assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,start", [("advdiff", False), ("advdiff", True), ("random", False)])
def test_qmr_device_bit_exact(pkg, orc, ctx, dtype, name, start):
    rng = np.random.default_rng(43)
    if name == "advdiff":
        S = orc.advdiff(10, 200.0)[0].astype(dtype).to_scipy()               # non-symmetric: A and adjoint(A) differ
        b = pkg.fixtures.advection_dominated(10, 200.0)[4].astype(dtype)
    else:
        S = sprand_plus(rng, 500, 0.02, 3, dtype)
        b = rng.standard_normal(500).astype(dtype)
    n = S.shape[0]
    x0 = rng.standard_normal(n).astype(dtype) if start else None
    xo, ho = orc.qmr(S, b, x0, maxiter=120, mode="tree", shape=ctx.reduce_shape(dtype))
    dA = pkg.extras.with_adjoint_from_scipy(S)
    for fused in (True, False):                  # the fused sweeps (mik_axpy2_dot, mik_scal2, mik_qmr_update) and one L1 call per statement: same bits
        if start:
            x, ch = pkg.extras.qmr_(pkg.HipVector.from_numpy(x0), dA, pkg.HipVector.from_numpy(b), maxiter=120, log=True, fused=fused)
        else:
            x, ch = pkg.extras.qmr(dA, pkg.HipVector.from_numpy(b), maxiter=120, log=True, fused=fused)
        assert ch.iters == ho["iters"] > 10 and ch.isconverged == ho["isconverged"], fused
        assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo), fused
    if ho["isconverged"] and dtype == np.float64:                            # resnorm is the QUASI-residual: the true one is within sqrt(k + 1) of it at best
        assert np.linalg.norm(S @ xo - b) / np.linalg.norm(b) <= 1e-3
