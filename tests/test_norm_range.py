"""Over-/underflow-safe norms: the reference's norm() is BLAS nrm2 / generic_norm2, so a badly scaled system
must not "converge" on a residual whose sum of squares underflowed to 0 (or blow up on one that overflowed).
include/mik.h "Norms" defines the semantics; the oracle mirrors them (oracle/orc_impl.inc: safe_nrm_)."""
import numpy as np
import pytest

SCALES = {np.float64: (1e-200, 1e-160, 1e160, 1e200), np.float32: (1e-30, 1e-22, 1e20, 1e30)}


def ref_norm(v):
    v = np.asarray(v, np.float64)
    m = np.max(np.abs(v))
    return 0.0 if m == 0 else float(m * np.sqrt(np.sum((v / m) ** 2)))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", ["seq", "pair", "tree", "blas"])
def test_oracle_norm_is_scale_safe(orc, dtype, mode):
    rng = np.random.default_rng(4)
    W, L = (2, 2) if dtype == np.float64 else (4, 2)
    for n in (7, 31, 32, 1000, 5000):
        v = rng.standard_normal(n)
        for s in (1.0,) + SCALES[dtype]:
            x = (v * s).astype(dtype)
            got = float(orc.nrm2(x, mode=mode, W=W, L=L))
            assert got == pytest.approx(ref_norm(x), rel=2e-15 * n if dtype == np.float64 else 2e-7 * max(n, 8) ** 0.5)
    assert orc.nrm2(np.zeros(100, dtype), mode=mode, W=W, L=L) == 0.0
    bad = np.ones(100, dtype)
    bad[17] = np.inf
    assert np.isinf(orc.nrm2(bad, mode=mode, W=W, L=L))
    bad[3] = np.nan
    assert np.isnan(orc.nrm2(bad, mode=mode, W=W, L=L))


def test_oracle_unscaled_norm_is_unchanged(orc):
    """inside the safe range the value is exactly sqrt(tree sum of squares): the bit-exact goldens stay valid"""
    v = np.random.default_rng(5).standard_normal(3000)
    assert orc.nrm2(v, mode="tree", W=2, L=2) == np.sqrt(orc.dot(v, v, mode="tree", W=2, L=2))
    assert orc.nrm2(v, mode="seq") == np.sqrt(orc.dot(v, v, mode="seq"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_nrm2_scaled_vectors(pkg, orc, ctx, dtype):
    rng = np.random.default_rng(6)
    W, L = ctx.reduce_shape(dtype)
    for n in (5, 1000, 70001):
        v = rng.standard_normal(n)
        for s in (1.0,) + SCALES[dtype]:
            x = (v * s).astype(dtype)
            got = pkg.norm(pkg.HipVector.from_numpy(x))
            assert got == orc.nrm2(x, mode="tree", W=W, L=L)                       # same documented semantics, same bits
            assert float(got) == pytest.approx(ref_norm(x), rel=1e-13 if dtype == np.float64 else 2e-5)
    assert pkg.norm(pkg.HipVector.from_numpy(np.zeros(300, dtype))) == 0.0
    bad = np.ones(300, dtype)
    bad[100] = np.inf
    assert np.isinf(pkg.norm(pkg.HipVector.from_numpy(bad)))
    bad[7] = np.nan
    assert np.isnan(pkg.norm(pkg.HipVector.from_numpy(bad)))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,scale", [(np.float32, 1e-14), (np.float32, 1e16), (np.float64, 1e-140), (np.float64, 1e140)])
def test_cg_on_a_badly_scaled_rhs(pkg, orc, ctx, dtype, scale):
    """|r|^2 leaves the safe range in every step (the scaled norm runs each time): the solve takes the iterations of the
    unscaled system and its history is the unscaled one times the scale -- instead of `residual = 0 <= tol` at once"""
    A = orc.laplace(9, 3).astype(dtype)
    b1 = orc.hashed_rhs(A.n).astype(dtype)
    b = (b1.astype(np.float64) * scale).astype(dtype)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    x1, h1 = pkg.cg(dA, pkg.HipVector.from_numpy(b1), log=True)
    x, h = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True)
    xo, ho = orc.cg(A, b, mode="tree", shape=ctx.cg_shape(dtype))
    assert h.iters == ho["iters"] > 5 and h.isconverged
    assert np.array_equal(h["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    assert abs(h.iters - h1.iters) <= 1
    m = min(h.iters, h1.iters) - 1
    np.testing.assert_allclose(np.asarray(h["resnorm"][:m]) / scale, h1["resnorm"][:m], rtol=2e-2 if dtype == np.float32 else 1e-6)
    # batched stepping goes through the same host fix-up
    it = pkg.cg_iterator_(pkg.zerox(dA, pkg.HipVector.from_numpy(b)), dA, pkg.HipVector.from_numpy(b), initially_zero=True)
    assert it.residual == ho["res0"] and it.residual > it.tol > 0
    got = it.iterate_many(0, 7)
    assert np.array_equal(got, ho["resnorm"][:7])


@pytest.mark.gpu
@pytest.mark.parametrize("orth", ["mgs", "cgs", "dgks"])
def test_gmres_on_a_badly_scaled_system(pkg, orc, ctx, orth):
    """operator and rhs scaled by 1e-160: beta and every Gram-Schmidt norm take the scaled path"""
    A, b = orc.advdiff(8, 200.0)
    s = 1e-160
    As = orc.CSC(A.n, A.colptr, A.rowval, A.nzval * s, A.index_base)
    bs = b * s
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    dA = pkg.HipCSR(A.n, A.n, As.colptr, As.rowval, As.nzval)
    x, h = pkg.gmres(dA, pkg.HipVector.from_numpy(bs), restart=12, log=True, orth_meth=M)
    xo, ho = orc.gmres(As, bs, restart=12, orth_meth=orth, mode="tree", shape=ctx.reduce_shape(np.float64))
    assert h.iters == ho["iters"] and h.isconverged and np.array_equal(h["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    x1, h1 = pkg.gmres(pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval), pkg.HipVector.from_numpy(b), restart=12, log=True, orth_meth=M)
    assert abs(h.iters - h1.iters) <= 2
    np.testing.assert_allclose(x.to_numpy(), x1.to_numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1.0, 1e-140])
@pytest.mark.parametrize("knob9", [0, 1])
@pytest.mark.parametrize("batch", [1, 4])
def test_cg_stepping_at_128_cubed_through_freezes_and_maxiter(pkg, orc, ctx, scale, knob9, batch):
    """The production-size step (2 M rows: spread finalisers, look-ahead, x riding on the next sweep over u) stepped one by one and in
    batches, with look-ahead on and off (MIK_KNOB_NO_LOOKAHEAD), on a right-hand side whose |r|^2 leaves the safe range in every step (each step frozen
    and finished by the host: x must not be updated twice, whichever kernel applied it), into maxiter and beyond: bit for bit the
    oracle's history and x."""
    A = orc.laplace(128, 3)
    b = orc.hashed_rhs(A.n) * scale
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    steps = 8 if scale != 1.0 else 12
    ctx.set_tuning(7, knob9)        # MIK_KNOB_NO_LOOKAHEAD
    db = pkg.HipVector.from_numpy(b)
    x = pkg.zerox(dA, db)
    it = pkg.cg_iterator_(x, dA, db, initially_zero=True, reltol=0.0, maxiter=steps + 3)
    hist = []
    k = 0
    while k < steps:
        got = it.iterate_many(k, min(batch, steps - k))
        assert got.size == min(batch, steps - k)
        hist.extend(got.tolist())
        k += got.size
    tail = it.iterate_many(k, 10)                            # runs into maxiter; asked again, nothing may move
    assert tail.size == 3 and it.iterate_many(k + 3, 5).size == 0
    xo, ho = orc.cg(A, b, maxiter=steps + 3, reltol=0.0, mode="tree", shape=ctx.cg_shape(np.float64))
    assert np.array_equal(np.array(hist + tail.tolist()), np.asarray(ho["resnorm"])) and np.array_equal(x.to_numpy(), xo)
