"""BASELINE.json configs[4] stand-in: fp32 gmres(restart=50) on an irregular CSR operator (row lengths from 5 to
thousands) -- the load-imbalance / long-row case of the SpMV.  GPU box only."""
import numpy as np
import pytest

from conftest import KN
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def long_row_shape(orc, ctx):
    """rows longer than mik_spmv_long_row() use the wave-shaped row sum; the oracle's spmv mirrors it"""
    orc.set_long_row(ctx.spmv_long_row(), ctx.spmv_long_segment(), ctx.spmv_long_group())
    yield
    orc.set_long_row(0)


def as_oracle_csc(orc, n, rowptr, colidx, val):
    M = sp.csr_matrix((val, colidx, rowptr), shape=(n, n)).tocsc()
    M.sort_indices()
    return orc.CSC(n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.copy(), 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_irregular_spmv_bit_exact(pkg, orc, ctx, dtype):
    n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(20000, dtype)
    lens = np.diff(rowptr)
    assert lens.min() >= 6 and lens.max() > 2048          # rows longer than one LDS tile are present
    A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
    x = np.random.default_rng(2).standard_normal(n).astype(dtype)
    y = (A @ pkg.HipVector.from_numpy(x)).to_numpy()
    Ao = as_oracle_csc(orc, n, rowptr, colidx, val)
    yo = orc.spmv(Ao, x)
    assert np.array_equal(y, yo)
    # rows up to the threshold keep the reference's strictly sequential order; longer rows differ from it by rounding only
    orc.set_long_row(0)
    yseq = orc.spmv(Ao, x)
    short = lens <= ctx.spmv_long_row()
    assert np.array_equal(y[short], yseq[short]) and not np.array_equal(y[~short], yseq[~short])
    np.testing.assert_allclose(y, yseq, rtol=1e-4 if dtype == np.float32 else 1e-12, atol=1e-4 if dtype == np.float32 else 1e-12)
    # ... and both orders sit inside the forward-error bound of the reference's own loop (len * eps * sum |a_ij x_j|, Higham 3.1)
    # around the sum in extended precision; over the long rows the wave shape is the more accurate of the two
    S = sp.csr_matrix((val.astype(np.longdouble).astype(np.float64), colidx, rowptr), shape=(n, n))
    exact = np.array([np.sum(val[rowptr[i]:rowptr[i + 1]].astype(np.longdouble) * x[colidx[rowptr[i]:rowptr[i + 1]]].astype(np.longdouble))
                      for i in np.flatnonzero(~short)])
    mag = np.asarray(abs(S) @ np.abs(x.astype(np.float64)))[~short]
    bound = lens[~short] * np.finfo(dtype).eps * mag
    err_dev = np.abs(y[~short].astype(np.longdouble) - exact).astype(np.float64)
    err_seq = np.abs(yseq[~short].astype(np.longdouble) - exact).astype(np.float64)
    assert np.all(err_dev <= bound) and np.all(err_seq <= bound)
    assert np.sqrt(np.mean((err_dev / mag) ** 2)) <= np.sqrt(np.mean((err_seq / mag) ** 2))


def test_irregular_gmres_fp32_restart50(pkg, orc, ctx):
    n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(20000, np.float32)
    A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
    b = pkg.fixtures.hashed_rhs(n, dtype=np.float32)
    x, ch = pkg.gmres(A, pkg.HipVector.from_numpy(b), restart=50, log=True)
    Ao = as_oracle_csc(orc, n, rowptr, colidx, val)
    xo, ho = orc.gmres(Ao, b, restart=50, mode="tree", shape=ctx.reduce_shape(np.float32))
    assert ch.iters == ho["iters"] and ch.mvps == ho["mvps"] and ch.isconverged and ho["isconverged"]
    assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
    S = sp.csr_matrix((val.astype(np.float64), colidx, rowptr), shape=(n, n))
    assert np.linalg.norm(S @ x.to_numpy().astype(np.float64) - b) / np.linalg.norm(b) <= 5e-4      # sqrt(eps_f32) = 3.5e-4
    assert np.all(np.diff(ch["resnorm"]) <= 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("segment", [0, 300])
def test_cut_rows_segment_sums(pkg, orc, ctx, dtype, segment):
    """rows longer than mik_spmv_long_segment() are cut into segments summed by separate waves; segment sums are added in
    segment order whatever the order the waves finish in: lengths around the segment boundaries (a last segment of 1 or 4
    entries, exactly one / two segments), both CSR kernels, repeated launches (the tickets reset themselves)"""
    L = pkg.lib()
    L.mik_set_tuning(KN.LONG_SEGMENT, segment)
    try:
        seg = ctx.spmv_long_segment()
        assert seg == (segment or 1024)
        orc.set_long_row(ctx.spmv_long_row(), seg, ctx.spmv_long_group())
        rng = np.random.default_rng(9)
        n = 12000
        lens = rng.integers(3, 40, size=n)
        for i, l in enumerate((seg - 1, seg, seg + 1, 2 * seg, 2 * seg + 4, 5 * seg + seg // 2, 65, 64, 256, 257, 10 * seg)):
            lens[37 + 501 * i] = min(l, n)
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]).astype(np.int64)
        val = rng.standard_normal(cols.size).astype(dtype)
        Ao = as_oracle_csc(orc, n, rowptr, cols, val)
        x = rng.standard_normal(n).astype(dtype)
        want = orc.spmv(Ao, x)
        for variant in (2, 1):
            L.mik_set_tuning(KN.CSR_KERNEL, variant)
            A = pkg.HipCSR(n, n, rowptr, cols, val, index_base=0, is_csc=False)
            dx = pkg.HipVector.from_numpy(x)
            for _ in range(3):
                assert np.array_equal((A @ dx).to_numpy(), want), (variant, segment)
            b = pkg.HipVector.from_numpy(orc.hashed_rhs(n).astype(dtype))
            xs, ch = pkg.cg(A, b, log=True, maxiter=3)                   # fused-dot launches (matrix not SPD: bits only)
            xo, ho = orc.cg(Ao, orc.hashed_rhs(n).astype(dtype), maxiter=3, mode="tree", shape=ctx.cg_shape(dtype))
            assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(xs.to_numpy(), xo)
    finally:
        L.mik_set_tuning(KN.LONG_SEGMENT, 0)
        L.mik_set_tuning(KN.CSR_KERNEL, 0)


def banded_irregular(n, dtype, halfwidth, seed=3, long_every=0, touch_last_column=False):
    """rows of 3 ... 60 entries with columns inside [row - halfwidth, row + halfwidth] clipped to the matrix (no wrap-around:
    the first and last row-blocks have one-sided bands), optionally a few rows beyond the long-row threshold"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(3, 60, size=n)
    if long_every:
        lens[5::long_every] = rng.integers(300, 3000, size=lens[5::long_every].size)
    cols = []
    for r, l in enumerate(lens):
        w = max(halfwidth, 2 * l)
        lo, hi = max(0, r - w), min(n - 1, r + w)
        l = min(l, hi - lo + 1)
        lens[r] = l
        cr = np.sort(rng.choice(np.arange(lo, hi + 1), size=l, replace=False))
        if touch_last_column and r >= n - 600 and cr[-1] != n - 1:      # the last row-blocks all reference x[n - 1]
            cr[-1] = n - 1
        cols.append(cr)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cols = np.concatenate(cols).astype(np.int64)
    return rowptr, cols, rng.standard_normal(cols.size).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("long_every,n", [(0, 9000), (997, 9000), (0, 9001), (997, 9003)])
def test_x_window_in_lds_bit_exact(pkg, orc, ctx, dtype, long_every, n):
    """irregular rows inside a band -- the product-tile kernel serves x from an LDS window per 256-row block
    (k_spmv_rowblock XWIN, csr_build_xwin): same products, same order, same bits as the oracle, with the windows (default), with
    the windows and the by-length row permutation of a block's threads (RPERM) built but not used (development knob MIK_KNOB_LAYOUTS bit 64) and
    never built (bit 32); plain SpMV (long rows merged into the launch)
    and the fused-dot launches of a CG step; one-sided bands at both ends of the matrix (the last window slides down).
    n = 9001 / 9003: x is not a whole number of 16-byte groups, so the slid-down window of the last row-blocks
    cannot reach x[n - 1] from an aligned start -- those blocks must gather from memory (mik_xwin_plan), and every one of them
    references the last column."""
    rowptr, cols, val = banded_irregular(n, dtype, 700, long_every=long_every, touch_last_column=n != 9000)
    Ao = as_oracle_csc(orc, n, rowptr, cols, val)
    x = np.random.default_rng(8).standard_normal(n).astype(dtype)
    want = orc.spmv(Ao, x)
    b = orc.hashed_rhs(n).astype(dtype)
    xo, ho = orc.cg(Ao, b, maxiter=3, mode="tree", shape=ctx.cg_shape(dtype))
    L = pkg.lib()
    # MIK_KNOB_LAYOUTS bit 32: neither windows nor the row permutation are built -- without long rows the operator is then back on the LDS-DMA tile
    for knob, kern in ((0, "k_spmv_rowblock+xwin"), (KN.XWIN_UNUSED, "k_spmv_rowblock"), (KN.NO_XWIN, "k_spmv_rowblock" if long_every else "k_spmv_rowgather")):
        L.mik_set_tuning(KN.LAYOUTS, knob)
        try:
            A = pkg.HipCSR(n, n, rowptr, cols, val, index_base=0, is_csc=False)
            assert A.spmv_kernel() == kern, (A.spmv_kernel(), A.layout())
            dx = pkg.HipVector.from_numpy(x)
            for _ in range(2):
                assert np.array_equal((A @ dx).to_numpy(), want), knob
            xs, ch = pkg.cg(A, pkg.HipVector.from_numpy(b), log=True, maxiter=3)          # fused-dot launches (matrix not SPD: bits only)
            assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(xs.to_numpy(), xo)
        finally:
            L.mik_set_tuning(KN.LAYOUTS, 0)


def test_x_window_blocks_that_wrap_around_gather_from_memory(pkg, orc, ctx):
    """fixtures.irregular_matrix(bandwidth > 0) wraps its bands around the matrix: the first and last row-blocks span all of x,
    carry no window (win_lo = -1) and gather from memory inside the same launch; every other block reads its LDS window"""
    n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(40000, np.float32, bandwidth=600)
    A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
    assert A.spmv_kernel() == "k_spmv_rowblock+xwin"
    x = np.random.default_rng(2).standard_normal(n).astype(np.float32)
    Ao = as_oracle_csc(orc, n, rowptr, colidx, val)
    assert np.array_equal((A @ pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(Ao, x))
    b = pkg.fixtures.hashed_rhs(n, dtype=np.float32)
    xg, ch = pkg.gmres(A, pkg.HipVector.from_numpy(b), restart=50, log=True, maxiter=60)
    xo, ho = orc.gmres(Ao, b, restart=50, maxiter=60, mode="tree", shape=ctx.reduce_shape(np.float32))
    assert ch.iters == ho["iters"] and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(xg.to_numpy(), xo)
