"""The operator's device layouts (mik_csr_layout): CSR row-blocks, jagged slices, per-slice-offset forms, slice-constant forms.
The layout is picked at upload from the sparsity pattern; mul! and every solver must return the same bits in all."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from conftest import KN

# KN.CSR_KERNEL selects the CSR kernel: 0 = tile filled by LDS-DMA + per-row gather (k_spmv_rowgather, default), 1 = products
# staged through registers (k_spmv_rowblock)
FORMS = {"csr-rowblock": {KN.LAYOUTS: KN.CSR_ONLY}, "csr-rowblock/products": {KN.LAYOUTS: KN.CSR_ONLY, KN.CSR_KERNEL: 1},
         "jagged-slices": {KN.LAYOUTS: KN.NO_SLICE_OFFSETS | KN.JAGGED_ALWAYS}, "sliced-ell+slice-offsets+row-masks": {KN.LAYOUTS: KN.NO_SLICE_CONSTANT}, "best": {},
         # the kernels of the slice-constant layout: flat loads, buffer loads slot by slot, one row per lane
         "best/flat-loads": {KN.SDIA_KERNEL: 1}, "best/slot-by-slot": {KN.SDIA_KERNEL: 2}, "best/one-row-per-lane": {KN.SDIA_KERNEL: 3}}


def with_knobs(pkg, knobs, fn):
    L = pkg.lib()
    for k, v in knobs.items():
        L.mik_set_tuning(k, v)
    try:
        return fn()
    finally:
        for k in knobs:
            L.mik_set_tuning(k, 0)


def upload(pkg, A):
    return pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("case", ["laplace3d", "laplace2d", "advdiff", "banded_wide"])
def test_spmv_and_cg_identical_in_every_layout(pkg, orc, ctx, case, dtype):
    if case == "laplace3d":
        A = orc.laplace(11, 3)
    elif case == "laplace2d":
        A = orc.laplace(37, 2)
    elif case == "advdiff":
        A = orc.advdiff(9, 300.0)[0]
    else:   # 19 diagonals, SPD: slices wider than one 8-entry pass
        n = 3000
        offs = [0] + [o for d in (1, 2, 3, 7, 50, 51, 200, 333, 900) for o in (d, -d)]
        S = sp.diags([np.full(n - abs(o), 40.0 if o == 0 else -1.0 / (1 + abs(o) % 5)) for o in offs], offs, format="csc")
        A = orc.CSC.from_scipy(S)
    A = A.astype(dtype)
    x = np.random.default_rng(1).standard_normal(A.n).astype(dtype)
    want = orc.spmv(A, x)
    b = orc.hashed_rhs(A.n).astype(dtype)
    hist = {}
    for form, knobs in FORMS.items():
        def run():
            dA = upload(pkg, A)
            if form == "sliced-ell+slice-offsets+row-masks" and case == "banded_wide":
                assert dA.layout() == "wide-slice-values+row-masks"         # 19 constant diagonals: > 8 offsets per slice, <= 32
            elif not form.startswith("best"):
                assert dA.layout() == form.split("/")[0]
            elif case != "banded_wide":       # every slice of these constant-coefficient stencils uses <= 8 offsets, one value per slot
                assert dA.layout() == "slice-offsets+slice-values+row-masks"
            else:
                assert dA.layout() == "wide-slice-values+row-masks"
            y = pkg.mul_(pkg.HipVector(A.n, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy()
            xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=60) if case != "advdiff" else pkg.gmres(
                dA, pkg.HipVector.from_numpy(b), log=True, maxiter=40, restart=8)
            return y, ch["resnorm"], xs.to_numpy()
        y, res, xs = with_knobs(pkg, knobs, run)
        assert np.array_equal(y, want), form
        hist[form] = (res, xs)
    ref = hist["csr-rowblock"]
    for form, (res, xs) in hist.items():
        assert np.array_equal(res, ref[0]) and np.array_equal(xs, ref[1]), form


def test_layout_choice_follows_the_pattern(pkg, orc, ctx):
    # uneven row lengths (5-200): the CSR product tile;  even rows with > 255 distinct offsets and no group padding: jagged
    # slices;  short rows whose group padding costs > 10 %: the CSR tile;  finite-element rows (24-54 entries): jagged slices
    n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(20000, np.float32, long_rows=False)
    assert pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False).layout() == "csr-rowblock"
    for dt in (np.float32, np.float64):
        nf, rp, ci, vv = pkg.fixtures.fe_matrix((21, 17), 6, dt)
        dF = pkg.HipCSR(nf, nf, rp, ci, vv, index_base=0, is_csc=False)
        assert dF.layout() == "jagged-slices" and dF.spmv_kernel() == "k_spmv_jds"
        xf = np.random.default_rng(3).standard_normal(nf).astype(dt)
        F = orc.CSC.from_scipy(sp.csr_matrix((vv, ci, rp), shape=(nf, nf)).tocsc())
        assert np.array_equal(pkg.mul_(pkg.HipVector(nf, dt), dF, pkg.HipVector.from_numpy(xf)).to_numpy(), orc.spmv(F, xf))
        bf = orc.hashed_rhs(nf).astype(dt)
        xs, ch = pkg.cg(dF, pkg.HipVector.from_numpy(bf), log=True)           # SPD: cg! converges; dot(u, c) formed inside k_spmv_jds
        xo, ho = orc.cg(F, bf, mode="tree", shape=ctx.cg_shape(dt))
        assert ch.isconverged and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(xs.to_numpy(), xo)
    rng = np.random.default_rng(0)
    n = 4096
    cols = (np.arange(n)[:, None] + rng.integers(-1500, 1500, size=(n, 6))) % n     # 6 entries per row, ~3000 distinct offsets
    S = sp.csr_matrix((rng.standard_normal(6 * n), cols.ravel(), np.arange(0, 6 * n + 1, 6)), shape=(n, n))
    S.sum_duplicates()
    A = orc.CSC.from_scipy(S.tocsc())
    dA = upload(pkg, A)
    assert dA.layout() == "jagged-slices"
    x = rng.standard_normal(n)
    assert np.array_equal(pkg.mul_(pkg.HipVector(n), dA, pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(A, x))
    cols = (np.arange(n)[:, None] + np.sort(rng.choice(3000, size=(n, 3)), axis=1) + 1) % n            # 3 entries per row: 33 % padding in fp64
    S = sp.csr_matrix((rng.standard_normal(3 * n), cols.ravel(), np.arange(0, 3 * n + 1, 3)), shape=(n, n))
    S.sum_duplicates()
    A = orc.CSC.from_scipy(S.tocsc())
    dA = upload(pkg, A)
    assert dA.layout() == "csr-rowblock"
    assert np.array_equal(pkg.mul_(pkg.HipVector(n), dA, pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(A, x))
    # empty rows, a slice of empty rows, an empty last slice
    D = sp.lil_matrix((1000, 1000))
    D[5, 7] = 2.0; D[5, 5] = 1.0; D[700, 3] = -4.0; D[999, 999] = 3.0
    A = orc.CSC.from_scipy(D.tocsc())
    dA = upload(pkg, A)
    x = rng.standard_normal(1000)
    assert np.array_equal(pkg.mul_(pkg.HipVector(1000), dA, pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(A, x))


def test_rectangular_block_with_halo_columns(pkg, orc, ctx, dist):
    """a rank's n_loc x n_ext block (ghost columns behind the owned ones) takes the coded layout too"""
    N, NZ, P = 12, 12, 3
    n = N * N * NZ
    A = orc.laplace(N, 3)
    S = A.to_scipy().tocsr()
    offsets = dist.partition_rows(n, P, align=N * N)
    x = np.random.default_rng(2).standard_normal(n)
    want = orc.spmv(A, x)
    for r in range(P):
        r0, r1 = int(offsets[r]), int(offsets[r + 1])
        blk = S[r0:r1]
        li, plan = dist.localize_block(blk.indptr.astype(np.int64), blk.indices.astype(np.int64), offsets, r)
        dA = pkg.HipCSR(plan.n_loc, plan.n_loc + plan.n_ghost, blk.indptr.astype(np.int64), li, blk.data, index_base=0, is_csc=False)
        assert dA.layout() == "slice-offsets+slice-values+row-masks"
        xe = np.concatenate([x[r0:r1], x[plan.ghost_gids]])
        y = pkg.mul_(pkg.HipVector(plan.n_loc), dA, pkg.HipVector.from_numpy(xe)).to_numpy()
        assert np.array_equal(y, want[r0:r1])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csr_kernel_variants_on_ragged_rows(pkg, orc, ctx, dtype):
    """rows of 0..64 entries (several LDS passes per wave, even and odd lengths, empty rows, a ragged last block):
    every CSR kernel variant and the jagged slices (forced: their lanes idle on such uneven rows) return the oracle's bits,
    with and without the fused dot"""
    rng = np.random.default_rng(5)
    n = 5 * 256 + 77
    lens = rng.integers(0, 65, size=n)
    lens[rng.integers(0, n, 40)] = 0
    lens[300:364] = 64                                    # one wave whose 64 rows are all full
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]).astype(np.int64)
    val = rng.standard_normal(cols.size).astype(dtype)
    S = sp.csr_matrix((val, cols, rowptr), shape=(n, n))
    A = orc.CSC.from_scipy(S.tocsc())
    x = rng.standard_normal(n).astype(dtype)
    want = orc.spmv(A, x)
    b = orc.hashed_rhs(n).astype(dtype)
    ref = None
    for variant in (0, 1, "jagged-slices"):
        def run():
            dA = pkg.HipCSR(n, n, rowptr, cols, val, index_base=0, is_csc=False)
            assert dA.layout() == ("csr-rowblock" if variant != "jagged-slices" else variant)
            y = pkg.mul_(pkg.HipVector(n, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy()
            # 3 CG steps exercise the fused-dot epilogue (the matrix is not SPD; only the bits matter)
            xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=3)
            return y, ch["resnorm"], xs.to_numpy()
        y, res, xs = with_knobs(pkg, {KN.LAYOUTS: KN.CSR_ONLY, KN.CSR_KERNEL: variant} if variant != "jagged-slices" else {KN.LAYOUTS: KN.JAGGED_ALWAYS}, run)
        assert np.array_equal(y, want), variant
        if ref is None:
            ref = (res, xs)
        assert np.array_equal(res, ref[0]) and np.array_equal(xs, ref[1]), variant


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_slice_constant_values_only_when_the_bits_agree(pkg, orc, ctx, dtype):
    """constant-coefficient stencil -> one value per slot and slice; perturb ONE entry (or flip the sign of a zero) and the
    operator keeps per-row value slots; either way mul! returns the oracle's bits"""
    A = orc.laplace(13, 3).astype(dtype)
    x = np.random.default_rng(3).standard_normal(A.n).astype(dtype)
    assert upload(pkg, A).layout() == "slice-offsets+slice-values+row-masks"
    B = orc.CSC(A.n, A.colptr, A.rowval, A.nzval.copy(), A.index_base)
    B.nzval[1234] = np.nextafter(B.nzval[1234], dtype(10))          # one ulp in one entry
    dB = upload(pkg, B)
    assert dB.layout() == "sliced-ell+slice-offsets+row-masks"
    assert np.array_equal(pkg.mul_(pkg.HipVector(A.n, dtype), dB, pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(B, x))
    # variable coefficients that are constant per SLICE still qualify: scale rows slice by slice
    S = A.to_scipy().tocsr()
    scale = np.repeat(np.arange(1, (A.n + 255) // 256 + 1, dtype=np.float64), 256)[:A.n]
    D = orc.CSC.from_scipy(sp.diags(scale) @ S)
    D = D.astype(dtype)
    dD = upload(pkg, D)
    assert dD.layout() == "slice-offsets+slice-values+row-masks"
    assert np.array_equal(pkg.mul_(pkg.HipVector(A.n, dtype), dD, pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(D, x))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,far", [(1500, 40), (1500, 41), (1501, 40), (1282, 255)])
def test_slice_constant_kernels_absent_slots_and_missing_diagonals(pkg, orc, ctx, dtype, n, far):
    """k_spmv_sdiab / k_spmv_sdiab2 read 0.0 for a slot a row does not have (buffer range check) and add value * 0: x entries
    that are Inf / NaN must only reach the rows that really reference them; rows without a diagonal entry (the fused dot takes
    x[r] from the centre slot otherwise); an Inf coefficient keeps the flat-load kernel.  The two-rows-per-lane kernel needs an
    even n (1501: one row per lane) and row pairs that agree on the presence of a gathered slot (far = 41, 255: the pairs at the
    ends of the far diagonals do not -- those waves run slot by slot); its lane-neighbour slots may be absent in any row."""
    rng = np.random.default_rng(11)
    S = sp.diags([np.full(n - far, -1.0), np.full(n - 1, -2.0), np.full(n, 5.0), np.full(n - 1, -3.0), np.full(n - far, -0.5)],
                 [-far, -1, 0, 1, far], format="lil")
    for r in (0, 3, 255, 256, 700, n - 1):              # rows without a diagonal entry, also at slice edges
        S[r, r] = 0.0
    for r in (39, 40, 41, 64, 128, 900):                # rows without the slot below the centre, also first in their wave
        S[r, r - 1] = 0.0
    for r in (63, 127, 500, 501):                       # ... without the slot above it, also last in their wave
        S[r, r + 1] = 0.0
    S[700, 700 - far] = 0.0                             # one row of a pair without a far slot
    S = S.tocsc()
    S.eliminate_zeros()
    A = orc.CSC.from_scipy(S).astype(dtype)
    x = rng.standard_normal(n).astype(dtype)
    x[[0, 38, 62, 128, 255, 256, 499, 502, 899, n - 1]] = [np.inf, np.nan, np.inf, np.nan, -np.inf, np.nan, np.inf, np.nan, np.inf, np.nan]
    want = orc.spmv(A, x)
    xf = rng.standard_normal(n).astype(dtype)
    b = orc.hashed_rhs(n).astype(dtype)
    ref = None
    for form, knobs in (("csr-rowblock", {KN.LAYOUTS: KN.CSR_ONLY}), ("best", {}), ("best/flat-loads", {KN.SDIA_KERNEL: 1}), ("best/slot-by-slot", {KN.SDIA_KERNEL: 2}),
                        ("best/one-row-per-lane", {KN.SDIA_KERNEL: 3})):
        def run():
            dA = upload(pkg, A)
            if form.startswith("best"):
                assert dA.layout() == "slice-offsets+slice-values+row-masks"
                assert dA.spmv_kernel() == ("k_spmv_sdiac" if knobs.get(KN.SDIA_KERNEL) == 1 else "k_spmv_sdiab2" if not knobs and n % 2 == 0 else "k_spmv_sdiab")
            y = pkg.mul_(pkg.HipVector(n, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy()
            xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=4)     # fused dot; only the bits matter
            return y, ch["resnorm"], xs.to_numpy(), pkg.mul_(pkg.HipVector(n, dtype), dA, pkg.HipVector.from_numpy(xf)).to_numpy()
        y, res, xs, yf = with_knobs(pkg, knobs, run)
        assert np.array_equal(y, want, equal_nan=True), form
        assert np.array_equal(yf, orc.spmv(A, xf)), form
        if ref is None:
            ref = (res, xs)
        assert np.array_equal(res, ref[0]) and np.array_equal(xs, ref[1]), form
    # a non-finite coefficient: value * 0.0 would be NaN, so the buffer kernel is not used
    S2 = S.copy().tolil()
    S2[10, 11] = np.inf
    A2 = orc.CSC.from_scipy(S2.tocsc()).astype(dtype)
    dA2 = upload(pkg, A2)
    if dA2.layout() == "slice-offsets+slice-values+row-masks":
        assert dA2.spmv_kernel() == "k_spmv_sdiac"
    assert np.array_equal(pkg.mul_(pkg.HipVector(n, dtype), dA2, pkg.HipVector.from_numpy(xf)).to_numpy(), orc.spmv(A2, xf), equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("rows", [1300, 1301, 1282, 65, 2])
def test_rectangular_blocks_with_lane_neighbour_slots(pkg, ctx, dtype, rows):
    """The local block of a row-partitioned operator has more columns than rows (halo entries behind the owned ones).  The
    slice-constant kernels take the slots next to the centre from neighbouring lanes; the last row's upper neighbour is then
    column `rows` -- outside the rows, inside the columns -- and a wave may consist of that one row (65) or pair (1282 = 10 x 128
    + 2).  Every layout against the row-by-row sums in column order."""
    cols, far = rows + 50, 40
    S = sp.diags([np.full(cols - far, -1.0), np.full(cols - 1, -2.0), np.full(cols, 5.0), np.full(cols - 1, -3.0), np.full(cols - far, -0.5)],
                 [-far, -1, 0, 1, far], format="csr")[:rows].astype(dtype).tocsr()
    S.sort_indices()
    x = np.random.default_rng(5).standard_normal(cols).astype(dtype)
    want = np.zeros(rows, dtype)
    lens = np.diff(S.indptr)
    for k in range(int(lens.max())):                     # a = a + val * x, entries in column order, one rounding each
        has = lens > k
        at = S.indptr[:-1][has] + k
        want[has] = (want[has] + (S.data[at] * x[S.indices[at]]).astype(dtype)).astype(dtype)
    C = S.tocsc()
    C.sort_indices()
    seen = set()
    for form, knobs in (("csr-rowblock", {KN.LAYOUTS: KN.CSR_ONLY}), ("best", {}), ("best/one-row-per-lane", {KN.SDIA_KERNEL: 3}), ("best/flat-loads", {KN.SDIA_KERNEL: 1}),
                        ("best/slot-by-slot", {KN.SDIA_KERNEL: 2})):
        def run():
            dA = pkg.HipCSR(rows, cols, C.indptr.astype(np.int64), C.indices.astype(np.int64), C.data, index_base=0)
            seen.add(dA.spmv_kernel())
            return pkg.mul_(pkg.HipVector(rows, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy()
        assert np.array_equal(with_knobs(pkg, knobs, run), want), form
    if rows % 2 == 0 and rows >= 256:
        assert "k_spmv_sdiab2" in seen and "k_spmv_sdiab" in seen


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_jagged_slices_with_split_off_long_rows(pkg, orc, ctx, dtype):
    """a finite-element operator with a few dense constraint rows: the FE rows run in the jagged slices, the rows beyond
    mik_spmv_long_row() in the wave-per-row workgroups that lead the same launch (MERGE_LONG), and dot(u, c) of the CG step comes
    from k_rowdot -- mul! and the CG history equal the oracle's bit for bit (long rows in the documented wave shape)"""
    n, rp, ci, vv = pkg.fixtures.fe_matrix((19, 23), 6, dtype)
    S = sp.csr_matrix((vv.astype(np.float64), ci, rp), shape=(n, n)).tolil()
    rng = np.random.default_rng(12)
    for row, cnt in ((7, 300), (n // 2, 2600), (n - 3, 257)):
        cols = np.sort(rng.choice(n, size=cnt, replace=False))
        S[row, cols] = rng.standard_normal(cnt) * 0.01
        S[row, row] = 50.0
    S = (S + S.T).tocsr() * 0.5                                   # keep it symmetric (the dense rows become dense columns too)
    S = S + sp.diags(np.full(n, 5.0))
    S = S.astype(dtype).tocsr()
    S.sort_indices()
    lens = np.diff(S.indptr)
    assert (lens > ctx.spmv_long_row()).sum() >= 2
    dA = pkg.HipCSR(n, n, S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data, index_base=0, is_csc=False)
    assert dA.layout() == "jagged-slices" and dA.spmv_kernel() == "k_spmv_jds"
    A = orc.CSC.from_scipy(S.tocsc())
    orc.set_long_row(ctx.spmv_long_row(), ctx.spmv_long_segment(), ctx.spmv_long_group())
    try:
        x = rng.standard_normal(n).astype(dtype)
        for _ in range(2):                                        # the segment tickets reset themselves
            assert np.array_equal(pkg.mul_(pkg.HipVector(n, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(A, x))
        b = orc.hashed_rhs(n).astype(dtype)
        xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=25)
        xo, ho = orc.cg(A, b, maxiter=25, mode="tree", shape=ctx.cg_shape(dtype))
        assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(xs.to_numpy(), xo)
    finally:
        orc.set_long_row(0)


def test_development_knobs_are_per_context(pkg, orc, ctx):
    """a knob set on one context (mik_ctx_set_tuning) does not reach another; the process-wide mik_set_tuning
    of the test suite writes the defaults and every live context"""
    A = orc.laplace(9, 3)
    other = pkg.HipContext(0)
    try:
        other.set_tuning(KN.LAYOUTS, KN.CSR_ONLY)                    # CSR only -- on `other`
        dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base, ctx=ctx)
        dB = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base, ctx=other)
        assert dA.layout() == "slice-offsets+slice-values+row-masks" and dB.layout() == "csr-rowblock"
        x = np.random.default_rng(0).standard_normal(A.n)
        ya = pkg.mul_(pkg.HipVector(A.n, ctx=ctx), dA, pkg.HipVector.from_numpy(x, ctx=ctx)).to_numpy()
        yb = pkg.mul_(pkg.HipVector(A.n, ctx=other), dB, pkg.HipVector.from_numpy(x, ctx=other)).to_numpy()
        assert np.array_equal(ya, yb) and np.array_equal(ya, orc.spmv(A, x))
        del dB
        pkg.lib().mik_set_tuning(KN.LAYOUTS, KN.CSR_ONLY)            # process-wide: reaches both
        try:
            assert pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base, ctx=ctx).layout() == "csr-rowblock"
        finally:
            pkg.lib().mik_set_tuning(KN.LAYOUTS, 0)
        assert pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base, ctx=other).layout() == "slice-offsets+slice-values+row-masks"
    finally:
        other.close() if hasattr(other, "close") else None


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("lo,hi", [(-16, 15), (-15, 15), (-20, 11)])
def test_wide_layout_at_its_slot_limit(pkg, orc, dtype, lo, hi):
    """31 and 32 constant offsets per slice (the mask word's capacity): 32 leave no spare mask bit, so the slice is summed slot by
    slot; 31 decompose into runs and lone offsets.  Same bits as the oracle either way."""
    n = 1536
    offs = np.arange(lo, hi + 1)
    rows = np.repeat(np.arange(n), offs.size)
    cols = rows + np.tile(offs, n)
    ok = (cols >= 0) & (cols < n)
    rows, cols = rows[ok], cols[ok]
    vals = np.where(cols == rows, 40.0, -1.0 - ((cols - rows) % 13) / 16.0).astype(dtype)
    M = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    M.sort_indices()
    dA = pkg.HipCSR(n, n, M.indptr, M.indices, M.data, index_base=0, is_csc=False)
    assert dA.layout() == "wide-slice-values+row-masks"
    A = orc.CSC.from_scipy(M.tocsc())
    x = np.random.default_rng(2).standard_normal(n).astype(dtype)
    x[100] = np.inf
    assert np.array_equal(pkg.mul_(pkg.HipVector(n, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(A, x), equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(128, 6, 5), (256, 5, 4), (128, 9), (384, 7), (64, 8, 6), (128, 3, 3)])
def test_wide_layout_slot_tests_per_wave(pkg, orc, ctx, dtype, shape):
    """k_spmv_sdiaw2 tests a slot once per wave where its 128 rows agree (k_sdiaw_chunk_bits, built at upload): grid lines of 128 /
    256 / 384 nodes put whole waves into the interior (every slot present: the multiply-add-only loop, 9 or 3 items in flight),
    onto a y / z face (whole runs absent: skipped) and -- lines of 64 -- across line ends (some rows lack a slot: per-lane
    select).  The first / last row of a wave takes its outer value from the edge load, which must read 0 where the row has no
    such neighbour: the line ends of x carry Inf, which may only reach the rows that reference them.  Distinct values per offset."""
    n, rp, ci, _ = pkg.fixtures.fe_matrix(shape, 1, np.float64, renumber=False)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    off = ci - rows
    vv = np.where(off == 0, 30.0, -1.0 - ((off % 97) / 128.0)).astype(dtype)
    dA = pkg.HipCSR(n, n, rp, ci, vv, index_base=0, is_csc=False)
    assert dA.layout() == "wide-slice-values+row-masks" and dA.spmv_kernel() == "k_spmv_sdiaw2"
    A = orc.CSC.from_scipy(sp.csr_matrix((vv, ci, rp), shape=(n, n)).tocsc())
    rng = np.random.default_rng(11)
    for trial in range(3):
        x = rng.standard_normal(n).astype(dtype)
        if trial == 1:
            x[shape[0] - 1::shape[0]] = np.inf                         # last node of every line
        if trial == 2:
            x[::shape[0]] = -np.inf                                    # first node of every line
        want = orc.spmv(A, x)
        dx = pkg.HipVector.from_numpy(x)
        got = pkg.mul_(pkg.HipVector(n, dtype), dA, dx).to_numpy()
        assert np.array_equal(got, want, equal_nan=True)
    # the dot fused into the SpMV: six steps of the CG recurrence (the operator is not symmetric -- only the arithmetic matters)
    b = orc.hashed_rhs(n).astype(dtype)
    xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=6)
    xo, ho = orc.cg(A, b, maxiter=6, mode="tree", shape=ctx.cg_shape(dtype))
    assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(xs.to_numpy(), xo)
    dA.set_layout("csr")
    xc, cc = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=6)
    dA.set_layout("auto")
    assert np.array_equal(cc["resnorm"], ch["resnorm"]) and np.array_equal(xc.to_numpy(), xs.to_numpy())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N,dims", [(13, 3), (40, 2), (21, 3), (16, 3), (22, 3), (7, 2)])
def test_wide_slice_constant_layout_for_box_stencils(pkg, orc, ctx, dtype, N, dims):
    """constant-coefficient 9-point (2-D) / 27-point (3-D) stencils exceed the 8 offsets per slice of the mask-byte
    layouts; they get slice patterns of up to 32 {offset, value} pairs and one 32-bit mask per row, chosen automatically: mul!,
    the fused dot and the CG history equal the oracle's and the CSR layout's bit for bit.  One perturbed coefficient (or an Inf)
    and the operator falls back to a general layout with the same bits."""
    n, rp, ci, vv = pkg.fixtures.box_stencil_matrix(N, dims, dtype)
    dA = pkg.HipCSR(n, n, rp, ci, vv, index_base=0, is_csc=False)
    # an even number of rows: two rows per lane (even grid sizes: every run of line neighbours has an even centre -> one 16-byte
    # gather per run; k_spmv_sdiaw2); odd: one row per lane
    assert dA.layout() == "wide-slice-values+row-masks" and dA.spmv_kernel() == ("k_spmv_sdiaw2" if n % 2 == 0 else "k_spmv_sdiaw")
    assert n < 3000 or dA.spmv_stored_bytes() < 0.2 * dA.spmv_algorithmic_bytes()
    A = orc.CSC.from_scipy(sp.csr_matrix((vv, ci, rp), shape=(n, n)).tocsc())
    rng = np.random.default_rng(4)
    x = rng.standard_normal(n).astype(dtype)
    x[::17] = np.inf                                                   # an Inf in x must only reach the rows that reference it
    want = orc.spmv(A, x)
    got = pkg.mul_(pkg.HipVector(n, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy()
    assert np.array_equal(got, want, equal_nan=True)
    b = orc.hashed_rhs(n).astype(dtype)
    xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True)
    xo, ho = orc.cg(A, b, mode="tree", shape=ctx.cg_shape(dtype))
    assert ch.isconverged and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(xs.to_numpy(), xo)
    dA.set_layout("csr")
    xc, cc = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True)
    assert np.array_equal(cc["resnorm"], ch["resnorm"]) and np.array_equal(xc.to_numpy(), xs.to_numpy())
    dA.set_layout("auto")
    ctx.set_tuning(KN.SDIA_KERNEL, 3)                                  # one row per lane
    try:
        assert dA.spmv_kernel() == "k_spmv_sdiaw"
        assert np.array_equal(pkg.mul_(pkg.HipVector(n, dtype), dA, pkg.HipVector.from_numpy(x)).to_numpy(), want, equal_nan=True)
        x1, c1 = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True)
        assert np.array_equal(c1["resnorm"], ch["resnorm"]) and np.array_equal(x1.to_numpy(), xs.to_numpy())
    finally:
        ctx.set_tuning(KN.SDIA_KERNEL, 0)
    assert dA.set_layout("auto").compact() and dA.layout() == "wide-slice-values+row-masks"
    for bad in (np.nextafter(vv[5], dtype(10)), np.inf):
        v2 = vv.copy()
        v2[5] = bad
        dB = pkg.HipCSR(n, n, rp, ci, v2, index_base=0, is_csc=False)
        assert dB.layout() != "wide-slice-values+row-masks"
        B = orc.CSC.from_scipy(sp.csr_matrix((v2, ci, rp), shape=(n, n)).tocsc())
        xf = rng.standard_normal(n).astype(dtype)
        assert np.array_equal(pkg.mul_(pkg.HipVector(n, dtype), dB, pkg.HipVector.from_numpy(xf)).to_numpy(), orc.spmv(B, xf), equal_nan=True)
