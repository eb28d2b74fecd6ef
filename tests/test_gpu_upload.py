"""mik_csr_create: the device-side upload pipeline (csrc/mik_upload.hip: raw CSC -> validated Int32 CSR + layout analysis on
the device) against the host path (MIK_KNOB_UPLOAD = 1) and the oracle: same layout choice, same stored bytes, same bits
out of mul_ -- for symmetric and nonsymmetric CSC input, CSR input, a rank's rectangular block, empty rows, both dtypes --
and the matrices that are handed back to the host path (long rows, duplicate entries).  GPU box only."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def both_paths(pkg, make):
    L = pkg.lib()
    dev = make()
    L.mik_set_tuning(4, 1)          # MIK_KNOB_UPLOAD
    try:
        host = make()
    finally:
        L.mik_set_tuning(4, 0)
    return dev, host


def same_operator(pkg, orc, dev, host, A, x):
    assert dev.layout() == host.layout()
    assert dev.spmv_kernel() == host.spmv_kernel()
    assert dev.spmv_stored_bytes() == host.spmv_stored_bytes()
    yd = pkg.mul_(pkg.HipVector(dev.n_rows, x.dtype), dev, pkg.HipVector.from_numpy(x)).to_numpy()
    yh = pkg.mul_(pkg.HipVector(host.n_rows, x.dtype), host, pkg.HipVector.from_numpy(x)).to_numpy()
    assert np.array_equal(yd, yh)
    if A is not None:
        assert np.array_equal(yd, orc.spmv(A, x))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("case", ["laplace3d", "laplace2d", "advdiff", "varying", "banded_wide", "random", "empty_rows", "fe", "box27"])
def test_device_upload_equals_host_upload(pkg, orc, ctx, case, dtype):
    rng = np.random.default_rng(3)
    if case == "laplace3d":
        A = orc.laplace(13, 3)
    elif case == "laplace2d":
        A = orc.laplace(41, 2)
    elif case == "advdiff":                                  # nonsymmetric: the transpose matters
        A = orc.advdiff(9, 300.0)[0]
    elif case == "varying":                                  # stencil with varying coefficients: per-row value slots
        n = 2000
        S = sp.diags([rng.standard_normal(n - abs(o)) for o in (-45, -1, 0, 1, 45)], (-45, -1, 0, 1, 45), format="csc")
        A = orc.CSC.from_scipy(S)
    elif case == "fe":                                       # finite-element rows: jagged slices, built on the device / by the host builder
        nf, rp, ci, vv = pkg.fixtures.fe_matrix((17, 19), 6, np.float64)
        A = orc.CSC.from_scipy(sp.csr_matrix((vv, ci, rp), shape=(nf, nf)).tocsc())
    elif case == "box27":                                    # constant-coefficient 27-point stencil: wide slice-constant form, both builders
        nf, rp, ci, vv = pkg.fixtures.box_stencil_matrix(14, 3, np.float64)
        A = orc.CSC.from_scipy(sp.csr_matrix((vv, ci, rp), shape=(nf, nf)).tocsc())
    elif case == "banded_wide":                              # 19 constant diagonals (> 8 offsets per slice): wide slice-constant form
        n = 3000
        offs = [0] + [o for d in (1, 2, 3, 7, 50, 51, 200, 333, 900) for o in (d, -d)]
        S = sp.diags([np.full(n - abs(o), 40.0 if o == 0 else -1.0 / (1 + abs(o) % 5)) for o in offs], offs, format="csc")
        A = orc.CSC.from_scipy(S)
    elif case == "random":                                   # rows of 0..40 entries, random columns: CSR layout
        n = 3000
        S = sp.random(n, n, density=0.004, random_state=5, format="csc")
        S.sort_indices()
        A = orc.CSC.from_scipy(S)
    else:
        D = sp.lil_matrix((1000, 1000))
        D[5, 7] = 2.0; D[5, 5] = 1.0; D[700, 3] = -4.0; D[999, 999] = 3.0
        A = orc.CSC.from_scipy(D.tocsc())
    A = A.astype(dtype)
    x = rng.standard_normal(A.n).astype(dtype)
    dev, host = both_paths(pkg, lambda: pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base))
    same_operator(pkg, orc, dev, host, A, x)
    # the same matrix handed over as CSR (0-based)
    S = A.to_scipy().tocsr()
    S.sort_indices()
    dev, host = both_paths(pkg, lambda: pkg.HipCSR(A.n, A.n, S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data.astype(dtype), index_base=0,
                                                   is_csc=False))
    same_operator(pkg, orc, dev, host, A, x)


def test_device_upload_of_a_rank_block_with_halo_columns(pkg, orc, ctx, dist):
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
        dev, host = both_paths(pkg, lambda: pkg.HipCSR(plan.n_loc, plan.n_loc + plan.n_ghost, blk.indptr.astype(np.int64), li, blk.data, index_base=0,
                                                       is_csc=False))
        xe = np.concatenate([x[r0:r1], x[plan.ghost_gids]])
        same_operator(pkg, orc, dev, host, None, xe)
        assert np.array_equal(pkg.mul_(pkg.HipVector(plan.n_loc), dev, pkg.HipVector.from_numpy(xe)).to_numpy(), want[r0:r1])


def test_matrices_handed_back_to_the_host_path(pkg, orc, ctx):
    """long rows (the wave-shaped row sum) and duplicate (row, column) entries in CSC input keep the host path's semantics"""
    orc.set_long_row(ctx.spmv_long_row(), ctx.spmv_long_segment(), ctx.spmv_long_group())
    try:
        n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(6000, np.float64)
        dev, host = both_paths(pkg, lambda: pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False))
        x = np.random.default_rng(4).standard_normal(n)
        M = sp.csr_matrix((val, colidx, rowptr), shape=(n, n)).tocsc()
        M.sort_indices()
        same_operator(pkg, orc, dev, host, orc.CSC(n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.copy(), 0), x)
    finally:
        orc.set_long_row(0)
    # duplicates: column 1 lists row 2 twice; the host path adds both in the order given (Julia's column scatter would too)
    colptr = np.array([0, 1, 4, 5], np.int64)
    rowval = np.array([0, 2, 1, 2, 2], np.int64)
    nz = np.array([1.0, 0.1, 5.0, 0.7, 3.0])
    dev, host = both_paths(pkg, lambda: pkg.HipCSR(3, 3, colptr, rowval, nz, index_base=0))
    x = np.array([1.0, 3.0, -2.0])
    yd = pkg.mul_(pkg.HipVector(3), dev, pkg.HipVector.from_numpy(x)).to_numpy()
    yh = pkg.mul_(pkg.HipVector(3), host, pkg.HipVector.from_numpy(x)).to_numpy()
    assert np.array_equal(yd, yh)
    assert yd[2] == (0.1 * 3.0 + 0.7 * 3.0) + 3.0 * -2.0


def test_invalid_input_is_rejected_by_the_device_path(pkg, ctx):
    ptr = np.array([0, 1, 2], np.int64)
    with pytest.raises(pkg.MikError):
        pkg.HipCSR(2, 2, ptr, np.array([0, 5], np.int64), np.ones(2), index_base=0)       # index out of range
    with pytest.raises(pkg.MikError):
        pkg.HipCSR(3, 3, np.array([0, 2, 1, 3], np.int64), np.array([0, 1, 2], np.int64), np.ones(3), index_base=0)   # ptr not monotone
    A = pkg.HipCSR(2, 2, ptr, np.array([0, 1], np.int64), np.array([2.0, 3.0]), index_base=0)   # the context is still usable
    y = pkg.mul_(pkg.HipVector(2), A, pkg.HipVector.from_numpy(np.array([1.0, 1.0]))).to_numpy()
    assert np.array_equal(y, [2.0, 3.0])


def test_device_resident_input_arrays(pkg, orc, ctx, dist):
    """mik_csr_create on arrays that already live in device memory (torch tensors), incl. a matrix the host path must take;
    the row-partitioned bench builds its slab this way (dist._laplace_rows_torch / localize_block_torch)"""
    import torch
    A = orc.advdiff(9, 300.0)[0]
    dev = torch.device("cuda", 0)
    tp, ti, tv = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (A.colptr.astype(np.int64), A.rowval.astype(np.int64), A.nzval))
    torch.cuda.synchronize()
    dA = pkg.HipCSR.from_device(A.n, A.n, A.nzval.size, tp.data_ptr(), ti.data_ptr(), tv.data_ptr(), np.float64, index_base=A.index_base)
    hA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)
    x = np.random.default_rng(8).standard_normal(A.n)
    same_operator(pkg, orc, dA, hA, A, x)
    # long rows: handed back to the host path, which stages the device arrays once
    orc.set_long_row(ctx.spmv_long_row(), ctx.spmv_long_segment(), ctx.spmv_long_group())
    try:
        n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(6000, np.float64)
        tp, ti, tv = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rowptr.astype(np.int64), colidx.astype(np.int64), val))
        torch.cuda.synchronize()
        dA = pkg.HipCSR.from_device(n, n, val.size, tp.data_ptr(), ti.data_ptr(), tv.data_ptr(), np.float64, index_base=0, is_csc=False)
        hA = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
        same_operator(pkg, orc, dA, hA, None, np.random.default_rng(9).standard_normal(n))
    finally:
        orc.set_long_row(0)
    # mixed placement is refused
    with pytest.raises(pkg.MikError):
        pkg.HipCSR.from_device(A.n, A.n, A.nzval.size, tp.data_ptr(), A.rowval.ctypes.data, tv.data_ptr(), np.float64)
    # the slab generator and the localisation on the device equal the numpy ones
    N, NZ, P = 12, 9, 3
    offsets = np.arange(P + 1, dtype=np.int64) * (N * N * (NZ // P))
    for r in range(P):
        _, ptr, idx, val = dist._laplace_rows(pkg, N, NZ, offsets[r], offsets[r + 1], np.float64)
        _, tptr, tidx, tval = dist._laplace_rows_torch(N, NZ, offsets[r], offsets[r + 1], np.float64, 0)
        assert np.array_equal(ptr, tptr.cpu().numpy()) and np.array_equal(idx, tidx.cpu().numpy()) and np.array_equal(val, tval.cpu().numpy())
        li, plan = dist.localize_block(ptr, idx, offsets, r)
        tli, tplan = dist.localize_block_torch(tptr, tidx, offsets, r)
        assert np.array_equal(li, tli.cpu().numpy()) and np.array_equal(plan.ghost_gids, tplan.ghost_gids) and plan.recv == tplan.recv
        assert dist.interior_row_blocks(ptr, li, plan.n_loc) == dist.interior_row_blocks(tptr, tli, plan.n_loc)


def test_compact_releases_the_csr_arrays(pkg, orc, ctx):
    A = orc.laplace(12, 3)
    x = np.random.default_rng(1).standard_normal(A.n)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)
    want = orc.spmv(A, x)
    assert dA.compact() and dA.compact()                              # idempotent
    assert dA.layout() == "slice-offsets+slice-values+row-masks"
    assert np.array_equal(pkg.mul_(pkg.HipVector(A.n), dA, pkg.HipVector.from_numpy(x)).to_numpy(), want)
    L = pkg.lib()
    L.mik_set_tuning(0, 1)                                            # "CSR only" has nothing to fall back to any more
    try:
        assert dA.layout() == "slice-offsets+slice-values+row-masks"
        assert np.array_equal(pkg.mul_(pkg.HipVector(A.n), dA, pkg.HipVector.from_numpy(x)).to_numpy(), want)
    finally:
        L.mik_set_tuning(0, 0)
    xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(orc.hashed_rhs(A.n)), log=True)
    assert ch.isconverged
    # an operator that runs on its CSR arrays keeps them
    rng = np.random.default_rng(4)
    n = 3000
    colidx = ((np.arange(n)[:, None] + np.sort(rng.choice(2000, size=(n, 3)), axis=1) + 1) % n)
    colidx.sort(axis=1)
    keep = np.ones(colidx.shape, bool)
    keep[:, 1:] = colidx[:, 1:] != colidx[:, :-1]
    rowptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int64)
    dB = pkg.HipCSR(n, n, rowptr, colidx[keep].astype(np.int64), rng.standard_normal(int(keep.sum())), index_base=0, is_csc=False)
    assert dB.layout() == "csr-rowblock" and dB.compact() is False                 # 3 entries per row: the CSR tile
