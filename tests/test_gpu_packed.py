"""Dictionary-coded operator (mik_csr_pack, csrc/mik_packed.h): opt-in, lossless -- every result must be
bit-identical to the plain CSR path (and therefore to the oracle).  GPU box only."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def up(pkg, A):
    return pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("case", ["lap3d_20", "lap2d_33", "lap1d_1000", "advdiff_12", "lap3d_48"])
def test_packed_spmv_bit_exact(pkg, orc, ctx, case, dtype):
    if case.startswith("lap"):
        A = orc.laplace(int(case.split("_")[1]), int(case[3]))
    else:
        A, _ = orc.advdiff(12, 1000.0)
    A = A.astype(dtype)
    x = np.random.default_rng(1).standard_normal(A.n).astype(dtype)
    dA = up(pkg, A)
    y_plain = (dA @ pkg.HipVector.from_numpy(x)).to_numpy()
    assert dA.pack()
    y_packed = (dA @ pkg.HipVector.from_numpy(x)).to_numpy()
    assert np.array_equal(y_packed, y_plain) and np.array_equal(y_packed, orc.spmv(A, x))


def test_pack_declines_matrices_that_do_not_qualify(pkg, orc, ctx):
    rng = np.random.RandomState(5)
    M = sp.random(2000, 2000, 0.01, random_state=rng, format="csc") + sp.eye(2000, format="csc")     # thousands of distinct values
    A = orc.CSC.from_scipy(M)
    dA = up(pkg, A)
    x = rng.standard_normal(2000)
    y0 = (dA @ pkg.HipVector.from_numpy(x)).to_numpy()
    assert dA.pack() is False
    assert np.array_equal((dA @ pkg.HipVector.from_numpy(x)).to_numpy(), y0)                          # operator unchanged
    n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(5000, np.float64)                          # long rows
    B = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
    assert B.pack() is False


def test_packed_rows_longer_than_a_batch_and_empty_rows(pkg, orc, ctx):
    """banded 0/1-valued matrix: 41 entries per row (> 8-entry batches), a few empty rows"""
    n = 3000
    offs = list(range(-20, 21))
    M = sp.diags([np.ones(n - abs(o)) * (1.0 if o % 2 else -0.5) for o in offs], offs, format="lil")
    M[7, :] = 0
    M[2999, :] = 0
    M = M.tocsc()
    M.eliminate_zeros()
    A = orc.CSC.from_scipy(M)
    dA = up(pkg, A)
    assert dA.pack()
    x = np.random.default_rng(0).standard_normal(n)
    y = (dA @ pkg.HipVector.from_numpy(x)).to_numpy()
    assert np.array_equal(y, orc.spmv(A, x)) and y[7] == 0


@pytest.mark.parametrize("N", [16, 32])
def test_packed_cg_history_bit_exact(pkg, orc, ctx, N):
    A = orc.laplace(N, 3)
    b = orc.hashed_rhs(A.n)
    dA = up(pkg, A)
    assert dA.pack()
    x, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True)
    xo, ho = orc.cg(A, b, mode="tree", shape=ctx.cg_shape(np.float64))
    assert ch.iters == ho["iters"] and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)


def test_packed_gmres_bit_exact(pkg, orc, ctx):
    A, b = orc.advdiff(12, 1000.0)
    dA = up(pkg, A)
    assert dA.pack()
    x, ch = pkg.gmres(dA, pkg.HipVector.from_numpy(b), restart=10, log=True)
    xo, ho = orc.gmres(A, b, restart=10, mode="tree", shape=ctx.reduce_shape(np.float64))
    assert ch.iters == ho["iters"] and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
