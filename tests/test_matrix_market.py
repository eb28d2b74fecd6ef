"""MatrixMarket reader (fixtures.read_matrix_market): the on-disk input of BASELINE.json configs[4]
(benchmark/matrixmarket.jl:5-10).  Host logic only -- checked against scipy.io."""
import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp


def as_scipy(t):
    n_rows, n_cols, colptr, rowval, nzval = t
    return sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(n_rows, n_cols))


@pytest.mark.parametrize("symmetry", ["general", "symmetric"])
@pytest.mark.parametrize("field", ["real", "integer", "pattern"])
def test_reader_matches_scipy(pkg, tmp_path, symmetry, field):
    rng = np.random.RandomState(3)
    M = sp.random(40, 40 if symmetry == "symmetric" else 25, 0.1, random_state=rng, format="coo")
    if field != "real":
        M.data = np.round(M.data * 10) + 1
    if symmetry == "symmetric":
        M = (M + M.T).tocoo()
    if field == "pattern":
        M.data[:] = 1.0
    path = str(tmp_path / "m.mtx")
    scipy.io.mmwrite(path, M, field=field, symmetry=symmetry)
    got = as_scipy(pkg.fixtures.read_matrix_market(path))
    ref = scipy.io.mmread(path).tocsc()
    assert got.shape == ref.shape and abs(got - ref).max() == 0
    n_rows, n_cols, colptr, rowval, nzval = pkg.fixtures.read_matrix_market(path)
    assert colptr[0] == 1 and colptr[-1] - 1 == nzval.size
    for j in range(n_cols):                                      # rows ascending inside every column, like Julia stores them
        r = rowval[colptr[j] - 1: colptr[j + 1] - 1]
        assert np.all(np.diff(r) > 0)


def test_reader_rejects_complex_and_array(pkg, tmp_path):
    p = tmp_path / "c.mtx"
    p.write_text("%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 2.0\n")
    with pytest.raises(ValueError):
        pkg.fixtures.read_matrix_market(str(p))
    p.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(ValueError):
        pkg.fixtures.read_matrix_market(str(p))


@pytest.mark.gpu
def test_mtx_file_through_the_device_path(pkg, orc, ctx, tmp_path):
    """an SPD .mtx file -> read_matrix_market -> HipCSR -> cg, bit-exact against the oracle on the same arrays"""
    A = orc.laplace(9, 3)
    path = str(tmp_path / "lap.mtx")
    scipy.io.mmwrite(path, sp.tril(A.to_scipy()).tocoo(), symmetry="symmetric")      # stored as a lower triangle
    n_rows, n_cols, colptr, rowval, nzval = pkg.fixtures.read_matrix_market(path)
    assert np.array_equal(colptr, A.colptr) and np.array_equal(rowval, A.rowval) and np.array_equal(nzval, A.nzval)
    b = orc.hashed_rhs(A.n)
    x, ch = pkg.cg(pkg.HipCSR(n_rows, n_cols, colptr, rowval, nzval), pkg.HipVector.from_numpy(b), log=True)
    xo, ho = orc.cg(A, b, mode="tree", shape=ctx.cg_shape(np.float64))
    assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)
