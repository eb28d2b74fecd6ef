"""A second, independent restatement of the reference in pure Python scalar loops (IEEE double arithmetic, one rounded
operation per statement, strictly left-to-right sums) for tiny systems, compared BIT FOR BIT with the C oracle's SEQ
mode.  Two restatements written separately from the same reference lines agreeing exactly is the strongest pin of
the oracle available without a Julia runtime (the reference stores no residual histories)."""
import math

import numpy as np
import pytest


def csc_mul(A, x):
    """mul!(y, A::SparseMatrixCSC, x): column scatter, y[rowval[k]] += nzval[k] * x[col]  (SparseArrays)"""
    y = [0.0] * A.n
    for j in range(A.n):
        for k in range(int(A.colptr[j]) - A.index_base, int(A.colptr[j + 1]) - A.index_base):
            i = int(A.rowval[k]) - A.index_base
            y[i] = y[i] + float(A.nzval[k]) * x[j]
    return y


def dot(x, y):
    s = 0.0
    for a, b in zip(x, y):
        s = s + a * b
    return s


def py_cg(A, b, reltol, maxiter):
    """src/cg.jl:120-155 (cg_iterator!, initially_zero) + :43-66 (iterate) + :209-242 (cg!)"""
    n = A.n
    x, u, r = [0.0] * n, [0.0] * n, list(map(float, b))
    residual = math.sqrt(dot(r, r))                          # :140
    prev_residual = 1.0                                      # :146
    tol = max(reltol * residual, 0.0)                        # :141
    hist = []
    iteration = 0
    while not (iteration >= maxiter or residual <= tol):     # :36
        beta = residual * residual / (prev_residual * prev_residual)   # :50
        u = [ri + beta * ui for ri, ui in zip(r, u)]         # :51
        c = csc_mul(A, u)                                    # :54
        alpha = residual * residual / dot(u, c)              # :55
        x = [xi + alpha * ui for xi, ui in zip(x, u)]        # :58
        r = [ri - alpha * ci for ri, ci in zip(r, c)]        # :59
        prev_residual = residual                             # :61
        residual = math.sqrt(dot(r, r))                      # :62
        hist.append(residual)
        iteration += 1
    return x, hist


def py_gmres_mgs(A, b, restart, reltol, maxiter):
    """src/gmres.jl:108-136, :57-106, :224-304 with ModifiedGramSchmidt (src/orthogonalize.jl:67-79); x0 = 0.
    The least-squares solve goes through numpy.linalg.lstsq, so only the residual history (which never touches the
    Givens rotations, src/gmres.jl:224-233) is compared exactly, and x approximately."""
    n = A.n
    x = [0.0] * n
    V = [[0.0] * n for _ in range(restart + 1)]
    H = np.zeros((restart + 1, restart))
    nullvec = [1.0] * (restart + 1)

    def init():
        r = [bi - yi for bi, yi in zip(map(float, b), csc_mul(A, x))] if any(x) else list(map(float, b))
        beta = math.sqrt(dot(r, r))
        inv = 1.0 / beta
        V[0][:] = [ri * inv for ri in r]
        return beta

    beta = init()
    current, accumulator, res_beta = beta, 1.0, beta
    tol = max(reltol * beta, 0.0)
    k, iteration, hist = 1, 0, []
    done = lambda it: it >= maxiter or current <= tol
    while not done(iteration):
        w = csc_mul(A, V[k - 1])
        for i in range(k):
            h = dot(V[i], w)
            H[i, k - 1] = h
            w = [wi - h * vi for wi, vi in zip(w, V[i])]
        nrm = math.sqrt(dot(w, w))
        inv = 1.0 / nrm
        V[k][:] = [wi * inv for wi in w]
        H[k, k - 1] = nrm
        if H[k, k - 1] == 0.0:
            current = 0.0
        else:
            d = 0.0
            for i in range(k):
                d = d + nullvec[i] * H[i, k - 1]
            nullvec[k] = -(d / H[k, k - 1])
            accumulator = accumulator + nullvec[k] * nullvec[k]
            current = res_beta / math.sqrt(accumulator)
        k += 1
        if k == restart + 1 or done(iteration + 1):
            rhs = np.zeros(k)
            rhs[0] = beta
            y = np.linalg.lstsq(H[:k, :k - 1], rhs, rcond=None)[0]
            for j in range(k - 1):
                x = [xi + float(y[j]) * vi for xi, vi in zip(x, V[j])]
            k = 1
            if not done(iteration):
                beta = init()
                accumulator, res_beta = 1.0, beta
        hist.append(current)
        iteration += 1
    return x, hist


@pytest.mark.parametrize("N,dims", [(3, 3), (7, 2), (30, 1)])
def test_python_cg_equals_the_c_oracle_bit_for_bit(orc, N, dims):
    A = orc.laplace(N, dims)
    b = orc.hashed_rhs(A.n)
    x, hist = py_cg(A, b, 1.4901161193847656e-8, A.n)
    xo, ho = orc.cg(A, b, mode="seq")
    assert len(hist) == ho["iters"] and hist == list(ho["resnorm"]) and x == list(xo)


def test_python_gmres_history_equals_the_c_oracle_bit_for_bit(orc):
    A, b = orc.advdiff(4, 30.0)
    x, hist = py_gmres_mgs(A, b, 6, 1.4901161193847656e-8, 40)
    xo, ho = orc.gmres(A, b, restart=6, maxiter=40, orth_meth="mgs", mode="seq")
    assert len(hist) == ho["iters"]
    first_cycle = 6
    assert hist[:first_cycle] == list(ho["resnorm"][:first_cycle])          # before the first least-squares solve: exact
    np.testing.assert_allclose(hist, ho["resnorm"], rtol=1e-6)              # afterwards x comes from lstsq vs Givens
    np.testing.assert_allclose(x, xo, rtol=1e-7, atol=1e-12)


def wave_tree(v):
    """64 values -> shuffle-down tree with offsets 32, 16, ..., 1 (lane i adds lane i + offset); lane 0's result"""
    v = list(v)
    off = 32
    while off >= 1:
        v = [v[i] + (v[i + off] if i + off < 64 else v[i]) for i in range(64)]
        off //= 2
    return v[0]


def py_tree_dot(x, y, W, L):
    """The device's fixed-shape reduction written from its documentation (include/mik.h "Reduction semantics",
    DESIGN.md section 3), independently of oracle/orc_impl.inc."""
    n = len(x)
    seg = 256 * W * L
    nseg = (n + seg - 1) // seg
    S = []
    for s in range(nseg):
        lanes = []
        for t in range(256):
            acc = 0.0
            for e in range(W * L):
                i = s * seg + (e // W) * (256 * W) + W * t + e % W
                if i < n:
                    acc = acc + x[i] * y[i]
            lanes.append(acc)
        ws = [wave_tree(lanes[64 * w:64 * w + 64]) for w in range(4)]
        tot = ws[0]
        for w in range(1, 4):
            tot = tot + ws[w]
        S.append(tot)
    # level 2: 1024 virtual threads, stride-1024 serial sums, wave trees, 16 wave sums left to right
    vt = []
    for t in range(1024):
        acc = 0.0
        j = t
        while j < nseg:
            acc = acc + S[j]
            j += 1024
        vt.append(acc)
    ws = [wave_tree(vt[64 * w:64 * w + 64]) for w in range(16)]
    tot = ws[0]
    for w in range(1, 16):
        tot = tot + ws[w]
    return tot


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 1024, 1025, 4097, 70001])
@pytest.mark.parametrize("W,L", [(2, 2), (1, 1)])
def test_tree_reduction_shape_written_from_the_documentation(orc, n, W, L):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    assert py_tree_dot(list(map(float, x)), list(map(float, y)), W, L) == orc.dot(x, y, "tree", W, L)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("group", [1, 4])
def test_long_row_shape_against_a_python_restatement(orc, dtype, group):
    """The oracle's long-row mode (the device's documented row-sum shape for rows beyond mik_spmv_long_row(), include/mik.h):
    segments of `seg` entries; in a segment entry q goes to virtual lane (q / group) % 64, lanes add their products in ascending
    order from +0, wave-64 shuffle-down tree, segment sums left to right -- restated here in scalar Python, bit for bit."""
    rng = np.random.default_rng(41)
    n, thr, seg = 700, 40, 96
    lens = rng.integers(1, 30, size=n)
    for i, l in enumerate((41, 95, 96, 97, 192, 193, 500, 700)):
        lens[11 + 80 * i] = l
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens])
    vals = rng.standard_normal(cols.size).astype(dtype)
    import scipy.sparse as sp
    M = sp.csr_matrix((vals, cols, np.concatenate([[0], np.cumsum(lens)])), shape=(n, n)).tocsc()
    M.sort_indices()
    A = orc.CSC(n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(dtype), 0)
    x = rng.standard_normal(n).astype(dtype)
    orc.set_long_row(thr, seg, group)
    try:
        y = orc.spmv(A, x)
    finally:
        orc.set_long_row(0)
    yseq = orc.spmv(A, x)
    T = dtype
    start = np.concatenate([[0], np.cumsum(lens)])
    for r in range(n):
        if lens[r] <= thr:
            assert y[r] == yseq[r]
            continue
        prods = [T(vals[k]) * T(x[cols[k]]) for k in range(start[r], start[r + 1])]
        tot = None
        for s0 in range(0, len(prods), seg):
            lanes = [T(0)] * 64
            for q, pq in enumerate(prods[s0:s0 + seg]):
                lanes[(q // group) % 64] = T(lanes[(q // group) % 64] + pq)
            off = 32
            while off >= 1:
                for l in range(off):
                    lanes[l] = T(lanes[l] + lanes[l + off])
                off //= 2
            tot = lanes[0] if tot is None else T(tot + lanes[0])
        assert y[r] == tot, (r, lens[r])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_long_row_shape_stays_inside_the_error_bound_of_the_sequential_loop(orc, dtype):
    """What "rows beyond mik_spmv_long_row() differ from src's sequential mul! by rounding only" means in numbers: against the row
    sum in extended precision, the wave shape AND the reference's left-to-right loop both sit inside the forward-error bound of that
    loop (len * eps * sum |a_ij x_j|), and over the long rows the wave shape is the more accurate one."""
    import scipy.sparse as sp
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mik_fixtures", os.path.join(os.path.dirname(__file__), "..", "iterativesolvers.jl_amd", "fixtures.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    n, rowptr, colidx, val = fx.irregular_matrix(20000, dtype)
    lens = np.diff(rowptr)
    M = sp.csr_matrix((val, colidx, rowptr), shape=(n, n)).tocsc()
    M.sort_indices()
    A = orc.CSC(n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.copy(), 0)
    x = np.random.default_rng(2).standard_normal(n).astype(dtype)
    orc.set_long_row(256, 1024, 4)             # the library's defaults (mik_spmv_long_row / _segment / _group)
    try:
        y = orc.spmv(A, x)
    finally:
        orc.set_long_row(0)
    yseq = orc.spmv(A, x)
    long_rows = np.flatnonzero(lens > 256)
    assert long_rows.size >= 10 and np.array_equal(np.delete(y, long_rows), np.delete(yseq, long_rows))
    exact = np.array([np.sum(val[rowptr[i]:rowptr[i + 1]].astype(np.longdouble) * x[colidx[rowptr[i]:rowptr[i + 1]]].astype(np.longdouble)) for i in long_rows])
    S = sp.csr_matrix((np.abs(val.astype(np.float64)), colidx, rowptr), shape=(n, n))
    mag = np.asarray(S @ np.abs(x.astype(np.float64)))[long_rows]
    bound = lens[long_rows] * np.finfo(dtype).eps * mag
    err_wave = np.abs(y[long_rows].astype(np.longdouble) - exact).astype(np.float64)
    err_seq = np.abs(yseq[long_rows].astype(np.longdouble) - exact).astype(np.float64)
    assert np.all(err_wave <= bound) and np.all(err_seq <= bound)
    assert np.sqrt(np.mean((err_wave / mag) ** 2)) <= np.sqrt(np.mean((err_seq / mag) ** 2))
